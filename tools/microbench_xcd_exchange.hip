// Microbenchmark: latency of an in-kernel all-to-all exchange of one 8-byte {tag, value} granule per workgroup
// (one 128-B line each), as the persistent single-utterance decoder does three times per sample --
//   (a) NW workgroups spread over all XCDs, agent-scope (sc1) stores and loads        [what ar_persist_kernel does]
//   (b) NW workgroups on ONE XCD (workgroup id % 8 == 0 of a larger grid), sc1 stores and loads
//   (c) the same workgroups on one XCD, group-scope traffic: plain stores (L1 is write-through), sc0 loads (miss L1,
//       hit the XCD's L2) -- valid only because all parties share that L2
// Prints us per exchange (publish own granule -> see everyone's) and the XCC_ID each worker ran on.
// hipcc --offload-arch=gfx950 -O3 tools/microbench_xcd_exchange.hip -o build/mbx
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;

template <int MODE>
__device__ __forceinline__ void put(u64 *p, u64 v) {
    if (MODE == 2 || MODE == 4) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 3 || MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // sc1
}
template <int MODE>
__device__ __forceinline__ u64 get(const u64 *p) {
    if (MODE == 2 || MODE == 3 || MODE == 5) {
        u64 v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stride: participating workgroups are blockIdx.x % stride == 0 (stride 8 = one XCD if dispatch is round-robin)
template <int MODE>
__global__ __launch_bounds__(64) void k_exchange(u64 *g, int nw, int stride, int iters, unsigned *xcc, u64 *ticks) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride, lane = threadIdx.x;
    if (me >= nw) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (lane == 0) xcc[me] = id & 0xf;
    const u64 t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (lane == 0) put<MODE>(g + (size_t)me * 16, ((u64)it << 32) | (unsigned)me);
        bool ok;
        unsigned spins = 0;
        do {
            ok = true;
            for (int w = lane; w < nw; w += 64) ok &= (unsigned)(get<MODE>(g + (size_t)w * 16) >> 32) >= (unsigned)it;
            ok = __all(ok);
        } while (!ok && ++spins < (1u << 22));
        if (!ok) break;
    }
    if (lane == 0) ticks[me] = wall_clock64() - t0;
}

int main() {
    const int iters = 2000;
    u64 *g, *ticks; unsigned *xcc;
    CK(hipMalloc(&g, 256 * 128)); CK(hipMalloc(&ticks, 256 * 8)); CK(hipMalloc(&xcc, 256 * 4));
    std::vector<u64> ht(256); std::vector<unsigned> hx(256);
    printf("config,workers,us_per_exchange_mean,us_per_exchange_max,xcc_ids\n");
    struct Cfg { const char *name; int mode, nw, stride, grid; };
    const Cfg cfgs[] = {
        {"64 workgroups over all XCDs, sc1 (agent scope)", 1, 64, 1, 64},
        {"32 workgroups over all XCDs, sc1 (agent scope)", 1, 32, 1, 32},
        {"32 workgroups on one XCD, sc1 (agent scope)", 1, 32, 8, 256},
        {"32 workgroups on one XCD, plain stores + sc0 loads (group scope)", 2, 32, 8, 256},
        {"32 workgroups on one XCD, sc0 stores + sc0 loads", 3, 32, 8, 256},
        {"32 workgroups on one XCD, plain stores + sc1 loads", 4, 32, 8, 256},
        {"32 workgroups on one XCD, sc1 stores + sc0 loads", 5, 32, 8, 256},
        {"32 workgroups on one XCD, sc0 stores + sc1 loads", 6, 32, 8, 256},
    };
    for (const Cfg &c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(g, 0, 256 * 128));
#define L(M) hipLaunchKernelGGL(k_exchange<M>, dim3(c.grid), dim3(64), 0, 0, g, c.nw, c.stride, c.mode == 2 ? 20 : iters, xcc, ticks)
            switch (c.mode) { case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; default: L(6); }
#undef L
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(ht.data(), ticks, c.nw * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx.data(), xcc, c.nw * 4, hipMemcpyDeviceToHost));
        double mean = 0, mx = 0;
        unsigned seen = 0;
        for (int w = 0; w < c.nw; ++w) { const double us = ht[w] * 0.01 / (c.mode == 2 ? 20 : iters); mean += us / c.nw; mx = us > mx ? us : mx; seen |= 1u << hx[w]; }
        printf("%s,%d,%.3f,%.3f,0x%x\n", c.name, c.nw, mean, mx, seen);
    }
    return 0;
}
