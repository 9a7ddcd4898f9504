// Microbenchmark: cost of dependent kernel boundaries on MI355X (graph vs eager), to size the
// per-sample decode loop.  hipcc --offload-arch=gfx950 -O3 tools/microbench_launch.hip -o build/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }
__global__ void k_chain(const float *in, float *out, int n) {      // every block reads all of `in` (n floats), writes 1 float
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[blockIdx.x] = s * 1e-9f;
}

template <class F> float time_graph(hipStream_t s, int nodes, int reps, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (nodes * reps);
}
template <class F> float time_eager(hipStream_t s, int n, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) launch(i);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *buf; CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20));
    float *b2 = buf + (8 << 20);
    printf("config,us_per_kernel\n");
    for (int grid : {1, 16, 240, 1024}) {
        for (int blk : {64, 256, 1024}) {
            float g = time_graph(s, 480, 20, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(blk), 0, s); });
            float e = time_eager(s, 5000, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(blk), 0, s); });
            printf("empty grid=%d block=%d graph,%.3f\nempty grid=%d block=%d eager,%.3f\n", grid, blk, g, grid, blk, e);
        }
    }
    for (int grid : {1, 16, 240}) {
        float g = time_graph(s, 480, 20, [&](int) { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, buf); });
        printf("touch grid=%d block=256 graph,%.3f\n", grid, g);
    }
    // ping-pong all-gather chain: every block reads the whole previous vector (n floats)
    for (int n : {1024, 28672, 114688}) {       // 4 KB, 112 KB (896x32), 448 KB
        for (int grid : {16, 240}) {
            float g = time_graph(s, 480, 20, [&](int i) {
                hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, (i & 1) ? b2 : buf, (i & 1) ? buf : b2, n); });
            printf("chain n=%d grid=%d graph,%.3f\n", n, grid, g);
        }
    }
    // two alternating streams inside one graph? (fork/join): skipped
    return 0;
}
