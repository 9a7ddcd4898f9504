#!/usr/bin/env python3
"""One diagnostic run for the `rocprofv3 --pmc` + hipGraph replay SIGSEGV.

The profiler's own signal handler prints raw frame addresses only.  This target does the smallest thing
that crashed (one generate() call with graph replay) while a watcher thread keeps the newest
/proc/self/maps on disk, so the frames and the faulting address can be attributed to mappings -- and,
because the build container has the same image, symbolized offline from the same .so files.

    rocprofv3 --pmc SQ_WAVES --output-format csv -d gpurun_out/pmc_probe -- python3 tools/pmc_crash_probe.py graph
    (second argument `eager` = the same call with use_graph 0, for comparison)
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"


def watcher():
    i = 0
    while True:
        try:
            with open("/proc/self/maps") as f:
                txt = f.read()
            with open(os.path.join(OUT, f"pmc_probe_maps_{mode}_{i % 2}.txt"), "w") as f:
                f.write(txt)
            threads = sorted(os.listdir("/proc/self/task"))
            with open(os.path.join(OUT, f"pmc_probe_threads_{mode}.txt"), "w") as f:
                for t in threads:
                    try:
                        f.write(t + " " + open(f"/proc/self/task/{t}/comm").read())
                    except OSError:
                        pass
        except OSError:
            pass
        i += 1
        time.sleep(0.02)


threading.Thread(target=watcher, daemon=True).start()

import torch  # noqa: E402
import vectorquantizedcpc_amd as V  # noqa: E402
from vectorquantizedcpc_amd import synth  # noqa: E402

voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
if mode == "eager":
    voc.set_option("use_graph", 0)
steps = int(os.environ.get("PROBE_STEPS_PER_GRAPH", "0"))
if steps:
    voc.set_option("steps_per_graph", steps)
z = synth.randint("probe/z", (32, int(os.environ.get("PROBE_CODES", "2"))), 512).cuda()
spk = (torch.arange(32) % 102).cuda()
print(f"[probe] pid {os.getpid()} mode {mode}: generate()", file=sys.stderr, flush=True)
if os.environ.get("PROBE_BENCH_SEQUENCE"):
    # what round 1's crashing command did: bench.py --steps 1 --warmup 1 = (encoder + generate) twice on cached graphs,
    # then 3 x 2020 back-to-back launches of the timing harness
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict())
    enc = enc.cuda().eval()
    mel = synth.mel("probe/mel", 32, 2 * z.shape[1]).cuda()
    for it in range(2):
        idx = enc.encode_indices(mel)
        wav = voc.generate(idx, spk, seed=13, utt_base=0)
        print(f"[probe] bench sequence: call {it} enqueued", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    print("[probe] bench sequence: both calls finished; timing harness", file=sys.stderr, flush=True)
    print("[probe] kernel_times", voc.kernel_times(2000), file=sys.stderr, flush=True)
wav = voc.generate(z, spk, seed=13, utt_base=0)
torch.cuda.synchronize()
print(f"[probe] mode {mode} finished, |wav| max {float(wav.abs().max()):.3f}", file=sys.stderr, flush=True)
