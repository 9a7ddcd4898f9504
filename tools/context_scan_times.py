import sys, time, torch
sys.path.insert(0, '/root/repo')
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth
enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256)); enc.load_state_dict(synth.encoder_state_dict()); enc = enc.cuda().eval()
mel = synth.mel("bench/c1", 1, 200).cuda()
def wall(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
base = wall(lambda: enc.encode_indices(mel))
for mode, name in ((1, "resident, plain stores (one XCD)"), (2, "resident, agent-scope stores forced"), (0, "one launch per time step")):
    enc.set_option("persistent_context", mode)
    us = wall(lambda: enc.encode(mel))
    print(f"{name}: {us:.1f} us per encode() of 1 x 200 frames; context part {us - base:.1f} us = {(us - base) / 100:.2f} us per time step")
enc.set_option("persistent_context", 1)
