import ctypes as C, os, sys
import numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT)
from vectorquantizedcpc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "stamps", "libvqcpc_hip.so")
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth
enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256)); enc.load_state_dict(synth.encoder_state_dict()); enc = enc.cuda().eval()
mel = synth.mel("bench/c1", 1, 200).cuda()
for _ in range(5): enc.encode_indices(mel)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
assert _lib.load().vqcpc_debug_res_stamps(buf) == 0
s = np.array(buf, dtype=np.int64) * 0.01
print("workgroup (row tile 0, column group 1): us since the XCC check")
b = s - s[0]
print(f"conv   : start {b[1]:6.2f}  computed {b[3]:6.2f}  published {b[4]:6.2f}")
for L in range(1, 5):
    print(f"Linear{L}: enter {b[1 + 4 * L]:6.2f}  inputs ready {b[2 + 4 * L]:6.2f}  computed {b[3 + 4 * L]:6.2f}  published {b[4 + 4 * L]:6.2f}")
print(f"tail   : enter {b[21]:6.2f}  inputs ready {b[22]:6.2f}  (column group 0 goes on alone)")
