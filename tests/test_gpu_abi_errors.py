"""-m gpu: the C ABI's error contract (include/vqcpc.h): negative status + message, never a crash;
and bench.py's contract line on a tiny workload."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_status_codes_and_messages():
    lib = _lib.load()
    assert lib.vqcpc_device_count() >= 1
    h = C.c_void_p()
    assert lib.vqcpc_encoder_create(None, C.byref(h)) == -1 and b"null" in lib.vqcpc_last_error()
    w = _lib.EncoderWeights()
    w.channels, w.z_dim, w.in_channels, w.n_embeddings, w.c_dim = 256, 64, 80, 512, 256
    assert lib.vqcpc_encoder_create(C.byref(w), C.byref(h)) == -1 and b"channels must be 512" in lib.vqcpc_last_error()
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict())
    enc = enc.cuda().eval()
    hn = enc._native()
    mel = torch.zeros(1, 80, 8, device="cuda")
    z = torch.empty(1, 4, 64, device="cuda")
    idx = torch.empty(1, 4, dtype=torch.int64, device="cuda")
    assert lib.vqcpc_encoder_encode(hn, mel.data_ptr(), 1, 1, 0, z.data_ptr(), None, idx.data_ptr(), None, None) == -1
    assert lib.vqcpc_encoder_encode(hn, mel.data_ptr(), 0, 8, 0, z.data_ptr(), None, idx.data_ptr(), None, None) == -1
    assert lib.vqcpc_encoder_encode(hn, mel.data_ptr(), 1, 8, 7, z.data_ptr(), None, idx.data_ptr(), None, None) == -1
    assert lib.vqcpc_encoder_encode(hn, None, 1, 8, 0, z.data_ptr(), None, idx.data_ptr(), None, None) == -1
    assert lib.vqcpc_encoder_encode(hn, mel.data_ptr(), 1, 8, 0, z.data_ptr(), None, idx.data_ptr(), None, None) == 0
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(synth.vocoder_state_dict())
    voc = voc.cuda().eval()
    hv = voc._native()
    assert lib.vqcpc_vocoder_set_option(hv, b"steps_per_graph", 7) == -1        # must be even
    assert lib.vqcpc_vocoder_set_option(hv, b"no_such_option", 1) == -1
    nc = (C.c_int * 1)(9)
    wav = torch.empty(1, 1280, device="cuda")
    zz = torch.zeros(1, 4, dtype=torch.int64, device="cuda")
    sp = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert lib.vqcpc_vocoder_generate(hv, zz.data_ptr(), sp.data_ptr(), 1, 4, nc, 1, 0, None, wav.data_ptr(), None, 0, None) == -1
    assert b"n_codes" in lib.vqcpc_last_error()
    with pytest.raises(RuntimeError, match="libvqcpc_hip"):
        _lib.check(-1)


def test_bench_contract_line_on_a_tiny_workload():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--utterances-per-gpu", "2", "--frames", "8",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"],
                         capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-800:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "samples/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["value"] > 0 and "workload" in d["config"]
