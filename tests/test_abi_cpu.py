"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/vqcpc.h declares, the Python mirror keeps the reference's surface, and the product
path fails loudly without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vqcpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqcpc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vqcpc.h but not exported"
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.load().vqcpc_abi_version() == 1


def test_header_is_plain_c(tmp_path):
    """include/vqcpc.h is the drop-in boundary: it must compile as C99 with the calls INTEGRATION.md shows."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('''#include "vqcpc.h"
#include <stddef.h>
int use(vqcpc_encoder *enc, vqcpc_vocoder *voc, float *p, int64_t *i, void *s) {
    int rc = vqcpc_encoder_encode(enc, p, 1, 32, VQCPC_CONV_AUTO, p, NULL, i, NULL, s);
    rc |= vqcpc_vocoder_generate(voc, i, i, 1, 16, NULL, 13u, 0u, NULL, p, NULL, 0, s);
    return rc;
}
''')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_state_dict_surface_matches_reference():
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    sd = synth.encoder_state_dict()
    assert list(enc.state_dict().keys()) == list(sd.keys())        # SURVEY 8a-1 key set and order
    assert all(enc.state_dict()[k].shape == sd[k].shape for k in sd)
    enc.load_state_dict(sd)
    voc = V.Vocoder(V.ConfVocoder())
    sv = synth.vocoder_state_dict()
    assert set(voc.state_dict().keys()) == set(sv.keys())
    assert all(voc.state_dict()[k].shape == sv[k].shape for k in sv)
    voc.load_state_dict(sv)
    assert isinstance(enc.encoder[-1], torch.nn.Linear)             # encode.py:40 hooks this module


def test_no_cpu_fallback():
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256)).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc.encode(torch.zeros(1, 80, 32))
    voc = V.Vocoder(V.ConfVocoder()).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        voc.generate(torch.zeros(1, 4, dtype=torch.long), torch.zeros(1, dtype=torch.long))
    from vectorquantizedcpc_amd import loudness, preprocess         # the front-end drop-ins too
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        preprocess.wave_to_mel(torch.zeros(16000))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loudness.Meter(16000).integrated_loudness(torch.zeros(16000))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loudness.normalize.loudness(torch.zeros(16000), -20.0, -23.0)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import or load it."""
    pkg = os.path.join(ROOT, "vectorquantizedcpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "libvqcpc_oracle" not in text, f
            if f.endswith((".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "#include \"../../oracle" not in text and "dlopen" not in text, f


def test_synth_is_deterministic():
    a = synth.encoder_state_dict()["encoder.2.weight"]
    b = synth.encoder_state_dict()["encoder.2.weight"]
    assert torch.equal(a, b)
    assert abs(float(synth.mel("x", 2, 8).mean()) - 0.5) < 0.05


def test_weight_slots_follow_the_module_without_state_dict():
    """The native handles are keyed on (data_ptr, _version) of the weights, read per call through WeightSlots instead of
    state_dict() (71 us vs 5 us): the slots must name exactly the reference's state_dict keys (minus the EMA buffers the
    inference path never reads) and must see load_state_dict, dtype/device moves and buffer replacement."""
    import copy

    import torch

    import vectorquantizedcpc_amd as V
    from vectorquantizedcpc_amd import _lib, synth
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    assert set(enc._WEIGHT_NAMES) == set(enc.state_dict()) - {"codebook.ema_count", "codebook.ema_weight"}
    slots = _lib.WeightSlots(enc, enc._WEIGHT_NAMES)
    k0 = _lib.WeightSlots.key(slots.tensors())
    assert k0 == _lib.WeightSlots.key(slots.tensors())
    enc.load_state_dict(synth.encoder_state_dict())                  # copy_ in place: versions move
    k1 = _lib.WeightSlots.key(slots.tensors())
    assert k1 != k0
    enc.codebook.embedding = torch.zeros(512, 64)                    # a buffer REPLACED by assignment
    k2 = _lib.WeightSlots.key(slots.tensors())
    assert k2 != k1 and slots.tensors()[slots.names.index("codebook.embedding")] is enc.codebook.embedding
    enc.double()                                                     # .to(): parameters keep identity, storage moves
    assert _lib.WeightSlots.key(slots.tensors()) != k2
    voc = V.Vocoder(V.ConfVocoder())
    vs = _lib.WeightSlots(voc, list(voc.state_dict().keys()))
    assert [t.shape for t in vs.tensors()] == [t.shape for t in voc.state_dict().values()]
    assert "_slots" not in copy.deepcopy(enc).__dict__ and copy.deepcopy(enc).codebook._owner() is not enc
