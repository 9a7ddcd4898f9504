#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself (runs in the build container only).

Imports ``/root/reference/model.py`` (needs the single constant ``omegaconf.MISSING``,
``model.py:6``; omegaconf is not installed, so a 2-attribute in-process module stands
in for it, as SURVEY 8c documents), loads this project's seeded random-init weights
(``vectorquantizedcpc_amd.synth``) through ``load_state_dict`` and records what
``Encoder.encode`` / ``Encoder.forward`` (``model.py:59-86``) return on the PyTorch-CPU
path.  Fixtures are DATA only: case parameters, code indices, argmin margins, SHA-256
of the bit patterns of each stage, a few full rows, float64 checksums.  Nothing of the
reference's source travels; the GPU box rebuilds inputs and weights from the seed.

Vocoder half: the reference's arithmetic (third-party ``rnnms``) is absent.  The wrapper's own
glue (``network_vocoder.py:41-78``: embed, x2 nearest upsample, speaker broadcast, concat) IS
importable once ``rnnms.networks.vocoder`` is replaced by an in-process capture stub:
``vocoder_glue.npz`` records what the reference's ``Vocoder.generate`` / ``Vocoder.forward`` hand to
``rnnms`` -- that pins the glue (a-9) by the reference, nothing more.
``preprocess.npz``: the reference's own ``preemphasis`` / ``mulaw_encode`` / ``mulaw_decode`` (``preprocess.py:16-35``).
``vocoder_selforacle.npz`` comes from this project's own CPU oracle (``tests/golden/make_vocoder_fixtures.py``): self-oracle,
parity unpinned.

Usage:  python tools/gen_golden.py            (writes tests/golden/)
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectorquantizedcpc_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name -> (B, T, ln_affine, codebook)
ENCODER_CASES = {
    "c1_init": (1, 200, "init", "init"),          # BASELINE configs[0]: one 2 s utterance
    "c2_init": (64, 128, "init", "init"),         # BASELINE configs[1]: 64 x 128 frames
    "c2_random_data": (64, 128, "random", "data"),  # perturbed LN affine, data-scale codebook
    "ragged_3x32": (3, 32, "random", "init"),     # 48 rows, smallest MKL "large-M" regime
    "tiny_1x16": (1, 16, "init", "init"),         # 8 rows: reference takes MKL's small-M path
    "odd_2x33": (2, 33, "random", "data"),        # odd T: floor((T-2)/2)+1 = 16 frames each, last frame used
    "long_1x300": (1, 300, "init", "init"),       # B = 1 but 80*T > 20480: ATen switches to oneDNN
    # the edge of the bit-exact contract (>= 16 output rows): encode.py:42-46 is batch 1, so 0.32-0.62 s utterances live here
    "edge_1x32": (1, 32, "init", "init"),         # 16 rows: the smallest call the contract covers
    "edge_1x34": (1, 34, "random", "data"),       # 17 rows: one row past a whole 16-row tile
    "edge_1x62": (1, 62, "init", "data"),         # 31 rows
    "edge_2x16": (2, 16, "random", "init"),       # 2 x 8 = 16 rows from a batched call (oneDNN conv order)
}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def import_reference():
    for name in ("omegaconf", "omegaconf.omegaconf"):
        m = types.ModuleType(name)
        m.MISSING = "???"
        sys.modules[name] = m
    sys.path.insert(0, "/root/reference")
    import model  # noqa
    return model


def encoder_fixture(model, name, B, T, ln_affine, codebook):
    torch.manual_seed(0)
    enc = model.Encoder(model.ConfEncoder(80, 512, 512, 64, 256))
    sd = synth.encoder_state_dict(ln_affine=ln_affine, codebook=codebook)
    assert list(sd.keys()) == list(enc.state_dict().keys())
    enc.load_state_dict(sd)
    enc.eval()
    mel = synth.mel(name, B, T)
    stages = {}
    enc.conv.register_forward_hook(lambda m, i, o: stages.__setitem__("conv", o.transpose(1, 2).contiguous().clone()))
    for i, mod in enumerate(enc.encoder):
        if not isinstance(mod, torch.nn.ReLU):       # ReLU is in-place: its input hook output is overwritten
            mod.register_forward_hook(lambda m, inp, o, i=i: stages.__setitem__(f"enc{i}", o.clone()))
    with torch.no_grad():
        z, c, idx = enc.encode(mel)
        z_pre = stages["enc14"]
        # argmin margins from the reference's own distance matrix (model.py:107-110)
        E = enc.codebook.embedding
        xf = z_pre.reshape(-1, 64)
        dist = torch.addmm(torch.sum(E ** 2, dim=1) + torch.sum(xf ** 2, dim=1, keepdim=True),
                           xf, E.t(), alpha=-2.0, beta=1.0)
        top2 = torch.topk(dist, 2, dim=1, largest=False).values
        zf, cf, loss, ppl = enc(mel)
    out = {
        "case": np.array([B, T]), "ln_affine": np.array(ln_affine), "codebook": np.array(codebook),
        "indices": idx.numpy().astype(np.int16),
        "d_best": top2[:, 0].numpy(), "d_second": top2[:, 1].numpy(),
        "n_exact_ties": np.array(int((top2[:, 0] == top2[:, 1]).sum())),
        "loss": np.array(loss.item(), np.float32), "perplexity": np.array(ppl.item(), np.float32),
    }
    named = {"conv": stages["conv"], "z_pre": z_pre, "z": z, "c": c, "z_fwd": zf, "c_fwd": cf}
    for i in (0, 2, 3, 5, 6, 8, 9, 11, 12):
        named[f"enc{i}"] = stages[f"enc{i}"]
    for k, v in named.items():
        a = v.detach().numpy()
        a2 = a.reshape(-1, a.shape[-1])
        out[f"sha_{k}"] = np.array(sha(a))
        out[f"sum_{k}"] = np.array([a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum()])
        out[f"rows_{k}"] = a2[:: max(1, a2.shape[0] // 4)][:4].copy()   # 4 spread-out full rows
    np.savez_compressed(os.path.join(GOLD, f"encoder_{name}.npz"), **out)
    ulp = np.spacing(np.abs(out["d_best"]).astype(np.float32))
    print(f"{name}: rows={B * ((T - 2) // 2 + 1)} exact ties={out['n_exact_ties']} "
          f"rows with margin<=4ulp={(out['d_second'] - out['d_best'] <= 4 * ulp).sum()} "
          f"loss={out['loss']:.6g} ppl={out['perplexity']:.6g}")


def glue_fixture_from_reference():
    """Run the reference's own ``Vocoder.generate`` / ``Vocoder.forward`` (network_vocoder.py:41-78) with
    ``rnnms.networks.vocoder`` replaced by a stub that records what it is called with."""
    from dataclasses import dataclass
    seen = {}

    @dataclass
    class ConfRNNMSVocoder:            # network_vocoder.py:24 only needs a default-constructible class
        dim_i_feature: int = 128

    class RNNMSVocoder(torch.nn.Module):
        def __init__(self, conf):
            super().__init__()

        def forward(self, x, latent_series):                   # network_vocoder.py:67
            seen["forward"] = (x.clone(), latent_series.clone())
            return torch.zeros(x.shape[0], x.shape[1], 256)

        def generate(self, z_spk_series):                      # network_vocoder.py:78
            seen["generate"] = z_spk_series.clone()
            return torch.zeros(z_spk_series.shape[0], 160 * z_spk_series.shape[1])

    for name in ("rnnms", "rnnms.networks", "rnnms.networks.vocoder"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["rnnms.networks.vocoder"].ConfRNNMSVocoder = ConfRNNMSVocoder
    sys.modules["rnnms.networks.vocoder"].RNNMSVocoder = RNNMSVocoder
    import network_vocoder as nv                               # /root/reference (sys.path set by import_reference)
    voc = nv.Vocoder(nv.ConfVocoder(512, 64, 102, 64))
    sd = synth.vocoder_state_dict()
    own = {k: sd[k] for k in ("code_embedding.weight", "speaker_embedding.weight")}
    assert sorted(voc.state_dict().keys()) == sorted(own.keys())          # the stub has no parameters
    voc.load_state_dict(own)
    voc.eval()
    z = synth.randint("glue/z", (2, 5), 512)
    spk = synth.randint("glue/spk", (2,), 102)
    x = synth.randint("glue/x", (2, 37), 256)
    with torch.no_grad():
        voc.generate(z, spk)
        voc(x, z, spk)
    series = seen["generate"]
    assert series.shape == (2, 10, 128) and torch.equal(seen["forward"][1], series) and torch.equal(seen["forward"][0], x)
    np.savez_compressed(os.path.join(GOLD, "vocoder_glue.npz"), z=z.numpy(), speaker=spk.numpy(),
                        series=series.numpy(), x=x.numpy(),
                        source=np.array("reference network_vocoder.py Vocoder.generate/forward, rnnms replaced by a capture stub"))
    print("vocoder_glue: reference Vocoder.generate/forward hand rnnms a", tuple(series.shape), "series; forward passes x through")


def preprocess_fixture_from_reference():
    """The reference's own ``preprocess.py`` functions that sit on the path (``:16-17`` pre-emphasis = the first stage of
    ``wave_to_mel``; ``:20-35`` mu-law encode / decode = the last step of the sample loop).  The module imports ``librosa`` at
    the top (absent offline); none of these three functions uses it, so an empty in-process module stands in for the name."""
    for name in ("librosa",):
        sys.modules.setdefault(name, types.ModuleType(name))
    import scipy.signal  # noqa: F401  (preprocess.py:17 calls scipy.signal.lfilter through `import scipy`)
    import preprocess as pp                                    # /root/reference
    cls = np.arange(256)
    y = 2.0 * cls / 255.0 - 1.0
    dec = pp.mulaw_decode(y, 256)                              # what the sample loop emits for class x: mulaw_decode(2x/255 - 1)
    x = np.asarray(synth.uniform01("pp/x", 2048, synth.SEED), np.float64) * 2.0 - 1.0
    x[:4] = (-1.0, 1.0, 0.0, -0.0)
    enc = pp.mulaw_encode(x, 256)
    x32 = x.astype(np.float32)
    enc32 = pp.mulaw_encode(x32, 256)
    sig = (np.asarray(synth.uniform01("pp/sig", 4096, synth.SEED), np.float64) * 2.0 - 1.0).astype(np.float32)
    pre = pp.preemphasis(sig, 0.97)
    np.savez_compressed(os.path.join(GOLD, "preprocess.npz"), mulaw_decode_256=dec, mulaw_encode_in=x, mulaw_encode_out=enc,
                        mulaw_encode_out_f32_in=enc32, preemph_in=sig, preemph_out=pre,
                        source=np.array("reference preprocess.py preemphasis / mulaw_encode / mulaw_decode (librosa replaced by an empty module)"))
    print("preprocess: mulaw_decode", dec.dtype, dec[:2], dec[-2:], "| mulaw_encode range", enc.min(), enc.max(), "| preemphasis", pre.dtype, pre.shape)


def main():
    os.makedirs(GOLD, exist_ok=True)
    model = import_reference()
    print("reference imported from", model.__file__, "| torch", torch.__version__,
          "| cpu capability", torch.backends.cpu.get_cpu_capability(), "| threads", torch.get_num_threads())
    for name, args in ENCODER_CASES.items():
        encoder_fixture(model, name, *args)
    glue_fixture_from_reference()
    preprocess_fixture_from_reference()
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    main()
