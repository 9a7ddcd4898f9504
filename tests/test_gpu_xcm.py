"""-m gpu: the matrix-core per-XCD resident decoders (csrc/ar_xcm.hip: 16 decode slots per XCD, the default decode path
between 69 and 511 utterances in flight).

Checked like the VALU form in test_gpu_xcd.py: (1) bit for bit against the launch-per-step kernels (`xcd` = 0), which
test_gpu_vocoder.py checks against the oracles on their own; (2) draw by draw against the C oracle on the same history
(the parity-unpinned self-oracle, tolerance 2e-5 on the Gumbel-max score); (3) size-independent properties at configs[3]'s
single-GPU form (256 x 32 000 through 128 slots): an utterance alone equals itself in the batch, fewer slots give the same
samples; (4) the abort path of its in-kernel hand-offs.
"""
import numpy as np
import pytest
import torch

import oracle
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import driver, synth

pytestmark = pytest.mark.gpu
_cache = {}


def vocoder(fresh=False):
    if fresh or "v" not in _cache:
        sd = synth.vocoder_state_dict()
        v = V.Vocoder(V.ConfVocoder())
        v.load_state_dict(sd)
        v = v.to("cuda").eval()
        if fresh:
            return v, sd
        _cache["v"] = (v, sd)
    return _cache["v"]


def _both_paths(voc, z, spk, **kw):
    out = {}
    for name, (xcd, xcm) in (("xcm", (-1, 1)), ("launch", (0, 0))):
        voc.set_option("xcd", xcd)
        voc.set_option("xcm", xcm)
        wav, mu = voc.generate(z, spk, return_mulaw=True, **kw)
        voc.check()
        assert voc.last_path() == (3 if name == "xcm" else voc.last_path())
        out[name] = (wav.cpu(), mu.cpu())
    voc.set_option("xcd", -1)
    voc.set_option("xcm", -1)
    return out


@pytest.mark.parametrize("B,Tc,ragged", [(1, 2, False), (7, 3, True), (16, 2, False), (17, 3, True), (40, 2, True), (128, 2, False),
                                         (150, 3, True)])
def test_same_samples_as_the_launch_path(B, Tc, ragged):
    voc, _ = vocoder()
    z = synth.randint(f"xcm/z{B}", (B, Tc), 512).cuda()
    spk = synth.randint(f"xcm/s{B}", (B,), 102).cuda()
    n_codes = [max(1, Tc - (b % Tc)) for b in range(B)] if ragged else None
    out = _both_paths(voc, z, spk, n_codes=n_codes, seed=13, utt_base=3)
    assert torch.equal(out["xcm"][1], out["launch"][1]) and torch.equal(out["xcm"][0], out["launch"][0])
    assert int((out["xcm"][1] != 0).sum()) > 0.9 * 320 * sum(n_codes or [Tc] * B)


def test_draw_by_draw_against_the_oracle():
    voc, sd = vocoder()
    voc.set_option("xcm", 1)
    try:
        B, Tc, steps = 19, 2, 400
        z = synth.randint("xcm/oz", (B, Tc), 512)
        spk = synth.randint("xcm/os", (B,), 102)
        wav, mu = voc.generate(z.cuda(), spk.cuda(), seed=13, utt_base=11, return_mulaw=True, max_steps=steps)
        voc.check()
        assert voc.last_path() == 3
        wav, mu = wav.cpu().numpy(), mu.cpu().numpy()
        exact = total = 0
        for b in (0, 7, 16, 18):            # slots on XCDs 0, 7, 0 (second local slot), 2
            s_gpu = mu[b, :steps]
            inputs = np.concatenate([[128], s_gpu[:-1]])
            r = oracle.vocoder_generate(sd, z[b].numpy(), int(spk[b]), seed=13, utterance=11 + b, n_steps=steps, inputs=inputs,
                                        want_logits=True)
            for t in np.nonzero(r["samples"] != s_gpu)[0]:
                pick, sc = oracle.sample_from_logits(r["logits"][t], 13, 11 + b, int(t))
                assert sc[pick] - sc[int(s_gpu[t])] <= 2e-5, (b, int(t))
            exact += int((r["samples"] == s_gpu).sum())
            total += steps
            assert np.array_equal(wav[b, :steps], np.array([oracle.mulaw_decode(int(s)) for s in s_gpu], np.float32))
            assert not wav[b, steps:].any()
        assert exact >= 0.999 * total
    finally:
        voc.set_option("xcm", -1)


def test_more_utterances_than_slots_and_alone_equals_in_batch():
    """Continuous batching: 200 ragged utterances through 128 decode slots and through 24 (three per XCD); every utterance
    equals the same utterance decoded alone (its sampling stream does not depend on the placement)."""
    voc, _ = vocoder()
    n, Tc = 200, 3
    z = synth.randint("xcm/cz", (n, Tc), 512).cuda()
    spk = synth.randint("xcm/cs", (n,), 102).cuda()
    n_codes = [1 + (7 * i) % Tc for i in range(n)]
    ids = list(range(100, 100 + n))
    voc.set_option("xcm", 1)
    try:
        res = {}
        for slots in (128, 24):
            voc.set_option("xcm_slots", slots)
            wav, mu = voc.generate(z, spk, n_codes=n_codes, seed=5, utt_ids=ids, return_mulaw=True)
            voc.check()
            assert voc.last_path() == 3
            res[slots] = (wav.cpu(), mu.cpu())
        assert torch.equal(res[128][1], res[24][1]) and torch.equal(res[128][0], res[24][0])
        voc.set_option("xcm", 0)
        for i in (0, 77, 199):
            w1, m1 = voc.generate(z[i:i + 1, :n_codes[i]], spk[i:i + 1], seed=5, utt_ids=[ids[i]], return_mulaw=True)
            L = 320 * n_codes[i]
            assert torch.equal(m1[0].cpu(), res[128][1][i, :L]) and torch.equal(w1[0].cpu(), res[128][0][i, :L])
            assert not res[128][1][i, L:].any()
    finally:
        voc.set_option("xcm_slots", 128)
        voc.set_option("xcm", -1)


def test_path_by_utterances_in_flight():
    """The default choice: VALU per-XCD decoders up to 68 utterances in flight, the matrix-core ones from 69 to 511, launches above."""
    voc, _ = vocoder()
    for B, want in ((32, 2), (68, 2), (69, 3), (130, 3), (400, 3), (520, 0)):
        z = synth.randint(f"xcm/pz{B}", (B, 1), 512).cuda()
        spk = torch.zeros(B, dtype=torch.long, device="cuda")
        voc.generate(z, spk, seed=1, max_steps=8)
        voc.check()
        assert voc.last_path() == want, (B, voc.last_path())
    voc.set_option("slots", 100)                 # 600 utterances through 100 decode slots: 100 in flight
    try:
        z = synth.randint("xcm/pz600", (600, 1), 512).cuda()
        voc.generate(z, torch.zeros(600, dtype=torch.long, device="cuda"), seed=1, max_steps=8)
        voc.check()
        assert voc.last_path() == 3
    finally:
        voc.set_option("slots", 0)


def test_configs3_single_gpu_form_properties():
    """configs[3] on one GPU: 256 x 32 000 through the 128 slots (two utterances per slot, back to back): finite, in range, and
    three rows equal the same utterances decoded alone."""
    voc, _ = vocoder()
    z = synth.randint("xcm/bz", (256, 100), 512).cuda()
    spk = (torch.arange(256, device="cuda") % 102)
    wav, mu = voc.generate(z, spk, seed=13, utt_base=0, return_mulaw=True)
    voc.check()
    assert voc.last_path() == 3
    assert wav.shape == (256, 32000) and torch.isfinite(wav).all() and wav.abs().max() <= 1.0
    assert mu.min() >= 0 and mu.max() <= 255 and mu.unique().numel() > 32
    ms, n = voc.last_timing()
    print("256 x 32000 through 128 slots on the matrix-core per-XCD decoders: %.2f us per sample step, %.2f M samples/s"
          % (ms * 1e3 / n, 256 * 32000 / ms / 1e3))
    for i in (0, 129, 255):
        w1, m1 = voc.generate(z[i:i + 1], spk[i:i + 1], seed=13, utt_base=i, return_mulaw=True)
        voc.check()
        assert torch.equal(m1[0], mu[i]) and torch.equal(w1[0], wav[i])


def _reference_bits(z, spk, **kw):
    ref, _ = vocoder(fresh=True)
    ref.set_option("xcd", 0)
    ref.set_option("fuse_fc2", 0)
    wav, mu = ref.generate(z, spk, return_mulaw=True, **kw)
    ref.check()
    return wav.cpu(), mu.cpu()


def test_handoff_timeout_is_reported_by_the_same_call_and_the_rerun_is_right():
    """One worker skips a candidate publish: every wait behind it gives up after the (shortened) deadline, check() raises for
    THAT call, the handle falls back to launches, and the repeated call gives the samples of the undisturbed paths."""
    voc, _ = vocoder(fresh=True)
    B = 80
    z = synth.randint("xcm/ez", (B, 2), 512).cuda()
    spk = synth.randint("xcm/es", (B,), 102).cuda()
    voc.set_option("xcm", 1)
    voc.set_option("xcd_timeout_ms", 20)
    voc.set_option("xcd_debug_drop_step", 200)
    wav, mu = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True, async_=True)
    with pytest.raises(RuntimeError, match="timed out"):
        voc.check()
    assert not bool((mu[0, 260:] != 0).any())            # the call that suffered it is incomplete (and said so)
    voc.set_option("xcd_debug_drop_step", -1)
    wav2, mu2 = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True)      # the handle has fallen back
    voc.check()
    assert voc.last_path() == 0
    want = _reference_bits(z, spk, seed=9, utt_base=0)
    assert torch.equal(mu2.cpu(), want[1]) and torch.equal(wav2.cpu(), want[0])
    voc3, _ = vocoder(fresh=True)
    voc3.set_option("xcm", 1)
    voc3.set_option("xcd_timeout_ms", 20)
    voc3.set_option("xcd_debug_drop_step", 100)
    with pytest.warns(UserWarning, match="decode repeated"):
        wav3 = driver.generate_checked(voc3, z, spk, seed=9, utt_base=0)
    assert torch.equal(wav3.cpu(), want[0])


def test_128_x_32000_bit_equal_to_the_launch_path():
    """128 utterances of 2 s through the 128 matrix-core slots: the same 4 096 000 samples as the launch-per-step kernels (one
    run each)."""
    voc, _ = vocoder()
    z = synth.randint("xcm/fz128", (128, 100), 512).cuda()
    spk = (torch.arange(128, device="cuda") % 102)
    out = {}
    for name, (xcd, xcm) in {"xcm": (-1, 1), "launch": (0, 0)}.items():
        voc.set_option("xcd", xcd)
        voc.set_option("xcm", xcm)
        wav, mu = voc.generate(z, spk, seed=13, utt_base=0, return_mulaw=True)
        assert voc.last_path() == (3 if name == "xcm" else 0)
        out[name] = (wav.cpu(), mu.cpu())
    voc.set_option("xcd", -1)
    voc.set_option("xcm", -1)
    assert torch.equal(out["xcm"][1], out["launch"][1]) and torch.equal(out["xcm"][0], out["launch"][0])


def test_2400_consecutive_steps_draw_by_draw_against_the_oracle():
    """2 400 consecutive steps of two utterances inside a batch of 80 on the matrix-core decoders, every draw checked against
    the C oracle on the same history (a late-step divergence cannot hide behind 400-step windows)."""
    voc, sd = vocoder()
    voc.set_option("xcm", 1)
    try:
        B, Tc, steps = 80, 8, 2400
        z = synth.randint("xcm/lz", (B, Tc), 512)
        spk = synth.randint("xcm/ls", (B,), 102)
        wav, mu = voc.generate(z.cuda(), spk.cuda(), seed=21, utt_base=5, return_mulaw=True, max_steps=steps)
        assert voc.last_path() == 3
        mu = mu.cpu().numpy()
        for b in (0, 79):
            inputs = np.concatenate([[128], mu[b, :steps - 1]])
            r = oracle.vocoder_generate(sd, z[b].numpy(), int(spk[b]), seed=21, utterance=5 + b, n_steps=steps, inputs=inputs,
                                        want_logits=True)
            diff = np.nonzero(r["samples"] != mu[b, :steps])[0]
            assert len(diff) <= 0.001 * steps, (b, len(diff))
            for t in diff:
                pick, sc = oracle.sample_from_logits(r["logits"][t], 21, 5 + b, int(t))
                assert sc[pick] - sc[int(mu[b, t])] <= 2e-5, (b, int(t))
    finally:
        voc.set_option("xcm", -1)
