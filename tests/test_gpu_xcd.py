"""-m gpu: the per-XCD resident decoders (csrc/ar_xcd.hip, the default decode path up to 64 utterances in flight) and the
error path of every in-kernel hand-off (vqcpc_vocoder_check / vqcpc_encoder_check).

Checked three ways: (1) draw by draw against the C oracle on the same history (the parity-unpinned self-oracle of
test_gpu_vocoder.py, tolerance 2e-5 on the Gumbel-max score); (2) bit for bit against the launch-per-step kernels
(`xcd` = 0), which test_gpu_vocoder.py checks against the oracles on their own; (3) size-independent properties at the
BASELINE sizes (configs[2]: 1 x 32 000; configs[3]'s per-GPU shard: 32 x 32 000): an utterance alone equals itself in the
batch, continuous batching through few slots equals one slot per utterance.
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
    for mode in (1, 0):
        voc.set_option("xcd", mode)
        wav, mu = voc.generate(z, spk, return_mulaw=True, **kw)
        voc.check()
        out[mode] = (wav.cpu(), mu.cpu())
    voc.set_option("xcd", -1)
    return out


@pytest.mark.parametrize("B,Tc,ragged", [(1, 3, False), (2, 2, False), (3, 4, True), (4, 2, False), (9, 3, True), (17, 2, True),
                                         (32, 2, False)])
def test_same_samples_as_the_launch_path(B, Tc, ragged):
    voc, _ = vocoder()
    z = synth.randint(f"xcd/z{B}", (B, Tc), 512).cuda()
    spk = synth.randint(f"xcd/s{B}", (B,), 102).cuda()
    n_codes = [max(1, Tc - (b % Tc)) for b in range(B)] if ragged else None
    out = _both_paths(voc, z, spk, n_codes=n_codes, seed=13, utt_base=3)
    assert torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][0], out[0][0])
    assert int((out[1][1] != 0).sum()) > 0.9 * 320 * sum(n_codes or [Tc] * B)


@pytest.mark.parametrize("B", [5, 20])
def test_draw_by_draw_against_the_oracle(B):
    """B = 5: one slot per XCD (fp32 vector chains); B = 20: up to three slots per XCD = the four-slot kernel, whose W_hh / fc1 chains run
    on the matrix pipe (v_mfma_f32_4x4x1_16B_f32, ar_chain.h) with one column idle."""
    voc, sd = vocoder()
    voc.set_option("xcd", 1)
    try:
        Tc, steps = 2, 480
        z = synth.randint("xcd/oz", (B, Tc), 512)
        spk = synth.randint("xcd/os", (B,), 102)
        wav, mu = voc.generate(z.cuda(), spk.cuda(), seed=13, utt_base=11, return_mulaw=True, max_steps=steps)
        voc.check()
        wav, mu = wav.cpu().numpy(), mu.cpu().numpy()
        exact = total = 0
        for b in range(B):
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
        voc.set_option("xcd", -1)


def test_more_utterances_than_slots_and_alone_equals_in_batch():
    """Continuous batching: 40 ragged utterances through 8 decode slots (one per XCD) and through all 32; every utterance
    equals the same utterance decoded alone (its sampling stream does not depend on the placement)."""
    voc, _ = vocoder()
    n = 40
    Tc = 3
    z = synth.randint("xcd/cz", (n, Tc), 512).cuda()
    spk = synth.randint("xcd/cs", (n,), 102).cuda()
    n_codes = [1 + (7 * i) % Tc for i in range(n)]
    ids = list(range(100, 100 + n))
    voc.set_option("xcd", 1)
    try:
        res = {}
        for slots in (8, 32):
            voc.set_option("xcd_slots", slots)
            wav, mu = voc.generate(z, spk, n_codes=n_codes, seed=5, utt_ids=ids, return_mulaw=True)
            voc.check()
            res[slots] = (wav.cpu(), mu.cpu())
        assert torch.equal(res[8][1], res[32][1]) and torch.equal(res[8][0], res[32][0])
        for i in (0, 13, 39):
            w1, m1 = voc.generate(z[i:i + 1, :n_codes[i]], spk[i:i + 1], seed=5, utt_ids=[ids[i]], return_mulaw=True)
            L = 320 * n_codes[i]
            assert torch.equal(m1[0].cpu(), res[8][1][i, :L]) and torch.equal(w1[0].cpu(), res[8][0][i, :L])
            assert not res[8][1][i, L:].any()
    finally:
        voc.set_option("xcd_slots", 32)
        voc.set_option("xcd", -1)


def test_agent_scope_stores_same_bits():
    voc, _ = vocoder()
    z = synth.randint("xcd/az", (6, 2), 512).cuda()
    spk = synth.randint("xcd/as", (6,), 102).cuda()
    voc.set_option("xcd", 1)
    try:
        a = voc.generate(z, spk, seed=2, utt_base=0, return_mulaw=True)
        voc.set_option("xcd_agent_stores", 1)
        b = voc.generate(z, spk, seed=2, utt_base=0, return_mulaw=True)
        voc.check()
        assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
    finally:
        voc.set_option("xcd_agent_stores", 0)
        voc.set_option("xcd", -1)


def test_baseline_sizes_properties():
    """configs[2] (1 x 32 000) and configs[3]'s per-GPU shard (32 x 32 000) at full size on the default path: finite, in
    range, and three rows of the shard equal the same utterances decoded alone."""
    voc, _ = vocoder()
    z = synth.randint("xcd/bz", (32, 100), 512).cuda()
    spk = (torch.arange(32, device="cuda") % 102)
    wav, mu = voc.generate(z, spk, seed=13, utt_base=0, return_mulaw=True)
    voc.check()
    assert wav.shape == (32, 32000) and torch.isfinite(wav).all() and wav.abs().max() <= 1.0
    assert mu.min() >= 0 and mu.max() <= 255 and mu.unique().numel() > 32
    ms, n = voc.last_timing()
    assert n == 32000 and ms > 0
    print("32 x 32000 on the per-XCD decoders: %.2f us per sample step" % (ms * 1e3 / n))
    for i in (0, 17, 31):
        w1, m1 = voc.generate(z[i:i + 1], spk[i:i + 1], seed=13, utt_base=i, return_mulaw=True)
        voc.check()
        assert torch.equal(m1[0], mu[i]) and torch.equal(w1[0], wav[i])


# ------------------------------------------------------------------ the error path of the in-kernel hand-offs


def _reference_bits(z, spk, **kw):
    ref, _ = vocoder(fresh=True)
    ref.set_option("xcd", 0)
    ref.set_option("fuse_fc2", 0)
    wav, mu = ref.generate(z, spk, return_mulaw=True, **kw)
    ref.check()
    return wav.cpu(), mu.cpu()


@pytest.mark.parametrize("B", [1, 12, 20])
def test_xcd_handoff_timeout_is_reported_by_the_same_call_and_the_rerun_is_right(B):
    """One worker skips a candidate publish: every wait behind it gives up after the (shortened) deadline, check() raises
    for THAT call, the handle falls back to launches, and the repeated call gives the samples of the undisturbed paths."""
    voc, _ = vocoder(fresh=True)
    z = synth.randint("xcd/ez", (B, 2), 512).cuda()
    spk = synth.randint("xcd/es", (B,), 102).cuda()
    voc.set_option("xcd", 1)
    voc.set_option("xcd_timeout_ms", 20)
    voc.set_option("xcd_debug_drop_step", 200)
    wav, mu = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True, async_=True)      # only enqueued: the caller checks
    with pytest.raises(RuntimeError, match=r"timed out.*call #1 of this handle|call #1 of this handle.*timed out"):
        voc.check()
    assert not bool((mu[0, 260:] != 0).any())            # the call that suffered it is incomplete (and said so)
    voc.set_option("xcd_debug_drop_step", -1)
    wav2, mu2 = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True)      # the handle has fallen back
    assert voc.last_path() == 0
    want = _reference_bits(z, spk, seed=9, utt_base=0)
    assert torch.equal(mu2.cpu(), want[1]) and torch.equal(wav2.cpu(), want[0])
    # Vocoder.generate itself (the reference's call, convert.py:77) checks and repeats once: the drop-in caller gets the right bits
    voc3, _ = vocoder(fresh=True)
    voc3.set_option("xcd", 1)
    voc3.set_option("xcd_timeout_ms", 20)
    voc3.set_option("xcd_debug_drop_step", 100)
    with pytest.warns(UserWarning, match="decode repeated"):
        wav3 = voc3.generate(z, spk, seed=9, utt_base=0)
    assert torch.equal(wav3.cpu(), want[0])
    with pytest.warns(UserWarning, match="decode repeated"):
        voc3.set_option("xcd", 1)
        wav4 = driver.generate_checked(voc3, z, spk, seed=9, utt_base=0)
    assert torch.equal(wav4.cpu(), want[0])


def test_fused_launch_handoff_timeout_is_reported_and_the_rerun_is_right():
    """The fused fc2 || GRU launch (launch path, 32 utterances): one fc2 team skips its publish."""
    voc, _ = vocoder(fresh=True)
    z = synth.randint("xcd/fz", (32, 2), 512).cuda()
    spk = synth.randint("xcd/fs", (32,), 102).cuda()
    voc.set_option("xcd", 0)
    voc.set_option("handoff_timeout_ms", 20)
    voc.set_option("handoff_debug_drop_step", 150)
    voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True, async_=True)
    with pytest.raises(RuntimeError, match="timed out"):
        voc.check()
    voc.set_option("handoff_debug_drop_step", -1)
    wav2, mu2 = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True)
    voc.check()
    want = _reference_bits(z, spk, seed=9, utt_base=0)
    assert torch.equal(mu2.cpu(), want[1]) and torch.equal(wav2.cpu(), want[0])


def test_resident_context_scan_timeout_is_reported_and_the_rerun_is_right():
    """encode.py:42-46's one-utterance call: the resident LSTM scan loses one publish."""
    esd = synth.encoder_state_dict()
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(esd)
    enc = enc.to("cuda").eval()
    mel = synth.mel("xcd/ctx", 1, 64).cuda()
    enc.set_option("persistent_context", 0)
    want = enc.encode(mel)[1].cpu()
    enc.set_option("persistent_context", 1)
    enc.set_option("context_timeout_ms", 20)
    enc.set_option("context_debug_drop_step", 10)
    enc.encode(mel)
    with pytest.raises(RuntimeError, match="timed out"):
        enc.check()
    enc.set_option("context_debug_drop_step", -1)
    c2 = enc.encode(mel)[1]
    enc.check()
    assert torch.equal(c2.cpu(), want)


# ------------------------------------------------------------------ placement misses, re-arming, pipelined callers


@pytest.mark.parametrize("B", [4, 80])
def test_placement_miss_writes_nothing_is_reported_and_is_not_latched(B):
    """A shared GPU's first call: the 256 workgroups are not dealt 32 per XCD (here: workgroup 0 reports the wrong XCC_ID).
    The launch gives up before it touches any output (status 2), check() says so for THAT call, the handle keeps its
    options, and the repeat runs on the resident decoders again and gives the reference bits."""
    voc, _ = vocoder(fresh=True)
    z = synth.randint("xcd/pz", (B, 2), 512).cuda()
    spk = synth.randint("xcd/ps", (B,), 102).cuda()
    want_path = 2 if B <= 68 else 3
    voc.set_option("xcd_timeout_ms", 50)
    voc.set_option("xcd_debug_misplace", 1)
    wav, mu = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True, async_=True)
    with pytest.raises(RuntimeError, match="not dealt 32"):
        voc.check()
    assert not bool(mu.any()) and not bool(wav.any())                 # nothing was written
    wav2, mu2 = voc.generate(z, spk, seed=9, utt_base=0, return_mulaw=True)     # the misplacement was one shot; nothing latched
    assert voc.last_path() == want_path
    want = _reference_bits(z, spk, seed=9, utt_base=0)
    assert torch.equal(mu2.cpu(), want[1]) and torch.equal(wav2.cpu(), want[0])
    # the default call repeats on its own
    voc.set_option("xcd_debug_misplace", 1)
    with pytest.warns(UserWarning, match="not dealt 32"):
        wav3 = voc.generate(z, spk, seed=9, utt_base=0)
    assert voc.last_path() == want_path and torch.equal(wav3.cpu(), want[0])
    # two misses in a row: the handle gives the resident decoders up (and says so)
    voc.set_option("xcd_debug_misplace", 1)
    voc.generate(z, spk, seed=9, utt_base=0, async_=True)
    with pytest.raises(RuntimeError, match="not dealt 32"):
        voc.check()
    voc.set_option("xcd_debug_misplace", 1)
    voc.generate(z, spk, seed=9, utt_base=0, async_=True)
    with pytest.raises(RuntimeError, match="twice in a row"):
        voc.check()
    wav5 = voc.generate(z, spk, seed=9, utt_base=0)
    assert voc.last_path() == 0 and torch.equal(wav5.cpu(), want[0])


def test_a_timeout_latches_and_the_handle_rearms_after_clean_calls():
    voc, _ = vocoder(fresh=True)
    z = synth.randint("xcd/rz", (4, 1), 512).cuda()
    spk = synth.randint("xcd/rs", (4,), 102).cuda()
    voc.set_option("xcd_timeout_ms", 20)
    voc.set_option("xcd_debug_drop_step", 50)
    voc.generate(z, spk, seed=9, utt_base=0, async_=True)
    with pytest.raises(RuntimeError, match="re-arms after 16 clean calls"):
        voc.check()
    voc.set_option("xcd_debug_drop_step", -1)
    want = None
    for i in range(16):
        wav = voc.generate(z, spk, seed=9, utt_base=0)
        assert voc.last_path() == 0, i
        want = wav if want is None else want
        assert torch.equal(wav, want)
    wav = voc.generate(z, spk, seed=9, utt_base=0)                # the 17th call: resident decoders again, same bits
    assert voc.last_path() == 2 and torch.equal(wav, want)


def test_pipelined_calls_the_status_word_names_the_call():
    """Three calls enqueued without a synchronisation, the second one loses a publish: the check behind the sync names call #2;
    the third call (which saw a clean word when it was enqueued) is not blamed."""
    voc, _ = vocoder(fresh=True)
    z = synth.randint("xcd/qz", (4, 1), 512).cuda()
    spk = synth.randint("xcd/qs", (4,), 102).cuda()
    voc.set_option("xcd_timeout_ms", 20)
    voc.generate(z, spk, seed=9, utt_base=0, async_=True)
    voc.set_option("xcd_debug_drop_step", 50)
    voc.generate(z, spk, seed=9, utt_base=0, async_=True)
    voc.set_option("xcd_debug_drop_step", -1)
    try:
        voc.generate(z, spk, seed=9, utt_base=0, async_=True)     # may already see the word (then it raises for call #2 itself)
    except RuntimeError as e:
        assert "call #2 of this handle" in str(e)
    else:
        with pytest.raises(RuntimeError, match="call #2 of this handle"):
            voc.check()


# ------------------------------------------------------------------ full-size equalities and long oracle runs


def test_32_x_32000_bit_equal_to_the_launch_path():
    """BASELINE configs[3]'s per-GPU shard at full size: the per-XCD decoders and the launch-per-step kernels give the same
    1 024 000 samples (one run each)."""
    voc, _ = vocoder()
    z = synth.randint("xcd/fz32", (32, 100), 512).cuda()
    spk = (torch.arange(32, device="cuda") % 102)
    out = _both_paths(voc, z, spk, seed=13, utt_base=0)
    assert out[1][1].shape == (32, 32000)
    assert torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][0], out[0][0])


def test_2400_consecutive_steps_draw_by_draw_against_the_oracle():
    """A late-step divergence (tag wrap, conditioning-frame index, slot hand-over) cannot hide behind 400-step windows: 2 400
    consecutive steps (15 conditioning frames) of two utterances inside a batch of 9 on the per-XCD decoders, every draw
    checked against the C oracle on the same history."""
    voc, sd = vocoder()
    voc.set_option("xcd", 1)
    try:
        B, Tc, steps = 9, 8, 2400
        z = synth.randint("xcd/lz", (B, Tc), 512)
        spk = synth.randint("xcd/ls", (B,), 102)
        wav, mu = voc.generate(z.cuda(), spk.cuda(), seed=21, utt_base=5, return_mulaw=True, max_steps=steps)
        assert voc.last_path() == 2
        mu = mu.cpu().numpy()
        for b in (0, 8):
            inputs = np.concatenate([[128], mu[b, :steps - 1]])
            r = oracle.vocoder_generate(sd, z[b].numpy(), int(spk[b]), seed=21, utterance=5 + b, n_steps=steps, inputs=inputs,
                                        want_logits=True)
            diff = np.nonzero(r["samples"] != mu[b, :steps])[0]
            assert len(diff) <= 0.001 * steps, (b, len(diff))
            for t in diff:
                pick, sc = oracle.sample_from_logits(r["logits"][t], 21, 5 + b, int(t))
                assert sc[pick] - sc[int(mu[b, t])] <= 2e-5, (b, int(t))
    finally:
        voc.set_option("xcd", -1)
