"""-m gpu: the HIP encoder path through the C ABI against the oracle and the golden vectors.

Bar (north_star): code indices BIT-EXACT vs the reference's PyTorch-CPU path.  We assert more:
every stage's fp32 bit pattern equals the reference's (SHA-256 from tests/golden, full
arrays from the pinned oracle) for calls of >= 16 output frames.  Context `c` (LSTM) is
compared at 1e-6 absolute; loss / perplexity at 1e-6 / 1e-4 relative.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

import oracle
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth

pytestmark = pytest.mark.gpu
CASES = ["c1_init", "c2_init", "c2_random_data", "ragged_3x32", "tiny_1x16", "odd_2x33", "long_1x300",
         "edge_1x32", "edge_1x34", "edge_1x62", "edge_2x16"]      # edge_*: 16..31 rows, the edge of the bit-exact contract


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_models = {}


def encoder_for(ln_affine, codebook):
    key = (ln_affine, codebook)
    if key not in _models:
        sd = synth.encoder_state_dict(ln_affine=ln_affine, codebook=codebook)
        enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
        enc.load_state_dict(sd)
        _models[key] = (enc.to("cuda").eval(), sd)
    return _models[key]


def load_case(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"encoder_{name}.npz"))
    B, T = (int(v) for v in g["case"])
    enc, sd = encoder_for(str(g["ln_affine"]), str(g["codebook"]))
    mel = synth.mel(name, B, T)
    return g, enc, sd, mel


@pytest.mark.parametrize("name", CASES)
def test_indices_bit_exact_vs_reference(name, golden_dir):
    g, enc, _, mel = load_case(name, golden_dir)
    z, c, idx = enc.encode(mel.cuda())
    assert idx.dtype == torch.int64 and idx.shape == (mel.shape[0], (mel.shape[2] - 2) // 2 + 1)
    got = idx.cpu().numpy()
    want = g["indices"].astype(np.int64)
    bad = np.nonzero(got != want)
    margin = (g["d_second"] - g["d_best"]).reshape(want.shape)
    assert bad[0].size == 0, f"{bad[0].size} index mismatches; reference margins there: {margin[bad][:8]}"
    assert z.shape == (mel.shape[0], (mel.shape[2] - 2) // 2 + 1, 64) and c.shape[-1] == 256


@pytest.mark.parametrize("name", [c for c in CASES if c != "tiny_1x16"])
def test_every_stage_bit_exact(name, golden_dir):
    g, enc, sd, mel = load_case(name, golden_dir)
    melc = mel.cuda()
    x = oracle.conv1d_k4s2(mel.numpy(), sd["conv.weight"].numpy()).reshape(-1, 512)
    report = {}

    def check(stage, want, tag):
        got = enc.stage(melc, stage).cpu().numpy().reshape(want.shape)
        report[tag] = int((got.view(np.uint32) != want.view(np.uint32)).sum())

    check(0, x, "conv")
    assert sha(x.reshape(mel.shape[0], -1, 512)) == str(g["sha_conv"])
    x = np.maximum(oracle.layernorm(x, sd["encoder.0.weight"], sd["encoder.0.bias"], relu=False), 0)
    check(1, x, "ln0")
    for l, (lin, ln) in enumerate(((2, 3), (5, 6), (8, 9), (11, 12))):
        y = oracle.linear(x, sd[f"encoder.{lin}.weight"].numpy())
        check(2 + 2 * l, y, f"fc{lin}")
        x = np.maximum(oracle.layernorm(y, sd[f"encoder.{ln}.weight"], sd[f"encoder.{ln}.bias"], relu=False), 0)
        check(3 + 2 * l, x, f"ln{ln}")
    zp = oracle.linear(x, sd["encoder.14.weight"].numpy(), sd["encoder.14.bias"].numpy())
    check(10, zp, "z_pre")
    assert all(v == 0 for v in report.values()), f"bitwise mismatches per stage: {report}"
    z, _, _ = enc.encode(melc)
    assert sha(zp.reshape(mel.shape[0], -1, 64)) == str(g["sha_z_pre"])
    assert sha(z.cpu().numpy()) == str(g["sha_z"])


@pytest.mark.parametrize("name", CASES)
def test_context_and_forward_stats(name, golden_dir):
    g, enc, sd, mel = load_case(name, golden_dir)
    melc = mel.cuda()
    _, c, _ = enc.encode(melc)
    want_c = oracle.encoder_encode(sd, mel.numpy())["c"]
    assert np.abs(c.cpu().numpy() - want_c).max() <= 1e-6
    zf, cf, loss, ppl = enc(melc)
    rows = lambda a: a.reshape(-1, a.shape[-1])[:: max(1, a.reshape(-1, a.shape[-1]).shape[0] // 4)][:4]
    assert np.abs(rows(zf.cpu().numpy()) - g["rows_z_fwd"]).max() <= 1e-6
    assert np.abs(rows(cf.cpu().numpy()) - g["rows_c_fwd"]).max() <= 1e-6
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * max(1.0, float(g["loss"]))
    assert abs(float(ppl) - float(g["perplexity"])) <= 1e-4 * float(g["perplexity"])


def test_single_utterance_context_resident_scan_same_bits():
    """encode.py:42-46 encodes ONE utterance at a time: its context LSTM (model.py:57) then runs as one resident kernel
    (32 workgroups on one XCD, in-kernel exchanges of h_t) instead of one launch per time step.  Same bits as the launches,
    for 1, 7 and 100 time steps; <= 1e-6 against the oracle; a second call on the same handle gives the same again."""
    enc, sd = encoder_for("init", "init")
    for T in (2, 14, 200):
        mel = synth.mel("ctx1/%d" % T, 1, T)
        melc = mel.cuda()
        _, c_res, _ = enc.encode(melc)
        _, c_res2, _ = enc.encode(melc)
        try:
            enc.set_option("persistent_context", 2)      # the fallback: agent-scope stores (workers not on one XCD)
            _, c_agent, _ = enc.encode(melc)
            enc.set_option("persistent_context", 0)
            _, c_steps, _ = enc.encode(melc)
        finally:
            enc.set_option("persistent_context", 1)
        assert c_res.shape == (1, T // 2, 256)
        assert torch.equal(c_res, c_steps) and torch.equal(c_res, c_res2) and torch.equal(c_res, c_agent), T
        want = oracle.encoder_encode(sd, mel.numpy())["c"]
        assert np.abs(c_res.cpu().numpy() - want).max() <= 1e-6, T


def test_both_conv_backends_match_oracle():
    enc, sd = encoder_for("init", "init")
    mel = synth.mel("convmodes", 1, 64)
    for mode in (1, 2):
        want = oracle.conv1d_k4s2(mel.numpy(), sd["conv.weight"].numpy(), mode=mode)
        got = enc.stage(mel.cuda(), 0, conv_mode=mode).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), mode


@pytest.mark.parametrize("n_emb", [64, 192, 320, 1024])
def test_other_codebook_sizes_match_oracle(n_emb):
    """VQ kernel paths the 512-code fixtures do not reach: one 16-code tile per wave (64), odd tile counts
    (192, 320), many tiles (1024); 21 rows (not a multiple of the kernel's 16-row block).  Checked bit-exact
    against the pinned C oracle (a data-scale codebook, so the argmin is well separated from ties)."""
    sd = synth.encoder_state_dict(n_embeddings=n_emb, ln_affine="random", codebook="data")
    enc = V.Encoder(V.ConfEncoder(80, 512, n_emb, 64, 256))
    enc.load_state_dict(sd)
    enc = enc.cuda().eval()
    mel = synth.mel("cb%d" % n_emb, 3, 14)                     # 3 x 7 = 21 rows
    want = oracle.encoder_encode(sd, mel.numpy(), want_c=False, conv_mode=2)
    z, _, idx = enc.encode(mel.cuda())
    assert idx.shape == (3, 7) and int(idx.max()) < n_emb
    assert np.array_equal(idx.cpu().numpy().ravel(), want["indices"].ravel())
    assert np.array_equal(z.cpu().numpy().reshape(-1, 64).view(np.uint32), want["z"].reshape(-1, 64).view(np.uint32))


def test_hook_and_error_surface():
    enc, _ = encoder_for("init", "init")
    seen = []
    h = enc.encoder[-1].register_forward_hook(lambda m, i, o: seen.append(o.clone()))   # encode.py:34-40
    z, c, idx = enc.encode(synth.mel("hook", 2, 40).cuda())
    h.remove()
    assert len(seen) == 1 and seen[0].shape == (2, 20, 64)
    q = enc.codebook.embedding[idx]                      # F.embedding gather (model.py:113)
    assert torch.equal(q, z)
    z2, _, idx2 = enc.encode(synth.mel("hook", 2, 41).cuda())   # odd T: floor((T-2)/2)+1 frames
    assert idx2.shape == (2, 20)
    with pytest.raises(RuntimeError):
        enc.encode(torch.zeros(1, 40, 32, device="cuda"))
    enc.train()
    with pytest.raises(NotImplementedError):
        enc(synth.mel("hook", 1, 32).cuda())
    enc.eval()


def test_full_size_properties():
    """C2 size: idempotence, batch invariance under the batched conv order, gather identity."""
    enc, _ = encoder_for("init", "init")
    mel = synth.mel("c2_init", 64, 128).cuda()
    z1, c1, i1 = enc.encode(mel)
    z2, c2, i2 = enc.encode(mel)
    assert torch.equal(i1, i2) and torch.equal(z1, z2) and torch.equal(c1, c2)
    for b in (0, 17, 63):
        zb, _, ib = enc.encode(mel[b:b + 1], conv_mode=2)
        assert torch.equal(ib[0], i1[b]) and torch.equal(zb[0], z1[b])
    assert torch.equal(enc.encode_indices(mel), i1)


@pytest.mark.parametrize("name", ["c2_init", "odd_2x33", "c1_init", "c2_random_data"])
def test_layered_and_fused_front_ends_same_bits(name, golden_dir):
    """The one-launch fused front end (1), the six column-split launches for small calls (2) and the layered kernels
    (0, one launch per module) are three schedules of the same rounding sequence: every stage, z, z_pre and the indices
    carry the same bits, and they equal the reference's SHAs.  (Default: 2 up to 80 row tiles, else 1.)"""
    g, enc, sd, mel = load_case(name, golden_dir)
    melc = mel.cuda()
    outs = {}
    try:
        for fused in (1, 2, 0):
            enc.set_option("fused", fused)
            z, c, idx = enc.encode(melc)
            stages = [enc.stage(melc, s) for s in range(11)]
            outs[fused] = (z, idx, stages)
    finally:
        enc.set_option("fused", -1)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[2][1], outs[1][1])
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[2][0].view(torch.int32), outs[1][0].view(torch.int32))
    for s in range(11):
        assert torch.equal(outs[0][2][s].view(torch.int32), outs[1][2][s].view(torch.int32)), s
    assert sha(outs[1][0].cpu().numpy()) == str(g["sha_z"])
    assert sha(outs[1][2][10].cpu().numpy()) == str(g["sha_z_pre"])
    assert sha(outs[1][2][0].cpu().numpy()) == str(g["sha_conv"])


def test_vq_embedding_encode_and_forward_standalone(golden_dir):
    """VQEmbeddingEMA.encode / eval forward called on their own (model.py:103-155) equal what Encoder.encode /
    Encoder.forward compute from the same pre-VQ rows."""
    g, enc, sd, mel = load_case("c2_init", golden_dir)
    melc = mel.cuda()
    z, _, idx = enc.encode(melc)
    z_pre = enc.stage(melc, 10)
    q, i2 = enc.codebook.encode(z_pre)
    assert torch.equal(i2, idx) and torch.equal(q, z) and i2.shape == idx.shape
    zf, _, loss, ppl = enc(melc)
    q2, loss2, ppl2 = enc.codebook(z_pre)
    assert torch.equal(q2, zf) and float(loss2) == float(loss) and float(ppl2) == float(ppl)
    with pytest.raises(RuntimeError):
        enc.codebook.encode(z_pre.cpu())
    w = enc.encoder[14].weight                         # (a LayerNorm follows every other Linear and would undo a scale)
    w.data.mul_(2.0)                                   # a write through .data bumps no version: refresh() re-reads
    try:
        assert torch.equal(enc.encode(melc)[2], idx)
        enc.refresh()
        assert not torch.equal(enc.encode(melc)[2], idx)
    finally:
        w.data.mul_(0.5)
        enc.refresh()
    assert torch.equal(enc.encode(melc)[2], idx)
