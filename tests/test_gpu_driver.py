"""-m gpu: the batched, length-bucketed drivers return, per utterance, exactly what a batch-1 call
on that utterance returns (the reference's encode.py / convert.py loops are batch-1)."""
import pytest
import torch

import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import driver, synth

pytestmark = pytest.mark.gpu


def models():
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict())
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(synth.vocoder_state_dict())
    return enc.cuda().eval(), voc.cuda().eval()


def test_bucketing_rules():
    lengths = [40, 41, 300, 44, 256, 258, 36]
    modes = [driver.batch1_conv_mode(80, t) for t in lengths]
    assert modes == [1, 1, 2, 1, 1, 2, 1]
    buckets = driver.make_buckets(lengths, modes, max_batch=3, max_pad_frac=0.25)
    assert sorted(i for b in buckets for i in b) == list(range(7))
    for b in buckets:
        assert len(b) <= 3 and len({modes[i] for i in b}) == 1
        hi = max(lengths[i] for i in b)
        assert hi * len(b) - sum(lengths[i] for i in b) <= 0.25 * hi * len(b)
    assert driver.out_frames(33) == 16 and driver.out_frames(32) == 16 and driver.out_frames(2) == 1


def test_encode_utterances_equals_batch1_calls():
    enc, _ = models()
    Ts = [40, 33, 64, 300, 47, 258, 36]
    mels = [synth.mel(f"drv/{i}", 1, t)[0] for i, t in enumerate(Ts)]
    got = driver.encode_utterances(enc, mels, want_context=True, max_batch=4)
    for i, m in enumerate(mels):
        z, c, idx = enc.encode(m[None].cuda())                # what encode.py:44-46 does
        assert torch.equal(got[i]["indices"], idx[0]) and torch.equal(got[i]["z"], z[0]), i
        assert torch.allclose(got[i]["c"], c[0], atol=1e-6, rtol=0), i


def test_convert_utterances_equals_batch1_calls():
    enc, voc = models()
    Ts = [6, 9, 4, 8]
    mels = [synth.mel(f"cv/{i}", 1, t)[0] for i, t in enumerate(Ts)]
    spk = [3, 50, 7, 101]
    got = driver.convert_utterances(enc, voc, mels, spk, seed=13, max_batch=3, max_pad_frac=0.5)
    got_cb = driver.convert_utterances(enc, voc, mels, spk, seed=13, max_batch=3, slots=2)      # continuous batching
    assert all(torch.equal(a, b) for a, b in zip(got, got_cb))
    for i, m in enumerate(mels):
        idx = enc.encode_indices(m[None].cuda())
        wav = voc.generate(idx, torch.tensor([spk[i]], device="cuda"), seed=13, utt_ids=[i])
        assert got[i].shape == (320 * driver.out_frames(Ts[i]),)
        assert torch.equal(got[i], wav[0]), i


def test_manifest_workload_continuous_batching():
    """BASELINE configs[4]'s shape (bench.py's `manifest` leg) at test size: 96 ragged utterances of 1-4 s through the
    length-bucketed encoder and ONE continuous-batching decode over 48 slots (three tiles; every slot runs two utterances on
    average).  Every waveform has its utterance's length, is finite and in range and differs from its neighbours'; a spread
    of utterances equals the batch-1 calls of convert.py:72-77 bit for bit."""
    import bench
    enc, voc = models()
    frames, spk = bench.synthetic_manifest(96)
    frames = [min(f, 400) for f in frames]                                   # 1-4 s: about 4 M samples in all
    mels = [synth.mel(f"man/{i % 8}", 1, max(frames))[0][:, :f].contiguous() for i, f in enumerate(frames)]
    wavs = driver.convert_utterances(enc, voc, mels, spk, seed=13, max_batch=64, max_pad_frac=0.15, slots=48)
    assert len(wavs) == 96
    for i, w in enumerate(wavs):
        assert w.shape == (320 * driver.out_frames(frames[i]),), i
        assert torch.isfinite(w).all() and float(w.abs().max()) <= 1.0, i
    assert not torch.equal(wavs[0][:16000], wavs[1][:16000])
    ms, steps = voc.last_timing()
    assert steps >= max(int(w.numel()) for w in wavs) and ms > 0
    for i in (0, 17, 48, 95):
        idx = enc.encode_indices(mels[i][None].cuda())
        one = voc.generate(idx, torch.tensor([spk[i]], device="cuda"), seed=13, utt_ids=[i])
        assert torch.equal(wavs[i], one[0]), i


def test_cli_encode_and_convert_end_to_end(tmp_path):
    """encode.py / convert.py equivalents on a tiny synthetic dataset: files written in the
    reference's formats, contents equal to direct calls."""
    import json
    import numpy as np
    from vectorquantizedcpc_amd import cli, io
    ds = tmp_path / "datasets" / "eng"
    (ds / "test").mkdir(parents=True)
    names, Ts = ["S1_a", "S2_b", "S3_c"], [34, 40, 37]
    for n, t in zip(names, Ts):
        np.save(ds / "test" / f"{n}.mel.npy", synth.mel("cli/" + n, 1, t)[0].numpy())
    (ds / "test.json").write_text(json.dumps([["x", 0, 1, f"eng/test/{n}"] for n in names]))
    (ds / "speakers.json").write_text(json.dumps(["V001", "V002"]))
    assert cli.main(["encode", "--dataset", str(ds), "--out-dir", str(tmp_path / "z"), "--random-init"]) == 0
    enc, voc = models()
    for n, t in zip(names, Ts):
        z, _, _ = enc.encode(synth.mel("cli/" + n, 1, t).cuda())
        assert np.array_equal(io.load_frames_text(tmp_path / "z" / n), z[0].cpu().numpy())
    (tmp_path / "list.json").write_text(json.dumps([[f"test/{names[0]}", "V002", "o1"], [f"test/{names[1]}", "V001", "o2"]]))
    assert cli.main(["convert", "--dataset", str(ds), "--synthesis-list", str(tmp_path / "list.json"), "--in-dir", str(ds),
                     "--out-dir", str(tmp_path / "wav"), "--random-init", "--seed", "5"]) == 0
    from scipy.io import wavfile
    sr, w = wavfile.read(tmp_path / "wav" / "o1.wav")
    want = voc.generate(enc.encode_indices(synth.mel("cli/" + names[0], 1, Ts[0]).cuda()), torch.tensor([1], device="cuda"),
                        seed=5, utt_ids=[0])
    assert sr == 16000 and np.array_equal(w, want[0].cpu().numpy())
    # wav input: the mel front-end runs on the GPU (convert.py:54-70)
    tone = (0.2 * np.sin(2 * np.pi * 300 * np.arange(8000) / 16000)).astype(np.float32)
    wavfile.write(str(ds / "test" / "W1.wav"), 16000, tone)
    (tmp_path / "list2.json").write_text(json.dumps([["test/W1", "V001", "o3"]]))
    assert cli.main(["convert", "--dataset", str(ds), "--synthesis-list", str(tmp_path / "list2.json"), "--in-dir", str(ds),
                     "--out-dir", str(tmp_path / "wav"), "--random-init"]) == 0
    sr, w3 = wavfile.read(tmp_path / "wav" / "o3.wav")
    assert w3.shape == (320 * driver.out_frames(1 + 8000 // 160),)
    # convert.py:57,79-80: the output is re-normalised to the input's integrated loudness
    from oracle import loudness_ref
    assert abs(loudness_ref.integrated_loudness(w3, 16000) - loudness_ref.integrated_loudness(tone, 16000)) < 1e-4


def test_ragged_conditioning_allocates_the_utterances_own_frames_and_chunked_decodes_agree():
    """the conditioning rows (Gcond, 3 x 896 fp32 per frame) are computed for every utterance's own frames, not
    for B x T_max; and a decode cut into several calls by the driver's memory budget gives the same waveforms as one call."""
    enc, voc = models()
    n_codes = [60] + [3] * 7                                                   # 120 + 7 x 6 = 162 frames of 8 x 120 = 960
    z = synth.randint("rag/z", (8, 60), 512).cuda()
    spk = synth.randint("rag/s", (8,), 102).cuda()
    wav = voc.generate(z, spk, n_codes=n_codes, seed=13, utt_ids=list(range(8)))
    ws = voc.workspace_bytes()
    own, padded = 162 * driver.BYTES_PER_OWN_FRAME, 960 * 3 * 896 * 4
    assert own <= ws < padded, (own, ws, padded)                              # the padded Gcond alone would be 10.3 MB; everything now: 2.7 MB
    for i in (0, 3, 7):                                                        # ... and every utterance still equals itself decoded alone
        one = voc.generate(z[i:i + 1, : n_codes[i]], spk[i:i + 1], seed=13, utt_ids=[i])
        assert torch.equal(wav[i, : 320 * n_codes[i]], one[0]), i
        assert not bool(wav[i, 320 * n_codes[i]:].any())
    # the driver's chunked continuous batching (a budget that cuts 12 utterances into several calls) == one call
    Ts = [40, 12, 30, 8, 22, 36, 10, 18, 26, 14, 34, 20]
    mels = [synth.mel(f"rag/m{i}", 1, t)[0] for i, t in enumerate(Ts)]
    sp = [i % 102 for i in range(12)]
    whole = driver.convert_utterances(enc, voc, mels, sp, seed=13, slots=4)
    budget = 3 * (40 * driver.BYTES_PER_OWN_FRAME + 40 * (driver.BYTES_PER_PADDED_FRAME + 4 * 160))
    assert len(driver.decode_chunks([driver.out_frames(t) for t in Ts], budget)) >= 3
    parts = driver.convert_utterances(enc, voc, mels, sp, seed=13, slots=4, mem_budget_bytes=budget)
    assert all(torch.equal(a, b) for a, b in zip(whole, parts))


def test_batched_wav_front_end_equals_per_utterance_calls(tmp_path):
    """`cli convert`'s batched front end (driver.front_end_utterances: bucket by sample rate, resample with ceil lengths,
    loudness and log-mel with `lengths=` on padded rows) on wavs of different lengths and rates (16 k and 22.05 k, mono float and
    stereo int16) must give, per utterance, what the one-utterance calls give (io.load_wav + Meter + wave_to_mel), and
    cli.convert_files the outputs of single-utterance runs."""
    import numpy as np
    from scipy.io import wavfile
    from vectorquantizedcpc_amd import cli, io, loudness, preprocess
    enc, voc = models()
    specs = [(16000, 9000, 1, np.float32), (22050, 15000, 2, np.int16), (16000, 12345, 1, np.int16), (22050, 9876, 1, np.float32)]
    paths = []
    for i, (sr, n, ch, dt) in enumerate(specs):
        t = np.arange(n) / sr
        x = 0.3 * np.sin(2 * np.pi * (200 + 90 * i) * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 1.5 * t))
        if ch == 2:
            x = np.stack([x, 0.5 * x], axis=1)
        data = (x * 20000).astype(np.int16) if dt == np.int16 else x.astype(np.float32)
        p = tmp_path / f"in{i}.wav"
        wavfile.write(str(p), sr, data)
        paths.append(str(p)[:-4])
    rates, waves = zip(*[io.read_wav_file(p + ".wav") for p in paths])
    mels, lufs = driver.front_end_utterances(list(waves), list(rates), torch.device("cuda"))
    meter = loudness.Meter(16000)
    for i, p in enumerate(paths):
        w1 = io.load_wav(p + ".wav", 16000)                                       # one utterance: read + resample at load
        w1 = torch.as_tensor(w1, dtype=torch.float32, device="cuda")
        l1 = float(meter.integrated_loudness(w1[None])[0])
        m1 = preprocess.wave_to_mel(w1[None])[0]
        assert mels[i].shape == m1.shape, (i, mels[i].shape, m1.shape)
        assert torch.equal(mels[i], m1), i
        assert abs(lufs[i] - l1) < 1e-9, (i, lufs[i], l1)
    out_all = tmp_path / "all"
    out_all.mkdir()
    entries = [(p, 3 + i, f"o{i}") for i, p in enumerate(paths)]
    cli.convert_files(enc, voc, entries, str(out_all), 13, slots=2)
    for i in range(len(paths)):                        # the files equal one-utterance runs of convert.py:72-83 on the same sampling stream
        idx = enc.encode_indices(m_i := mels[i][None])
        one = voc.generate(idx, torch.tensor([3 + i], device="cuda"), seed=13, utt_ids=[i])
        n = 320 * driver.out_frames(int(m_i.shape[-1]))
        want = loudness.match_loudness([one[0, :n]], [lufs[i]])[0].cpu().numpy()
        sr, got = wavfile.read(str(out_all / f"o{i}.wav"))
        assert sr == 16000 and got.shape == want.shape and np.array_equal(got, want), i
