"""File formats of the reference's CLIs (encode.py / convert.py), CPU side."""
import json

import numpy as np
import torch

from vectorquantizedcpc_amd import io, synth


def test_frames_text_is_the_reference_format(tmp_path):
    z = synth.mel("io", 1, 7, n_mels=64)[0].T.contiguous()          # (7, 64) float32
    io.save_frames_text(tmp_path / "utt", z)
    ref = tmp_path / "ref.txt"
    with open(ref, "w") as f:
        np.savetxt(f, z.numpy(), fmt="%.16f")                       # encode.py:50-52 verbatim call
    assert (tmp_path / "utt.txt").read_bytes() == ref.read_bytes()
    assert np.array_equal(io.load_frames_text(tmp_path / "utt"), z.numpy())   # %.16f round-trips float32


def test_mel_wav_and_metadata(tmp_path):
    mel = synth.mel("io2", 1, 33)[0]
    np.save(tmp_path / "a.mel.npy", mel.numpy())
    assert torch.equal(io.load_mel(tmp_path / "a"), mel) and torch.equal(io.load_mel(tmp_path / "a.mel.npy"), mel)
    wav = torch.linspace(-1, 1, 1600)
    io.save_wav(tmp_path / "o", wav)
    from scipy.io import wavfile
    sr, back = wavfile.read(tmp_path / "o.wav")
    assert sr == 16000 and back.dtype == np.float32 and np.array_equal(back, wav.numpy())
    ds = tmp_path / "datasets" / "eng"
    ds.mkdir(parents=True)
    (ds / "test.json").write_text(json.dumps([["x", 0, 1, "eng/test/S1_a"], ["y", 0, 1, "eng/test/S2_b"]]))
    (ds / "speakers.json").write_text(json.dumps(["V002", "V001"]))
    assert [p.name for p in io.read_test_metadata(ds)] == ["S1_a", "S2_b"]
    assert io.read_test_metadata(ds)[0].parent == tmp_path / "datasets" / "eng" / "test"
    (tmp_path / "list.json").write_text(json.dumps([["eng/test/S1_a", "V002", "out1"]]))
    items, speakers = io.read_synthesis_list(tmp_path / "list.json", ds / "speakers.json")
    assert speakers == ["V001", "V002"] and items == [("eng/test/S1_a", 1, "out1")]


def test_checkpoint_layouts(tmp_path):
    sd = synth.encoder_state_dict()
    torch.save({"encoder": sd, "epoch": 3}, tmp_path / "cpc.pt")                 # train_cpc.py:23-29
    got = io.load_encoder_checkpoint(tmp_path / "cpc.pt")
    assert list(got.keys()) == list(sd.keys()) and torch.equal(got["conv.weight"], sd["conv.weight"])
    vsd = synth.vocoder_state_dict()
    torch.save({"state_dict": {**{"model." + k: v for k, v in vsd.items()}, "encoder.conv.weight": sd["conv.weight"]}},
               tmp_path / "last.ckpt")                                            # Lightning layout (vocoder.py:47-48)
    got = io.load_vocoder_checkpoint(tmp_path / "last.ckpt")
    assert set(got.keys()) == set(vsd.keys())
    torch.save({"vocoder": vsd}, tmp_path / "voc.pt")                             # convert.py:45
    assert set(io.load_vocoder_checkpoint(tmp_path / "voc.pt").keys()) == set(vsd.keys())


def test_vocoder_checkpoint_keys_are_matched_by_suffix_and_shape(tmp_path):
    """The `rnnms.*` key names are this project's guess at the absent third-party module's layout: a checkpoint
    whose module nests the GRUs differently still loads (unique suffix + shape), and one that does not fit
    raises with expected-vs-found lists instead of loading partially."""
    import pytest
    vsd = synth.vocoder_state_dict()
    renamed = {}
    for k, v in vsd.items():
        k2 = k.replace("rnnms.prenet.", "rnnms.prenet.net.").replace("rnnms.ar.rnn.", "rnnms.decoder.gru.")
        k2 = k2.replace("rnnms.ar.fc1", "rnnms.decoder.fc1").replace("rnnms.ar.fc2", "rnnms.decoder.fc2")
        renamed[k2.replace("rnnms.ar.embedding", "rnnms.decoder.embedding")] = v
    assert set(renamed) != set(vsd)
    torch.save({"vocoder": renamed}, tmp_path / "voc2.pt")
    got = io.load_vocoder_checkpoint(tmp_path / "voc2.pt", expected=vsd)
    assert list(got.keys()) == list(vsd.keys()) and all(torch.equal(got[k], vsd[k]) for k in vsd)
    bad = dict(renamed)
    bad.pop(next(k for k in bad if k.endswith("fc1.bias")))
    bad["rnnms.something.else"] = torch.zeros(3)
    with pytest.raises(KeyError) as ei:
        io.remap_state_dict(bad, vsd)
    assert "rnnms.ar.fc1.bias" in str(ei.value) and "rnnms.something.else" in str(ei.value)


def test_wav_decode_scales_integers_before_the_mono_mean(tmp_path):
    """librosa.load(mono=True) / soundfile semantics (convert.py:54-56): int16 / 2^15, int32 / 2^31, uint8 (a - 128) / 128, THEN
    the channel mean.  (Averaging first turned a stereo int16 file into unscaled floats around +-32768: ADVICE r2.)"""
    from scipy.io import wavfile
    rng = np.random.default_rng(3)
    st16 = rng.integers(-20000, 20000, size=(4000, 2), dtype=np.int16)
    wavfile.write(str(tmp_path / "st16.wav"), 16000, st16)
    rate, a = io.read_wav_file(tmp_path / "st16")
    assert rate == 16000 and a.dtype == np.float32 and a.shape == (4000,)
    assert np.allclose(a, st16.astype(np.float64).mean(axis=1) / 32768.0, atol=1e-7) and np.abs(a).max() < 1.0
    assert torch.equal(io.load_wav(tmp_path / "st16"), torch.from_numpy(a))
    m32 = (rng.integers(-2 ** 30, 2 ** 30, size=3000)).astype(np.int32)
    wavfile.write(str(tmp_path / "m32.wav"), 16000, m32)
    assert np.allclose(io.read_wav_file(tmp_path / "m32")[1], m32 / 2.0 ** 31, atol=1e-7)
    u8 = rng.integers(0, 256, size=(2000, 2), dtype=np.uint8)
    wavfile.write(str(tmp_path / "u8.wav"), 16000, u8)
    got = io.read_wav_file(tmp_path / "u8")[1]
    assert np.allclose(got, ((u8.astype(np.float64) - 128.0) / 128.0).mean(axis=1), atol=1e-7) and got.min() >= -1.0 and got.max() < 1.0
    f32 = rng.uniform(-0.5, 0.5, size=(1000, 2)).astype(np.float32)
    wavfile.write(str(tmp_path / "f32.wav"), 22050, f32)
    rate, a = io.read_wav_file(tmp_path / "f32")
    assert rate == 22050 and np.allclose(a, f32.mean(axis=1), atol=1e-7)
