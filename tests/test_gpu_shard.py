"""-m gpu: the sharded convert path end to end with the real HIP kernels -- two ranks (both on this
box's one GPU, gloo for the exchange since RCCL wants one device per rank) shard five utterances,
decode their own and gather on rank 0; the result equals a single-process run bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _models():
    import vectorquantizedcpc_amd as V
    from vectorquantizedcpc_amd import synth
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict())
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(synth.vocoder_state_dict())
    return enc.cuda().eval(), voc.cuda().eval()


def _inputs():
    from vectorquantizedcpc_amd import synth
    Ts = [10, 6, 8, 4, 7]
    return [synth.mel(f"shard/{i}", 1, t)[0] for i, t in enumerate(Ts)], [5, 17, 33, 80, 101]


def _decode_fn(enc, voc):
    from vectorquantizedcpc_amd import driver

    def fn(ids, mels, speakers):
        wavs = driver.convert_utterances(enc, voc, mels, speakers, seed=13, utt_ids=ids, max_batch=4, max_pad_frac=0.5)
        L = max((w.numel() for w in wavs), default=0)
        out = torch.zeros(len(wavs), L)
        for k, w in enumerate(wavs):
            out[k, : w.numel()] = w.cpu()
        return out                       # CPU tensor: gloo carries the gather in this test
    return fn


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vectorquantizedcpc_amd import shard
        enc, voc = _models()
        mels, spk = _inputs()
        res = shard.convert_sharded(mels, spk, _decode_fn(enc, voc))
        ret.put((rank, None if res is None else [r.clone() for r in res]))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_convert_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    assert got[1] is None and len(got[0]) == 5
    enc, voc = _models()
    mels, spk = _inputs()
    want = _decode_fn(enc, voc)(list(range(5)), mels, spk)
    from vectorquantizedcpc_amd import driver
    for i, m in enumerate(mels):
        n = 320 * driver.out_frames(m.shape[-1])
        assert got[0][i].shape == (n,) and torch.equal(got[0][i], want[i, :n]), i
