"""N > 1 path on CPU: world_size-2 gloo processes shard utterances and gather on rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vectorquantizedcpc_amd import shard


def test_partition_lpt_balances_and_is_deterministic():
    lengths = [900, 100, 500, 500, 300, 700, 200, 800]
    parts = shard.partition_lpt(lengths, 3)
    assert sorted(i for p in parts for i in p) == list(range(8))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 300
    assert parts == shard.partition_lpt(lengths, 3)
    assert shard.partition_contiguous(5, 2) == [[0, 1, 2], [3, 4]]
    assert shard.partition_lpt([], 2) == [[], []]


def _fake_decode(ids, mels, speakers):
    # deterministic stand-in for encode+generate: waveform value encodes (utterance, speaker, position)
    L = max([m.shape[-1] // 2 * 2 * 4 for m in mels], default=0)
    out = torch.zeros(len(ids), L)
    for k, (i, m, s) in enumerate(zip(ids, mels, speakers)):
        n = m.shape[-1] // 2 * 2 * 4
        out[k, :n] = i * 1000 + s * 10 + torch.arange(n) * 1e-3
    return out


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mels = [torch.zeros(80, t) for t in (8, 3, 6, 2, 5)]      # ragged, one odd length
        spk = [1, 2, 3, 4, 5]
        res = shard.convert_sharded(mels, spk, _fake_decode, samples_per_frame=4)
        if rank == 0:
            ok = len(res) == 5
            for i, (m, s) in enumerate(zip(mels, spk)):
                n = m.shape[-1] // 2 * 2 * 4
                ok &= res[i].shape == (n,) and torch.allclose(res[i], i * 1000 + s * 10 + torch.arange(n) * 1e-3)
            ret.put(bool(ok))
        else:
            assert res is None
            ret.put(True)
    finally:
        dist.destroy_process_group()


def test_sharded_convert_gathers_on_rank0_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    results = [ret.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(results) and all(p.exitcode == 0 for p in procs)
