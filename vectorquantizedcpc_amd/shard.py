"""Utterance sharding across the GPUs of one node.

Utterances are independent (the reference itself loops over them one by one,
``encode.py:42`` / ``convert.py:52``), so the path shards with NO data-path collective:
every rank (one process per GPU) holds replicated weights, encodes and decodes its own
utterances, and ONE exchange at the end collects the waveforms on rank 0 -- a gather over
RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).
"""
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def partition_lpt(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of utterance ids to ranks (decode cost is
    proportional to the sample count).  Deterministic: ties broken by utterance id / rank id."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(p) for p in parts]


def partition_contiguous(n: int, world_size: int) -> List[List[int]]:
    """Equal contiguous blocks (the weak-scaling bench: utterances of equal length)."""
    per = (n + world_size - 1) // world_size
    return [list(range(r * per, min(n, (r + 1) * per))) for r in range(world_size)]


def gather_waveforms(wav: torch.Tensor, ids: Sequence[int], lengths: Sequence[int], n_total: int,
                     dst: int = 0, group=None, force_collective: bool = False) -> Optional[List[torch.Tensor]]:
    """Collect per-rank waveforms on ``dst``.

    wav: (n_local, L_local) this rank's padded waveforms, row k belongs to utterance ids[k] and
    has lengths[k] valid samples.  Returns, on ``dst`` only, a list of n_total 1-D tensors in
    utterance order.  Message plan: one all_gather of (count, L) per rank, then one gather of the
    max-padded (count_max, L_max) float32 blocks and one of their (count_max, 2) int64 side tables
    (utterance id, length) -- integers travel as integers (a float32 column is exact only below 2**24).
    A world of one rank needs no exchange and takes none, unless ``force_collective`` asks for the three collectives anyway
    (``bench.py --workload manifest --force-gather`` on a 1-GPU box: the ragged RCCL path executes before an 8-GPU node does).
    """
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return [wav[ids.index(i), : lengths[ids.index(i)]] for i in range(n_total)] if len(ids) == n_total else None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = wav.device
    meta = torch.tensor([wav.shape[0], wav.shape[1] if wav.dim() == 2 else 0], device=dev, dtype=torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    cmax = int(max(m[0] for m in metas))
    lmax = int(max(m[1] for m in metas))
    block = torch.zeros(cmax, lmax, device=dev, dtype=torch.float32)
    side = torch.full((cmax, 2), -1, device=dev, dtype=torch.int64)
    n = wav.shape[0]
    if n:
        side[:n, 0] = torch.tensor([int(i) for i in ids], device=dev, dtype=torch.int64)
        side[:n, 1] = torch.tensor([int(v) for v in lengths], device=dev, dtype=torch.int64)
        block[:n, : wav.shape[1]] = wav
    out = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
    sides = [torch.empty_like(side) for _ in range(world)] if rank == dst else None
    dist.gather(side, sides, dst=dst, group=group)
    dist.gather(block, out, dst=dst, group=group)
    if rank != dst:
        return None
    result: List[Optional[torch.Tensor]] = [None] * n_total
    for r in range(world):
        sd = sides[r].tolist()
        for k in range(int(metas[r][0])):
            i, ln = sd[k]
            result[i] = out[r][k, :ln]
    return result


def convert_sharded(mels: Sequence[torch.Tensor], speakers: Sequence[int],
                    decode_fn: Callable[[List[int], List[torch.Tensor], List[int]], torch.Tensor],
                    samples_per_frame: int = 160, group=None, dst: int = 0,
                    check_fn: Optional[Callable[[], None]] = None, force_collective: bool = False,
                    stats: Optional[dict] = None):
    """Batched ``convert.py:52-77`` over a node: LPT-shard, decode locally, gather on ``dst``.

    decode_fn(ids, mels, speakers) -> (n_local, L) padded waveforms for this rank's utterances
    (the HIP path: ``Encoder.encode_indices`` + ``Vocoder.generate``).  ``check_fn`` (``Vocoder.check``) runs between
    the local decode and the gather; on failure the local decode is repeated once on the fallback path.
    ``stats`` (a dict) receives this rank's utterance count, sample count, local decode seconds and gather seconds (both
    bracketed by a device synchronisation when the waveforms live on a GPU).
    """
    import time

    def sync():
        if wav is not None and wav.is_cuda:
            torch.cuda.synchronize(wav.device)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [int(m.shape[-1]) // 2 * 2 * samples_per_frame for m in mels]
    mine = partition_lpt(lengths, world)[rank]
    wav = None
    t0 = time.perf_counter()
    wav = decode_fn(mine, [mels[i] for i in mine], [speakers[i] for i in mine])
    if check_fn is not None:             # e.g. Vocoder.check: an aborted in-kernel hand-off must not reach the gather
        try:
            check_fn()
        except RuntimeError:
            wav = decode_fn(mine, [mels[i] for i in mine], [speakers[i] for i in mine])     # the handle has fallen back
            check_fn()
    sync()
    t1 = time.perf_counter()
    out = gather_waveforms(wav, mine, [lengths[i] for i in mine], len(mels), dst=dst, group=group, force_collective=force_collective)
    sync()
    if stats is not None:
        stats.update({"utterances": len(mine), "samples": int(sum(lengths[i] for i in mine)), "decode_s": t1 - t0,
                      "gather_s": time.perf_counter() - t1})
    return out
