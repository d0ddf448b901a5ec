"""Clip-level data parallelism over the GPUs of one node.

Clips are independent, so the batch is cut into contiguous blocks, one per rank (weights are
replicated), each rank runs the whole hot path on its block, and ONE all-gather over RCCL/xGMI
reassembles the output waveforms on every rank.  Ragged lengths: the int64 lengths are gathered
first and the waveforms are padded to the longest clip for the gather.

`backend` is whatever the process group was initialised with: "nccl" (= RCCL) on GPUs, "gloo" on
CPU (tests/test_dist_cpu.py runs world_size 2 with a stand-in enhance function).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Contiguous partition: item i belongs to rank floor(i * world / n_items)."""
    lo = -(-rank * n_items // world)
    hi = -(-(rank + 1) * n_items // world)
    return lo, min(hi, n_items)


def shard(items, world, rank):
    lo, hi = shard_bounds(len(items), world, rank)
    return items[lo:hi]


def gather_ragged(local, n_total, device, group=None):
    """local: list of 1-D float32 tensors (this rank's outputs, in order).  Returns the list of all
    n_total outputs in global order on every rank.  Two collectives: lengths, then padded data."""
    world = dist.get_world_size(group)
    per = max(shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world))
    lens = torch.zeros(per, dtype=torch.int64, device=device)
    for i, t in enumerate(local):
        lens[i] = t.numel()
    all_lens = torch.empty(world * per, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_lens, lens, group=group)
    maxlen = int(all_lens.max().item()) if all_lens.numel() else 0
    buf = torch.zeros((per, maxlen), dtype=torch.float32, device=device)
    for i, t in enumerate(local):
        buf[i, :t.numel()] = t
    out = torch.empty((world * per, maxlen), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    res = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        for j in range(hi - lo):
            res.append(out[r * per + j, :int(all_lens[r * per + j].item())])
    return res


def enhance_sharded(enhance_fn, mixes, ctx_a, ctx_b, device, group=None):
    """enhance_fn(mixes, ctx_a, ctx_b) -> list of 1-D float32 tensors on `device` (one per clip).
    Every rank passes the full batch description and gets the full list of outputs back."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(len(mixes), world, rank)
    local = enhance_fn(mixes[lo:hi], ctx_a[lo:hi], ctx_b[lo:hi]) if hi > lo else []
    return gather_ragged(local, len(mixes), device, group)
