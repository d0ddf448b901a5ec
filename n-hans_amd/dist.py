"""Clip-level data parallelism over the GPUs of one node.

Clips are independent, so the batch is cut into contiguous blocks, one per rank (weights are
replicated), each rank runs the whole hot path on its block, and ONE all-gather over RCCL/xGMI
reassembles the output waveforms on every rank.  Ragged lengths: the int64 lengths are gathered
first; each rank then contributes its clips back to back in one flat buffer (padded to the largest
per-rank total only).

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
    """local: list of 1-D float32 tensors (this rank's outputs, in order); an entry may be None -- an
    item this rank skipped (unreadable files) -- and comes back as None on every rank: it travels as
    length -1 in the lengths vector, so that an output that is legitimately EMPTY (a clip of 0 frames)
    stays an empty tensor.  Returns the list of all n_total outputs in global order on every rank.
    Two collectives: the lengths, then ONE all-gather of data.

    Each rank sends its clips back to back in one flat buffer, padded only to the largest per-rank
    TOTAL -- not every clip to the longest clip of the job: one ten-minute recording among thousands
    of short ones costs its own length once, not once per clip.  The lengths come to the host in one
    copy (no per-clip device read)."""
    world = dist.get_world_size(group)
    bounds = [shard_bounds(n_total, world, r) for r in range(world)]
    per = max(hi - lo for lo, hi in bounds) if bounds else 0
    lens = torch.zeros(max(per, 1), dtype=torch.int64)
    for i, t in enumerate(local):
        lens[i] = -1 if t is None else t.numel()
    local = [t for t in local if t is not None and t.numel()]
    lens = lens.to(device)
    all_lens = torch.empty(world * max(per, 1), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_lens, lens, group=group)
    all_lens = all_lens.cpu().view(world, max(per, 1))          # the one device -> host read
    rank_total = all_lens.clamp(min=0).sum(dim=1)
    width = int(rank_total.max()) if n_total else 0
    flat = torch.zeros(max(width, 1), dtype=torch.float32, device=device)
    if local:
        torch.cat([t.reshape(-1) for t in local], out=flat[:sum(t.numel() for t in local)])
    out = torch.empty(world * max(width, 1), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, flat, group=group)
    out = out.view(world, max(width, 1))
    res = []
    for r, (lo, hi) in enumerate(bounds):
        at = 0
        for j in range(hi - lo):
            n = int(all_lens[r, j])
            if n < 0:
                res.append(None)
                continue
            res.append(out[r, at:at + n])
            at += n
    return res


def enhance_sharded(enhance_fn, mixes, ctx_a, ctx_b, device, group=None):
    """enhance_fn(mixes, ctx_a, ctx_b) -> list of 1-D float32 tensors on `device` (one per clip).
    Every rank passes the full batch description and gets the full list of outputs back."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(len(mixes), world, rank)
    local = enhance_fn(mixes[lo:hi], ctx_a[lo:hi], ctx_b[lo:hi]) if hi > lo else []
    return gather_ragged(local, len(mixes), device, group)
