"""world_size-2 gloo tests of the clip sharding + single all-gather reassembly.
A stand-in enhance function replaces the HIP engine: the partition/gather logic is what is under
test and it is backend-independent (the GPU suite checks shard invariance of the real engine)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nhans_amd  # noqa: F401
from nhans_amd import dist as nd


def test_shard_bounds_partition():
    for n in (1, 2, 7, 8, 9, 256, 2048):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = nd.shard_bounds(n, world, r)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi))
                for i in range(lo, hi):
                    assert i * world // n == r              # clip i -> rank floor(i*G/N)
            assert got == list(range(n))
    assert nd.shard_bounds(2048, 8, 3) == (768, 1024)


def _fake_enhance(mixes, ca, cb):
    # deterministic function of all three inputs, ragged output lengths
    return [torch.from_numpy(m * 2.0 + a[:1] - b[:1]).float() for m, a, b in zip(mixes, ca, cb)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    lens = [400, 560, 1040, 400, 880]                      # ragged, 5 clips over 2 ranks (3 + 2)
    mixes = [rng.standard_normal(n).astype(np.float32) for n in lens]
    ca = [rng.standard_normal(8).astype(np.float32) for _ in lens]
    cb = [rng.standard_normal(8).astype(np.float32) for _ in lens]
    out = nd.enhance_sharded(_fake_enhance, mixes, ca, cb, torch.device("cpu"))
    ref = _fake_enhance(mixes, ca, cb)
    ok = len(out) == len(ref) and all(torch.equal(a, b) for a, b in zip(out, ref))
    # a batch smaller than the world: rank 1 owns nothing
    out1 = nd.enhance_sharded(_fake_enhance, mixes[:1], ca[:1], cb[:1], torch.device("cpu"))
    ok = ok and len(out1) == 1 and torch.equal(out1[0], ref[0])
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_batch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
