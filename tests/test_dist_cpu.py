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
    # one long recording among short ones: the data all-gather carries each rank's clips back to back, padded to the
    # largest per-rank TOTAL (not every clip to the longest clip); an EMPTY output (a clip of 0 frames) comes back as an
    # empty tensor and a skipped job (None: length -1 on the wire) as None -- the two are not the same thing
    lens2 = [400, 100000, 0, 400, None, 400]
    loc = [None if n is None else torch.full((n,), float(i + 1)) for i, n in enumerate(lens2)]
    lo, hi = nd.shard_bounds(len(lens2), world, rank)
    sizes = []
    real = dist.all_gather_into_tensor

    def spy(out, inp, group=None):
        sizes.append(out.numel())
        return real(out, inp, group=group)
    nd.dist.all_gather_into_tensor = spy
    try:
        got = nd.gather_ragged(loc[lo:hi], len(lens2), torch.device("cpu"))
    finally:
        nd.dist.all_gather_into_tensor = real
    ok = ok and [None if t is None else t.numel() for t in got] == lens2
    ok = ok and all((a is None and b is None) or torch.equal(a, b) for a, b in zip(got, loc))
    ok = ok and sizes == [world * 3, world * (400 + 100000 + 0)]            # lengths, then data: 2 x 100,400, not 6 x 100,000
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


# ------------------------------------------------------------------------------ the product entry point, 2 ranks on CPU
class _FakeEngine:
    """Stands in for engine.Engine on the CPU box: recognisable waveforms, records what each rank was given."""
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def enhance(self, mixes, ca, cb, want_mixed=True, taps=False):
        self.calls.append(len(mixes))
        return {"denoised_wav": [m * np.float32(0.5) + a[:1] for m, a in zip(mixes, cb)], "mixed_wav": [m.copy() for m in mixes]}


def _cli_worker(rank, world, port, argv, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", NHANS_DIST_BACKEND="gloo")
    from nhans_amd import apply
    fake = _FakeEngine()
    apply.set_engine("denoiser", fake)
    opened, real = [], apply.wavread

    def spy(path, *a, **k):
        opened.append(os.path.basename(os.path.dirname(path)) + "/" + os.path.basename(path))
        return real(path, *a, **k)
    apply.wavread = spy
    apply.main(argv)
    q.put((rank, fake.calls, dist.is_initialized(), sorted(n for n in opened if n.startswith("in/"))))


def test_cli_directory_mode_shards_clips_over_two_ranks(tmp_path):
    """`nhans_denoiser --input <dir>` under WORLD_SIZE=2: 6 jobs -> blocks of 3 and 3, every rank READS only its own
    block (one of rank 0's files is unreadable), ONE engine call per rank, one all-gather, rank 0 writes every
    file -- identical to what a single rank writes."""
    from scipy.io import wavfile
    from nhans_amd import apply, synth
    ind, negd = tmp_path / "in", tmp_path / "neg"
    ind.mkdir()
    negd.mkdir()
    names = ["c%d.wav" % i for i in range(5)]
    for i, n in enumerate(names):
        wavfile.write(str(ind / n), 16000, synth.mixture(70 + i, 0.05 + 0.03 * i))
        wavfile.write(str(negd / n), 16000, synth.noise_context(70 + i, 1.0))
    (ind / "bad.wav").write_bytes(b"this is not a wav file")          # job 0 of rank 0: reported, skipped, nothing written
    wavfile.write(str(negd / "bad.wav"), 16000, synth.noise_context(1, 1.0))
    common = ["--input", str(ind), "--neg", str(negd), "--pos", str(tmp_path / "Silent.wav"), "--weights", "synthetic"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, common + ["--output", str(tmp_path / "out2")], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, [2], False, ['in/bad.wav', 'in/c0.wav', 'in/c1.wav']), (1, [3], False, ['in/c2.wav', 'in/c3.wav', 'in/c4.wav'])]         # contiguous blocks, one batched call per rank; the CLI tore its process group down
    fake = _FakeEngine()
    apply.set_engine("denoiser", fake)
    try:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            os.environ.pop(k, None)
        apply.main(common + ["--output", str(tmp_path / "out1")])
    finally:
        apply._engines.pop("denoiser", None)
    assert fake.calls == [5]
    files = sorted(os.listdir(str(tmp_path / "out1")))
    assert files == sorted(os.listdir(str(tmp_path / "out2"))) and len(files) == 20
    for f in files:
        assert open(str(tmp_path / "out1" / f), "rb").read() == open(str(tmp_path / "out2" / f), "rb").read(), f
