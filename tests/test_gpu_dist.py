"""Multi-process runs of the REAL engine on the GPU box: two ranks, each with its own
libnhans_hip context, clips sharded by dist.shard_bounds, outputs reassembled by the path's one
all-gather -- and the result compared bit for bit with the single-process batch.

The driver's GPU box has one MI355X, so the two ranks share device 0 and the collective runs over
gloo (host); the RCCL variant of the same worker needs two devices and is skipped otherwise.  SCALE
numbers on 8 GPUs are the driver's to measure.

Workers are forked from the fork server started in conftest.py (never an exec from a process that
has initialised the GPU).
"""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CLIPS = 5
SECS = [0.3, 1.0, 0.025, 0.7, 0.45]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch():
    import nhans_amd  # noqa: F401
    from nhans_amd import apply, synth
    mixes = [apply.trim_to_frames(apply.normalise(synth.mixture(300 + i, SECS[i]))) for i in range(N_CLIPS)]
    ca = [apply.normalise(synth.silent()) for _ in range(N_CLIPS)]
    cb = [apply.normalise(synth.noise_context(300 + i)) for i in range(N_CLIPS)]
    return mixes, ca, cb


def _rank_main(rank, world, port, backend, q):
    """One rank: real Engine, dist.enhance_sharded, report every gathered waveform."""
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch
        import torch.distributed as tdist
        import nhans_amd  # noqa: F401
        from nhans_amd import dist as nd, engine, weights
        dev_index = rank if backend == "nccl" else 0
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            gdev = torch.device("cuda", dev_index)
        else:
            tdist.init_process_group("gloo", rank=rank, world_size=world)
            gdev = torch.device("cpu")
        eng = engine.Engine("denoiser", weights.synthetic_weights("denoiser", 7), device=dev_index, precision="f16x3")
        mixes, ca, cb = _batch()
        seen = []

        def run(m, a, b):
            seen.append(len(m))
            res = eng.enhance(m, a, b, want_mixed=False)
            return [torch.from_numpy(w).to(gdev) for w in res["denoised_wav"]]
        out = nd.enhance_sharded(run, mixes, ca, cb, gdev)
        lo, hi = nd.shard_bounds(len(mixes), world, rank)
        q.put((rank, seen == [hi - lo], [w.cpu().numpy() for w in out]))
        tdist.barrier()
        tdist.destroy_process_group()
        eng.close()
    except Exception as e:      # surface the failure instead of a timeout
        import traceback
        q.put((rank, False, traceback.format_exc() + repr(e)))


def _run_two_ranks(backend, world=2):
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    return res


def _single_process_reference():
    import nhans_amd  # noqa: F401
    from nhans_amd import engine, weights
    eng = engine.Engine("denoiser", weights.synthetic_weights("denoiser", 7), precision="f16x3")
    mixes, ca, cb = _batch()
    ref = eng.enhance(mixes, ca, cb, want_mixed=False)["denoised_wav"]
    eng.close()
    return ref


def _check(res, ref):
    for rank, own_shard_only, waves in res:
        assert own_shard_only is True, waves                       # each rank ran exactly its block of clips
        assert len(waves) == N_CLIPS
        for i in range(N_CLIPS):                                   # every rank holds the whole batch, bit-equal
            assert np.array_equal(waves[i], ref[i]), (rank, i)


def test_two_ranks_one_gpu_gloo_equals_single_process(lib_built):
    res = _run_two_ranks("gloo")              # workers first: this process has not touched the GPU yet
    _check(res, _single_process_reference())


def test_one_rank_rccl_collectives_equal_single_process(lib_built):
    """The RCCL leg of dist.enhance_sharded on the one GPU there is: process group over "nccl" with one rank, the
    lengths gather and the padded all-gather on DEVICE tensors (library load, communicator init, collectives,
    teardown) -- what the two-rank RCCL test below needs a second GPU for."""
    res = _run_two_ranks("nccl", world=1)
    _check(res, _single_process_reference())


def test_two_ranks_two_gpus_rccl_equals_single_process(lib_built):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    res = _run_two_ranks("nccl")
    _check(res, _single_process_reference())


def _cli_rank(rank, world, port, argv, q):
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0", NHANS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import nhans_amd  # noqa: F401
        from nhans_amd import apply
        apply.main(argv)
        import torch.distributed as tdist
        q.put((rank, not tdist.is_initialized(), ""))       # the CLI waits for rank 0's writes and tears its group down
    except Exception as e:
        import traceback
        q.put((rank, False, traceback.format_exc() + repr(e)))


def test_cli_directory_mode_sharded_over_two_ranks(lib_built, tmp_path):
    """`nhans_denoiser --input <dir> ...` under WORLD_SIZE=2 (what torch.distributed.run sets): the
    directory is one batch, sharded over the ranks, rank 0 writes every file; same bytes as one rank."""
    from scipy.io import wavfile
    import nhans_amd  # noqa: F401
    from nhans_amd import synth
    ind, negd = tmp_path / "in", tmp_path / "neg"
    ind.mkdir()
    negd.mkdir()
    names = ["clip%02d.wav" % i for i in range(N_CLIPS)]
    for i, n in enumerate(names):
        wavfile.write(str(ind / n), 16000, synth.mixture(300 + i, SECS[i]))
        wavfile.write(str(negd / n), 16000, synth.noise_context(300 + i))
    common = ["--input", str(ind), "--neg", str(negd), "--pos", str(tmp_path / "Silent.wav"), "--weights", "synthetic"]
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_rank, args=(r, 2, port, common + ["--output", str(tmp_path / "out2")], q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=600) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert [r[:2] for r in res] == [(0, True), (1, True)], res
    # one rank, same CLI
    p1 = ctx.Process(target=_cli_single, args=(common + ["--output", str(tmp_path / "out1")], q))
    p1.start()
    ok = q.get(timeout=600)
    p1.join(timeout=120)
    assert ok == "done", ok
    files = sorted(os.listdir(str(tmp_path / "out1")))
    assert files == sorted(os.listdir(str(tmp_path / "out2"))) and len(files) == 4 * N_CLIPS
    for f in files:
        a = open(str(tmp_path / "out1" / f), "rb").read()
        b = open(str(tmp_path / "out2" / f), "rb").read()
        assert a == b, f


def _cli_single(argv, q):
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        import nhans_amd  # noqa: F401
        from nhans_amd import apply
        apply.main(argv)
        q.put("done")
    except Exception as e:
        import traceback
        q.put(traceback.format_exc() + repr(e))


def _bench_launcher(argv, q):
    """`python bench.py <argv>` as the driver would start it, from a process that has not touched the GPU."""
    try:
        import contextlib
        import io
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            os.environ.pop(k, None)
        import bench
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rc = bench.main(argv)
        q.put((rc, buf.getvalue()))
    except BaseException as e:
        import traceback
        q.put((-1, traceback.format_exc() + repr(e)))


def test_bench_gpus_n_without_a_launcher_starts_n_ranks(lib_built):
    """`python bench.py --gpus 2 ...` with no torch.distributed.run around it (the shape the driver uses): the
    parent starts two rank processes itself, rank 0's JSON line says n_gpus 2 and counts both ranks' clips;
    a mismatch between --gpus and a launcher's WORLD_SIZE is an error, not a silently different run.  (Both
    ranks share device 0 over gloo here: the driver's box has one GPU.)"""
    import json
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    argv = ["--gpus", "2", "--share-device0", "--clips-per-gpu", "4", "--seconds", "1", "--steps", "2", "--warmup", "1",
            "--no-kernel-pass", "--no-cpu-baseline"]
    p = ctx.Process(target=_bench_launcher, args=(argv, q))
    p.start()
    rc, out = q.get(timeout=900)
    p.join(timeout=60)
    assert rc == 0, out
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["clips_per_gpu"] == 4 and line["steps"] == 2
    assert line["frames_per_s"] > 0 and line["status_flags"] == 0
    audio_per_step = 2 * 4 * (98 * 160 + 240) / 16000.0                  # both ranks' clips, trimmed to 98 frames
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - audio_per_step) < 1e-6 * audio_per_step
    # every rank's own figures travel to rank 0 (round 5): the whole-job step time is the slowest rank's, barrier to barrier
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(0 < r["ms_per_step"] <= line["ms_per_step"] * 1.001 for r in pr), pr


def test_bench_eight_ranks_on_one_device_and_a_rank_that_dies(lib_built):
    """Rehearsal of the 8-GPU launch the driver makes (`python bench.py --gpus 8`): eight rank processes, eight engines
    (small workspaces: all on device 0, the all-gather over gloo), one JSON line that counts all eight ranks' clips --
    and the same launch with rank 3 dying at start-up comes back as exit code 1 within seconds instead of hanging in
    the other ranks' init_process_group."""
    import json
    import time
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    argv = ["--gpus", "8", "--share-device0", "--clips-per-gpu", "2", "--seconds", "1", "--steps", "1", "--warmup", "1",
            "--frames-per-chunk", "196", "--no-kernel-pass", "--no-cpu-baseline", "--no-ceiling"]
    p = ctx.Process(target=_bench_launcher, args=(argv, q))
    p.start()
    rc, out = q.get(timeout=1500)
    p.join(timeout=60)
    assert rc == 0, out
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 8 and line["config"]["clips_per_gpu"] == 2 and line["status_flags"] == 0
    assert [r["rank"] for r in line["per_rank"]] == list(range(8)) and all(r["ms_per_step"] > 0 for r in line["per_rank"])
    audio_per_step = 8 * 2 * (98 * 160 + 240) / 16000.0
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - audio_per_step) < 1e-6 * audio_per_step
    t0 = time.time()
    p = ctx.Process(target=_bench_launcher, args=(argv + ["--debug-fail-rank", "3"], q))
    p.start()
    rc, out = q.get(timeout=600)
    p.join(timeout=60)
    assert rc == 1 and time.time() - t0 < 300, (rc, out)
