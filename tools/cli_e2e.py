"""End-to-end CLI figure: `nhans_denoiser --input <dir> --neg <dir> --output <dir>` on a directory of
N synthetic 10 s / 16 kHz int16 wavs -- files in -> files out, wall clock, beside the hot-path figure of bench.py (whose
timed region starts with the inputs resident in HBM).  The CLI is run in-process (apply.main) twice: the first run
pays library load, weight folding and the workspace allocation; the second is the steady-state figure.
    python tools/cli_e2e.py [clips=256] [seconds=10]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import apply, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    from scipy.io import wavfile
    tmp = tempfile.mkdtemp(prefix="nhans_cli_")
    ind, negd = os.path.join(tmp, "in"), os.path.join(tmp, "neg")
    os.makedirs(ind)
    os.makedirs(negd)
    t0 = time.perf_counter()
    for i in range(n):
        wavfile.write(os.path.join(ind, "clip%04d.wav" % i), 16000, synth.mixture(i, secs))
        wavfile.write(os.path.join(negd, "clip%04d.wav" % i), 16000, synth.noise_context(i))
    t_gen = time.perf_counter() - t0
    res = {"clips": n, "seconds_per_clip": secs, "generate_inputs_s": t_gen, "runs": []}
    for run in range(2):
        out = os.path.join(tmp, "out%d" % run)
        argv = ["--input", ind, "--neg", negd, "--pos", os.path.join(tmp, "Silent.wav"), "--output", out, "--weights", "synthetic"]
        stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")             # (the reference prints snr_est per clip)
        t0 = time.perf_counter()
        try:
            apply.main(argv)
        finally:
            sys.stdout = stdout
        dt = time.perf_counter() - t0
        files = os.listdir(out)
        res["runs"].append({"wall_s": dt, "x_realtime": n * secs / dt, "files_written": len(files),
                            "bytes_written": sum(os.path.getsize(os.path.join(out, f)) for f in files)})
    res["note"] = ("run 0 includes library load, weight folding (fold.py) and the first workspace allocation; run 1 is the "
                   "steady state: read + convert + normalise %d wavs, one ragged nhans_enhance_clips call, D2H, four float32 "
                   "wavs written per clip" % n)
    print(json.dumps(res))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
