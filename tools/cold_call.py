"""configs[1] as a user meets it -- `nhans_denoiser --input one.wav --neg noise.wav --output out.wav`
in a FRESH process (SN/apply.py:478-527, setup.py:44-50: one process per file), wall clock measured around the process
from outside, with the command's own --timing breakdown beside it.
    python tools/cold_call.py [runs=3] [seconds=10]
Legs: (a) --no-cache (fold in the process: what every call cost up to round 5, minus `import torch`), (b) the first call
with the cache (miss: fold + store), (c) calls with the cache warm (the steady state of a user's shell loop), (d) the
same through the full torch engine (NHANS_FORCE_TORCH_ENGINE=1) for comparison.  The very first process of a fresh box
also pages the libraries in from disk: reported separately as `first_process`."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(argv, env):
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import nhans_amd; from nhans_amd import apply; apply.main(sys.argv[1:])" % ROOT] + argv,
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    wall = time.perf_counter() - t0
    timing = None
    for line in p.stderr.decode().splitlines():
        if line.startswith("nhans timing: "):
            timing = json.loads(line[len("nhans timing: "):])
    if p.returncode != 0:
        sys.stderr.write(p.stderr.decode()[-2000:])
    return {"wall_s": wall, "rc": p.returncode, "breakdown": timing}


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    import numpy as np  # noqa: F401
    from scipy.io import wavfile
    import nhans_amd  # noqa: F401
    from nhans_amd import synth
    tmp = tempfile.mkdtemp(prefix="nhans_cold_")
    cache = os.path.join(tmp, "cache")
    wavfile.write(os.path.join(tmp, "in.wav"), 16000, synth.mixture(0, secs))
    wavfile.write(os.path.join(tmp, "neg.wav"), 16000, synth.noise_context(0))
    base = ["--input", os.path.join(tmp, "in.wav"), "--neg", os.path.join(tmp, "neg.wav"), "--pos", os.path.join(tmp, "Silent.wav"),
            "--weights", "synthetic", "--timing"]
    env = dict(os.environ, NHANS_CACHE_DIR=cache)
    res = {"audio_seconds": secs, "legs": {}}
    res["first_process"] = run(base + ["--output", os.path.join(tmp, "o0.wav"), "--no-cache"], env)
    res["legs"]["no_cache"] = [run(base + ["--output", os.path.join(tmp, "o1.wav"), "--no-cache"], env) for _ in range(runs)]
    res["legs"]["cache_miss_then_store"] = [run(base + ["--output", os.path.join(tmp, "o2.wav")], env)]
    res["legs"]["cache_hit"] = [run(base + ["--output", os.path.join(tmp, "o3.wav")], env) for _ in range(max(runs, 5))]
    res["legs"]["torch_engine_cache_irrelevant"] = [run(base + ["--output", os.path.join(tmp, "o4.wav")], dict(env, NHANS_FORCE_TORCH_ENGINE="1"))
                                                    for _ in range(runs)]
    a = wavfile.read(os.path.join(tmp, "o1.wav"))[1]
    b = wavfile.read(os.path.join(tmp, "o3.wav"))[1]
    c = wavfile.read(os.path.join(tmp, "o4.wav"))[1]
    res["outputs_bit_identical"] = bool((a == b).all() and (a == c).all())
    med = lambda xs: sorted(xs)[len(xs) // 2]
    res["summary_wall_s"] = {k: med([r["wall_s"] for r in v]) for k, v in res["legs"].items()}
    res["cache_bytes"] = sum(os.path.getsize(os.path.join(cache, f)) for f in os.listdir(cache)) if os.path.isdir(cache) else 0
    print(json.dumps(res, indent=1))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
