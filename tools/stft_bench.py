"""Times nhans_stft_features / nhans_istft alone on a batch of 10 s clips (hipEvents through torch).
    python tools/stft_bench.py [clips]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, spec, synth  # noqa: E402
from nhans_amd.apply import normalise, trim_to_frames  # noqa: E402


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    eng = engine.Engine("denoiser", precision="f16x3")
    base = [trim_to_frames(normalise(synth.mixture(i, 10.0))) for i in range(8)]
    mixes = [base[i % 8] for i in range(clips)]
    wav, off = eng._dev(mixes)
    frames = sum(spec.frames_for_samples(len(m))[1] for m in mixes)
    foff = [0]
    for m in mixes:
        foff.append(foff[-1] + spec.frames_for_samples(len(m))[1])
    for name, fn in (("stft_features", lambda: eng.stft_features(wav, off)),):
        lm, ph = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = min(ts)
        print("%s: %d frames, %.3f ms -> %.0f GB/s of 2,248 B/frame" % (name, frames, ms, frames * 2248 / ms / 1e6))
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); w, _ = eng.istft(lm, ph, foff); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = min(ts)
    print("istft_ola (incl. the output memset): %d frames, %.3f ms -> %.0f GB/s" % (frames, ms, frames * 2248 / ms / 1e6))
    err = float((w[:len(mixes[0])][240:-240] - wav[:len(mixes[0])][240:-240]).abs().max())
    print("round trip max err (interior, clip 0): %.2e" % err)


if __name__ == "__main__":
    main()
