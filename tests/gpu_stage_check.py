"""Stage-by-stage comparison of the HIP path against the float64 oracle on one short clip
(development aid, kept under tests/ because only tests may use the oracle; the pytest -m gpu suite is the real gate).
Usage on a GPU box:
    python tests/gpu_stage_check.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, spec, synth, weights  # noqa: E402
from oracle import nhans_oracle as O  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max()), float(np.abs(b).max())


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    kind = sys.argv[2] if len(sys.argv) > 2 else "denoiser"
    W = weights.synthetic_weights(kind, 7)
    mix = O.trim_to_frames(O.normalise(synth.mixture(0, secs)))
    ca = O.normalise(synth.silent())
    cb = O.normalise(synth.noise_context(0))
    prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
    eng = engine.Engine(kind, W, precision=prec)
    dev = eng.device

    # --- STFT
    spec_ref = O.stft(mix)
    lm_ref, ph_ref = O.logmag_phase(spec_ref)
    wav_t = torch.from_numpy(mix).to(dev)
    lm, ph = eng.stft_features(wav_t, [0, len(mix)])
    torch.cuda.synchronize()
    lm_h, ph_h = lm.cpu().numpy().astype(np.float64), ph.cpu().numpy().astype(np.float64)
    mag_h = np.exp(lm_h) - 1e-5
    print("stft frames", lm_h.shape, "| |X| err %.3e of max %.3e" % rel(mag_h, np.abs(spec_ref)))
    z_h = mag_h * np.exp(1j * ph_h)
    print("stft complex err %.3e of max %.3e" % (np.abs(z_h - spec_ref).max(), np.abs(spec_ref).max()))
    print("logmag err %.3e" % np.abs(lm_h - lm_ref).max())

    # --- contexts + tower
    ctx = []
    for w in (ca, cb):
        l, _ = O.logmag_phase(O.stft(w))
        ctx.append(O.context(l))
    ctx = np.stack(ctx)
    taps = {}
    emb_ref = O.embed_tower(ctx, W, taps)
    cl, _ = eng.stft_features(torch.from_numpy(np.concatenate([ca, cb])).to(dev), [0, len(ca), len(ca) + len(cb)], 200, False)
    torch.cuda.synchronize()
    print("ctx logmag err %.3e" % np.abs(cl.cpu().numpy().reshape(2, 200, 201) - ctx).max())
    emb = eng.embed(torch.from_numpy(ctx.astype(np.float32)).to(dev))
    torch.cuda.synchronize()
    print("emb err %.3e of max %.3e" % rel(emb.cpu().numpy(), emb_ref))

    # --- blocks (oracle features in, to isolate the network)
    lm32 = torch.from_numpy(lm_ref.astype(np.float32)).to(dev)
    T = lm_ref.shape[0]
    frames = sorted(set([0, min(17, T - 1), T // 2, T - 1]))
    win = O.strided_crop(lm_ref.astype(np.float32).astype(np.float64), O.MIX_WIN)
    ea = torch.from_numpy(emb_ref[0:1].astype(np.float32)).to(dev)
    eb = torch.from_numpy(emb_ref[1:2].astype(np.float32)).to(dev)
    ea64, eb64 = ea.cpu().numpy().astype(np.float64), eb.cpu().numpy().astype(np.float64)
    btaps = {}
    out_ref, den_ref = O.mask_net(win[frames], np.repeat(ea64, len(frames), 0), np.repeat(eb64, len(frames), 0), W, kind, btaps)
    names = [n for n, _ in O.STACK] + ["last_conv"]
    for b, name in enumerate(names):
        errs = []
        for i, f in enumerate(frames):
            got = eng.block_output(lm32, [0, T], ea, eb, f, 1, b)
            torch.cuda.synchronize()
            errs.append(np.abs(got.cpu().numpy()[0] - btaps[name][i]).max())
        print("block %d %-12s err %.3e (ref max %.3e)" % (b, name, max(errs), np.abs(btaps[name]).max()))
    t0 = time.time()
    lg, den = eng.mask_net(lm32, [0, T], ea, eb)
    torch.cuda.synchronize()
    print("mask_net %d frames in %.3f s" % (T, time.time() - t0))
    lg_h = lg.cpu().numpy()
    print("logits err on frames %s: %.3e (ref max %.3e)" % (frames, np.abs(lg_h[frames] - out_ref).max(), np.abs(out_ref).max()))
    print("denoised err: %.3e" % np.abs(den.cpu().numpy()[frames] - den_ref).max())

    # --- iSTFT
    ph32 = torch.from_numpy(ph_ref.astype(np.float32)).to(dev)
    wav, ooff = eng.istft(lm32, ph32, [0, T])
    torch.cuda.synchronize()
    wav_ref = O.recover_samples(lm_ref.astype(np.float32).astype(np.float64), ph_ref.astype(np.float32).astype(np.float64))
    print("istft len %d/%d err %.3e rms %.3e" % (wav.numel(), len(wav_ref), np.abs(wav.cpu().numpy() - wav_ref).max(),
                                                np.sqrt(np.mean((wav.cpu().numpy() - wav_ref) ** 2))))
    # --- whole path
    got = eng.enhance([mix], [ca], [cb], want_mixed=True, taps=True)
    print("e2e mixed_wav vs oracle roundtrip err %.3e" % np.abs(got["mixed_wav"][0] - wav_ref).max())
    print("e2e logits err on frames: %.3e" % np.abs(got["logits"][frames] - out_ref).max())
    print("e2e emb err %.3e" % np.abs(got["emb"] - emb_ref).max())


if __name__ == "__main__":
    main()
