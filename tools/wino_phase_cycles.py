"""Dev tool: where a workgroup of the Winograd conv kernel (conv_wino.hip) spends its cycles, from s_memtime stamps of
every wave (option debug_cycles_ptr; needs a developer build: `make -C n-hans_amd/csrc clean && make -C n-hans_amd/csrc DEV=1`):
    python tools/wino_phase_cycles.py [block 1|3] [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, spec, synth  # noqa: E402
from nhans_amd.apply import normalise, trim_to_frames  # noqa: E402


def main():
    block = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 998
    eng = engine.Engine("denoiser", precision="f16x3")
    eng.set_option("winograd", 1)
    mix = trim_to_frames(normalise(synth.mixture(0, max(10.0, frames / 100.0 + 0.1))))
    lm, _ = eng.stft_features(torch.from_numpy(mix).cuda(), [0, len(mix)])
    ea = torch.zeros(1, 512, device="cuda")
    dbg = torch.zeros(12 * (1 << 20), dtype=torch.int64, device="cuda")
    eng.set_option("debug_cycles_ptr", dbg.data_ptr())
    eng.set_option("frames_per_chunk", frames)
    for _ in range(2):
        dbg.zero_()
        eng.set_option("profile", 1)
        eng.profile_reset()
        eng.block_output(lm, [0, lm.shape[0]], ea, ea, 0, frames, block)
        torch.cuda.synchronize()
        prof = eng.profile()
    g = spec.main_geometry()[block]
    raw = dbg.cpu().numpy()
    d = raw[:(4 << 20) // 48 * 48].reshape(-1, 12, 4).astype(np.float64)
    # every Winograd launch of the call stamps the same buffer: the records beyond the LAST launch's grid belong to earlier,
    # larger launches (until late in round 5 this tool averaged them in: its "block 3" figures were half block 1's)
    nblk = int(raw[(8 << 20) + 19])
    assert 0 < nblk <= int((d[:, 0, 0] > 0).sum())
    d = d[:nblk]
    m = d.mean(0)
    e = raw[(4 << 20):(4 << 20) + nblk * 96].reshape(nblk, 12, 8).astype(np.float64).mean(0)
    nc = g["cout"] // 8
    print("block %d conv2 (the last launch), %d workgroups, %d periods of 8 channels; MFMA floor per period %d cycles (per 128 tile-pixels)"
          % (block, nblk, nc, 2 * 48 * 32))
    print("  profile:", {k: round(v["ms"], 3) for k, v in prof.items() if "wino" in k or "halo" in k})
    f = raw[(6 << 20):(6 << 20) + nblk * 48].reshape(nblk, 12, 4).astype(np.float64).mean(0)
    for w in range(8):
        print("  wave %d: transform per period: tile reads landed %.0f, arithmetic done %.0f, writes drained %.0f"
              % (w, f[w, 0] / (nc - 1), f[w, 1] / (nc - 1), e[w, 6] / nc))
    for w in range(8):
        print("  wave %d: first request after %.0f, two of four issued %.0f, setup %.0f, first tile %.0f, prologue %.0f | K loop %.0f = %.0f per period: multiply %.0f, request + transform %.0f, "
              "tile wait %.0f, barrier %.0f | epilogue %.0f (pass 0: tiles in LDS %.0f, barrier %.0f, transformed %.0f)"
              % (w, f[w, 2], f[w, 3], e[w, 0], e[w, 1], m[w, 1], m[w, 0], m[w, 0] / nc, e[w, 5] / nc, e[w, 6] / nc, e[w, 7] / nc, m[w, 3] / nc, m[w, 2],
                 e[w, 2], e[w, 3], e[w, 4]))
    nb4 = min(nblk, 4096)                                                  # (the epilogue and per-period records: first 4,096 workgroups)
    h = raw[(8 << 20):(8 << 20) + nb4 * 288].reshape(nb4, 12, 24).astype(np.float64).mean(0)
    for w in range(8):
        print("  wave %d epilogue from the loop's end: pass 0 entry %.0f, residual requested %.0f, tiles in LDS %.0f, barrier %.0f, transformed %.0f, columns stored %.0f | "
              "pass 1 entry %.0f, requested %.0f, tiles %.0f, barrier %.0f, transformed %.0f, stored %.0f | drained %.0f"
              % (w, h[w, 3], h[w, 4], h[w, 0], h[w, 1], h[w, 2], h[w, 5], h[w, 11], h[w, 12], h[w, 8], h[w, 9], h[w, 10], h[w, 13], h[w, 16]))
    pv = raw[(10 << 20):(10 << 20) + nb4 * 384].reshape(nb4, 12, 32)[:, :, :nc].astype(np.float64)
    per = np.diff(np.concatenate([np.zeros((nb4, 12, 1)), pv], axis=2), axis=2).mean(0)
    print("  wave 0, cycles of each period (barrier to barrier):", " ".join("%.0f" % x for x in per[0]))
    print("  s_memtime ticks per s_memrealtime tick (100 MHz) over loop + epilogue: %.2f (= the shader clock in MHz / 100)" % (h[0, 18] / max(h[0, 17], 1)))


if __name__ == "__main__":
    main()
