"""Dev tool: where the K loop of the halo conv kernel (conv_igemm_halo.hip) spends its cycles, from
s_memtime stamps of waves 0 and 7 of every workgroup (option debug_cycles_ptr; needs a developer build of the library:
`make -C n-hans_amd/csrc clean && make -C n-hans_amd/csrc DEV=1`; NHANS_ABLATE=<mask> selects a timing ablation):
    python tools/halo_phase_cycles.py [block 0..7] [frames]"""
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
    block = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 998
    os.environ.setdefault("NHANS_CONV_VARIANT", "2")
    eng = engine.Engine("denoiser", precision="f16x3")
    mix = trim_to_frames(normalise(synth.mixture(0, 10.0)))
    lm, _ = eng.stft_features(torch.from_numpy(mix).cuda(), [0, len(mix)])
    ea = torch.zeros(1, 512, device="cuda")
    dbg = torch.zeros(8 * (1 << 20), dtype=torch.int64, device="cuda")
    eng.set_option("debug_cycles_ptr", dbg.data_ptr())
    eng.set_option("frames_per_chunk", frames)
    for _ in range(2):
        dbg.zero_()
        eng.block_output(lm, [0, lm.shape[0]], ea, ea, 0, frames, block)
        torch.cuda.synchronize()
    g = spec.main_geometry()[block]
    bn = 128 if g["cout"] >= 128 else 64
    tile = 256 if g["cout"] >= 128 or os.environ["NHANS_CONV_VARIANT"] != "2" else 512
    nblk = -(-(frames * g["hout"] * g["wout"]) // tile) * (g["cout"] // bn)
    d = dbg.cpu().numpy()[:nblk * 48].reshape(nblk, 12, 4).astype(np.float64)   # the last launch = this block's conv2
    taps = g["kh"] * g["kw"] * g["cout"] // 32 + (g["cin"] // 32 if g["cin"] not in (1, g["cout"]) else 0)
    m = d.mean(0) / taps
    print("block %d conv2, %d taps, %d workgroups; cycles per tap" % (block, taps, nblk))
    for w in range(8):
        print("  consumer wave %d: loop %.0f | lgkm+barrier %.0f | reads+MFMA %.0f   || per workgroup: prologue %.0f, K loop %.0f, epilogue %.0f cycles"
              % (w, m[w, 0], m[w, 3], m[w, 0] - m[w, 3], m[w, 1] * taps, m[w, 0] * taps, m[w, 2] * taps))
    e = dbg.cpu().numpy()[(4 << 20):(4 << 20) + nblk * 96].reshape(nblk, 12, 8).astype(np.float64).mean(0)
    for w in (0, 7):
        print("  epilogue of consumer wave %d, cycles from its start: accumulators in LDS %.0f | barrier passed %.0f | first group of loads "
              "arrived %.0f | all stores issued %.0f | stores drained %.0f" % (w, e[w, 0], e[w, 1], e[w, 2], e[w, 3], e[w, 4]))
    for w in range(8, 12):
        print("  producer wave %d: loop %.0f | DMA issue %.0f | vmcnt wait %.0f | barrier %.0f" % (w, m[w, 0], m[w, 1], m[w, 2], m[w, 3]))


if __name__ == "__main__":
    main()
