"""Dev tool: per-workgroup cycle breakdown (prologue / K loop / epilogue) of the LDS-DMA conv
kernel for one layer, from s_memtime stamps written by the kernel (option debug_cycles_ptr; needs a developer build of the library:
`make -C n-hans_amd/csrc clean && make -C n-hans_amd/csrc DEV=1`; NHANS_ABLATE=<mask> selects a timing ablation).
    NHANS_CONV_VARIANT=1 python tools/conv_phase_cycles.py [block 0..7] [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, synth  # noqa: E402
from nhans_amd.apply import normalise, trim_to_frames  # noqa: E402


def main():
    block = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 998
    os.environ.setdefault("NHANS_CONV_VARIANT", "1")
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
    from nhans_amd import spec
    g = spec.main_geometry()[block]
    bn = 128 if g["cout"] >= 128 else 64
    nblk = -(-(frames * g["hout"] * g["wout"]) // 256) * (g["cout"] // bn)
    raw = dbg.cpu().numpy()
    d = raw[:4 << 20].reshape(-1, 4)[:nblk]               # the last launch is this block's conv2
    ph = raw[4 << 20:].reshape(-1, 4)[:nblk]
    nch = g["kh"] * g["kw"] * g["cout"] // 32 + (g["cin"] // 32 if g["cin"] not in (1, g["cout"]) else 0)
    print("in-loop cycles per chunk (wave 0): issue %.0f | frag reads %.0f | mfma issue %.0f | dma wait + barrier %.0f" % tuple(ph.mean(0) / nch))
    pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    span = d[:, 3].max() - d[:, 0].min()
    print("block %d conv2: %d workgroups; s_memtime ticks (100 MHz) median pro/loop/epi = %d / %d / %d ; "
          "mean %.0f / %.0f / %.0f ; kernel span %d" % (block, len(d), np.median(pro), np.median(loop), np.median(epi),
                                                    pro.mean(), loop.mean(), epi.mean(), span))


if __name__ == "__main__":
    main()
