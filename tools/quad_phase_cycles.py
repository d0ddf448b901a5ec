"""Dev tool (needs `make -C n-hans_amd/csrc clean && make -C n-hans_amd/csrc DEV=1`): per-workgroup phase ticks of the
four-wave conv kernel (conv_igemm_quad.hip, option quad_workgroups) and how many of its workgroups are resident
per CU, from s_memtime stamps and HW_REG_HW_ID of every wave.
    python tools/quad_phase_cycles.py [block 0..7] [frames]"""
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
    eng = engine.Engine("denoiser", precision="f16x3")
    mix = trim_to_frames(normalise(synth.mixture(0, 10.0)))
    lm, _ = eng.stft_features(torch.from_numpy(mix).cuda(), [0, len(mix)])
    ea = torch.zeros(1, 512, device="cuda")
    dbg = torch.zeros(8 * (1 << 20), dtype=torch.int64, device="cuda")
    eng.set_option("debug_cycles_ptr", dbg.data_ptr())
    eng.set_option("frames_per_chunk", frames)
    eng.set_option("quad_workgroups", 1)
    for _ in range(2):
        dbg.zero_()
        eng.block_output(lm, [0, lm.shape[0]], ea, ea, 0, frames, block)
        torch.cuda.synchronize()
    g = spec.main_geometry()[block]
    nblk = -(-(frames * g["hout"] * g["wout"]) // 128) * (g["cout"] // 128)
    taps = g["kh"] * g["kw"] * g["cout"] // 32 + (g["cin"] // 32 if g["cin"] not in (1, g["cout"]) else 0)
    d = dbg.cpu().numpy()[:nblk * 32].reshape(nblk, 4, 8)
    f = d.astype(np.float64)
    print("block %d conv2, %d taps, %d workgroups of 128 x 128" % (block, taps, nblk))
    for w in range(4):
        print("  wave %d: prologue %.0f | K loop %.0f (%.0f per tap, of which vmcnt+barrier %.0f) | epilogue %.0f | lifetime %.0f ticks"
              % (w, f[:, w, 1].mean(), f[:, w, 0].mean(), f[:, w, 0].mean() / taps, f[:, w, 3].mean() / taps, f[:, w, 2].mean(),
                 (f[:, w, 5] - f[:, w, 4]).mean()))
    t_in, t_out, hw = d[:, 0, 4], d[:, 0, 5], d[:, 0, 6]
    cu = ((hw >> 32) & 0xF) * 100000 + ((hw >> 8) & 0xFF)       # XCC, (SE, SH, CU): the clocks of different XCCs are unrelated
    share, loop_overlap = {}, []
    t_loop0 = t_in + d[:, 0, 1]                                  # K loop window of wave 0
    t_loop1 = t_loop0 + d[:, 0, 0]
    for c in set(cu.tolist()):
        idx = np.nonzero(cu == c)[0]
        ev = sorted([(int(t_in[i]), 1) for i in idx] + [(int(t_out[i]), -1) for i in idx])
        live, last = 0, ev[0][0]
        for t, s in ev:
            share[live] = share.get(live, 0) + (t - last)
            live += s
            last = t
        # fraction of each workgroup's non-loop time (prologue + epilogue) that lies inside a K loop of another one
        for i in idx:
            non = [(int(t_in[i]), int(t_loop0[i])), (int(t_loop1[i]), int(t_out[i]))]
            cov = tot = 0
            for a0, a1 in non:
                tot += a1 - a0
                for k in idx:
                    if k != i:
                        cov += max(0, min(a1, int(t_loop1[k])) - max(a0, int(t_loop0[k])))
            loop_overlap.append(cov / max(tot, 1))
    s = sum(share.values())
    print("  %d CUs; time share by resident workgroups per CU: " % len(set(cu.tolist()))
          + ", ".join("%d: %.1f %%" % (k, 100 * v / s) for k, v in sorted(share.items())))
    print("  share of a workgroup's prologue + epilogue time that runs beside another workgroup's K loop: %.0f %%" % (100 * np.mean(loop_overlap)))


if __name__ == "__main__":
    main()
