"""Dev measurement: shader clock and socket power while ONE kind of conv launch runs back to back for a few seconds
(amdgpu hwmon, as bench.py's device_state), next to the wall time per launch -- which layers run at the socket's
power limit and which do not.  Blocks 0..k of the stack are run through nhans_debug_block_output; the difference between
consecutive k isolates a block.   python tools/clock_by_kernel.py [frames] [seconds]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
import bench  # noqa: E402
from nhans_amd import engine, synth  # noqa: E402
from nhans_amd.apply import normalise, trim_to_frames  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    eng = engine.Engine("denoiser", precision="f16x3")
    mix = trim_to_frames(normalise(synth.mixture(0, frames / 100.0 + 0.1)))
    lm, _ = eng.stft_features(torch.from_numpy(mix).cuda(), [0, len(mix)])
    ea = torch.zeros(1, 512, device="cuda")
    eng.set_option("frames_per_chunk", frames)
    out = {}
    for wino in (1, 0):
        eng.set_option("winograd", wino)
        for block in (0, 1, 3, 5, 7, 8):
            eng.block_output(lm, [0, lm.shape[0]], ea, ea, 0, frames, block)      # warm
            torch.cuda.synchronize()
            smp = bench.DeviceSampler(0, period=0.05)
            smp.start()
            t0 = time.time()
            n = 0
            while time.time() - t0 < seconds:
                eng.block_output(lm, [0, lm.shape[0]], ea, ea, 0, frames, block)
                torch.cuda.synchronize()
                n += 1
            dt = (time.time() - t0) / n
            st = smp.stop() or {}
            out["winograd=%d blocks 0..%d" % (wino, block)] = {
                "ms_per_pass": round(dt * 1e3, 2), "sclk_mhz_mean": round(st.get("sclk_mhz_mean", 0)),
                "sclk_mhz_min": st.get("sclk_mhz_min"), "socket_power_w_mean": round(st.get("socket_power_w_mean", 0)),
                "socket_power_w_max": st.get("socket_power_w_max"), "power_cap_w": st.get("power_cap_w")}
            if wino == 0 and block >= 5:
                pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
