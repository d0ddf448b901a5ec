"""Within-process A/B of conv kernel choices on the bench workload (interleaved rounds, per-kernel
hipEvent times from the library's profiler).
    python tools/ab_variants.py [--clips 32] [--rounds 3] [--variants 2 3]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, weights  # noqa: E402
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--clips", type=int, default=32)
    p.add_argument("--rounds", type=int, default=3)
    p.add_argument("--variants", type=int, nargs="+", default=[2, 3])
    p.add_argument("--kind", default="denoiser")
    p.add_argument("--option", default="conv_variant", help="the nhans_set_option key to A/B (values: --variants)")
    a = p.parse_args()
    bench._rank_imports()
    W = weights.synthetic_weights(a.kind, 7)
    eng = engine.Engine(a.kind, W, precision="f16x3")
    mixes, ca, cb, _ = bench.make_batch(a.kind, 0, a.clips, 10.0, min(a.clips, 8))
    mix_t, mix_off = eng._dev(mixes)
    ca_t, ca_off = eng._dev(ca)
    cb_t, cb_off = eng._dev(cb)
    run = lambda: eng.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off)
    run()
    torch.cuda.synchronize()
    # the knob must not change a bit of the result
    outs = []
    for v in a.variants:
        eng.set_option(a.option, v)
        outs.append(run()["denoised_wav"].clone())
    torch.cuda.synchronize()
    for v, o in zip(a.variants[1:], outs[1:]):
        print("variant %d vs %d: bit-identical = %s (max |diff| %.3g)" % (v, a.variants[0], bool(torch.equal(o, outs[0])),
                                                                       float((o - outs[0]).abs().max())))
    del outs
    eng.set_option("profile", 1)
    res = {v: [] for v in a.variants}
    for r in range(a.rounds):
        for v in a.variants:
            eng.set_option(a.option, v)
            eng.profile_reset()
            run()
            torch.cuda.synchronize()
            res[v].append(eng.profile())
    for v in a.variants:
        names = sorted({k for pr in res[v] for k in pr})
        print("variant %d" % v)
        tot = []
        for pr in res[v]:
            tot.append(sum(x["ms"] for x in pr.values()))
        print("  total kernel ms per pass: " + " ".join("%.1f" % t for t in tot))
        for k in names:
            ms = [pr[k]["ms"] for pr in res[v] if k in pr]
            fl = res[v][0][k]["flops"]
            best = min(ms)
            print("  %-32s ms %s  calls %d  TF(best) %s" % (k, " ".join("%.2f" % m for m in ms), res[v][0][k]["calls"],
                                                          "%.1f" % (fl / best / 1e9) if fl else "-"))
    print(json.dumps({str(v): [sum(x["ms"] for x in pr.values()) for pr in res[v]] for v in a.variants}))


if __name__ == "__main__":
    main()
