"""Sum rocprofv3 --pmc counter_collection CSVs per kernel and derive MFMA-busy / HBM bytes per launch.
    python tools/pmc_summary.py out.json passA_dir passC_dir passD_dir
The passes (each its own run, counters only -- MI355X_MICROARCH.md, HBM/rocprofv3 section):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d passA -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d passC -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d passD -- python3 bench.py ...
Derived: MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD x 1024 SIMDs);
HBM read bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction), write bytes = WRITE_SIZE x 1024."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys


def kernel_source_sha16(root):
    """Fingerprint of what the counters describe: every kernel source of the library and the weight packing.  bench.py
    recomputes it and quotes `traffic` only from a summary whose fingerprint equals its own."""
    h = hashlib.sha256()
    d = os.path.join(root, "n-hans_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(root, "n-hans_amd", "fold.py"), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].strip()


def main():
    if sys.argv[1] == "--stamp-commit":                    # python tools/pmc_summary.py --stamp-commit summary.json  (on the dev box)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = json.load(open(sys.argv[2]))
        assert d["_meta"]["kernel_source_sha16"] == kernel_source_sha16(root), "the tree's kernel sources are not the profiled ones"
        d["_meta"]["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"]).decode().strip()
        json.dump(d, open(sys.argv[2], "w"), indent=1, sort_keys=True)
        print(d["_meta"])
        return
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for tag, d in zip("acd", dirs):
        seen = collections.defaultdict(set)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r["Dispatch_Id"])
        for k, s in seen.items():
            agg[k]["dispatches_pass_" + tag] = len(s)
    res = {}
    for k, c in agg.items():
        c = dict(c)
        n = max(c.get("dispatches_pass_a", 0), 1)
        if c.get("GRBM_GUI_ACTIVE"):
            c["derived_mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "FETCH_SIZE" in c:
            c["derived_hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0 * 2.0 / max(c.get("dispatches_pass_c", n), 1)
        if "WRITE_SIZE" in c:
            c["derived_hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / max(c.get("dispatches_pass_d", n), 1)
        res[k] = c
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = {"kernel_source_sha16": kernel_source_sha16(root)}
    try:                                                   # (the GPU box has no .git: the commit is stamped afterwards by the same script)
        meta["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        meta["commit"] = None
    res["_meta"] = meta
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    res.pop("_meta")
    for k, c in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        print("%-46s launches %3d  mfma_busy %.3f  hbm_rd/launch %.3e  hbm_wr/launch %.3e" % (
            k[-46:], c.get("dispatches_pass_a", 0), c.get("derived_mfma_busy_frac", 0.0),
            c.get("derived_hbm_read_bytes_per_launch", 0.0), c.get("derived_hbm_write_bytes_per_launch", 0.0)))


if __name__ == "__main__":
    main()
