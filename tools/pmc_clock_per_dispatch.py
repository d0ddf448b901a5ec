"""Dev tool: effective shader clock of the individual conv launches, from one rocprofv3 --pmc pass:
    (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_clk -- python3 $REPO/bench.py \
         --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-pass --no-ceiling) ; python tools/pmc_clock_per_dispatch.py /tmp/pmc_clk
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration = the clock the launch actually ran at; grouped by
kernel and grid size (the 64- and 128-channel Winograd layers are the same kernel with different grids)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cnt = collections.defaultdict(dict)
meta = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        cnt[k][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[k] = (r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size"), r.get("Start_Timestamp"), r.get("End_Timestamp"))
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
grp = collections.defaultdict(list)
for k, c in cnt.items():
    name, grid, st, en = meta[k]
    ns = dur.get(k)
    if ns is None and st and en:
        ns = float(en) - float(st)
    if not ns or "conv_" not in name:
        continue
    grp[(name, grid)].append((c.get("GRBM_GUI_ACTIVE", 0) / 8 / ns * 1e3, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8, 1) / 256 / 4, ns / 1e6))
for (name, grid), v in sorted(grp.items()):
    n = len(v)
    print("%-62s grid %-8s n=%3d  clock %6.0f MHz  mfma-busy %.3f  %.3f ms" % (name, grid, n, sum(a for a, _, _ in v) / n, sum(b for _, b, _ in v) / n, sum(c for _, _, c in v) / n))
