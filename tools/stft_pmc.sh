#!/bin/bash
# rocprofv3 --pmc passes over tools/stft_bench.py (STFT / iSTFT alone, 256 clips); per-kernel sums -> gpurun_out/<tag>/stft_pmc.txt
set -u
TAG=${1:-stftpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/spmc_$i -- python3 $GRAFT_REPO_ROOT/tools/stft_bench.py 256 > /dev/null 2> $OUT/pass_$i.err); echo "pass $i rc=$?"
done
python3 - <<'PY' > $OUT/stft_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/spmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'stft' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in acc:
    print(k)
    n = None
    for c, v in sorted(acc[k].items()):
        # each dispatch contributes one row per counter (summed over XCDs by the tool or one row per dimension)
        print("   %-28s %.4e  (rows %d)" % (c, v, cnt[k][c]))
PY
cat $OUT/stft_pmc.txt
