#!/bin/bash
# dev helper: issue-side PMC passes (VALU / LDS / VMEM utilisation) over a short bench run
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_issue
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_issue'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'waldo::warp' not in r['Kernel_Name']: continue
        k = r['Kernel_Name'].split('(')[0][-50:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print('   %-32s %.5g' % (c, sum(v) / len(v)))
PY
