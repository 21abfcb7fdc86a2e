#!/bin/bash
# dev helper: kernel stats + HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + issue-side
# counters of every waldo:: kernel an arbitrary script launches.
#   tools_dev/pmc_any.sh NAME script.py [args]   ->  gpurun_out/pmc_NAME/summary.txt
# FETCH_SIZE is printed x2 (gfx950 tallies 128-byte requests at 64 bytes for wide coalesced reads:
# MI355X_MICROARCH.md; narrow gathers are uncalibrated -- read ratios, not absolutes).
cd /tmp && export TMPDIR=/tmp
name=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rm -rf $OUT; mkdir -p $OUT
script=$GRAFT_REPO_ROOT/$1; shift
export PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $script "$@" > $OUT/stats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $script "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
out = "$OUT"
dur = {}
for f in glob.glob(out + '/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'waldo' in r['Name']:
            dur[r['Name'].split('(')[0]] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'waldo' not in r['Kernel_Name']: continue
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg, key=lambda k: -dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1]):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    calls, us = dur.get(k, (0, 0.0))
    fetch, write = 2.0 * c.get('FETCH_SIZE', 0) * 1024, c.get('WRITE_SIZE', 0) * 1024
    print(f"{k[-70:]}\n    calls {calls}  avg {us:.1f} us   fetch(x2) {fetch / 1e6:.1f} MB  write {write / 1e6:.1f} MB"
          f"  -> {(fetch + write) / max(us, 1e-9) / 1e6:.2f} TB/s moved")
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8.0   # summed over the 8 XCDs
    if cyc > 0:
        valu = c.get('SQ_INSTS_VALU', 0) * 4 / 1024 / cyc
        lds = c.get('SQ_ACTIVE_INST_LDS', 0) * 4 / 256 / cyc
        print(f"    VALU issue {100 * valu:.0f} % of the SIMD cycles, LDS pipe {100 * lds:.0f} % of the CU cycles, "
              f"bank-conflict cycles {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / 256 / cyc:.0f} %, "
              f"waves {c.get('SQ_WAVES', 0):.0f}, VALU / wave {c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_WAVES', 1), 1):.0f}")
PY
cat $OUT/summary.txt
