cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/st_p; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_p -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --pipeline --steps 3 --warmup 1 --no-cpu-baseline > /tmp/st_p.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/st_p/*/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    steps = 4
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total ms/step', tot / 1e6 / steps)
    for r in rows:
        if 'waldo' in r['Name']: continue
        ms = float(r['TotalDurationNs']) / 1e6 / steps
        if ms > 0.03: print(f"{r['Name'][:130]:130s} {int(r['Calls'])/steps:6.1f} {ms:7.3f}")
PY
