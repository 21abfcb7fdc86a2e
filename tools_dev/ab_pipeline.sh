#!/bin/bash
# dev: A/B of library builds on the C5 pipeline (one box, interleaved, two rounds):  tools_dev/ab_pipeline.sh ENTRY lib1.so lib2.so ...
# prints ms per step of the whole pipeline and of the entry point ENTRY (substring of its C-ABI name) per library
entry=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    python bench.py --config ${CFG:-C5} --pipeline --steps 4 --warmup 2 --lib "$lib" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
e = {k: v['ms_per_step'] for k, v in d['pipeline']['entry_points'].items() if '$entry' in k}
print('round $round %-40s step %7.3f ms  %s' % ('$lib'.split('/')[-1], d['ms_per_step'], e))"
  done
done
