#!/bin/bash
# HBM traffic of the step kernel with and without the XCD-aware workgroup -> environment mapping (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
for mode in map nomap; do
  [ $mode = nomap ] && export LM_NO_XCD_MAP=1 || unset LM_NO_XCD_MAP
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$GRAFT_REPO_ROOT/gpurun_out/xcd_${mode}_$c
    rocprofv3 --pmc $c -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --fuse 0 > $OUT.json 2> $OUT.err
    python - $OUT $c $mode <<'PY'
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
tot, n = db.execute("select sum(value), count(*) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", (sys.argv[2],)).fetchone()
print("%s %s: %.1f KiB per launch (%d launches)" % (sys.argv[3], sys.argv[2], tot / n, n))
PY
  done
  python -c "import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/xcd_${mode}_WRITE_SIZE.json')); print('$mode', d['ms_per_step'], 'ms')"
done
