#!/bin/bash
# Round 2, experiment 1: which source-level change (if any) makes the -O2 / -O3 builds pass the GPU parity suite?
#   variants/O3_base.so            -O3, source as shipped in round 1
#   variants/O3_volatile.so        -O3, lane memory accessed through `volatile float*` (no caching / forwarding / merging)
#   variants/O3_updpp.so           -O3, quad sums with update_dpp(old = 0) instead of mov_dpp (undefined old)
#   variants/O2_base.so, O2_volatile_updpp.so   the -O2 failure (per-environment parameters, 4 replicas)
OUT=gpurun_out/r2_ab
mkdir -p $OUT
for v in O3_base O3_volatile O3_updpp; do
  [ -f variants/$v.so ] || continue
  LOCOHIP_LIB=$PWD/variants/$v.so timeout 900 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider > $OUT/$v.log 2>&1
  echo "=== $v: $(tail -1 $OUT/$v.log)"
  grep -E "^(FAILED|ERROR)|Error|assert" $OUT/$v.log | head -40
done
for v in O2_base O2_volatile_updpp; do
  [ -f variants/$v.so ] || continue
  LOCOHIP_LIB=$PWD/variants/$v.so timeout 600 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider -k "per_environment or redraw or fused" > $OUT/$v.log 2>&1
  echo "=== $v: $(tail -1 $OUT/$v.log)"
  grep -E "^(FAILED|ERROR)" $OUT/$v.log | head -20
done
# instruction-cache counters of the shipped build
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$OUT/avail.txt 2>&1
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$OUT/pmc_icache -o pmc -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_icache.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_icache.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES -d $GRAFT_REPO_ROOT/$OUT/pmc_wait -o pmc -- $CMD > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_wait.err
)
python - <<'PY'
import glob, sqlite3, json
for tag in ("pmc_icache", "pmc_wait"):
    for p in glob.glob("gpurun_out/r2_ab/%s/**/*.db" % tag, recursive=True):
        d = sqlite3.connect(p)
        try:
            rows = d.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        except Exception as e:
            print(tag, "query failed", e); continue
        for k, c, s, n in rows:
            if "step_kernel" in k:
                print(tag, k.split("step_kernel")[1][:40], c, "per dispatch %.4g" % (s / n), "dispatches", n)
PY
