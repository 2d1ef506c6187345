#!/bin/bash
# rocprofv3 evidence for the bench command (run on the GPU box through gpurun). Outputs under gpurun_out/prof_<tag>/
TAG=${1:-r1}
STEPS=${2:-200}
EXTRA=${3:-}            # e.g. "--task HumanoidTorque.run" or "--task Atlas.walk --dr --envs-per-gpu 2048"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --configs off --sustained 0 --fuse 0 $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- $CMD > /dev/null 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_IFETCH SQ_WAIT_INST_LDS -d $OUT/pmc2 -o pmc2 -- $CMD > /dev/null 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $CMD > /dev/null 2> $OUT/pmc3.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $CMD > /dev/null 2> $OUT/pmc4.err
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $OUT/pmc5 -o pmc5 -- $CMD > /dev/null 2> $OUT/pmc5.err
cd $OUT
find . -name "*.csv" | head -30
python - <<'PY'
import csv, glob, collections, json, os
res = {}
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    res["kernel_stats"] = list(csv.DictReader(open(f)))
for tag in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
    for f in glob.glob(tag + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "step_kernel" not in row.get("Kernel_Name", ""): continue
            acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
        res[tag] = {k: dict(sum=v[0], dispatches=v[1], per_dispatch=v[0] / max(v[1], 1)) for k, v in acc.items()}
json.dump(res, open("summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY

cd $GRAFT_REPO_ROOT && python tools/summarize_rocprof.py gpurun_out/prof_$TAG $TAG > /dev/null 2>&1 && mkdir -p gpurun_out/profiles && cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json gpurun_out/profiles/
