#!/bin/bash
# rocprofv3 counters of the pair-family kernel under a random policy (I-cache, waits, instruction mix). gpurun_out/r3_pmc/
TASK=${1:-HumanoidTorque.run}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3_pmc
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/probes/r3/ht_step_times.py $TASK 8 1"
cd $GRAFT_REPO_ROOT
(cd /tmp; rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $OUT/pmc5 -o pmc5 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/pmc5.out 2> $OUT/pmc5.err)
(cd /tmp; rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/pmc1.out 2> $OUT/pmc1.err)
(cd /tmp; rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_LDS -d $OUT/pmc2 -o pmc2 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/pmc2.out 2> $OUT/pmc2.err)
cd $OUT
python - <<'PY'
import csv, glob, collections, json
res = {}
for tag in ("pmc1", "pmc2", "pmc5"):
    for f in glob.glob(tag + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        rows = list(csv.DictReader(open(f)))
        # the last dispatches are the contact-rich ones: keep per-dispatch lists
        per = collections.defaultdict(list)
        for row in rows:
            if "step_kernel" not in row.get("Kernel_Name", ""): continue
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
        res[tag] = {k: v for k, v in per.items()}
json.dump(res, open("summary.json", "w"))
for tag, d in res.items():
    for k, v in d.items():
        print(tag, k, ["%.3g" % x for x in v])
PY
