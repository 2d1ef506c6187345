#!/bin/bash
# rocprofv3 counters of the replay kernel against the regular kernel: the same small batch with every control step through the replay kernel
TASK=${1:-Talos.walk}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/replay_prof_$TASK; mkdir -p $OUT
cat > /tmp/rp.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from loco_mujoco_amd import LocoEnv
from loco_mujoco_amd.backend import HipBatch, HipModel
task = "$TASK"
np.random.seed(0)
env = LocoEnv.make(task, debug=True)
table = env._reset_table(); hm = HipModel(env._chain_model()); nv = env._model.nv
for mode in (0, 2):
    b = HipBatch(hm, 64); b.set_replay({0: 0, 2: 4}[mode])
    rows = table[np.random.RandomState(0).randint(0, len(table), 64)]
    b.set_reset_table(table, seed=0); b.set_auto_reset(True, horizon=1000)
    b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
    if rows.shape[1] > 2 * nv: b.set_goal(rows[:, 2 * nv:])
    st = b.rollout(20, action_mode=0 if task.startswith("UnitreeA1") else 1, seed=12)
    print(mode, st["kernel_ms"] / 20)
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /tmp/rp.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python /tmp/rp.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $OUT/pmc2 -o pmc2 -- python /tmp/rp.py > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_CYCLES_VMEM -d $OUT/pmc3 -o pmc3 -- python /tmp/rp.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/trace/*.db")[0])
for r in db.execute("select name, count(*), avg(duration), min(duration), max(duration), max(lds_size), max(scratch_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(grid_x), max(workgroup_x) from kernels where name like '%step_kernel%' group by name"): print(r)
for p in sorted(glob.glob("$OUT/pmc*/*.db")):
    d = sqlite3.connect(p)
    for r in d.execute("select kernel_name, counter_name, sum(value)/count(*), count(*) from counters_collection where kernel_name like '%step_kernel%' group by kernel_name, counter_name"): print(r[0].split("step_kernel")[1][:40], r[1], r[2], r[3])
PY
