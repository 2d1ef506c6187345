#!/bin/bash
# resume-at-substep A/B (same box): restart (round 4, LM_NO_RESUME=1 on the probes build) vs resume
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
B="python bench.py --steps 150 --warmup 30 --sustained 0 --fuse 25 --no-cpu-baseline --configs off"
P='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms %.3f value %.4g fused %.4g replayed %d overflow %d nan %d parity %s" % (d["ms_per_step"], d["value"], d.get("rollout_fused", {}).get("value", 0), d["stats"]["replayed_env_steps"], d["stats"]["overflow_contacts"], d["stats"]["nan_resets"], d.get("parity")))
'
for t in HumanoidTorque.run Atlas.walk UnitreeG1.walk UnitreeH1.run Talos.walk; do
  for mode in restart resume; do
    echo "== $t $mode" >> $O/ab.txt
    if [ $mode = restart ]; then export LM_NO_RESUME=1; else unset LM_NO_RESUME; fi
    LOCOHIP_LIB=$PWD/loco_mujoco_amd/csrc/liblocohip_probes.so timeout 300 $B --task $t 2>&1 | python -c "$P" >> $O/ab.txt 2>&1
  done
done
unset LM_NO_RESUME
cat $O/ab.txt
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "replay or fused or handoff or hand_off or root_dof or sharding or bench_line or folded" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.txt
