#!/bin/bash
# Round-4 measurement set (GPU box, through gpurun): the bench line, the other configurations, rocprofv3 evidence. Outputs: gpurun_out/profiles/
# (rocprofv3 runs one kernel at a time: the profiled commands use --no-pollers, the replay kernel only as the pass behind the launch)
mkdir -p gpurun_out/profiles gpurun_out/r4_bench
bash tools/probes/prof_run.sh r4 200 > gpurun_out/r4_bench/prof_r4.log 2>&1
bash tools/probes/prof_run.sh r4_HumanoidTorque.run 60 "--task HumanoidTorque.run --no-pollers --fuse 0" > gpurun_out/r4_bench/prof_r4_ht.log 2>&1
cp gpurun_out/profiles/r4_pmc.json gpurun_out/profiles/r4_kernel_stats.csv profiles/ 2>/dev/null      # the bench line below quotes the profile of this build
python bench.py > gpurun_out/profiles/r4_bench.json 2> gpurun_out/r4_bench/bench.err
for t in HumanoidTorque.run UnitreeH1.run UnitreeH1.walk Atlas.walk HumanoidMuscle.run Talos.walk UnitreeG1.walk; do
  python bench.py --task $t --steps 300 --warmup 50 > gpurun_out/profiles/r4_bench_$t.json 2>> gpurun_out/r4_bench/bench.err
done
python bench.py --task Atlas.walk --dr --envs-per-gpu 2048 --steps 300 --warmup 50 > gpurun_out/profiles/r4_bench_Atlas.walk.dr2048.json 2>> gpurun_out/r4_bench/bench.err
python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 300 --warmup 50 > gpurun_out/profiles/r4_bench_HumanoidMuscle.run.2048.json 2>> gpurun_out/r4_bench/bench.err
python bench.py --envs-per-gpu 16384 --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/profiles/r4_bench_a1_16384.json 2>> gpurun_out/r4_bench/bench.err
python bench.py --envs-per-gpu 65536 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/profiles/r4_bench_a1_65536.json 2>> gpurun_out/r4_bench/bench.err
for f in gpurun_out/profiles/r4_bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.0f" % d["value"], "%.3f ms" % d["ms_per_step"], "fused %.3f" % d.get("rollout_fused", {}).get("ms_per_step", 0), "overflow %d" % d["stats"]["overflow_contacts"],
          "replayed %d" % d["stats"].get("replayed_env_steps", -1), "selfcon %d" % d["stats"]["self_contacts"], "parity", d.get("parity", {}).get("within_tolerance"),
          "cpu %s" % (d.get("cpu_baseline", {}).get("value")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
