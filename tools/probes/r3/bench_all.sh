#!/bin/bash
# Round-3 measurement set (GPU box, through gpurun): the bench line, the other configurations, rocprofv3 evidence. Outputs: gpurun_out/profiles/
mkdir -p gpurun_out/profiles gpurun_out/r3_bench
python bench.py > gpurun_out/profiles/r3_bench.json 2> gpurun_out/r3_bench/bench.err
for t in HumanoidTorque.run UnitreeH1.run UnitreeH1.walk Atlas.walk HumanoidMuscle.run Talos.walk UnitreeG1.walk; do
  python bench.py --task $t --steps 300 --warmup 50 > gpurun_out/profiles/r3_bench_$t.json 2>> gpurun_out/r3_bench/bench.err
done
python bench.py --task Atlas.walk --dr --envs-per-gpu 2048 --steps 300 --warmup 50 > gpurun_out/profiles/r3_bench_Atlas.walk.dr2048.json 2>> gpurun_out/r3_bench/bench.err
python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 300 --warmup 50 > gpurun_out/profiles/r3_bench_HumanoidMuscle.run.2048.json 2>> gpurun_out/r3_bench/bench.err
python bench.py --envs-per-gpu 16384 --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/profiles/r3_bench_a1_16384.json 2>> gpurun_out/r3_bench/bench.err
python bench.py --envs-per-gpu 65536 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/profiles/r3_bench_a1_65536.json 2>> gpurun_out/r3_bench/bench.err
bash tools/probes/prof_run.sh r3 200 > gpurun_out/r3_bench/prof_r3.log 2>&1
bash tools/probes/prof_run.sh r3_HumanoidTorque.run 100 "--task HumanoidTorque.run" > gpurun_out/r3_bench/prof_r3_ht.log 2>&1
for f in gpurun_out/profiles/r3_bench*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "%.0f" % d["value"], "%.3f ms" % d["ms_per_step"], "fused %.3f" % d.get("rollout_fused", {}).get("ms_per_step", 0), "overflow %d" % d["stats"]["overflow_contacts"],
      "cpu %s" % (d.get("cpu_baseline", {}).get("value")))
PY
done
