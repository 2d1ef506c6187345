#!/bin/bash
# Round-6 closing measurement set (GPU box, through gpurun): rocprofv3 passes of the four BASELINE configurations (stamped with the library's
# sha256, copied into profiles/ on the box so that the bench lines that follow quote them), then the bench line in its default and in the
# driver's form, then every other configuration. Outputs: gpurun_out/profiles/r6_*.
mkdir -p gpurun_out/profiles gpurun_out/r6_bench
bash tools/probes/prof_run.sh r6 200 > gpurun_out/r6_bench/prof_r6.log 2>&1
bash tools/probes/prof_run.sh r6_HumanoidTorque.run 60 "--task HumanoidTorque.run --no-pollers --fuse 0" > gpurun_out/r6_bench/prof_r6_ht.log 2>&1
bash tools/probes/prof_run.sh r6_Atlas.walk.dr2048 100 "--task Atlas.walk --dr --envs-per-gpu 2048 --no-pollers --fuse 0" > gpurun_out/r6_bench/prof_r6_atlas.log 2>&1
bash tools/probes/prof_run.sh r6_HumanoidMuscle.run2048 100 "--task HumanoidMuscle.run --envs-per-gpu 2048 --no-pollers --fuse 0" > gpurun_out/r6_bench/prof_r6_hm.log 2>&1
ls -la profiles/r6_* gpurun_out/profiles/ 2>&1 | tail -12
python bench.py > gpurun_out/profiles/r6_bench.json 2> gpurun_out/r6_bench/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles/r6_bench_driver_form.json 2>> gpurun_out/r6_bench/bench.err
for t in HumanoidTorque.run UnitreeH1.run UnitreeH1.walk Atlas.walk HumanoidMuscle.run Talos.walk UnitreeG1.walk; do
  python bench.py --task $t --steps 300 --warmup 50 --configs off > gpurun_out/profiles/r6_bench_$t.json 2>> gpurun_out/r6_bench/bench.err; echo "$t rc $?" >> gpurun_out/r6_bench/rc.txt
done
python bench.py --task Atlas.walk --dr --envs-per-gpu 2048 --steps 300 --warmup 50 --configs off > gpurun_out/profiles/r6_bench_Atlas.walk.dr2048.json 2>> gpurun_out/r6_bench/bench.err
python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 300 --warmup 50 --configs off > gpurun_out/profiles/r6_bench_HumanoidMuscle.run.2048.json 2>> gpurun_out/r6_bench/bench.err
python bench.py --envs-per-gpu 16384 --steps 300 --warmup 50 --no-cpu-baseline --configs off > gpurun_out/profiles/r6_bench_a1_16384.json 2>> gpurun_out/r6_bench/bench.err
python bench.py --envs-per-gpu 65536 --steps 100 --warmup 20 --no-cpu-baseline --configs off > gpurun_out/profiles/r6_bench_a1_65536.json 2>> gpurun_out/r6_bench/bench.err
cat gpurun_out/r6_bench/rc.txt
for f in gpurun_out/profiles/r6_bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.0f" % d["value"], "%.3f ms" % d["ms_per_step"], "fused %.3f" % d.get("rollout_fused", {}).get("ms_per_step", 0), "overflow %d" % d["stats"]["overflow_contacts"],
          "replayed %d" % d["stats"].get("replayed_env_steps", -1), "selfcon %d own %d" % (d["stats"]["self_contacts"], d["stats"].get("own_manifold_contacts", -1)), "parity", d.get("parity", {}).get("within_tolerance"), d.get("parity", {}).get("ill_conditioned"),
          "cpu %s" % (d.get("cpu_baseline", {}).get("value")), "traffic", d["roofline"].get("traffic"), "binding", (d["roofline"].get("binding") or {}).get("valu_issue_frac"), "surface", (d.get("python_surface") or {}).get("over_kernel_rate"))
    for k, c in (d.get("configs") or {}).items():
        print("   ", k, "%.0f" % c.get("value", 0), "%.3f ms" % c.get("ms_per_step", 0), "fused", (c.get("rollout_fused") or {}).get("ms_per_step"), "parity", (c.get("parity") or {}).get("within_tolerance"), "traffic", (c.get("roofline") or {}).get("traffic"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
