#!/bin/bash
# Round-5 measurement set (GPU box, through gpurun): the bench line (with the committed profiles of this build next to it) and the other
# configurations. Outputs: gpurun_out/profiles/r5_bench*.json
mkdir -p gpurun_out/profiles gpurun_out/r5_bench
python bench.py > gpurun_out/profiles/r5_bench.json 2> gpurun_out/r5_bench/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles/r5_bench_driver_form.json 2>> gpurun_out/r5_bench/bench.err
for t in HumanoidTorque.run UnitreeH1.run UnitreeH1.walk Atlas.walk HumanoidMuscle.run Talos.walk UnitreeG1.walk; do
  python bench.py --task $t --steps 300 --warmup 50 --configs off > gpurun_out/profiles/r5_bench_$t.json 2>> gpurun_out/r5_bench/bench.err
done
python bench.py --task Atlas.walk --dr --envs-per-gpu 2048 --steps 300 --warmup 50 --configs off > gpurun_out/profiles/r5_bench_Atlas.walk.dr2048.json 2>> gpurun_out/r5_bench/bench.err
python bench.py --task HumanoidMuscle.run --envs-per-gpu 2048 --steps 300 --warmup 50 --configs off > gpurun_out/profiles/r5_bench_HumanoidMuscle.run.2048.json 2>> gpurun_out/r5_bench/bench.err
python bench.py --envs-per-gpu 16384 --steps 300 --warmup 50 --no-cpu-baseline --configs off > gpurun_out/profiles/r5_bench_a1_16384.json 2>> gpurun_out/r5_bench/bench.err
python bench.py --envs-per-gpu 65536 --steps 100 --warmup 20 --no-cpu-baseline --configs off > gpurun_out/profiles/r5_bench_a1_65536.json 2>> gpurun_out/r5_bench/bench.err
for f in gpurun_out/profiles/r5_bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.0f" % d["value"], "%.3f ms" % d["ms_per_step"], "fused %.3f" % d.get("rollout_fused", {}).get("ms_per_step", 0), "overflow %d" % d["stats"]["overflow_contacts"],
          "replayed %d" % d["stats"].get("replayed_env_steps", -1), "selfcon %d own %d" % (d["stats"]["self_contacts"], d["stats"].get("own_manifold_contacts", -1)), "parity", d.get("parity", {}).get("within_tolerance"), d.get("parity", {}).get("ill_conditioned"),
          "cpu %s" % (d.get("cpu_baseline", {}).get("value")), "traffic", d["roofline"].get("traffic"), "binding", (d["roofline"].get("binding") or {}).get("valu_issue_frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
