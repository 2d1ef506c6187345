"""LDS capacity probe: HumanoidTorque one-step KAT error for several envs-per-workgroup settings."""
import os, sys, subprocess
for epb in ("1", "4", "8", "16"):
    env = dict(os.environ, LM_ENVS_PER_BLOCK=epb)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-q", "-s", "-k", "humanoid_torque_one_control_step_kats and run"], env=env, capture_output=True, text=True)
    print("epb", epb, [l for l in r.stdout.splitlines() if "KAT errors" in l or "passed" in l or "failed" in l or "Error" in l][:4])
