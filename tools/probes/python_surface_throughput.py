"""Throughput of the Python surface: LocoEnv.step() with numpy actions at n_envs=4096 (host buffers, float64 in/out) and
step_device with torch tensors, against the raw rollout."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loco_mujoco_amd import LocoEnv
import __graft_entry__ as g
g.smoke()
for task, nu in (("UnitreeA1.simple", 12), ("HumanoidTorque.run", 13)):
    np.random.seed(0)
    env = LocoEnv.make(task, debug=True, n_envs=4096)
    env.reset()
    env.enable_auto_reset(seed=0)
    a = np.zeros((4096, nu)) if task.startswith("Unitree") else np.random.uniform(-1, 1, (4096, nu))
    for _ in range(20): env.step(a)
    t0 = time.perf_counter()
    for _ in range(200): obs, r, d, _ = env.step(a)
    dt = (time.perf_counter() - t0) / 200
    print("%s LocoEnv.step (numpy, host buffers): %.3f ms/step, %.0f env-steps/s" % (task, dt * 1e3, 4096 / dt))
    b = env.backend
    ta = torch.tensor(a, dtype=torch.float32, device="cuda")
    to = torch.empty((4096, obs.shape[1]), dtype=torch.float32, device="cuda"); tr = torch.empty(4096, dtype=torch.float32, device="cuda"); td = torch.empty(4096, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(20): b.step_device(ta, to, tr, td, stream=s, sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): b.step_device(ta, to, tr, td, stream=s, sync=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("%s step_device (torch tensors, caller's stream): %.3f ms/step, %.0f env-steps/s" % (task, dt * 1e3, 4096 / dt))
