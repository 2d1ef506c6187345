"""The README's usage example, executed."""
import numpy as np
import torch
from loco_mujoco_amd import LocoEnv
env = LocoEnv.make("UnitreeA1.simple", n_envs=4096)
obs = env.reset()
obs, reward, absorbing, info = env.step(np.zeros((4096, 12)))
env.enable_auto_reset(seed=0)
b = env.backend
act = torch.zeros(4096, 12, device="cuda"); obs_t = torch.empty(4096, 37, device="cuda")
rew = torch.empty(4096, device="cuda"); done = torch.empty(4096, dtype=torch.uint8, device="cuda")
for _ in range(5):
    b.step_device(act, obs_t, rew, done, stream=torch.cuda.current_stream().cuda_stream, sync=False)
torch.cuda.synchronize()
terminal, ended = (done & 1).bool(), (done & 2).bool()        # the done byte is a bit field (include/locohip.h)
st = b.rollout(100, action_mode=1, steps_per_launch=25)
print("ok", obs.shape, float(obs_t.abs().max()), int(terminal.sum()), int(ended.sum()), st["env_steps"], st["nan_resets"])
