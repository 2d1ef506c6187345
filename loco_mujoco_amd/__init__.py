"""
loco_mujoco_amd — MI355X-native batched drop-in for the ``LocoEnv.step()`` hot path of loco-mujoco.

    from loco_mujoco_amd import LocoEnv
    env = LocoEnv.make("UnitreeA1.simple", n_envs=4096)
    obs = env.reset(); obs, reward, absorbing, info = env.step(action)
"""

__version__ = "0.1.0"

from .environments import (Atlas, GymnasiumWrapper, HumanoidMuscle, HumanoidMuscle4Ages, HumanoidTorque,
                           HumanoidTorque4Ages, LocoEnv, Talos, UnitreeA1, UnitreeG1, UnitreeH1)


def get_all_task_names():
    return LocoEnv.get_all_task_names()
