from .base import LocoEnv, ValidTaskConf
from .unitree_a1 import UnitreeA1
from .atlas import Atlas
from .humanoids import BaseHumanoid, HumanoidMuscle, HumanoidTorque
from .gymnasium import GymnasiumWrapper

UnitreeA1.register()
Atlas.register()
HumanoidTorque.register()
HumanoidMuscle.register()
