from .base import LocoEnv, ValidTaskConf
from .unitree_a1 import UnitreeA1
from .atlas import Atlas
from .humanoids import BaseHumanoid, BaseHumanoid4Ages, HumanoidMuscle, HumanoidMuscle4Ages, HumanoidTorque, HumanoidTorque4Ages
from .talos import Talos
from .unitree_h1 import UnitreeG1, UnitreeH1
from .gymnasium import GymnasiumWrapper

UnitreeA1.register()
Atlas.register()
Talos.register()
UnitreeH1.register()
UnitreeG1.register()
HumanoidTorque.register()
HumanoidMuscle.register()
HumanoidTorque4Ages.register()
HumanoidMuscle4Ages.register()
