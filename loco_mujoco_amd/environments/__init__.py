from .base import LocoEnv, ValidTaskConf
from .unitree_a1 import UnitreeA1
from .gymnasium import GymnasiumWrapper

UnitreeA1.register()
