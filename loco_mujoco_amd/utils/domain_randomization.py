"""
Per-episode randomisation of joint parameters — host-side mirror of the reference's
``loco_mujoco/utils/domain_randomization.py`` for the part that the shipped configurations use
(``environments/data/*/domain_randomization_*.yaml``: joint ``damping`` ranges, everything else sigma 0).

The reference re-compiles a MuJoCo model per reset from a randomised XML (``base.py:183-185``,
``domain_randomization.py:219-227``). Here the three joint parameters that do not change any compile-time constant
(``damping``, ``stiffness``, ``frictionloss``; ``dof_invweight0`` and the row regularisers only depend on inertia and
``armature``) are **per-environment arrays on the device**: the host draws them at ``reset()`` with ``np.random`` and the
step kernel redraws them itself when it restarts an episode (counter-based RNG). ``armature``, ``Inertial`` and ``Geoms``
randomisation would need the compile-time constants per environment and raise ``NotImplementedError`` when a
configuration asks for a non-zero spread.

Distributions, including the reference's quirks (``domain_randomization.py:299-383``):
  sigma s               -> clip(N(default, s), 0, inf)
  uniform_range [a, b]  -> U(a, b) for damping, **N(a, b)** for stiffness / frictionloss
  uniform_range_delta d -> U(default-d, default+d) for damping, **N(default-d, default+d)** for the others
"""

import numpy as np
import yaml

PARAMS = ("damping", "stiffness", "frictionloss")
KIND_NONE, KIND_CLIPPED_NORMAL, KIND_UNIFORM, KIND_NORMAL = 0, 1, 2, 3


class JointRandomization:
    """Parsed configuration: ``spec[param_index, dof] = (kind, a, b)`` for the dofs of a compiled model."""

    def __init__(self, model, config_path):
        with open(config_path, "r") as f:
            config = yaml.safe_load(f) or {}
        for section in ("Inertial", "Geoms"):
            if self._has_spread(config.get(section)) or self._has_spread((config.get("Default") or {}).get(section)):
                raise NotImplementedError("domain randomisation of <%s> needs per-environment compile-time constants "
                                          "(not built)" % section.lower())
        joints = config.get("Joints") or {}
        default = config.get("Default") or {}
        nominal = dict(damping=model.dof_damping, stiffness=model.jnt_stiffness, frictionloss=model.dof_frictionloss)
        self.spec = np.zeros((len(PARAMS), model.nv, 3))
        for d, name in enumerate(model.jnt_names):
            if name in joints:
                conf = joints[name]
            elif "Joints" in default and "exclude" in default and name not in default["exclude"]:
                conf = default["Joints"]
            else:
                continue
            for param, rule in (conf or {}).items():
                kinds = [k for k in ("sigma", "uniform_range", "uniform_range_delta") if k in rule]
                assert len(kinds) == 1, "Exactly one parameter should be provided for joint %s (%s)" % (name, param)
                if param == "armature":
                    if self._has_spread({param: rule}):
                        raise NotImplementedError("armature randomisation changes dof_invweight0 (not built)")
                    continue
                if param not in PARAMS:
                    raise ValueError("Parameter %s currently nor supported for domain randomization." % param)
                p, base = PARAMS.index(param), float(nominal[param][d])
                if "sigma" in rule:
                    if float(rule["sigma"]) != 0.0:
                        self.spec[p, d] = (KIND_CLIPPED_NORMAL, base, float(rule["sigma"]))
                elif "uniform_range" in rule:
                    low, high = rule["uniform_range"]
                    assert high > low and low >= 0.0, "uniform_range for %s wrongly specified" % name
                    self.spec[p, d] = (KIND_UNIFORM if param == "damping" else KIND_NORMAL, low, high)
                else:
                    delta = rule["uniform_range_delta"]
                    assert type(delta) == float, "uniform_range_delta parameter for %s should be a float" % name
                    assert base - delta > 0.0, "uniform_range_delta param (%g) for joint %s is bigger than %s" % (delta, name, param)
                    self.spec[p, d] = (KIND_UNIFORM if param == "damping" else KIND_NORMAL, base - delta, base + delta)
        self.nominal = np.stack([np.asarray(nominal[p], dtype=np.float64) for p in PARAMS])

    @staticmethod
    def _has_spread(section):
        """True if any rule below ``section`` asks for a non-degenerate distribution."""
        if not isinstance(section, dict):
            return False
        for key, value in section.items():
            if key == "sigma":
                if np.any(np.asarray(value, dtype=np.float64) != 0.0):
                    return True
            elif key in ("uniform_range", "uniform_range_delta"):
                return True
            elif JointRandomization._has_spread(value):
                return True
        return False

    @property
    def active(self):
        return bool((self.spec[:, :, 0] != KIND_NONE).any())

    def sample(self, n=1):
        """(3, n, nv) damping / stiffness / frictionloss drawn with ``np.random`` (the reference's generator)."""
        out = np.repeat(self.nominal[:, None, :], n, axis=1)
        for p in range(len(PARAMS)):
            for d in np.nonzero(self.spec[p, :, 0])[0]:
                kind, a, b = self.spec[p, d]
                if kind == KIND_CLIPPED_NORMAL:
                    out[p, :, d] = np.clip(np.random.normal(a, b, n), 0.0, np.inf)
                elif kind == KIND_UNIFORM:
                    out[p, :, d] = np.random.uniform(a, b, n)
                else:
                    # the reference's N(a, b) quirk can draw negative values, which no joint parameter may take: clipped at 0
                    # here and in the kernel's redraw
                    out[p, :, d] = np.clip(np.random.normal(a, b, n), 0.0, np.inf)
        return out
