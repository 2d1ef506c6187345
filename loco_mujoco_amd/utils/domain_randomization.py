"""
Per-episode domain randomisation — host-side mirror of the reference's ``loco_mujoco/utils/domain_randomization.py``.

The reference re-compiles a MuJoCo model per reset from a randomised XML (``base.py:183-185``,
``domain_randomization.py:219-227``). Here

* the three joint parameters that do not change any compile-time constant (``damping``, ``stiffness``, ``frictionloss``) are
  **per-environment arrays on the device**: the host draws them at ``reset()`` with ``np.random`` and the step kernel redraws
  them itself when it restarts an episode (counter-based RNG);
* everything that DOES change compile-time constants — joint ``armature``, ``Inertial`` (mass, diaginertia, fullinertia) and
  ``Geoms`` friction: ``dof_invweight0`` / ``body_invweight0`` and every regulariser derived from them — comes from a **pool of
  model variants**: ``n_model_variants`` randomised models are compiled on the host once (``mjcf.model_variant``: the
  reference's recompile, without the XML round trip), lowered, and uploaded as per-variant tables
  (``lowering.variant_tables``); every environment holds the index of its variant, drawn at ``reset()`` and redrawn by the
  step kernel at each device-side restart. The reference's worker pool pre-builds models the same way
  (``domain_randomization.py:196-217``), with a fresh draw per model instead of a fixed pool.
  ``Geoms`` ``mass`` / ``density`` only matter for bodies without an ``<inertial>`` element; none of the suite's robots
  has one, and a rule on such a body raises ``NotImplementedError``.

Distributions, including the reference's quirks (``domain_randomization.py:299-383,386-514``):
  joints   sigma s               -> clip(N(default, s), 0, inf)
           uniform_range [a, b]  -> U(a, b) for damping, **N(a, b)** for stiffness / frictionloss / armature
           uniform_range_delta d -> U(default-d, default+d) for damping, **N(default-d, default+d)** for the others
  inertial mass: sigma / uniform_range / uniform_range_delta as written; diaginertia: U(value-d, value+d) per component;
           fullinertia: the SVD of the UPPER-TRIANGULAR matrix [[xx xy xz] [0 yy yz] [0 0 zz]] gets its singular values
           redrawn U(s-d, s+d) and the six numbers are read back from U diag(s') V^T (``:500-513``) — a body whose
           ``<inertial>`` has no ``fullinertia`` attribute fails the reference's assertion, and fails it here
  geoms    friction: sigma (3 numbers) -> clip(N(friction, sigma), 0, inf); uniform_range_delta (3 numbers) -> U(f-d, f+d)
"""

import numpy as np
import yaml

PARAMS = ("damping", "stiffness", "frictionloss")
KIND_NONE, KIND_CLIPPED_NORMAL, KIND_UNIFORM, KIND_NORMAL = 0, 1, 2, 3


class JointRandomization:
    """Parsed configuration: ``spec[param_index, dof] = (kind, a, b)`` for the dofs of a compiled model (joint damping /
    stiffness / frictionloss, per environment on the device) + the rules that need a pool of model variants."""

    def __init__(self, model, config_path):
        with open(config_path, "r") as f:
            config = yaml.safe_load(f) or {}
        self._model = model
        joints = config.get("Joints") or {}
        default = config.get("Default") or {}
        nominal = dict(damping=model.dof_damping, stiffness=model.jnt_stiffness, frictionloss=model.dof_frictionloss)
        self.spec = np.zeros((len(PARAMS), model.nv, 3))
        self.armature_rules = []          # (dof, kind, a, b)
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
                    base = float(model.dof_armature[d])
                    if "sigma" in rule:
                        if float(rule["sigma"]) != 0.0:
                            self.armature_rules.append((d, KIND_CLIPPED_NORMAL, base, float(rule["sigma"])))
                    elif "uniform_range" in rule:
                        low, high = rule["uniform_range"]
                        assert high > low and low >= 0.0, "uniform_range for %s wrongly specified" % name
                        self.armature_rules.append((d, KIND_NORMAL, float(low), float(high)))
                    else:
                        delta = rule["uniform_range_delta"]
                        assert type(delta) == float, "uniform_range_delta parameter for %s should be a float" % name
                        assert base - delta > 0.0, "uniform_range_delta param (%g) for joint %s is bigger than armature" % (delta, name)
                        self.armature_rules.append((d, KIND_NORMAL, base - delta, base + delta))
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
        # ---- rules that change compile-time constants (reference :273-291: per body, first its inertial, then its geoms)
        self.body_rules = []              # (body, [(param, rule dict), ...], [(geom, [(param, rule), ...]), ...])
        inertial, geoms = config.get("Inertial"), config.get("Geoms")
        for b, name in enumerate(model.body_names):
            iconf = gconf = None
            if inertial is not None and name in inertial:
                iconf = inertial[name]
            elif "Inertial" in default:
                iconf = default["Inertial"]
            if geoms is not None and name in geoms:
                gconf = geoms[name]
            elif "Geoms" in default:
                gconf = default["Geoms"]
            has_inertial = int(model.body_inertial_kind[b]) != 0
            irules = [(p, r) for p, r in (iconf or {}).items()] if has_inertial else []
            grules = []
            if gconf:
                for g in np.nonzero(np.asarray(model.geom_body) == b)[0]:
                    grules.append((int(g), list(gconf.items())))
            for p, r in irules + [pr for _, prs in grules for pr in prs]:
                assert len(set(r.keys()) & {"sigma", "uniform_range", "uniform_range_delta"}) == 1, \
                    "Exactly one parameter should be provided for body %s (%s)" % (name, p)
            if not (self._has_spread(dict(irules)) or any(self._has_spread(dict(prs)) for _, prs in grules)):
                continue
            for p, r in [pr for _, prs in grules for pr in prs]:
                if p in ("mass", "density") and self._has_spread({p: r}) and not has_inertial:
                    raise NotImplementedError("geom %s randomisation on body %s, which has no <inertial> element" % (p, name))
            # a body whose only spread is a geom mass / density rule draws nothing and changes nothing (its <inertial> element
            # wins in the engine's compiler, sample_model_variant): it is NOT a model rule — with nothing else in the file the
            # environment needs neither the model compiler nor a variant pool (an empty draw program is refused by the library)
            if not (self._has_spread(dict(irules)) or
                    any(self._has_spread({p: r}) for _, prs in grules for p, r in prs if p == "friction")):
                continue
            self.body_rules.append((b, irules, grules))

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
        return bool((self.spec[:, :, 0] != KIND_NONE).any()) or self.has_model_rules

    @property
    def has_model_rules(self):
        """True if the configuration randomises something that changes compile-time constants (a variant pool is needed)."""
        return bool(self.armature_rules or self.body_rules)

    @staticmethod
    def _draw_scalar(value, rule, what, name):
        if "sigma" in rule:
            return float(np.clip(np.random.normal(value, rule["sigma"]), 0.0, np.inf))
        if "uniform_range" in rule:
            low, high = rule["uniform_range"]
            assert high > low, "uniform_range for body %s wrongly specified, because high < low" % name
            assert low >= 0.0, "uniform_range for body %s wrongly specified, because low < 0.0" % name
            return float(np.random.uniform(low, high))
        delta = rule["uniform_range_delta"]
        assert type(delta) == float, "uniform_range_delta parameter for %s should be a float, but found %s." % (name, type(delta))
        assert value - delta > 0.0, "uniform_range_delta param (%g) for body %s is bigger than %s (%g)." % (delta, name, what, value)
        return float(np.random.uniform(value - delta, value + delta))

    def sample_model_variant(self, model=None):
        """One randomised model (``np.random`` draws in the reference's order: joints' armature, then per body its
        inertial and its geoms), compiled: ``mjcf.model_variant`` of the nominal model (``model``: another model of the same
        robot, e.g. with another carried weight — same bodies, joints and geoms in the same order)."""
        from .. import mjcf
        m = self._model if model is None else model
        if model is not None:
            for b, _, _ in self.body_rules:
                assert m.body_names[b] == self._model.body_names[b], "models of one environment must share their body order"
        armature, mass, inertial, friction = {}, {}, {}, {}
        for d, kind, a, b in self.armature_rules:
            v = np.random.normal(a, b)
            armature[d] = float(np.clip(v, 0.0, np.inf))     # N(low, high) can go negative (reference quirk): clipped at 0
        for b, irules, grules in self.body_rules:
            name = m.body_names[b]
            for param, rule in irules:
                if param == "mass":
                    mass[b] = self._draw_scalar(float(m.body_xml_mass[b]), rule, "mass", name)
                    continue
                # the reference's ``elif param_name == "diaginertia" or "fullinertia"`` takes every other key here
                assert "uniform_range_delta" in rule, ("domain randomization of inertia only allowed using uniform_range_delta, "
                                                       "but found %s." % list(rule.keys()))
                delta = rule["uniform_range_delta"]
                assert type(delta) == float, "uniform_range_delta parameter for %s should be a float, but found %s." % (name, type(delta))
                kind = int(m.body_inertial_kind[b])
                if param == "diaginertia":
                    assert kind == 1, "Randomizing diaginertia not allowed if not specified in the xml."
                    d0 = m.body_inertial_vals[b, :3]
                    lows, highs = d0 - delta, d0 + delta
                    assert np.all(lows > 0.0), "Error for body %s. uniform_range_delta param (%g) is bigger than the smallest singular values (%g)." % (name, delta, d0.min())
                    inertial[b] = np.concatenate([np.random.uniform(lows, highs), np.zeros(3)])
                elif param == "fullinertia":
                    assert kind == 2, "Randomizing fullinertia not allowed if not specified in the xml."
                    fi = m.body_inertial_vals[b]
                    triu = np.array([[fi[0], fi[3], fi[4]], [0.0, fi[1], fi[5]], [0.0, 0.0, fi[2]]])
                    u, sv, vh = np.linalg.svd(triu, compute_uv=True)
                    lows, highs = sv - delta, sv + delta
                    assert np.all(lows > 0.0), "Error for body %s. uniform_range_delta param (%g) is bigger than the smallest singular values (%g)." % (name, delta, sv.min())
                    t = u @ np.diag(np.random.uniform(lows, highs)) @ vh
                    inertial[b] = np.array([t[0, 0], t[1, 1], t[2, 2], t[0, 1], t[0, 2], t[1, 2]])
            for g, prs in grules:
                for param, rule in prs:
                    if param == "friction":
                        fr = np.asarray(m.geom_friction[g], dtype=np.float64)
                        if "sigma" in rule:
                            assert len(rule["sigma"]) == 3, "sigma for randomizing friction in geom of body %s needs to be 3-dimensional" % name
                            friction[g] = np.clip(np.random.normal(fr, rule["sigma"]), 0.0, np.inf)
                        elif "uniform_range_delta" in rule:
                            delta = np.asarray(rule["uniform_range_delta"], dtype=np.float64)
                            assert len(delta) == 3, "uniform_range_delta for randomizing friction in geom of body %s needs to be 3-dimensional" % name
                            assert np.all(fr >= delta), "uniform_delta range is bigger than friction coefficient. Error occurred in body %s." % name
                            friction[g] = np.random.uniform(fr - delta, fr + delta)
                    # "mass" / "density": the body has an <inertial> element (checked in __init__), which the engine's
                    # compiler prefers over geom masses — the rule changes nothing
        return mjcf.model_variant(m, body_mass=mass, body_inertial=inertial, dof_armature=armature, geom_friction=friction)

    # ---- the same rules as a flat list of scalar draws: what the device-side model compiler executes per restart
    TARGET_ARMATURE, TARGET_MASS, TARGET_DIAGINERTIA, TARGET_SINGULAR, TARGET_FRICTION = 0, 1, 2, 3, 4

    def model_draw_ops(self, model=None):
        """
        The rules of :meth:`sample_model_variant` as scalar draws in the SAME order (``np.random`` draws a vector as consecutive
        scalars): ``[(kind, a, b, target, index, component)]`` with ``kind`` KIND_CLIPPED_NORMAL / KIND_UNIFORM / KIND_NORMAL (both
        normals are clipped at 0), ``target`` TARGET_*: armature of dof ``index``; mass / diaginertia component / singular value
        (of the upper-triangular fullinertia matrix) of body ``index``; friction component of geom ``index``. Second result:
        ``{body: (U, Vt)}`` of the fullinertia rules. Raises what the sampler asserts.
        """
        m = self._model if model is None else model
        ops, svd = [], {}
        for d, kind, a, b in self.armature_rules:
            ops.append((kind, a, b, self.TARGET_ARMATURE, d, 0))

        def scalar(value, rule, what, name):
            if "sigma" in rule:
                return (KIND_CLIPPED_NORMAL, float(value), float(rule["sigma"]))
            if "uniform_range" in rule:
                low, high = rule["uniform_range"]
                assert high > low, "uniform_range for body %s wrongly specified, because high < low" % name
                assert low >= 0.0, "uniform_range for body %s wrongly specified, because low < 0.0" % name
                return (KIND_UNIFORM, float(low), float(high))
            delta = rule["uniform_range_delta"]
            assert type(delta) == float, "uniform_range_delta parameter for %s should be a float, but found %s." % (name, type(delta))
            assert value - delta > 0.0, "uniform_range_delta param (%g) for body %s is bigger than %s (%g)." % (delta, name, what, value)
            return (KIND_UNIFORM, float(value - delta), float(value + delta))

        for b, irules, grules in self.body_rules:
            name = m.body_names[b]
            for param, rule in irules:
                if param == "mass":
                    ops.append(scalar(float(m.body_xml_mass[b]), rule, "mass", name) + (self.TARGET_MASS, b, 0))
                    continue
                assert "uniform_range_delta" in rule, ("domain randomization of inertia only allowed using uniform_range_delta, "
                                                       "but found %s." % list(rule.keys()))
                delta = rule["uniform_range_delta"]
                assert type(delta) == float, "uniform_range_delta parameter for %s should be a float, but found %s." % (name, type(delta))
                kind = int(m.body_inertial_kind[b])
                if param == "diaginertia":
                    assert kind == 1, "Randomizing diaginertia not allowed if not specified in the xml."
                    d0 = m.body_inertial_vals[b, :3]
                    assert np.all(d0 - delta > 0.0), "Error for body %s. uniform_range_delta param (%g) is bigger than the smallest singular values (%g)." % (name, delta, d0.min())
                    ops += [(KIND_UNIFORM, float(d0[k] - delta), float(d0[k] + delta), self.TARGET_DIAGINERTIA, b, k) for k in range(3)]
                elif param == "fullinertia":
                    assert kind == 2, "Randomizing fullinertia not allowed if not specified in the xml."
                    fi = m.body_inertial_vals[b]
                    triu = np.array([[fi[0], fi[3], fi[4]], [0.0, fi[1], fi[5]], [0.0, 0.0, fi[2]]])
                    u, sv, vh = np.linalg.svd(triu, compute_uv=True)
                    assert np.all(sv - delta > 0.0), "Error for body %s. uniform_range_delta param (%g) is bigger than the smallest singular values (%g)." % (name, delta, sv.min())
                    svd[b] = (u, vh)
                    ops += [(KIND_UNIFORM, float(sv[k] - delta), float(sv[k] + delta), self.TARGET_SINGULAR, b, k) for k in range(3)]
            for g, prs in grules:
                for param, rule in prs:
                    if param != "friction":
                        continue
                    fr = np.asarray(m.geom_friction[g], dtype=np.float64)
                    if "sigma" in rule:
                        assert len(rule["sigma"]) == 3, "sigma for randomizing friction in geom of body %s needs to be 3-dimensional" % name
                        ops += [(KIND_CLIPPED_NORMAL, float(fr[k]), float(rule["sigma"][k]), self.TARGET_FRICTION, g, k) for k in range(3)]
                    elif "uniform_range_delta" in rule:
                        delta = np.asarray(rule["uniform_range_delta"], dtype=np.float64)
                        assert len(delta) == 3, "uniform_range_delta for randomizing friction in geom of body %s needs to be 3-dimensional" % name
                        assert np.all(fr >= delta), "uniform_delta range is bigger than friction coefficient. Error occurred in body %s." % name
                        ops += [(KIND_UNIFORM, float(fr[k] - delta[k]), float(fr[k] + delta[k]), self.TARGET_FRICTION, g, k) for k in range(3)]
        return ops, svd

    def variant_from_draws(self, values, model=None):
        """The compiled model of ONE set of draws (``values[i]`` = the value op ``i`` of :meth:`model_draw_ops` drew): what
        :meth:`sample_model_variant` returns when ``np.random`` draws these numbers. The device-side model compiler reports its draws
        (``lm_get_model_draws``); this is the host's (and the oracle's) model of that environment."""
        from .. import mjcf
        m = self._model if model is None else model
        ops, svd = self.model_draw_ops(m)
        assert len(values) == len(ops)
        armature, mass, diag, sing, friction = {}, {}, {}, {}, {}
        for (kind, a, b, target, index, comp), v in zip(ops, values):
            v = float(v)
            if target == self.TARGET_ARMATURE:
                armature[index] = v
            elif target == self.TARGET_MASS:
                mass[index] = v
            elif target == self.TARGET_DIAGINERTIA:
                diag.setdefault(index, m.body_inertial_vals[index, :3].copy())[comp] = v
            elif target == self.TARGET_SINGULAR:
                sing.setdefault(index, np.zeros(3))[comp] = v
            else:
                friction.setdefault(index, np.asarray(m.geom_friction[index], dtype=np.float64).copy())[comp] = v
        inertial = {b: np.concatenate([d, np.zeros(3)]) for b, d in diag.items()}
        for b, sv in sing.items():
            t = svd[b][0] @ np.diag(sv) @ svd[b][1]
            inertial[b] = np.array([t[0, 0], t[1, 1], t[2, 2], t[0, 1], t[0, 2], t[1, 2]])
        return mjcf.model_variant(m, body_mass=mass, body_inertial=inertial, dof_armature=armature, geom_friction=friction)

    @staticmethod
    def draw_op(op):
        """One scalar draw of an op with ``np.random`` (the reference's generator)."""
        kind, a, b = op[:3]
        if kind == KIND_UNIFORM:
            return float(np.random.uniform(a, b))
        return float(np.clip(np.random.normal(a, b), 0.0, np.inf))

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
