"""
MJCF reader + "mini-compiler" (host side, cold path).

The reference builds its models with two third-party packages that are not part of this
framework: ``dm_control.mjcf`` (XML object model used for model surgery, e.g.
``loco_mujoco/environments/quadrupeds/unitreeA1.py:756-776``) and ``mujoco.MjModel.from_xml_string``
(the MuJoCo model compiler, invoked by mushroom-rl from ``loco_mujoco/environments/base.py:109-111``).
This module replaces both for the subset of MJCF the BASELINE models use:

* default classes (nested ``<default class=...>``, ``childclass``), ``autolimits``,
* bodies with hinge/slide joints, explicit ``<inertial>`` (``diaginertia``+``quat`` or ``fullinertia``),
* primitive geoms (plane, sphere, capsule, cylinder, box; ``fromto``), sites,
* ``<motor>`` and ``<position>`` actuators with joint transmission,
* ``<option>`` timestep / cone / impratio / integrator / iterations / tolerance.

The result is a :class:`CompiledModel`: flat numpy arrays (body tree, joints/dofs, geoms, actuators,
options) plus the ``qpos0``-derived constants the constraint regulariser needs
(``dof_invweight0``, ``body_invweight0``).

Everything here is fp64 numpy; nothing here is on the hot path.
"""

import copy
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# geom types (numbering is private to this framework; shared with csrc/ and oracle/ via the blob)
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 1, 2, 3, 4, 5
GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
              "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH}
JNT_SLIDE, JNT_HINGE = 0, 1
CONE_PYRAMIDAL, CONE_ELLIPTIC = 0, 1
ACT_MOTOR, ACT_MUSCLE, ACT_POSITION = 0, 1, 2
INT_EULER, INT_RK4 = 0, 1

_DEFAULT_SOLREF = (0.02, 1.0)
_DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


# --------------------------------------------------------------------------------------
# small math helpers
# --------------------------------------------------------------------------------------

def _floats(s, n=None):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None:
        assert len(v) == n, (s, n)
    return v


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-300:
        return np.array([1.0, 0, 0, 0])
    s = np.sin(0.5 * angle) / n
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def z_to_quat(vec):
    """Quaternion that rotates the z-axis onto ``vec`` (shortest arc)."""
    vec = np.asarray(vec, dtype=np.float64)
    vec = vec / np.linalg.norm(vec)
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis)
    c = vec[2]
    if s < 1e-10:
        if c > 0:
            return np.array([1.0, 0, 0, 0])
        return np.array([0.0, 1.0, 0, 0])
    ang = np.arctan2(s, c)
    return axis_angle_quat(axis / s, ang)


# --------------------------------------------------------------------------------------
# XML object model (what the reference gets from dm_control.mjcf)
# --------------------------------------------------------------------------------------

class MjcfHandle:
    """
    A thin mutable handle on an MJCF document, with the few operations the reference's environment
    classes perform through ``dm_control.mjcf`` (``find``, ``add``, remove): see
    ``unitreeA1.py:768-774``, ``base_humanoid.py:86-127``, ``atlas.py:338-364``.
    """

    def __init__(self, root, base_dir=None):
        self.root = root
        self.base_dir = base_dir      # where relative mesh files are resolved (None: meshes are not read)

    @classmethod
    def from_path(cls, path):
        return cls(ET.parse(str(path)).getroot(), os.path.dirname(os.path.abspath(str(path))))

    @classmethod
    def from_string(cls, s):
        return cls(ET.fromstring(s))

    def copy(self):
        return MjcfHandle(copy.deepcopy(self.root), self.base_dir)

    def find(self, tag, name):
        for el in self.root.iter(tag):
            if el.get("name") == name:
                return el
        return None

    def find_all(self, tag):
        return list(self.root.iter(tag))

    def add(self, parent, tag, **attrs):
        el = ET.SubElement(parent, tag)
        for k, v in attrs.items():
            el.set(k, v if isinstance(v, str) else " ".join(repr(float(x)) for x in np.atleast_1d(v)))
        return el

    def remove(self, el):
        for parent in self.root.iter():
            if el in list(parent):
                parent.remove(el)
                return True
        return False

    def to_xml_string(self):
        return ET.tostring(self.root, encoding="unicode")


# --------------------------------------------------------------------------------------
# compiled model
# --------------------------------------------------------------------------------------

@dataclass
class CompiledModel:
    # options
    timestep: float = 0.002
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    cone: int = CONE_PYRAMIDAL
    impratio: float = 1.0
    integrator: int = INT_EULER
    iterations: int = 100
    tolerance: float = 1e-8
    # sizes
    nbody: int = 0
    njnt: int = 0
    nv: int = 0
    ngeom: int = 0
    nu: int = 0
    nsite: int = 0
    na: int = 0
    ntendon: int = 0
    # names
    body_names: list = field(default_factory=list)
    jnt_names: list = field(default_factory=list)
    geom_names: list = field(default_factory=list)
    act_names: list = field(default_factory=list)
    site_names: list = field(default_factory=list)
    tendon_names: list = field(default_factory=list)
    # arrays are attached dynamically (see compile_mjcf)

    _SCALARS = ("timestep", "cone", "impratio", "integrator", "iterations", "tolerance", "nbody", "njnt", "nv",
                "ngeom", "nu", "na", "nsite", "ntendon", "meaninertia", "n_dropped_mesh_geoms")
    _NAMES = ("body_names", "jnt_names", "geom_names", "act_names", "site_names", "tendon_names")

    def save(self, path):
        """Serialise to an ``.npz`` (arrays + a JSON header); see :meth:`load`."""
        import json
        arrays = {k: v for k, v in self.__dict__.items() if isinstance(v, np.ndarray)}
        meta = {k: getattr(self, k) for k in self._SCALARS}
        meta.update({k: list(getattr(self, k)) for k in self._NAMES})
        np.savez_compressed(path, __meta__=np.array(json.dumps(meta)), **arrays)

    @classmethod
    def load(cls, path):
        import json
        m = cls()
        with np.load(path, allow_pickle=False) as f:
            meta = json.loads(str(f["__meta__"]))
            for k in f.files:
                if k != "__meta__":
                    setattr(m, k, f[k])
        for k, v in meta.items():
            setattr(m, k, v)
        return m

    def jnt_id(self, name):
        return self.jnt_names.index(name)

    def act_id(self, name):
        return self.act_names.index(name)

    def body_id(self, name):
        return self.body_names.index(name)


class _Defaults:
    """Resolved ``<default>`` tree: class name -> {tag -> attribute dict}."""

    def __init__(self, root):
        self.classes = {"main": {}}
        self.parent = {"main": None}
        for d in root.findall("default"):
            self._walk(d, "main", top=True)

    def _walk(self, el, parent_cls, top=False):
        if top:
            cls = el.get("class", "main")
        else:
            cls = el.get("class")
            assert cls is not None, "nested <default> needs a class"
        if cls not in self.classes:
            self.classes[cls] = copy.deepcopy(self.classes[parent_cls]) if cls != parent_cls else {}
            self.parent[cls] = parent_cls if cls != parent_cls else None
        table = self.classes[cls]
        for child in el:
            if child.tag == "default":
                continue
            table.setdefault(child.tag, {}).update(child.attrib)
        for child in el.findall("default"):
            self._walk(child, cls)

    def resolve(self, tag, el, childclass):
        cls = el.get("class", childclass if childclass is not None else "main")
        attrs = dict(self.classes.get(cls, {}).get(tag, {}))
        attrs.update({k: v for k, v in el.attrib.items() if k != "class"})
        return attrs


def inertia_from_spec(mass, kind, vals, quat, bounds):
    """(mass, 3x3 inertia tensor in the BODY frame about the COM) of an ``<inertial>`` element: ``kind`` 2 = fullinertia
    (xx yy zz xy xz yz), 1 = diaginertia + quat; then the compiler bounds ``(boundmass, boundinertia, balanceinertia)``,
    which act on the principal moments."""
    boundmass, boundinertia, balanceinertia = bounds
    if kind == 2:
        f = vals
        inertia = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
    else:
        r = quat_to_mat(quat)
        inertia = r @ np.diag(vals[:3]) @ r.T
    ev, evec = np.linalg.eigh(inertia)
    if boundinertia > 0:
        ev = np.maximum(ev, boundinertia)          # lower bound first ...
    if balanceinertia and (ev[0] + ev[1] < ev[2]):
        ev[:] = ev.mean()                          # ... then the triangle-inequality repair
    inertia = evec @ np.diag(ev) @ evec.T
    if boundmass > 0:
        mass = max(mass, boundmass)
    return mass, inertia


def _bounding_capsule(v):
    """Bounding capsule of a point cloud: (centre, unit axis, radius, half length); half = 0 for stubby clouds."""
    c0 = 0.5 * (v.min(0) + v.max(0))
    r_sphere = float(np.linalg.norm(v - c0, axis=1).max())
    ev, evec = np.linalg.eigh(np.cov((v - v.mean(0)).T))
    axis = evec[:, 2]
    t = (v - c0) @ axis
    perp = (v - c0) - np.outer(t, axis)
    c = c0 + 0.5 * (perp.min(0) + perp.max(0))          # recentre the axis across the section
    t = (v - c) @ axis
    d = np.linalg.norm((v - c) - np.outer(t, axis), axis=1)
    r = float(d.max()) * (1 + 1e-9)
    reach = np.sqrt(np.maximum(r * r - d * d, 0.0))     # how far each point may lie beyond a segment end
    lo, hi = float(np.min(t + reach)), float(np.max(t - reach))
    if hi <= lo or r > 0.75 * r_sphere:
        return c0, np.array([0.0, 0.0, 1.0]), r_sphere, 0.0
    return c + 0.5 * (lo + hi) * axis, axis, r, 0.5 * (hi - lo)


def _mesh_triangles(root, comp, base_dir):
    """{mesh name: (ntri, 3, 3) float64 vertices, scaled} from binary STL files."""
    out = {}
    if base_dir is None:
        return out
    meshdir = comp.get("meshdir", "")
    for el in root.iter("mesh"):
        f = el.get("file")
        if f is None or not f.lower().endswith(".stl"):
            continue
        path = os.path.join(base_dir, meshdir, f)
        if not os.path.exists(path):
            continue
        raw = open(path, "rb").read()
        ntri = int(np.frombuffer(raw[80:84], "<u4")[0])
        if len(raw) != 84 + 50 * ntri:
            continue
        tri = np.frombuffer(raw[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")]), count=ntri)
        v = tri["v"].reshape(-1, 3, 3).astype(np.float64) * _floats(el.get("scale", "1 1 1"), 3)
        out[el.get("name", os.path.splitext(os.path.basename(f))[0])] = v
    return out


def _mesh_com(v):
    """Centre of mass of a triangle soup (ntri, 3, 3) as the engine's compiler finds it (pyramids from the area-weighted mean
    of the triangle centres, |volume| each): the origin of a mesh geom's frame, and the interior point the convex-convex
    collider (MPR) starts from."""
    a, b, c = v[:, 0], v[:, 1], v[:, 2]
    nrm = np.cross(b - a, c - a)
    area = 0.5 * np.linalg.norm(nrm, axis=1)
    nrm = nrm / np.maximum(2.0 * area, 1e-300)[:, None]
    cen = (a + b + c) / 3.0
    facecen = (area[:, None] * cen).sum(0) / area.sum()
    pv = np.abs(np.einsum("ij,ij->i", cen - facecen, nrm) * area / 3.0)
    return (pv[:, None] * (0.75 * cen + 0.25 * facecen)).sum(0) / pv.sum()


def _geom_inertia(g, tris):
    """(mass, centre, inertia about the centre) of one geom in the BODY frame; None if it carries no mass.
    Primitives: closed formulas; meshes: the engine's pyramid sums over the triangles (see below)."""
    density, mass_attr = g["density"], g["mass"]
    if (mass_attr is not None and mass_attr <= 0) or (mass_attr is None and density <= 0):
        return None
    t, s = g["type"], g["size"]
    r = quat_to_mat(g["quat"])
    if t == GEOM_MESH:
        v = tris.get(g["mesh"])
        if v is None:
            raise NotImplementedError("mass of mesh geom %s: mesh file not available" % g["name"])
        # the engine's default ("legacy", exactmeshinertia=false) integration: pyramids from an apex to every
        # triangle with |volume| — apex = area-weighted mean of the triangle centres for volume and centre,
        # apex = that centre for the second moments. Equal to the exact integral for a convex closed mesh only.
        a, b, c = v[:, 0], v[:, 1], v[:, 2]
        nrm = np.cross(b - a, c - a)
        area = 0.5 * np.linalg.norm(nrm, axis=1)
        nrm = nrm / np.maximum(2.0 * area, 1e-300)[:, None]
        cen = (a + b + c) / 3.0
        facecen = (area[:, None] * cen).sum(0) / area.sum()
        pv = np.abs(np.einsum("ij,ij->i", cen - facecen, nrm) * area / 3.0)
        vol = pv.sum()
        com = (pv[:, None] * (0.75 * cen + 0.25 * facecen)).sum(0) / vol
        a, b, c = a - com, b - com, c - com
        pv = np.abs(np.einsum("ij,ij->i", (a + b + c) / 3.0, nrm) * area / 3.0)
        ssum = a + b + c
        second = np.einsum("i,ijk->jk", pv / 20.0, np.einsum("ij,ik->ijk", a, a) + np.einsum("ij,ik->ijk", b, b)
                           + np.einsum("ij,ik->ijk", c, c) + np.einsum("ij,ik->ijk", ssum, ssum))
        inertia = np.trace(second) * np.eye(3) - second
        # the compiler then treats the mesh geom as its equivalent inertia box (same principal axes and same
        # inertia / mass ratio): mass = density x BOX volume, not x mesh volume (pinned by the Talos golden rollouts)
        ev, axes = np.linalg.eigh(inertia)
        half = np.array([np.sqrt(max(6.0 * (ev[(i + 1) % 3] + ev[(i + 2) % 3] - ev[i]) / vol, 0.0)) / 2.0 for i in range(3)])
        m = mass_attr if mass_attr is not None else density * 8.0 * half.prod()
        inertia = axes @ np.diag(m / 3.0 * np.array([half[1] ** 2 + half[2] ** 2, half[0] ** 2 + half[2] ** 2,
                                                    half[0] ** 2 + half[1] ** 2])) @ axes.T
        return m, g["pos"] + r @ com, r @ inertia @ r.T
    if t == GEOM_SPHERE:
        vol = 4.0 / 3.0 * np.pi * s[0] ** 3
        diag = np.full(3, 0.4 * s[0] ** 2)
    elif t == GEOM_BOX:
        vol = 8.0 * s[0] * s[1] * s[2]
        diag = np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 3.0
    elif t == GEOM_CYLINDER:
        vol = np.pi * s[0] ** 2 * 2 * s[1]
        diag = np.array([(3 * s[0] ** 2 + 4 * s[1] ** 2) / 12.0, (3 * s[0] ** 2 + 4 * s[1] ** 2) / 12.0, 0.5 * s[0] ** 2])
    elif t == GEOM_CAPSULE:
        rad, h = s[0], 2 * s[1]
        vc, vs = np.pi * rad ** 2 * h, 4.0 / 3.0 * np.pi * rad ** 3
        vol = vc + vs
        izz = (vc * 0.5 * rad ** 2 + vs * 0.4 * rad ** 2) / vol
        ixx = (vc * (rad ** 2 / 4 + h ** 2 / 12) + vs * (0.4 * rad ** 2 + 0.375 * rad * h + h ** 2 / 4)) / vol
        diag = np.array([ixx, ixx, izz])
    else:
        return None
    m = mass_attr if mass_attr is not None else density * vol
    return m, g["pos"].copy(), r @ np.diag(m * diag) @ r.T


def _mesh_bounds(root, comp, base_dir, hulls=None, graphs=None):
    """{mesh name: bounding capsule (centre, axis, radius, half length)} in the mesh's own frame, from binary STL files."""
    out = {}
    if base_dir is None:
        return out
    meshdir = comp.get("meshdir", "")
    for el in root.iter("mesh"):
        f = el.get("file")
        if f is None or not f.lower().endswith(".stl"):
            continue
        path = os.path.join(base_dir, meshdir, f)
        if not os.path.exists(path):
            continue
        raw = open(path, "rb").read()
        ntri = int(np.frombuffer(raw[80:84], "<u4")[0])
        if len(raw) != 84 + 50 * ntri:
            continue                      # ASCII STL: not read
        tri = np.frombuffer(raw[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")]), count=ntri)
        v = tri["v"].reshape(-1, 3).astype(np.float64) * _floats(el.get("scale", "1 1 1"), 3)
        name = el.get("name", os.path.splitext(os.path.basename(f))[0])
        out[name] = _bounding_capsule(v)
        if hulls is not None:
            # vertices of the convex hull, in the order of their first appearance in the file (the support search of the
            # plane-mesh collider breaks ties by that order): the engine collides the hull of a mesh, not the mesh. The hull's
            # vertex graph comes from the same qhull run over ALL distinct points of the file (triangulated facets; which
            # diagonal a nearly planar quad gets depends on the run, and this is the variant that reproduces most UnitreeH1
            # golden rows: profiles/r3_notes.md §2)
            pts, first = np.unique(v, axis=0, return_index=True)
            u = v[np.sort(first)]
            try:
                from scipy.spatial import ConvexHull
                h = ConvexHull(u, qhull_options="Qt")
                keep = np.sort(h.vertices)
                hulls[name] = u[keep]
                local = {int(k): i for i, k in enumerate(keep)}
                nb = [set() for _ in keep]
                for tri in h.simplices:
                    for i in tri:
                        nb[local[int(i)]].update(local[int(j)] for j in tri if j != i)
                if graphs is not None:
                    graphs[name] = [sorted(s_, key=lambda j, i=i: (float(np.linalg.norm(u[keep[j]] - u[keep[i]])), j))
                                    for i, s_ in enumerate(nb)]
            except Exception:             # degenerate (flat) mesh or no scipy: every distinct vertex, no graph
                hulls[name] = u
    return out


def compile_mjcf(handle, timestep=None, drop_mesh_geoms=False):
    """
    Compile an :class:`MjcfHandle` into a :class:`CompiledModel`.

    ``timestep`` overrides ``<option timestep>`` like the reference does (``base.py:33,109-111``).
    ``drop_mesh_geoms`` (historical name: the flag says "mesh geoms are expected"): collidable mesh geoms collide as their CONVEX HULLS
    (against the floor: support vertex + hull-graph neighbours; against other geoms: the convex collider), which needs the mesh files.
    When they can be read (``handle.base_dir``), each such geom is kept as a ``GEOM_MESH`` geom with its hull vertices, vertex graph,
    centre of mass and a bounding capsule (``geom_pos``/``geom_quat`` = capsule frame in the body frame, z along the capsule;
    ``geom_size`` = radius, half length — the broad phases); without files the geom is removed and counted in
    ``m.n_dropped_mesh_geoms``. Without the flag a model with collidable meshes raises.
    """
    root = handle.root
    comp = root.find("compiler")
    comp = comp.attrib if comp is not None else {}
    assert comp.get("angle", "degree") == "radian", "only angle=radian models are supported"
    assert comp.get("coordinate", "local") == "local"
    autolimits = comp.get("autolimits", "false") == "true"
    balanceinertia = comp.get("balanceinertia", "false") == "true"
    boundmass = float(comp.get("boundmass", 0))
    boundinertia = float(comp.get("boundinertia", 0))

    m = CompiledModel()
    opt = root.find("option")
    opt = opt.attrib if opt is not None else {}
    m.timestep = float(opt.get("timestep", 0.002)) if timestep is None else float(timestep)
    m.cone = CONE_ELLIPTIC if opt.get("cone", "pyramidal") == "elliptic" else CONE_PYRAMIDAL
    m.impratio = float(opt.get("impratio", 1.0))
    m.integrator = {"Euler": INT_EULER, "RK4": INT_RK4}[opt.get("integrator", "Euler")]
    m.iterations = int(opt.get("iterations", 100))
    m.tolerance = float(opt.get("tolerance", 1e-8))
    assert opt.get("solver", "Newton") == "Newton"
    if "gravity" in opt:
        m.gravity = _floats(opt["gravity"], 3)

    defaults = _Defaults(root)

    bodies = [dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mass=0.0,
                   ipos=np.zeros(3), inertia=np.zeros((3, 3)), jntadr=-1, jntnum=0)]
    joints, geoms, sites = [], [], []

    def walk_body(el, parent_id, childclass):
        if el.tag == "worldbody":
            bid = 0
        else:
            cc = el.get("childclass", childclass)
            childclass = cc
            b = dict(name=el.get("name", "body%d" % len(bodies)), parent=parent_id,
                     pos=_floats(el.get("pos", "0 0 0"), 3), quat=_orientation(el.attrib),
                     mass=0.0, ipos=np.zeros(3), inertia=np.zeros((3, 3)), jntadr=-1, jntnum=0)
            bodies.append(b)
            bid = len(bodies) - 1
            inertial = el.find("inertial")
            b["explicit_inertial"] = inertial is not None
            b["inertial_kind"], b["inertial_vals"], b["inertial_quat"] = 0, np.zeros(6), np.array([1.0, 0, 0, 0])
            if inertial is not None:
                ia = inertial.attrib
                # the XML-level numbers are kept: domain randomisation acts on THEM (reference
                # utils/domain_randomization.py:460-514: mass, diaginertia, singular values of the fullinertia triangle)
                if "fullinertia" in ia:
                    b["inertial_kind"], b["inertial_vals"] = 2, _floats(ia["fullinertia"], 6)
                else:
                    q = _floats(ia.get("quat", "1 0 0 0"), 4)
                    b["inertial_kind"], b["inertial_quat"] = 1, q / np.linalg.norm(q)
                    b["inertial_vals"] = np.concatenate([_floats(ia["diaginertia"], 3), np.zeros(3)])
                b["ipos"] = _floats(ia.get("pos", "0 0 0"), 3)
                b["mass"], b["inertia"] = inertia_from_spec(float(ia["mass"]), b["inertial_kind"], b["inertial_vals"],
                                                            b["inertial_quat"], (boundmass, boundinertia, balanceinertia))
            for j in el.findall("joint"):
                a = defaults.resolve("joint", j, childclass)
                jt = a.get("type", "hinge")
                assert jt in ("hinge", "slide"), "joint type %s not supported" % jt
                axis = _floats(a.get("axis", "0 0 1"), 3)
                axis = axis / np.linalg.norm(axis)
                has_range = "range" in a
                rng = _floats(a.get("range", "0 0"), 2)
                if "limited" in a and a["limited"] in ("true", "false"):
                    limited = a["limited"] == "true"
                else:
                    limited = has_range and autolimits
                assert float(a.get("ref", 0)) == 0.0 and float(a.get("springref", 0)) == 0.0
                jd = dict(name=a.get("name", "jnt%d" % len(joints)), type=JNT_HINGE if jt == "hinge" else JNT_SLIDE,
                          body=bid, pos=_floats(a.get("pos", "0 0 0"), 3), axis=axis,
                          limited=limited, range=rng, stiffness=float(a.get("stiffness", 0)),
                          damping=float(a.get("damping", 0)), armature=float(a.get("armature", 0)),
                          frictionloss=float(a.get("frictionloss", 0)), margin=float(a.get("margin", 0)),
                          solref_limit=_floats(a.get("solreflimit", "%g %g" % _DEFAULT_SOLREF), 2),
                          solimp_limit=_pad_solimp(a.get("solimplimit")),
                          solref_friction=_floats(a.get("solreffriction", "%g %g" % _DEFAULT_SOLREF), 2),
                          solimp_friction=_pad_solimp(a.get("solimpfriction")))
                if b["jntnum"] == 0:
                    b["jntadr"] = len(joints)
                b["jntnum"] += 1
                joints.append(jd)
        for g in el.findall("geom"):
            a = defaults.resolve("geom", g, childclass)
            gt = GEOM_TYPES[a.get("type", "sphere")]
            size = np.zeros(3)
            sz = _floats(a["size"]) if "size" in a else np.zeros(0)
            size[:len(sz)] = sz
            pos = _floats(a.get("pos", "0 0 0"), 3)
            q = _orientation(a)
            if "fromto" in a and gt in (GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX):
                ft = _floats(a["fromto"], 6)
                vec = ft[0:3] - ft[3:6]
                half = 0.5 * np.linalg.norm(vec)
                pos = 0.5 * (ft[0:3] + ft[3:6])
                q = z_to_quat(vec)
                if gt == GEOM_BOX:
                    size = np.array([size[0], size[0], half])
                else:
                    size = np.array([size[0], half, 0.0])
            fr = np.array([1.0, 0.005, 0.0001])
            if "friction" in a:
                f = _floats(a["friction"])
                fr[:len(f)] = f
            geoms.append(dict(name=a.get("name", ""), type=gt, body=bid, pos=pos, quat=q, size=size,
                              contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
                              condim=int(a.get("condim", 3)), priority=int(a.get("priority", 0)),
                              friction=fr, solmix=float(a.get("solmix", 1.0)),
                              solref=_floats(a.get("solref", "%g %g" % _DEFAULT_SOLREF), 2),
                              solimp=_pad_solimp(a.get("solimp")),
                              margin=float(a.get("margin", 0)), gap=float(a.get("gap", 0)),
                              mesh=a.get("mesh"), density=float(a.get("density", 1000.0)),
                              mass=(float(a["mass"]) if "mass" in a else None)))
        for s in el.findall("site"):
            a = defaults.resolve("site", s, childclass)
            pos = _floats(a.get("pos", "0 0 0"), 3)
            q = _orientation(a)
            if "fromto" in a:
                ft = _floats(a["fromto"], 6)
                pos = 0.5 * (ft[0:3] + ft[3:6])
                q = z_to_quat(ft[0:3] - ft[3:6])
            sites.append(dict(name=a.get("name", ""), body=bid, pos=pos, quat=q / np.linalg.norm(q)))
        for child in el.findall("body"):
            walk_body(child, bid, childclass)

    walk_body(root.find("worldbody"), 0, None)

    # ---------------- bodies
    # ---------------- bodies without <inertial>: mass, centre and inertia from their geoms (compiler inertiafromgeom="auto")
    if comp.get("inertiafromgeom", "auto") != "false":
        tris = None
        for bid, b in enumerate(bodies):
            if bid == 0 or b.get("explicit_inertial", True):
                continue
            parts = []
            for gd in geoms:
                if gd["body"] != bid:
                    continue
                if tris is None and gd["type"] == GEOM_MESH:
                    tris = _mesh_triangles(root, comp, handle.base_dir)
                part = _geom_inertia(gd, tris or {})
                if part is not None and part[0] > 0:
                    parts.append(part)
            if not parts:
                continue
            mass = sum(p_[0] for p_ in parts)
            com = sum(p_[0] * p_[1] for p_ in parts) / mass
            inertia = np.zeros((3, 3))
            for m_, c_, i_ in parts:
                d = c_ - com
                inertia += i_ + m_ * (d @ d * np.eye(3) - np.outer(d, d))
            b["mass"], b["ipos"], b["inertia"] = mass, com, inertia

    m.nbody = len(bodies)
    m.body_names = [b["name"] for b in bodies]
    m.body_parent = np.array([b["parent"] for b in bodies], dtype=np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_quat = np.array([b["quat"] for b in bodies])
    m.body_mass = np.array([b["mass"] for b in bodies])
    m.body_ipos = np.array([b["ipos"] for b in bodies])
    m.body_inertia = np.array([b["inertia"] for b in bodies])          # (nbody,3,3) body frame, about COM
    # XML-level <inertial> numbers (0 = none / from geoms, 1 = diaginertia + quat, 2 = fullinertia) and the compiler bounds
    m.body_inertial_kind = np.array([b.get("inertial_kind", 0) for b in bodies], dtype=np.int32)
    m.body_inertial_vals = np.array([b.get("inertial_vals", np.zeros(6)) for b in bodies], dtype=np.float64)
    m.body_inertial_quat = np.array([b.get("inertial_quat", np.array([1.0, 0, 0, 0])) for b in bodies], dtype=np.float64)
    m.body_xml_mass = np.array([b["mass"] if b.get("inertial_kind", 0) else 0.0 for b in bodies], dtype=np.float64)
    m.compiler_bounds = np.array([boundmass, boundinertia, 1.0 if balanceinertia else 0.0])
    m.body_jntadr = np.array([b["jntadr"] for b in bodies], dtype=np.int32)
    m.body_jntnum = np.array([b["jntnum"] for b in bodies], dtype=np.int32)
    # weld id: the nearest ancestor-or-self that has joints (0 = welded to the world)
    weld = np.zeros(m.nbody, dtype=np.int32)
    for i in range(1, m.nbody):
        weld[i] = i if bodies[i]["jntnum"] > 0 else weld[bodies[i]["parent"]]
    m.body_weldid = weld

    # ---------------- joints / dofs (all joints are 1-dof, so dof id == joint id)
    m.njnt = m.nv = len(joints)
    m.jnt_names = [j["name"] for j in joints]
    m.jnt_type = np.array([j["type"] for j in joints], dtype=np.int32)
    m.jnt_body = np.array([j["body"] for j in joints], dtype=np.int32)
    m.jnt_pos = np.array([j["pos"] for j in joints]).reshape(-1, 3)
    m.jnt_axis = np.array([j["axis"] for j in joints]).reshape(-1, 3)
    m.jnt_limited = np.array([j["limited"] for j in joints], dtype=np.int32)
    m.jnt_range = np.array([j["range"] for j in joints]).reshape(-1, 2)
    m.jnt_stiffness = np.array([j["stiffness"] for j in joints])
    m.jnt_margin = np.array([j["margin"] for j in joints])
    m.jnt_solref = np.array([j["solref_limit"] for j in joints]).reshape(-1, 2)
    m.jnt_solimp = np.array([j["solimp_limit"] for j in joints]).reshape(-1, 5)
    m.dof_damping = np.array([j["damping"] for j in joints])
    m.dof_armature = np.array([j["armature"] for j in joints])
    m.dof_frictionloss = np.array([j["frictionloss"] for j in joints])
    m.dof_solref = np.array([j["solref_friction"] for j in joints]).reshape(-1, 2)
    m.dof_solimp = np.array([j["solimp_friction"] for j in joints]).reshape(-1, 5)
    # dof parent: previous dof in the same body, else last dof of the nearest jointed ancestor
    dof_parent = -np.ones(m.nv, dtype=np.int32)
    body_lastdof = -np.ones(m.nbody, dtype=np.int32)
    for i in range(1, m.nbody):
        last = body_lastdof[bodies[i]["parent"]]
        for k in range(bodies[i]["jntnum"]):
            d = bodies[i]["jntadr"] + k
            dof_parent[d] = last
            last = d
        body_lastdof[i] = last
    m.dof_parent = dof_parent
    m.qpos0 = np.zeros(m.nv)

    # ---------------- geoms (drop purely visual ones: contype == conaffinity == 0)
    geoms = [g for g in geoms if (g["contype"] != 0 or g["conaffinity"] != 0)]
    m.n_dropped_mesh_geoms = sum(1 for g in geoms if g["type"] == GEOM_MESH)
    if m.n_dropped_mesh_geoms and not drop_mesh_geoms:
        raise NotImplementedError("this model has collidable mesh geoms: compile it with drop_mesh_geoms=True (they collide as convex hulls, "
                                  "which needs the mesh files next to the XML)")
    hulls, graphs = {}, {}
    bounds = _mesh_bounds(root, comp, handle.base_dir, hulls, graphs) if m.n_dropped_mesh_geoms else {}
    kept = []
    hull_of = {}                      # index in `kept` -> hull vertices in the frame of the geom's body
    graph_of = {}                     # index in `kept` -> neighbour lists of the hull's vertices (indices into the hull)
    mesh_tris = _mesh_triangles(root, comp, handle.base_dir) if m.n_dropped_mesh_geoms else {}
    for g in geoms:
        g["center"] = np.array(g["pos"], dtype=np.float64)
        if g["type"] == GEOM_MESH:
            if g["mesh"] not in bounds:
                continue
            if g["mesh"] in hulls:
                hull_of[len(kept)] = g["pos"] + hulls[g["mesh"]] @ quat_to_mat(g["quat"]).T
                graph_of[len(kept)] = graphs.get(g["mesh"])
            if g["mesh"] in mesh_tris:
                g["center"] = g["pos"] + quat_to_mat(g["quat"]) @ _mesh_com(mesh_tris[g["mesh"]])
            centre, axis, radius, half = bounds[g["mesh"]]
            g = dict(g, pos=g["pos"] + quat_to_mat(g["quat"]) @ centre, quat=quat_mul(g["quat"], z_to_quat(axis)),
                     size=np.array([radius, half, 0.0]))
        kept.append(g)
    geoms = kept
    m.ngeom = len(geoms)
    m.geom_names = [g["name"] for g in geoms]
    m.geom_type = np.array([g["type"] for g in geoms], dtype=np.int32)
    m.geom_body = np.array([g["body"] for g in geoms], dtype=np.int32)
    m.geom_pos = np.array([g["pos"] for g in geoms]).reshape(-1, 3)
    # centre of every geom for the convex-convex collider, body frame: the geom frame origin — for a mesh geom that is the
    # mesh's centre of mass (geom_pos of a mesh geom is the centre of its bounding capsule instead)
    m.geom_center = np.array([g["center"] for g in geoms]).reshape(-1, 3)
    m.geom_quat = np.array([g["quat"] for g in geoms]).reshape(-1, 4)
    m.geom_size = np.array([g["size"] for g in geoms]).reshape(-1, 3)
    m.geom_contype = np.array([g["contype"] for g in geoms], dtype=np.int32)
    m.geom_conaffinity = np.array([g["conaffinity"] for g in geoms], dtype=np.int32)
    m.geom_condim = np.array([g["condim"] for g in geoms], dtype=np.int32)
    m.geom_priority = np.array([g["priority"] for g in geoms], dtype=np.int32)
    m.geom_friction = np.array([g["friction"] for g in geoms]).reshape(-1, 3)
    m.geom_solmix = np.array([g["solmix"] for g in geoms])
    m.geom_solref = np.array([g["solref"] for g in geoms]).reshape(-1, 2)
    m.geom_solimp = np.array([g["solimp"] for g in geoms]).reshape(-1, 5)
    m.geom_margin = np.array([g["margin"] for g in geoms])
    m.geom_gap = np.array([g["gap"] for g in geoms])
    # convex hulls of the mesh geoms (plane-mesh collider: one contact at the hull's support vertex): vertices of geom g are
    # hull_vert[geom_hull_adr[g] : + geom_hull_num[g]], in the frame of the geom's body; -1 / 0 for the other geoms
    m.geom_hull_adr = -np.ones(m.ngeom, dtype=np.int32)
    m.geom_hull_num = np.zeros(m.ngeom, dtype=np.int32)
    chunks, n = [], 0
    for gi in sorted(hull_of):
        m.geom_hull_adr[gi], m.geom_hull_num[gi] = n, len(hull_of[gi])
        chunks.append(hull_of[gi])
        n += len(hull_of[gi])
    m.hull_vert = (np.concatenate(chunks) if chunks else np.zeros((0, 3))).astype(np.float32)
    # the hull's vertex graph (plane-mesh collider: further contacts at the neighbours of the support vertex, DESIGN.md §2 item
    # 10): neighbours of hull vertex i of geom g, as indices INTO the geom's hull, nearest first, are
    # hull_nbr[hull_nbr_adr[geom_hull_adr[g] + i] : hull_nbr_adr[geom_hull_adr[g] + i + 1]]
    adr, nbr = [0], []
    for gi in sorted(hull_of):
        lists = graph_of.get(gi) or [[] for _ in range(len(hull_of[gi]))]
        for lst in lists:
            nbr += [int(j) for j in lst]
            adr.append(len(nbr))
    m.hull_nbr_adr, m.hull_nbr = np.array(adr, dtype=np.int32), np.array(nbr, dtype=np.int32)

    # ---------------- sites
    m.nsite = len(sites)
    m.site_names = [s["name"] for s in sites]
    m.site_body = np.array([s["body"] for s in sites], dtype=np.int32)
    m.site_pos = np.array([s["pos"] for s in sites]).reshape(-1, 3)
    m.site_quat = np.array([s["quat"] for s in sites]).reshape(-1, 4)

    # ---------------- tendons: spatial paths through sites only (no wrapping geoms, no pulleys)
    tendons, wrap = [], []
    ten_root = root.find("tendon")
    if ten_root is not None:
        for t_el in ten_root:
            if t_el.tag != "spatial":
                raise NotImplementedError("tendon type <%s>" % t_el.tag)
            path = []
            for w_el in t_el:
                if w_el.tag != "site":
                    raise NotImplementedError("tendon path element <%s> (wrapping / pulleys)" % w_el.tag)
                path.append(m.site_names.index(w_el.get("site")))
            a = defaults.resolve("tendon", t_el, None)
            if float(a.get("stiffness", 0)) != 0 or float(a.get("damping", 0)) != 0 or a.get("limited", "false") == "true" \
                    or float(a.get("frictionloss", 0)) != 0:
                raise NotImplementedError("tendon springs / dampers / limits / friction")
            tendons.append(dict(name=t_el.get("name", ""), adr=len(wrap), num=len(path)))
            wrap += path
    m.ntendon = len(tendons)
    m.tendon_names = [t["name"] for t in tendons]
    m.tendon_adr = np.array([t["adr"] for t in tendons], dtype=np.int32)
    m.tendon_num = np.array([t["num"] for t in tendons], dtype=np.int32)
    m.wrap_site = np.array(wrap, dtype=np.int32)

    # ---------------- actuators: motors on joints, muscles on tendons
    acts = []
    act_root = root.find("actuator")
    if act_root is not None:
        for a_el in act_root:
            if a_el.tag not in ("motor", "muscle", "position"):
                raise NotImplementedError("actuator type <%s>" % a_el.tag)
            a = defaults.resolve(a_el.tag, a_el, None)
            gear = _floats(a.get("gear", "1"))[0]
            cr = _floats(a.get("ctrlrange", "0 0"), 2)
            if "ctrllimited" in a and a["ctrllimited"] in ("true", "false"):
                cl = a["ctrllimited"] == "true"
            else:
                cl = ("ctrlrange" in a) and autolimits
            act = dict(name=a.get("name", ""), kind=ACT_MOTOR, dof=-1, tendon=-1, gear=gear, ctrlrange=cr, ctrllimited=cl,
                       dynprm=np.zeros(3), gainprm=np.zeros(9), lengthrange=np.zeros(2), biasprm=np.zeros(3),
                       forcerange=_floats(a.get("forcerange", "0 0"), 2))
            if "forcelimited" in a and a["forcelimited"] in ("true", "false"):
                act["forcelimited"] = a["forcelimited"] == "true"
            else:
                act["forcelimited"] = ("forcerange" in a) and autolimits
            if a_el.tag == "motor":
                act["dof"] = m.jnt_names.index(a["joint"])
            elif a_el.tag == "position":
                # <position kp>: force = kp * ctrl - kp * length (gain [kp, 0, 0], bias [0, -kp, 0]), clamped to forcerange
                kp = float(a.get("kp", 1))
                act["kind"], act["dof"] = ACT_POSITION, m.jnt_names.index(a["joint"])
                act["gainprm"][0], act["biasprm"][1] = kp, -kp
            else:
                # <muscle> shortcut: activation dynamics (timeconst, tausmooth) and the force-length-velocity curve
                # parameters (range, force, scale, lmin, lmax, vmax, fpmax, fvmax), shared by gain and bias
                act["kind"] = ACT_MUSCLE
                if "tendon" not in a:
                    raise NotImplementedError("muscle %s: only tendon transmission is supported" % act["name"])
                act["tendon"] = m.tendon_names.index(a["tendon"])
                tc = _floats(a.get("timeconst", "0.01 0.04"), 2)
                act["dynprm"] = np.array([tc[0], tc[1], float(a.get("tausmooth", 0))])
                rng = _floats(a.get("range", "0.75 1.05"), 2)
                force = float(a.get("force", -1))
                if force < 0:
                    raise NotImplementedError("muscle %s: force from scale/acc0 is not built" % act["name"])
                if "lengthrange" not in a:
                    raise NotImplementedError("muscle %s: automatic length-range computation is not built" % act["name"])
                act["gainprm"] = np.array([rng[0], rng[1], force, float(a.get("scale", 200)), float(a.get("lmin", 0.5)),
                                           float(a.get("lmax", 1.6)), float(a.get("vmax", 1.5)), float(a.get("fpmax", 1.3)),
                                           float(a.get("fvmax", 1.2))])
                act["lengthrange"] = _floats(a["lengthrange"], 2)
            acts.append(act)
    m.nu = len(acts)
    m.na = sum(1 for a in acts if a["kind"] == ACT_MUSCLE)
    m.act_names = [a["name"] for a in acts]
    m.act_kind = np.array([a["kind"] for a in acts], dtype=np.int32)
    m.act_dof = np.array([a["dof"] for a in acts], dtype=np.int32)
    m.act_tendon = np.array([a["tendon"] for a in acts], dtype=np.int32)
    m.act_gear = np.array([a["gear"] for a in acts])
    m.act_ctrlrange = np.array([a["ctrlrange"] for a in acts]).reshape(-1, 2)
    m.act_ctrllimited = np.array([a["ctrllimited"] for a in acts], dtype=np.int32)
    m.act_dynprm = np.array([a["dynprm"] for a in acts]).reshape(-1, 3)
    m.act_gainprm = np.array([a["gainprm"] for a in acts]).reshape(-1, 9)
    m.act_lengthrange = np.array([a["lengthrange"] for a in acts]).reshape(-1, 2)
    m.act_biasprm = np.array([a["biasprm"] for a in acts]).reshape(-1, 3)
    m.act_forcerange = np.array([a["forcerange"] for a in acts]).reshape(-1, 2)
    m.act_forcelimited = np.array([a["forcelimited"] for a in acts], dtype=np.int32)

    _set_const(m)
    return m


def _orientation(attrs):
    """Body/geom/site orientation from ``quat`` | ``axisangle`` | ``euler`` (radians, default xyz sequence)."""
    if "quat" in attrs:
        q = _floats(attrs["quat"], 4)
    elif "axisangle" in attrs:
        a = _floats(attrs["axisangle"], 4)
        q = axis_angle_quat(a[:3], a[3])
    elif "euler" in attrs:
        e = _floats(attrs["euler"], 3)
        q = np.array([1.0, 0, 0, 0])
        for ax, ang in zip(np.eye(3), e):          # intrinsic x-y-z
            q = quat_mul(q, axis_angle_quat(ax, ang))
    else:
        q = np.array([1.0, 0, 0, 0])
    for bad in ("xyaxes", "zaxis"):
        assert bad not in attrs, "orientation attribute %s not supported" % bad
    return q / np.linalg.norm(q)


def _pad_solimp(s):
    v = np.array(_DEFAULT_SOLIMP)
    if s is not None:
        f = _floats(s)
        v[:len(f)] = f
    return v


# --------------------------------------------------------------------------------------
# qpos0-derived constants
# --------------------------------------------------------------------------------------

def forward_kinematics(m, qpos):
    """
    Plain numpy forward kinematics (cold path; used for model constants and host-side checks).
    Joint composition rule: the joints of one body act in declaration order, each about its own
    body-local axis through its anchor.

    Returns dict with xpos (nbody,3), xmat (nbody,3,3), xipos (nbody,3), xanchor (nv,3), xaxis (nv,3).
    """
    xpos = np.zeros((m.nbody, 3))
    xquat = np.zeros((m.nbody, 4))
    xquat[0] = [1, 0, 0, 0]
    xanchor = np.zeros((m.nv, 3))
    xaxis = np.zeros((m.nv, 3))
    for i in range(1, m.nbody):
        p = m.body_parent[i]
        rp = quat_to_mat(xquat[p])
        pos = xpos[p] + rp @ m.body_pos[i]
        quat = quat_mul(xquat[p], m.body_quat[i])
        for k in range(m.body_jntnum[i]):
            j = m.body_jntadr[i] + k
            r = quat_to_mat(quat)
            xanchor[j] = pos + r @ m.jnt_pos[j]
            xaxis[j] = r @ m.jnt_axis[j]
            if m.jnt_type[j] == JNT_SLIDE:
                pos = pos + xaxis[j] * (qpos[j] - m.qpos0[j])
            else:
                quat = quat_mul(quat, axis_angle_quat(m.jnt_axis[j], qpos[j] - m.qpos0[j]))
                pos = xanchor[j] - quat_to_mat(quat) @ m.jnt_pos[j]
        quat = quat / np.linalg.norm(quat)
        xpos[i], xquat[i] = pos, quat
    xmat = np.array([quat_to_mat(q) for q in xquat])
    xipos = xpos + np.einsum("bij,bj->bi", xmat, m.body_ipos)
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, xanchor=xanchor, xaxis=xaxis)


def _dof_affects_body(m):
    """bool (nbody, nv): dof d moves body b."""
    aff = np.zeros((m.nbody, m.nv), dtype=bool)
    for b in range(1, m.nbody):
        a = b
        while a != 0:
            for k in range(m.body_jntnum[a]):
                aff[b, m.body_jntadr[a] + k] = True
            a = m.body_parent[a]
    return aff


def body_jacobians(m, kin, point):
    """6 x nv Jacobians (translational rows 0:3 at ``point[b]``, rotational rows 3:6) for every body."""
    aff = _dof_affects_body(m)
    jac = np.zeros((m.nbody, 6, m.nv))
    for b in range(1, m.nbody):
        for d in range(m.nv):
            if not aff[b, d]:
                continue
            ax = kin["xaxis"][d]
            if m.jnt_type[d] == JNT_SLIDE:
                jac[b, 0:3, d] = ax
            else:
                jac[b, 0:3, d] = np.cross(ax, point[b] - kin["xanchor"][d])
                jac[b, 3:6, d] = ax
    return jac


def mass_matrix(m, qpos):
    kin = forward_kinematics(m, qpos)
    jac = body_jacobians(m, kin, kin["xipos"])
    mm = np.zeros((m.nv, m.nv))
    for b in range(1, m.nbody):
        jp, jr = jac[b, 0:3], jac[b, 3:6]
        iw = kin["xmat"][b] @ m.body_inertia[b] @ kin["xmat"][b].T
        mm += m.body_mass[b] * jp.T @ jp + jr.T @ iw @ jr
    mm[np.diag_indices(m.nv)] += m.dof_armature
    return mm, kin, jac


def _set_const(m):
    """
    ``dof_invweight0`` / ``body_invweight0`` at ``qpos0`` (what MuJoCo's model compiler stores and its
    constraint regulariser reads): diagonal of M^-1 per dof, and for every
    moving body the mean diagonal of the translational / rotational 3x3 blocks of J M^-1 J^T with J taken
    at the body's centre of mass.
    """
    if m.nv == 0:
        m.dof_invweight0 = np.zeros(0)
        m.body_invweight0 = np.zeros((m.nbody, 2))
        m.meaninertia = 1.0
        return
    mm, kin, jac = mass_matrix(m, m.qpos0)
    minv = np.linalg.inv(mm)
    m.dof_invweight0 = np.diag(minv).copy()
    biw = np.zeros((m.nbody, 2))
    for b in range(1, m.nbody):
        if m.body_weldid[b] == 0:
            continue
        a = jac[b] @ minv @ jac[b].T
        biw[b, 0] = (a[0, 0] + a[1, 1] + a[2, 2]) / 3.0
        biw[b, 1] = (a[3, 3] + a[4, 4] + a[5, 5]) / 3.0
    m.body_invweight0 = biw
    m.meaninertia = float(np.trace(mm) / m.nv)


def model_variant(m, body_mass=None, body_inertial=None, dof_armature=None, geom_friction=None):
    """
    A copy of ``m`` with other inertial / armature / friction numbers — what recompiling the XML after
    ``apply_domain_randomization`` (reference utils/domain_randomization.py:228-294) gives: ``body_mass`` {body: mass},
    ``body_inertial`` {body: 6 XML-level numbers of its kind}, ``dof_armature`` {dof: value}, ``geom_friction`` {geom: 3
    numbers}. The derived constants (``dof_invweight0``, ``body_invweight0``, ``meaninertia``) are recomputed.
    """
    import copy
    v = copy.copy(m)
    for k, a in m.__dict__.items():
        if isinstance(a, np.ndarray):
            setattr(v, k, a.copy())
    bounds = (float(m.compiler_bounds[0]), float(m.compiler_bounds[1]), bool(m.compiler_bounds[2]))
    touched = set(body_mass or {}) | set(body_inertial or {})
    for b in touched:
        kind = int(m.body_inertial_kind[b])
        if kind == 0:
            raise ValueError("body %s has no <inertial> element" % m.body_names[b])
        if body_inertial and b in body_inertial:
            v.body_inertial_vals[b] = np.asarray(body_inertial[b], dtype=np.float64)
        if body_mass and b in body_mass:
            v.body_xml_mass[b] = float(body_mass[b])
        v.body_mass[b], v.body_inertia[b] = inertia_from_spec(float(v.body_xml_mass[b]), kind, v.body_inertial_vals[b],
                                                               v.body_inertial_quat[b], bounds)
    for d, a in (dof_armature or {}).items():
        v.dof_armature[d] = float(a)
    for g, fr in (geom_friction or {}).items():
        v.geom_friction[g] = np.asarray(fr, dtype=np.float64)
    _set_const(v)
    return v
