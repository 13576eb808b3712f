"""Closed-form known-answer tests that pin the CPU oracle (SPEC.md).

PARITY UNPINNED against the reference: /root/reference holds only README.md:1, so these KATs are
the anchor. They run on CPU (-m "not gpu").
"""
import numpy as np
import pytest

from softbodyunity_amd.mesh import jelly_cube

f32 = np.float32


def test_kat1_free_fall(oracle_mod):
    # semi-implicit Euler: v_k = v0 + k h g ; x_k = x0 + k h v0 + g h^2 k(k+1)/2
    x0 = np.array([[0.0, 1.0, 0.0], [0.5, 0.25, -0.75]], f32)
    v0 = np.array([[1.0, 0.0, 0.0], [0.0, 0.5, -0.25]], f32)
    o = oracle_mod.Oracle(x0, v0, np.ones(2, f32), gravity=(0, -9.81, 0))
    S, dt, ticks = 10, 0.02, 3
    for _ in range(ticks):
        o.step(dt, S)
    k = S * ticks
    h = dt / S
    g = np.array([0, -9.81, 0])
    v_exp = v0.astype(np.float64) + k * h * g
    x_exp = x0.astype(np.float64) + k * h * v0.astype(np.float64) + g * h * h * k * (k + 1) / 2
    # PBD derives v from positions: v carries ~ulp(|x|)/h of rounding per substep, x ~k ulps
    ulp = float(np.spacing(f32(np.abs(x_exp).max())))
    assert np.allclose(o.v, v_exp, rtol=0, atol=k * ulp / h)
    assert np.allclose(o.x, x_exp, rtol=0, atol=k * k * ulp)  # h * sum of the v errors


def _two(oracle_mod, w, L, L0, alpha=0.0, g=(0, 0, 0)):
    x = np.array([[0, 0, 0], [L, 0, 0]], f32)
    o = oracle_mod.Oracle(x, None, np.array(w, f32), gravity=g)
    o.set_distance([[0, 1]], [L0], compliance=alpha)
    return o


def test_kat2_rigid_distance_one_projection(oracle_mod):
    o = _two(oracle_mod, [1.0, 3.0], 2.0, 1.0)
    s = o.scalars(0.02, 1)
    o.project_range(s, 0, 1)
    # dx_i = -w_i/(w_i+w_j) (L-L0) n, with n = (x_i - x_j)/L = (-1,0,0)
    assert np.allclose(o.x[0], [0.25, 0, 0], atol=1e-7)
    assert np.allclose(o.x[1], [1.25, 0, 0], atol=1e-7)
    assert abs(np.linalg.norm(o.x[1] - o.x[0]) - 1.0) < 1e-6


def test_kat3_compliant_distance(oracle_mod):
    alpha, dt = 1e-3, 0.02
    o = _two(oracle_mod, [1.0, 1.0], 1.5, 1.0, alpha=alpha)
    s = o.scalars(dt, 1)
    o.project_range(s, 0, 1)
    dl = -(1.5 - 1.0) / (2.0 + alpha / dt ** 2)
    # x0 += w0*dl*n, n=(-1,0,0)
    assert np.allclose(o.x[0, 0], -dl, rtol=1e-6)
    assert np.allclose(o.x[1, 0], 1.5 + dl, rtol=1e-6)


def test_kat4_pinned_partner_takes_all(oracle_mod):
    o = _two(oracle_mod, [0.0, 2.0], 3.0, 1.0, g=(0, -9.81, 0))
    x_pin = o.x[0].copy()
    s = o.scalars(0.02, 1)
    o.project_range(s, 0, 1)
    assert np.array_equal(o.x[0], x_pin)
    assert np.allclose(o.x[1], [1.0, 0, 0], atol=1e-6)
    for _ in range(5):
        o.step(0.02, 10)
    assert np.array_equal(o.x[0], x_pin)
    assert np.array_equal(o.v[0], np.zeros(3, f32))


def test_kat5_rest_lattice_bitwise_fixed_point(oracle_mod):
    m = jelly_cube(6, perturb=0.0)
    o = oracle_mod.Oracle(m.pos, None, m.inv_mass, gravity=(0, 0, 0))
    o.set_distance(m.dist_ij, m.dist_rest)
    for _ in range(4):
        o.step(0.02, 10)
    assert np.array_equal(o.x.view(np.uint32), m.pos.view(np.uint32))
    assert not o.v.any()


def test_kat6_momentum_conserved(oracle_mod):
    m = jelly_cube(6)
    rng = np.random.default_rng(7)
    w = rng.uniform(0.5, 2.0, m.n).astype(f32)
    o = oracle_mod.Oracle(m.pos, None, w, gravity=(0, 0, 0))
    o.set_distance(m.dist_ij, m.dist_rest)
    mass = 1.0 / w.astype(np.float64)
    p0 = (mass[:, None] * o.x.astype(np.float64)).sum(0)
    s = o.scalars(0.02, 10)
    o.project_range(s, 0, len(m.dist_rest))
    p1 = (mass[:, None] * o.x.astype(np.float64)).sum(0)
    assert np.abs(p1 - p0).max() < 5e-4 * np.abs(p0).max() / 100  # rounding only


def test_kat7_volume_regular_tet(oracle_mod):
    # regular tet scaled by s: V = s^3 V0; one rigid projection restores V0 to first order
    a = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float64)
    V0 = abs(np.dot(a[1] - a[0], np.cross(a[2] - a[0], a[3] - a[0]))) / 6
    tet = np.array([[0, 1, 2, 3]])
    e = a
    if np.dot(e[1] - e[0], np.cross(e[2] - e[0], e[3] - e[0])) < 0:
        tet = np.array([[0, 2, 1, 3]])
    sc = 1.02
    o = oracle_mod.Oracle((a * sc).astype(f32), None, np.ones(4, f32), gravity=(0, 0, 0))
    o.set_volume(tet, [V0])
    s = o.scalars(0.02, 1)
    o.project_range(s, 0, 1)
    x = o.x.astype(np.float64)[tet[0]]
    V1 = np.dot(x[1] - x[0], np.cross(x[2] - x[0], x[3] - x[0])) / 6
    assert abs(V1 - V0) / V0 < 3 * (sc ** 3 - 1) ** 2  # second-order residual
    # closed form: symmetric tet, equal masses -> uniform scaling about the centroid by factor t,
    # lambda = -C/(sum |grad|^2): each vertex moves along its gradient by the same amount
    cen = x.mean(0)
    assert np.allclose(cen, 0, atol=1e-6)
    r = np.linalg.norm(x, axis=1)
    assert np.allclose(r, r[0], rtol=1e-6)


def _dihedral(xx):
    """Signed dihedral of SPEC.md §6 in float64 (independent restatement via atan2)."""
    a, b, c, d = xx
    e = b - a
    n1 = np.cross(a - c, b - c); n2 = np.cross(b - d, a - d)
    u1 = n1 / np.linalg.norm(n1); u2 = n2 / np.linalg.norm(n2)
    return np.arctan2(-np.dot(np.cross(u1, u2), e) / np.linalg.norm(e), np.dot(u1, u2))


def test_kat7b_bending_gradient_and_projection(oracle_mod):
    rng = np.random.default_rng(11)
    x = np.array([[0, 0, 0], [1, 0.1, 0], [0.4, 1, 0.2], [0.6, -0.9, 0.3]], np.float64)
    phi = _dihedral(x)
    # flat hinge (wings on opposite sides, coplanar) has phi = 0 and is a fixed point for rest (1,0)
    flat = np.array([[0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0.5, -1, 0]], f32)
    assert abs(_dihedral(flat.astype(np.float64))) < 1e-12
    o = oracle_mod.Oracle(flat, None, np.ones(4, f32), gravity=(0, 0, 0))
    o.set_bending([[0, 1, 2, 3]], [[1.0, 0.0]])
    s = o.scalars(0.02, 1)
    o.project_range(s, 0, 1)
    assert np.allclose(o.x, flat, atol=1e-7)
    # one projection == -w C / sum(w |grad phi|^2) * grad phi with grad phi by central differences
    w = np.array([1.0, 0.5, 2.0, 1.5])
    phi0 = phi + 0.05
    G = np.zeros((4, 3))
    for k in range(4):
        for c in range(3):
            xp = x.copy(); xm = x.copy(); xp[k, c] += 1e-6; xm[k, c] -= 1e-6
            G[k, c] = (_dihedral(xp) - _dihedral(xm)) / 2e-6
    Cval = np.sin(phi - phi0)
    lam = -Cval / (w * (G ** 2).sum(1)).sum()
    expect = x + (w * lam)[:, None] * G
    o = oracle_mod.Oracle(x.astype(f32), None, w.astype(f32), gravity=(0, 0, 0))
    o.set_bending([[0, 1, 2, 3]], [[np.cos(phi0), np.sin(phi0)]])
    o.project_range(s, 0, 1)
    assert np.allclose(o.x, expect, atol=5e-6)
    # repeated projection converges to the rest angle; centre of mass is preserved (sum of grads = 0)
    com0 = ((1 / w)[:, None] * o.x).sum(0)
    for _ in range(20):
        o.project_range(s, 0, 1)
    assert abs(_dihedral(o.x.astype(np.float64)) - phi0) < 1e-5
    assert np.allclose(((1 / w)[:, None] * o.x).sum(0), com0, atol=1e-5)


def test_kat8_order_invariance_within_colour(oracle_mod):
    # x-direction springs of even parity form a matching: any permutation gives bitwise-equal results;
    # swapping two different colours does not.
    n = 6
    m = jelly_cube(n)
    ij = m.dist_ij
    lo = ij[:, 0]
    is_x = (ij[:, 1] - ij[:, 0]) == 1
    even = is_x & ((lo % n) % 2 == 0)
    odd = is_x & ((lo % n) % 2 == 1)
    ids_e = np.nonzero(even)[0]; ids_o = np.nonzero(odd)[0]
    rest_ids = np.nonzero(~is_x)[0]

    def run(order):
        o = oracle_mod.Oracle(m.pos, None, m.inv_mass)
        o.set_distance(m.dist_ij, m.dist_rest)
        o.set_order(np.zeros(len(order), np.uint8), order)
        o.step(0.02, 10)
        return o.x.copy()
    rng = np.random.default_rng(3)
    a = run(np.concatenate([ids_e, ids_o, rest_ids]))
    b = run(np.concatenate([rng.permutation(ids_e), rng.permutation(ids_o), rest_ids]))
    c = run(np.concatenate([ids_o, ids_e, rest_ids]))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(a.view(np.uint32), c.view(np.uint32))


def test_natural_order_equals_identity_order(oracle_mod):
    m = jelly_cube(5)
    o1 = oracle_mod.Oracle(m.pos, None, m.inv_mass); o1.set_distance(m.dist_ij, m.dist_rest)
    o2 = oracle_mod.Oracle(m.pos, None, m.inv_mass); o2.set_distance(m.dist_ij, m.dist_rest)
    M = len(m.dist_rest)
    o2.set_order(np.zeros(M, np.uint8), np.arange(M))
    o1.step(0.02, 10); o2.step(0.02, 10)
    assert np.array_equal(o1.x.view(np.uint32), o2.x.view(np.uint32))


def test_cube_counts():
    m = jelly_cube(8)
    assert m.n == 512 and len(m.dist_rest) == 1344  # BASELINE.json:7 "~1.4k springs"
    assert np.all(m.dist_rest == 1.0)


def test_kat10_ground_plane(oracle_mod):
    # a free particle dropped on the plane y >= 0 ends up exactly on it with zero normal velocity
    x0 = np.array([[0.3, 0.05, -0.2], [1.0, 2.0, 0.0]], f32)
    o = oracle_mod.Oracle(x0, None, np.array([1.0, 0.0], f32), gravity=(0, -9.81, 0))
    o.set_ground_plane((0, 1, 0), 0.0)
    for _ in range(20):
        o.step(0.02, 10)
    assert o.x[0, 1] == 0.0 and abs(o.v[0, 1]) < 1e-6
    assert np.array_equal(o.x[0, [0, 2]], x0[0, [0, 2]])        # frictionless: tangential position untouched
    assert np.array_equal(o.x[1], x0[1])                         # pinned particles ignore the plane
    # a tilted plane: penetration is removed along the normal
    n = np.array([0.6, 0.8, 0.0]); p = np.array([[0.0, -1.0, 0.0]], f32)
    o = oracle_mod.Oracle(p, None, np.ones(1, f32), gravity=(0, 0, 0))
    o.set_ground_plane(n, 0.5)
    o.collide()
    assert abs(np.dot(n, o.x[0].astype(np.float64)) - 0.5) < 1e-6
    assert np.allclose(np.cross(o.x[0] - p[0], n), 0, atol=1e-6)
