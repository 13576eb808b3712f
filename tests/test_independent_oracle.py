"""The C oracle against a second, independent restatement of SPEC.md (oracle/oracle_np.py: numpy, vectorised by colour class, its
own greedy colouring), and against physical known answers none of the closed-form KATs of test_oracle_kat.py covers. CPU only.

Nothing here touches the plugin's planner except the one test that says so: the constraint order comes from
oracle_np.greedy_colour_order, so a SPEC misreading shared by oracle.c and the planner-ordered tests cannot hide behind the order.
PARITY stays UNPINNED by the reference (/root/reference/README.md:1 is the whole reference); this hardens the oracle, it does not pin it.
"""
import numpy as np
import pytest

from oracle import oracle_np
from softbodyunity_amd.mesh import SoftbodyMesh, bunny_surrogate, from_triangle_mesh, jelly_cube
from helpers import build_plan, make_oracle


def _pair(oracle_mod, mesh, gravity=(0.0, -9.81, 0.0), damping=0.0, compliance=(0.0, 0.0, 0.0), plane=None):
    """(C oracle, numpy solver) on the same mesh with the SAME independent order."""
    t, ids, off = oracle_np.greedy_colour_order(mesh.n, mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl)
    c = make_oracle(oracle_mod, mesh, None, gravity, damping, compliance, ground_plane=plane)
    c.set_order(t, ids)
    p = oracle_np.NumpySolver(mesh.pos, mesh.vel, mesh.inv_mass, gravity, damping)
    if len(mesh.dist_rest):
        p.set_distance(mesh.dist_ij, mesh.dist_rest, compliance[0])
    if len(mesh.vol_rest):
        p.set_volume(mesh.vol_ijkl, mesh.vol_rest, compliance[1])
    if len(mesh.bend_rest):
        p.set_bending(mesh.bend_ijkl, mesh.bend_rest, compliance[2])
    if plane is not None:
        p.set_ground_plane(plane[:3], plane[3])
    p.set_classes(t, ids, off)
    return c, p


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def grid_cloth(n, seed=2):
    g = np.arange(n, dtype=np.float64)
    V = np.stack(np.meshgrid(g, [0.0], g, indexing="ij"), axis=-1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    F = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
    m, _ = from_triangle_mesh(V, F)
    rng = np.random.default_rng(seed)
    m.pos = (m.rest_pos.astype(np.float64) + rng.normal(0, 0.05, m.rest_pos.shape)).astype(np.float32)
    return m


def test_colouring_made_here_is_a_proper_colouring_and_a_permutation():
    mesh = bunny_surrogate(target_verts=600, seed=3)
    t, ids, off = oracle_np.greedy_colour_order(mesh.n, mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl)
    arr = [mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl]
    for ty in range(3):
        assert np.array_equal(np.sort(ids[t == ty]), np.arange(len(arr[ty])))
    for a, b in zip(off[:-1], off[1:]):
        assert len(set(t[a:b])) == 1
        verts = arr[t[a]][ids[a:b]].ravel()
        assert len(np.unique(verts)) == len(verts)


@pytest.mark.parametrize("case", ["cube8", "cube8_pinned_damped_plane", "tets2k", "cloth"])
def test_c_oracle_equals_the_numpy_restatement_bit_for_bit(oracle_mod, case):
    kw = {}
    if case == "cube8":
        mesh, ticks, S = jelly_cube(8), 5, 10                     # BASELINE.json:7
    elif case == "cube8_pinned_damped_plane":
        mesh, ticks, S = jelly_cube(8, pin_top=True), 6, 7
        kw = dict(damping=0.7, compliance=(3e-6, 0.0, 0.0), plane=(0.0, 1.0, 0.0, -0.3))
    elif case == "tets2k":
        mesh, ticks, S = bunny_surrogate(target_verts=2000, seed=11), 3, 8
        assert mesh.n > 1500 and len(mesh.vol_rest) > 5000 and len(mesh.bend_rest) > 100
        mesh.inv_mass[::7] = 0.0; mesh.inv_mass[1::5] = 2.5        # pinned and heavy / light particles
        kw = dict(compliance=(1e-7, 2e-7, 1e-4), damping=0.2, plane=(0.0, 1.0, 0.0, -1.0))
    else:
        mesh, ticks, S = grid_cloth(20), 4, 10
        kw = dict(compliance=(0.0, 0.0, 1e-3))
    c, p = _pair(oracle_mod, mesh, **kw)
    for _ in range(ticks):
        c.step(0.02, S)
        p.step(0.02, S)
        assert np.array_equal(_bits(c.x), _bits(p.x)) and np.array_equal(_bits(c.v), _bits(p.v))
    assert np.isfinite(c.x).all() and not np.array_equal(_bits(c.x), _bits(mesh.pos))


def test_kinematic_particles_in_both_restatements(oracle_mod):
    # SPEC.md 2, kinematic particles: the pinned top layer of the cube is moved by the host between ticks; both restatements take the
    # assignment, agree bit for bit, keep the pins where they were put (at rest) and drag the body along; a free particle is refused
    mesh = jelly_cube(8, pin_top=True)
    pins = np.nonzero(mesh.inv_mass == 0)[0]
    rest = mesh.pos[pins].copy()
    c, p = _pair(oracle_mod, mesh, damping=0.3)
    for t in range(8):
        target = rest + np.array([0.25 * np.sin(0.5 * t), 0.0, 0.1 * t], np.float32)
        c.set_kinematic_positions(pins, target); p.set_kinematic_positions(pins, target)
        c.step(0.02, 6); p.step(0.02, 6)
        assert np.array_equal(_bits(c.x), _bits(p.x)) and np.array_equal(_bits(c.v), _bits(p.v))
        assert np.array_equal(_bits(c.x[pins]), _bits(target)) and not c.v[pins].any()
    free = np.nonzero(mesh.inv_mass > 0)[0]
    assert (c.x[free, 2] - mesh.pos[free, 2]).mean() > 0.03           # the body followed the handle along z
    for solver in (c, p):
        with pytest.raises(ValueError):
            solver.set_kinematic_positions(free[:1], np.zeros((1, 3), np.float32))


def test_numpy_restatement_on_the_planners_groups_equals_the_c_oracle_on_the_planners_order(oracle_mod):
    # the one test here that uses the plugin's planner: its published GROUPS (claimed vertex-disjoint) become the numpy solver's
    # classes, its flat order drives the sequential C oracle -- equal bits means the groups really commute, per substep parity
    mesh = bunny_surrogate(target_verts=1500, seed=5)
    plan = build_plan(mesh, tile_particles=128)
    comp = (1e-7, 1e-7, 1e-5)
    c = make_oracle(oracle_mod, mesh, plan, compliance=comp)
    p = oracle_np.NumpySolver(mesh.pos, mesh.vel, mesh.inv_mass)
    p.set_distance(mesh.dist_ij, mesh.dist_rest, comp[0]); p.set_volume(mesh.vol_ijkl, mesh.vol_rest, comp[1]); p.set_bending(mesh.bend_ijkl, mesh.bend_rest, comp[2])
    for parity in (0, 1):
        t, ids = plan.order(parity)
        p.set_classes(t, ids, plan.groups(parity), parity=parity)
    for _ in range(2):
        c.step(0.02, 5); p.step(0.02, 5)
    assert np.array_equal(_bits(c.x), _bits(p.x)) and np.array_equal(_bits(c.v), _bits(p.v))


# ---- physical known answers ---------------------------------------------------------------------------------------------------

def _chain(n, pinned, pos, L0):
    ij = np.stack([np.arange(n - 1), np.arange(1, n)], 1).astype(np.int32)
    w = np.ones(n, np.float32); w[list(pinned)] = 0.0
    pos = np.asarray(pos, np.float32)
    return SoftbodyMesh(rest_pos=pos.copy(), pos=pos.copy(), vel=np.zeros_like(pos), inv_mass=w, dist_ij=ij,
                        dist_rest=np.full(n - 1, L0, np.float32))


def _settle(o, seconds, S, dt=0.02):
    for _ in range(int(round(seconds / dt))):
        o.step(dt, S)


@pytest.mark.parametrize("solver", ["c", "numpy"])
def test_hanging_chain_static_sag(oracle_mod, solver):
    """A vertical chain of unit masses on compliant springs, top pinned. At rest spring k carries the weight of the particles
    below it, f_k = m g (n-1-k), and XPBD's compliance IS the inverse stiffness: extension = alpha f_k -- to first order. One
    Gauss-Seidel sweep per substep adds terms of order h^2 f that follow from SPEC.md alone (sweep order = colour 0: springs
    0, 2, 4, ... on the predicted positions, then colour 1: springs 1, 3, ... on what colour 0 left; every free particle must be
    moved back by exactly the h^2 g the integrate step added, i.e. spring k applies lambda_k = h^2 f_k):
        even k >= 2 : ext = f_k (2 h^2 + alpha)              (both ends free, both moved alike by the integrate step)
        k = 0       : ext = f_0 (h^2 + alpha) - h^2 g        (pinned top end: w_i + w_j = 1, the free end dropped by h^2 g)
        odd k       : ext = f_k (2 h^2 + alpha) - h^2 (f_(k-1) + f_(k+1))   (its ends were already pulled apart by its neighbours)
    binary32 leaves a dead band: a residual extension error E moves an end point by E / (2 + alpha / h^2) per substep, and below one
    ulp of the position (1.2e-7 for |y| < 2: the chain is kept short and near the origin) that is rounded away -- the solver stops
    within ulp * (2 + alpha / h^2) of the answer. Large substeps (4 per tick) keep alpha / h^2 = 80 and the band at 1e-5."""
    n, alpha, g, L0, S = 8, 2e-3, 9.81, 0.25, 4
    h2 = (0.02 / S) ** 2
    mesh = _chain(n, [0], [[0.0, -L0 * k, 0.0] for k in range(n)], L0)
    c, p = _pair(oracle_mod, mesh, damping=8.0, compliance=(alpha, 0.0, 0.0))
    o = c if solver == "c" else p
    _settle(o, 8.0, S)
    x = np.asarray(o.x, np.float64)
    ext = np.linalg.norm(x[1:] - x[:-1], axis=1) - np.float64(np.float32(L0))
    f = g * np.arange(n - 1, 0, -1.0)
    fpad = np.concatenate([[0.0], f, [0.0]])                   # f_(k-1), f_(k+1) with 0 beyond the ends
    want = f * (2 * h2 + alpha)
    want[0] = f[0] * (h2 + alpha) - h2 * g
    odd = np.arange(n - 1) % 2 == 1
    want[odd] -= h2 * (fpad[:-2] + fpad[2:])[odd]
    assert np.abs(np.asarray(o.v)).max() < 1e-3               # at rest up to binary32 granularity (one ulp of y over h)
    band = 4 * 1.2e-7 * (2 + alpha / h2)
    assert np.allclose(ext, want, rtol=2e-4, atol=band), (ext - want, band)
    assert np.abs(ext - alpha * f).max() > 20 * band           # (the h^2 terms are resolved: the plain alpha * f is measurably off)
    assert np.allclose(ext, alpha * f, rtol=5e-2)              # ... yet right to first order
    assert np.abs(x[:, [0, 2]]).max() < 1e-6                   # stays on the vertical line


def test_chain_between_two_pins_matches_the_minimum_of_its_potential_energy(oracle_mod):
    """The discrete catenary of springs: both ends pinned 0.9 * (rest length of the chain) apart, gravity, compliant springs.
    The resting shape must be THE minimiser of E = sum (L - L0)^2 / (2 alpha) + sum m g y, found here with scipy, independent of
    XPBD altogether (the one-sweep-per-substep terms above are of relative order h^2 / alpha = 0.2 %)."""
    from scipy.optimize import minimize
    n, alpha, g, L0, S = 15, 2e-3, 9.81, 0.2, 10
    span = 0.9 * (n - 1) * L0
    x0 = np.stack([np.linspace(0, span, n), -0.06 * np.sin(np.linspace(0, np.pi, n)), np.zeros(n)], 1)
    mesh = _chain(n, [0, n - 1], x0, L0)
    c, _ = _pair(oracle_mod, mesh, damping=6.0, compliance=(alpha, 0.0, 0.0))
    _settle(c, 12.0, S)
    got = np.asarray(c.x, np.float64)
    l0 = np.float64(np.float32(L0))

    def energy(q):
        x = x0.copy(); x[1:-1, :2] = q.reshape(-1, 2)
        L = np.linalg.norm(x[1:] - x[:-1], axis=1)
        return ((L - l0) ** 2).sum() / (2 * alpha) + g * x[1:-1, 1].sum()
    res = minimize(energy, got[1:-1, :2].ravel() + 0.01, method="BFGS", options={"gtol": 1e-10, "maxiter": 10000})
    want = x0.copy(); want[1:-1, :2] = res.x.reshape(-1, 2)
    sag = -want[:, 1].min()
    assert np.abs(np.asarray(c.v)).max() < 1e-3                             # (binary32 granularity, as above)
    assert sag > 0.5                                                        # it really sags (the springs stretch by up to 30 %)
    assert np.abs(got - want).max() < 5e-3 * sag, (np.abs(got - want).max(), sag)


def test_rigid_volume_blob_keeps_its_volume_under_gravity(oracle_mod):
    """A soft blob (compliant springs) dropped on the ground: with rigid volume constraints (alpha_v = 0) its total volume stays
    within 1 % of the rest volume while it is squashed; without them the same blob loses several times as much."""
    mesh = bunny_surrogate(target_verts=400, seed=7)
    T = mesh.vol_ijkl.astype(np.int64)

    def volume(x):
        x = np.asarray(x, np.float64)
        return np.einsum("ij,ij->i", x[T[:, 1]] - x[T[:, 0]], np.cross(x[T[:, 2]] - x[T[:, 0]], x[T[:, 3]] - x[T[:, 0]])).sum() / 6.0
    v0 = volume(mesh.rest_pos)
    floor = float(mesh.rest_pos[:, 1].min()) - 0.05
    worst = {}
    for with_volume in (True, False):
        m = SoftbodyMesh(rest_pos=mesh.rest_pos, pos=mesh.pos.copy(), vel=np.zeros_like(mesh.pos), inv_mass=mesh.inv_mass, dist_ij=mesh.dist_ij,
                         dist_rest=mesh.dist_rest, vol_ijkl=mesh.vol_ijkl if with_volume else np.zeros((0, 4), np.int32),
                         vol_rest=mesh.vol_rest if with_volume else np.zeros(0, np.float32))
        c, _ = _pair(oracle_mod, m, damping=1.0, compliance=(2e-3, 0.0, 0.0), plane=(0.0, 1.0, 0.0, floor))
        drift = 0.0
        for _ in range(60):
            c.step(0.02, 20)
            drift = max(drift, abs(volume(c.x) / v0 - 1.0))
        worst[with_volume] = drift
        assert np.isfinite(c.x).all() and float(np.asarray(c.x)[:, 1].min()) >= floor - 1e-5
    assert worst[True] <= 0.01, worst
    assert worst[False] >= 3 * worst[True], worst


def _dihedral(x, q):
    """Signed dihedral angle of hinge (a, b | c, d) in float64, atan2 form: 0 = flat, SPEC 6's sign."""
    a, b, c, d = (np.asarray(x, np.float64)[k] for k in q)
    e = b - a
    n1 = np.cross(a - c, b - c); n2 = np.cross(b - d, a - d)
    return float(np.arctan2(-np.dot(np.cross(n1, n2), e) / np.linalg.norm(e), np.dot(n1, n2)))


@pytest.mark.parametrize("rest_deg", [0.0, 35.0, -50.0])
def test_folded_hinge_converges_to_its_rest_angle(oracle_mod, rest_deg):
    """Two triangles on a shared edge, folded to +70 degrees, no gravity: the bending constraint (rigid) with the five edge
    springs must bring the signed dihedral angle to the rest angle -- sign included -- and keep every edge at its length."""
    a, b = np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0])
    c = np.array([0.5, 0.0, 0.9])

    def wing(phi):      # d such that the dihedral angle of (a, b | c, d) is phi
        return np.array([0.5, -0.9 * np.sin(phi), -0.9 * np.cos(phi)])
    rest = np.radians(rest_deg)
    quad = np.array([[0, 1, 2, 3]], np.int32)
    assert abs(_dihedral([a, b, c, wing(0.3)], quad[0]) - 0.3) < 1e-12 or abs(_dihedral([a, b, c, wing(0.3)], quad[0]) + 0.3) < 1e-12
    sign = np.sign(_dihedral([a, b, c, wing(0.3)], quad[0]))         # orientation of wing() against SPEC 6's sign, measured not assumed
    P0 = np.array([a, b, c, wing(sign * np.radians(70.0))], np.float32)
    ij = np.array([[0, 1], [0, 2], [1, 2], [0, 3], [1, 3]], np.int32)
    L0 = np.linalg.norm(P0[ij[:, 0]].astype(np.float64) - P0[ij[:, 1]], axis=1).astype(np.float32)
    mesh = SoftbodyMesh(rest_pos=P0.copy(), pos=P0.copy(), vel=np.zeros_like(P0), inv_mass=np.ones(4, np.float32), dist_ij=ij, dist_rest=L0,
                        bend_ijkl=quad, bend_rest=np.array([[np.cos(rest), np.sin(rest)]], np.float32))
    o, _ = _pair(oracle_mod, mesh, gravity=(0.0, 0.0, 0.0), damping=3.0)
    assert abs(_dihedral(o.x, quad[0]) - np.radians(70.0)) < 1e-6
    start = abs(_dihedral(o.x, quad[0]) - rest)
    for tick in range(150):
        o.step(0.02, 10)
        err = abs(_dihedral(o.x, quad[0]) - rest)
        if tick == 0:
            assert err < 0.05 * start          # a rigid hinge is (nearly) resolved by the first tick's ten projections
    assert err < 2e-4, err
    x = np.asarray(o.x, np.float64)
    assert np.allclose(np.linalg.norm(x[ij[:, 0]] - x[ij[:, 1]], axis=1), L0, atol=2e-5)
    # no external force: the centroid stays -- up to binary32 granularity of the velocities (one ulp of a position over h = 6e-5 per
    # particle and substep, a random walk over 3 s; measured drift velocity 1e-5 .. 1e-4)
    assert np.abs(x.mean(0) - P0.astype(np.float64).mean(0)).max() < 3e-3
