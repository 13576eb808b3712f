"""Edge cases through the C ABI on the GPU, each against the oracle (bit-exact): empty / tiny / degenerate inputs."""
import numpy as np
import pytest

from softbodyunity_amd import Softbody, native
from softbodyunity_amd.mesh import SoftbodyMesh, jelly_cube
from helpers import make_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32


def _mesh(pos, ij=None, rest=None, w=None, vel=None):
    pos = np.asarray(pos, f32).reshape(-1, 3)
    n = pos.shape[0]
    ij = np.zeros((0, 2), np.int32) if ij is None else np.asarray(ij, np.int32).reshape(-1, 2)
    rest = np.zeros(0, f32) if rest is None else np.asarray(rest, f32)
    return SoftbodyMesh(rest_pos=pos.copy(), pos=pos.copy(), vel=np.zeros_like(pos) if vel is None else np.asarray(vel, f32),
                        inv_mass=np.ones(n, f32) if w is None else np.asarray(w, f32), dist_ij=ij, dist_rest=rest)


def _pair(oracle_mod, mesh, ticks=3, S=7, **kw):
    sb = Softbody(mesh, substeps=S, **kw).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan())
        for _ in range(ticks):
            sb.step(); o.step(0.02, S)
        x, v = sb.get_positions(), sb.get_velocities()
    finally:
        sb.OnDestroy()
    assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32))
    assert np.array_equal(v.view(np.uint32), o.v.view(np.uint32))
    return x, v


def test_single_particle_no_constraints(oracle_mod):
    x, v = _pair(oracle_mod, _mesh([[0, 1, 0]], vel=[[1, 0, 0]]))
    assert x[0, 1] < 1.0 and x[0, 0] > 0.0


def test_particles_without_any_constraint_and_tiling_off(oracle_mod):
    rng = np.random.default_rng(0)
    _pair(oracle_mod, _mesh(rng.uniform(-1, 1, (1000, 3))), tile_particles=-1)
    _pair(oracle_mod, _mesh(rng.uniform(-1, 1, (1000, 3))), tile_particles=64)


def test_zero_length_spring_is_skipped_and_everything_pinned_is_static(oracle_mod):
    # coincident endpoints: L = 0 -> SPEC §4 skips the constraint (no NaN)
    m = _mesh([[0, 0, 0], [0, 0, 0], [1, 0, 0]], ij=[[0, 1], [1, 2]], rest=[0.5, 1.0])
    x, v = _pair(oracle_mod, m)
    assert np.isfinite(x).all()
    m = jelly_cube(6); m.inv_mass[:] = 0.0
    x, v = _pair(oracle_mod, m)
    assert np.array_equal(x, m.pos) and not v.any()


def test_one_substep_and_odd_substep_counts(oracle_mod):
    m = jelly_cube(9)
    for S in (1, 2, 3, 5):
        _pair(oracle_mod, m, ticks=4, S=S, tile_particles=64)


def test_ragged_cells_non_cubic_block_and_duplicate_springs(oracle_mod):
    # a 13 x 7 x 5 block (cells ragged at every face) with every x-spring listed twice
    nx, ny, nz = 13, 7, 5
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3).astype(f32)
    idx = lambda i, j, k: (i * ny + j) * nz + k
    ij = []
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                if i + 1 < nx: ij += [(idx(i, j, k), idx(i + 1, j, k))] * 2
                if j + 1 < ny: ij.append((idx(i, j, k), idx(i, j + 1, k)))
                if k + 1 < nz: ij.append((idx(i, j, k), idx(i, j, k + 1)))
    ij = np.array(ij, np.int32)
    m = _mesh(g + np.random.default_rng(1).uniform(-0.05, 0.05, g.shape).astype(f32), ij=ij, rest=np.ones(len(ij), f32))
    m.rest_pos = g
    for tile in (64, 512, -1):
        _pair(oracle_mod, m, tile_particles=tile)


@pytest.mark.parametrize("cluster_layers", [True, False])
def test_long_range_springs_go_to_cluster_tiles_or_global_colours(oracle_mod, monkeypatch, cluster_layers):
    # springs between random far-apart particles fit no grid cell of any tiling: their connected components become sparse
    # cluster tiles (one T2 layer); with SB_NO_CLUSTER_LAYERS they fall through to the global-colour kernels
    if not cluster_layers:
        monkeypatch.setenv("SB_NO_CLUSTER_LAYERS", "1")
    m = jelly_cube(10)
    rng = np.random.default_rng(2)
    extra = rng.integers(0, m.n, (300, 2)).astype(np.int32)
    extra = extra[extra[:, 0] != extra[:, 1]]
    m.dist_ij = np.concatenate([m.dist_ij, extra])
    m.dist_rest = np.concatenate([m.dist_rest, np.linalg.norm(m.rest_pos[extra[:, 0]] - m.rest_pos[extra[:, 1]], axis=1).astype(f32)])
    sb = Softbody(m, substeps=5, tile_particles=64).Start()
    try:
        st = sb.stats()
        if cluster_layers:
            assert st["n_global_colours"] == 0 and st["t2_constraints"] >= 250
        else:
            assert st["n_global_colours"] > 0
    finally:
        sb.OnDestroy()
    _pair(oracle_mod, m, S=5, tile_particles=64)


def test_bad_calls_are_rejected():
    import ctypes as C
    m = jelly_cube(4)
    sb = Softbody(m).Start()
    try:
        L = native.lib()
        assert L.sb_step(sb._h, -0.02, 10) == native.SB_ERR_INVALID_ARG
        assert L.sb_step(sb._h, 0.02, 0) == native.SB_ERR_INVALID_ARG
        assert L.sb_set_particles(sb._h, native.ptr(m.pos), None, native.ptr(m.inv_mass), m.n) == native.SB_ERR_STATE
        out = np.zeros((m.n + 1, 3), f32)
        assert L.sb_get_positions(sb._h, native.ptr(out), m.n + 1) == native.SB_ERR_INVALID_ARG
        assert L.sb_finalize(sb._h) == native.SB_ERR_STATE
        p = C.POINTER(C.c_float)()
        assert L.sb_readback_end(sb._h, C.byref(p)) == native.SB_ERR_STATE
    finally:
        sb.OnDestroy()
    bad = jelly_cube(4); bad.inv_mass[3] = -1.0
    with pytest.raises(native.SoftbodyError):
        Softbody(bad).Start()


@pytest.mark.parametrize("tile", [512, -1])
def test_spring_lengths_across_the_float_range(oracle_mod, tile):
    # the kernels use their own correctly rounded sqrt for squared lengths >= 2^-96 and SPEC §4 skips shorter springs:
    # independent two-particle springs whose lengths sweep 2^-60 .. 2^+55, a cluster right at the 2^-48 threshold
    rng = np.random.default_rng(5)
    n_pairs = 6000
    expo = np.concatenate([rng.uniform(-60, 55, n_pairs - 1000), rng.uniform(-48.6, -47.4, 1000)])
    length = np.exp2(expo)
    direction = rng.normal(size=(n_pairs, 3)); direction /= np.linalg.norm(direction, axis=1)[:, None]
    a = np.zeros((n_pairs, 3)); a[:, 0] = np.arange(n_pairs) * 0.0   # all pairs start at the origin side by side in index space
    b = a + direction * length[:, None]
    pos = np.empty((2 * n_pairs, 3), f32); pos[0::2] = a; pos[1::2] = b
    ij = np.stack([np.arange(n_pairs) * 2, np.arange(n_pairs) * 2 + 1], 1)
    rest = (length * rng.uniform(0.5, 1.5, n_pairs)).astype(f32)
    w = rng.choice(np.array([0.0, 0.5, 1.0, 3.0], f32), 2 * n_pairs)
    m = _mesh(pos, ij=ij, rest=rest, w=w)
    sb = Softbody(m, substeps=2, gravity=(0, 0, 0), tile_particles=tile).Start()
    try:
        o = make_oracle(oracle_mod, m, sb.plan(), gravity=(0, 0, 0))
        x0 = sb.get_positions().copy()
        sb.step(); o.step(0.02, 2)
        x, v = sb.get_positions(), sb.get_velocities()
    finally:
        sb.OnDestroy()
    assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), o.v.view(np.uint32))
    moved = (x != x0).any(1).reshape(-1, 2).any(1)
    L2 = ((pos[0::2].astype(np.float64) - pos[1::2]) ** 2).sum(1)
    assert not moved[L2 < 2.0 ** -97].any() and moved[(L2 > 2.0 ** -95) & (w.reshape(-1, 2).sum(1) > 0)].mean() > 0.9


@pytest.mark.parametrize("seed", range(6))
def test_random_irregular_meshes(oracle_mod, seed):
    # randomized clouds (anisotropic, mixed masses, springs + tets + hinges), random tile size: bit-exact against the oracle
    from test_plan import _random_mesh
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(60, 3000))
    stretch = [(1, 1, 1), (4, 1, 0.25), (1, 0.05, 1), (8, 8, 0.1)][seed % 4]
    tile = [32, 64, 256, -1, 512, 128][seed]
    mesh = _random_mesh(seed, n, stretch)
    kw = dict(distance_compliance=1e-6, volume_compliance=1e-6, bending_compliance=1e-4, tile_particles=tile)
    sb = Softbody(mesh, substeps=4, **kw).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan(), compliance=(1e-6, 1e-6, 1e-4))
        for _ in range(2):
            sb.step(); o.step(0.02, 4)
        x, v = sb.get_positions(), sb.get_velocities()
    finally:
        sb.OnDestroy()
    assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), o.v.view(np.uint32))


def test_create_destroy_cycles_do_not_leak_device_memory():
    # every handle owns streams, events, graphs, pinned snapshots and device buffers: 40 full life cycles must give the
    # memory back (hipMemGetInfo of the runtime the plugin itself is linked against)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    mesh = jelly_cube(24)

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def cycle(readback):
        sb = Softbody(mesh, substeps=4).Start()
        sb.step(); sb.step(0.01, 3)
        if readback:
            sb.readback_begin(); sb.step(); sb.readback_end()
        sb.get_positions()
        sb.OnDestroy()

    for k in range(3):
        cycle(True)
    free0 = free_bytes()
    for k in range(40):
        cycle(k % 2 == 0)
    free1 = free_bytes()
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 40 create/destroy cycles"


@pytest.mark.parametrize("tile", [512, -1])
def test_hub_particle_with_300_springs(oracle_mod, tile):
    # valence 300 -> 300+ rounds in one tile (round words read from memory, not LDS or lanes) / 300 global colours
    from test_plan import _hub_mesh
    _pair(oracle_mod, _hub_mesh(), ticks=2, S=3, tile_particles=tile)


def _fan_mesh(n_ring=150, n_hinges=40, seed=7):
    """A hub and an apex joined to a ring: tet (hub, r_i, r_i+1, apex) for every ring edge, springs hub - r_i and along the ring,
    hinges on the edge (hub, apex) with wings (r_i, r_i+1): hub and apex sit in every tet and hinge, so one tile walks n_ring + n_hinges
    groups (more than the 64 wave items a register holds)."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0.0, 2.0 * np.pi, n_ring, endpoint=False)
    ring = np.stack([np.cos(ang), np.zeros(n_ring), np.sin(ang)], axis=1) * rng.uniform(0.9, 1.1, (n_ring, 1))
    pos = np.concatenate([[[0.0, -0.4, 0.0]], [[0.0, 0.9, 0.0]], ring]).astype(f32)
    hub, apex, r = 0, 1, lambda i: 2 + i % n_ring
    ij = np.array([(hub, r(i)) for i in range(n_ring)] + [(r(i), r(i + 1)) for i in range(n_ring)] + [(hub, apex)], np.int32)
    rest = (np.linalg.norm(pos[ij[:, 0]] - pos[ij[:, 1]], axis=1) * 0.95).astype(f32)
    tets = np.array([(hub, r(i), r(i + 1), apex) for i in range(n_ring)], np.int32)
    e1, e2, e3 = pos[tets[:, 1]] - pos[tets[:, 0]], pos[tets[:, 2]] - pos[tets[:, 0]], pos[tets[:, 3]] - pos[tets[:, 0]]
    vol = (np.einsum("ij,ij->i", e1, np.cross(e2, e3)) / 6.0 * 0.97).astype(f32)
    step = max(1, n_ring // n_hinges)
    hinges = np.array([(hub, apex, r(i), r(i + 1)) for i in range(0, n_ring, step)][:n_hinges], np.int32)
    phi = np.full(len(hinges), 0.3, np.float64)
    bend = np.stack([np.cos(phi), np.sin(phi)], axis=1).astype(f32)
    return SoftbodyMesh(rest_pos=pos.copy(), pos=pos.copy(), vel=np.zeros_like(pos), inv_mass=np.ones(len(pos), f32), dist_ij=ij, dist_rest=rest,
                        vol_ijkl=tets, vol_rest=vol, bend_ijkl=hinges, bend_rest=bend)


@pytest.mark.parametrize("lanes", ["256", "512"])
@pytest.mark.parametrize("window", [None, "1024"])
def test_hub_with_tets_and_hinges_walks_more_than_64_steps(oracle_mod, monkeypatch, lanes, window):
    # the wave items of a tile are kept 64 steps at a time in a register: 190 dependent groups reload it twice per pass;
    # with a 4 KiB LDS window the tile's data does not fit and the generic group loop (window refills) runs instead
    monkeypatch.setenv("SB_QUAD_LANES", lanes)
    if window:
        monkeypatch.setenv("SB_WIN_DWORDS", window)
    mesh = _fan_mesh()
    kw = dict(distance_compliance=1e-6, volume_compliance=1e-6, bending_compliance=1e-3)
    sb = Softbody(mesh, substeps=4, **kw).Start()
    try:
        plan = sb.plan()
        assert len(plan.groups(0)) - 1 >= 150
        o = make_oracle(oracle_mod, mesh, plan, compliance=(1e-6, 1e-6, 1e-3))
        for _ in range(3):
            sb.step(); o.step(0.02, 4)
        x, v = sb.get_positions(), sb.get_velocities()
    finally:
        sb.OnDestroy()
    assert np.isfinite(x).all()
    assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), o.v.view(np.uint32))
