"""GPU parity: the HIP plugin, called through the C ABI, against the CPU oracle on identical meshes and
the published schedule. Stated tolerance (BASELINE.json:5): max_i |x_i - x_i^oracle|_2 / diag(bbox0) <= 1e-4;
SPEC.md §1 additionally makes bit-for-bit agreement the expected outcome, which is asserted where noted."""
import numpy as np
import pytest

from softbodyunity_amd import Softbody, native
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube
from helpers import make_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run_pair(oracle_mod, mesh, ticks, substeps, dt=0.02, compliance=(0.0, 0.0, 0.0), damping=0.0, **kw):
    sb = Softbody(mesh, substeps=substeps, fixed_delta_time=dt, damping=damping, distance_compliance=compliance[0],
                  volume_compliance=compliance[1], bending_compliance=compliance[2], **kw).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan(), damping=damping, compliance=compliance)
        for _ in range(ticks):
            sb.FixedUpdate(readback=False)
            o.step(dt, substeps)
        x = sb.get_positions(); v = sb.get_velocities()
        st = sb.stats()
        rep = sb.validate()          # every parity case also has the validator kernel re-read the tables its launches used
        assert rep["errors"] == [0] * 6 and rep["constraints_checked"] == sum(st["n_constraints_local"]), rep
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, o.x, mesh.pos)
    return rel, mabs, bit, x, v, o, st


@pytest.mark.parametrize("tile", [512, 64, -1])
def test_cfg1_cube8_10substeps(oracle_mod, tile):
    # BASELINE.json:7 — 8x8x8 jelly cube, 512 particles, 1344 springs, 10 substeps
    mesh = jelly_cube(8)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=20, substeps=10, tile_particles=tile)
    assert rel <= TOL, (rel, mabs)
    assert bit, f"expected bit-exact positions (SPEC.md §1), max abs diff {mabs}"
    assert np.array_equal(v.view(np.uint32), o.v.view(np.uint32))


def test_cfg1_golden_fixture(oracle_mod):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cube8_s10_t5.npz"))
    mesh = jelly_cube(8)
    sb = Softbody(mesh, substeps=10).Start()
    try:
        for parity in (0, 1):
            t, ids = sb.plan().order(parity)
            assert np.array_equal(ids, g[f"order_id{parity}"]), "schedule changed: regenerate tests/golden (make_golden.py)"
        for _ in range(5):
            sb.FixedUpdate(readback=False)
        x = sb.get_positions()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, g["x"], mesh.pos)
    assert rel <= TOL and bit


@pytest.mark.parametrize("tile,graph", [(512, True), (512, False), (-1, True)])
def test_cfg2_cube64_20substeps(oracle_mod, tile, graph):
    # BASELINE.json:8 — 64^3 jelly cube, 20 substeps, 1 GPU, fp32
    mesh = jelly_cube(64)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=3, substeps=20, tile_particles=tile, use_graph=graph)
    assert rel <= TOL, (rel, mabs)
    assert bit, f"max abs diff {mabs}"
    if tile > 0:
        # 8^3 aligned cells; 9^3 shifted cells of which the 386 partial rim cells share ~170 workgroups (tile packing)
        assert st["n_tilings"] == 2 and st["n_global_colours"] == 0
        assert st["n_tiles"][0] == 512 and 512 <= st["n_tiles"][1] <= 520


def test_cfg2_state_after_100_ticks(oracle_mod):
    # SURVEY.md 8c parity metric: state T with T = 100 ticks (2 000 substeps) -- the cube hangs from its pinned top layer and swings;
    # max_i |x_gpu - x_cpu| / diag(bbox_0) <= 1e-4 is the stated bar, identical bits the expected outcome
    mesh = jelly_cube(64, pin_top=True)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=100, substeps=20)
    assert rel <= TOL and bit, (rel, mabs)
    assert np.array_equal(v.view(np.uint32), o.v.view(np.uint32))
    assert np.abs(x - mesh.pos).max() > 0.25      # it really went somewhere


def test_pinned_top_layer_and_damping_and_compliance(oracle_mod):
    mesh = jelly_cube(20, pin_top=True)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=10, substeps=8, compliance=(1e-6, 0, 0), damping=0.5,
                                            tile_particles=128)
    assert rel <= TOL and bit
    top = mesh.inv_mass == 0
    assert np.array_equal(x[top], mesh.pos[top])


@pytest.mark.parametrize("third_tiling", [True, False])
def test_full_stencil_mixed_tile_and_global(oracle_mod, monkeypatch, third_tiling):
    # diagonal springs lie inside neither T0 nor T1: with the third tiling they run in the T2 layers' LDS tiles, without
    # it in global colours between the two tile phases
    if not third_tiling:
        monkeypatch.setenv("SB_NO_T2", "1")
    mesh = jelly_cube(14, stencil="full")
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=5, substeps=10, tile_particles=64, compliance=(1e-7, 0, 0))
    assert st["constraints_in_tiles"] > 0
    if third_tiling:
        assert st["n_t2_layers"] > 0 and st["t2_constraints"] > 0 and st["n_t2_tiles"] > 0
    else:
        assert st["n_t2_layers"] == 0 and st["n_global_colours"] > 0 and st["constraints_in_global"] > 0
    assert rel <= TOL and bit


def test_changing_dt_and_substeps_between_ticks(oracle_mod):
    mesh = jelly_cube(10)
    sb = Softbody(mesh).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan())
        for dt, S in ((0.02, 10), (0.01, 4), (0.02, 10), (1 / 60, 7), (0.02, 1), (0.02, 3)):
            sb.step(dt, S); o.step(dt, S)
        x = sb.get_positions()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, o.x, mesh.pos)
    assert rel <= TOL and bit


def test_graph_cache_evicts_when_the_host_varies_substeps(oracle_mod):
    # one hipGraphExec per (substeps, tick flavour), at most 8 kept (least recently used goes): more distinct substep
    # counts than that, revisited, must still give the oracle's bits
    mesh = jelly_cube(10)
    sb = Softbody(mesh).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan())
        for S in list(range(1, 14)) + [2, 4, 2, 13, 1, 8, 8]:
            sb.step(0.02, S); o.step(0.02, S)
        x = sb.get_positions(); v = sb.get_velocities()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, o.x, mesh.pos)
    assert rel <= TOL and bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32))


def test_state_round_trip(oracle_mod):
    # download -> upload -> step == step
    mesh = jelly_cube(12)
    a = Softbody(mesh, substeps=10).Start(); b = Softbody(mesh, substeps=10).Start()
    try:
        a.step(); b.step()
        b.set_state(b.get_positions(), b.get_velocities())
        a.step(); b.step()
        assert np.array_equal(a.get_positions().view(np.uint32), b.get_positions().view(np.uint32))
    finally:
        a.OnDestroy(); b.OnDestroy()


@pytest.fixture(scope="module")
def bunny20k():
    return bunny_surrogate(target_verts=20000, seed=1234)


def test_cfg5_surrogate_state_after_60_ticks_on_the_ground(oracle_mod, bunny20k):
    # a long run on the irregular mesh: springs + volumes + hinges, dropped on a ground plane, 60 ticks x 20 substeps, bit for bit
    mesh = bunny20k
    comp = (1e-7, 1e-7, 1e-5)
    plane = (0.0, 1.0, 0.0, float(mesh.pos[:, 1].min()) - 0.05)
    sb = Softbody(mesh, substeps=20, distance_compliance=comp[0], volume_compliance=comp[1], bending_compliance=comp[2], ground_plane=plane).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan(), compliance=comp, ground_plane=plane)
        for _ in range(60):
            sb.FixedUpdate(readback=False)
            o.step(0.02, 20)
        x = sb.get_positions(); v = sb.get_velocities()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, o.x, mesh.pos)
    assert np.isfinite(x).all() and rel <= TOL and bit, (rel, mabs)
    assert np.array_equal(v.view(np.uint32), o.v.view(np.uint32))
    assert x[:, 1].min() >= plane[3] - 1e-6 and np.abs(x - mesh.pos).max() > 0.02      # it landed


@pytest.mark.parametrize("tile", [256, -1])
def test_cfg5_surrogate_volume_bending(oracle_mod, bunny20k, tile):
    # BASELINE.json:11 (single-GPU part): irregular tet mesh SURROGATE with distance+volume+bending
    mesh = bunny20k
    comp = (1e-7, 1e-7, 1e-4)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=5, substeps=10, compliance=comp, tile_particles=tile)
    assert st["n_constraints_local"][1] == len(mesh.vol_rest) and st["n_constraints_local"][2] == len(mesh.bend_rest)
    assert np.isfinite(x).all()
    assert rel <= TOL, (rel, mabs)
    assert bit, f"max abs diff {mabs}"


def test_rest_lattice_is_a_bitwise_fixed_point_on_gpu():
    mesh = jelly_cube(16, perturb=0.0)
    sb = Softbody(mesh, gravity=(0, 0, 0), substeps=10).Start()
    try:
        for _ in range(5):
            sb.step()
        assert np.array_equal(sb.get_positions().view(np.uint32), mesh.pos.view(np.uint32))
        assert not sb.get_velocities().any()
    finally:
        sb.OnDestroy()


def test_error_paths_through_the_abi():
    import ctypes as C
    L = native.lib()
    d = native.SbDesc(); L.sb_desc_default(C.byref(d))
    h = C.c_void_p()
    assert L.sb_create(C.byref(d), C.byref(h)) == 0
    assert L.sb_step(h, 0.02, 10) == native.SB_ERR_STATE
    assert L.sb_finalize(h) == native.SB_ERR_STATE
    pos = np.zeros((4, 3), np.float32); w = np.ones(4, np.float32)
    assert L.sb_set_particles(h, native.ptr(pos), None, native.ptr(w), 4) == 0
    ij = np.array([[0, 9]], np.int32); r = np.ones(1, np.float32)
    assert L.sb_set_distance_constraints(h, native.ptr(ij), native.ptr(r), 1, 0.0) == native.SB_ERR_INVALID_ARG
    assert b"out of range" in L.sb_last_error()
    assert L.sb_destroy(h) == 0
    d.device = 99
    assert L.sb_create(C.byref(d), C.byref(h)) == native.SB_ERR_NO_DEVICE


def test_large_cube_properties():
    # 128^3 (2.1M particles): size-independent properties instead of an oracle run —
    # momentum: with g=0 and equal masses the centre of mass stays put; pinned lattice at rest is a fixed point
    mesh = jelly_cube(128)
    sb = Softbody(mesh, gravity=(0, 0, 0), substeps=20).Start()
    try:
        c0 = mesh.pos.astype(np.float64).mean(0)
        for _ in range(3):
            sb.step()
        x = sb.get_positions()
        assert np.isfinite(x).all()
        assert np.abs(x.astype(np.float64).mean(0) - c0).max() < 1e-4
        # springs relax towards rest length
        d0 = np.linalg.norm(mesh.pos[mesh.dist_ij[:, 0]] - mesh.pos[mesh.dist_ij[:, 1]], axis=1) - 1.0
        d1 = np.linalg.norm(x[mesh.dist_ij[:, 0]] - x[mesh.dist_ij[:, 1]], axis=1) - 1.0
        assert np.abs(d1).mean() < 0.5 * np.abs(d0).mean()
    finally:
        sb.OnDestroy()


@pytest.mark.parametrize("tile", [512, -1])
def test_ground_plane_drop(oracle_mod, tile):
    # a 12^3 jelly cube dropped on the plane y >= -0.5: collide step fused into the tile kernel's MARK round
    mesh = jelly_cube(12)
    plane = (0.0, 1.0, 0.0, -0.5)
    sb = Softbody(mesh, substeps=10, tile_particles=tile, ground_plane=plane, distance_compliance=1e-6).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan(), compliance=(1e-6, 0, 0), ground_plane=plane)
        for _ in range(40):
            sb.step(); o.step(0.02, 10)
        x = sb.get_positions(); v = sb.get_velocities()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, o.x, mesh.pos)
    assert x[:, 1].min() >= -0.5 and (x[:, 1] < -0.49).sum() > 10      # it landed and rests on the plane
    assert rel <= TOL and bit, (rel, mabs)
    assert np.array_equal(v.view(np.uint32), o.v.view(np.uint32))


def test_cfg5_surrogate_100k(oracle_mod):
    # BASELINE.json:11 at its stated size (~100k vertices), single GPU: distance + volume + bending, bit-exact
    mesh = bunny_surrogate(target_verts=100_000, seed=1234)
    assert 80_000 < mesh.n < 125_000
    comp = (1e-7, 1e-7, 1e-4)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, mesh, ticks=3, substeps=20, compliance=comp, tile_particles=256)
    assert st["constraints_in_tiles"] > 0.6 * (len(mesh.dist_rest) + len(mesh.vol_rest) + len(mesh.bend_rest))
    assert np.isfinite(x).all()
    assert rel <= TOL and bit, (rel, mabs)


def test_async_readback_matches_blocking_readback():
    # snapshots taken between ticks equal sb_get_positions at the same point, while later ticks
    # are already enqueued behind them
    mesh = jelly_cube(32)
    sb = Softbody(mesh, substeps=10).Start()
    try:
        ref = []
        for _ in range(3):
            sb.step(); ref.append(sb.get_positions().copy())
        sb.set_state(mesh.pos, mesh.vel)
        sb.step(); sb.readback_begin()
        sb.step(); sb.readback_begin()                 # two pending, both ticks enqueued before either is read
        a = sb.readback_end().copy()
        sb.step(); sb.readback_begin()
        b = sb.readback_end().copy(); c = sb.readback_end().copy()
        for got, want in zip((a, b, c), ref):
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        with pytest.raises(native.SoftbodyError):
            sb.readback_end()
        # pointer lifetime (softbody.h): a snapshot handed out by readback_end stays intact through the NEXT
        # readback_begin even when another snapshot was pending at the time (three buffers for two pending snapshots)
        sb.set_state(mesh.pos, mesh.vel)
        sb.step(); sb.readback_begin()
        sb.step(); sb.readback_begin()
        held = sb.readback_end()                        # view of the plugin's pinned buffer, not a copy
        sb.step(); sb.readback_begin()                  # must not land in the buffer `held` points at
        sb.synchronize()
        second = sb.readback_end().copy(); third = sb.readback_end().copy()
        assert np.array_equal(held.view(np.uint32), ref[0].view(np.uint32))
        assert np.array_equal(second.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(third.view(np.uint32), ref[2].view(np.uint32))
    finally:
        sb.OnDestroy()


def test_lazy_tick_boundary_is_invisible(oracle_mod, monkeypatch):
    # the deferred last kernel of a tick is fused into the next tick or flushed on any read / parameter change:
    # every interleaving must give the bits of the eager schedule (SB_NO_LAZY_TICK) and of the oracle
    mesh = jelly_cube(20)
    plan_steps = [(0.02, 10, False), (0.02, 10, False), (0.02, 10, True), (0.02, 10, False), (0.01, 10, False),
                  (0.01, 10, False), (0.01, 6, False), (0.01, 7, True), (0.01, 6, False), (0.01, 6, False)]

    def run(lazy):
        if lazy:
            monkeypatch.delenv("SB_NO_LAZY_TICK", raising=False)
        else:
            monkeypatch.setenv("SB_NO_LAZY_TICK", "1")
        sb = Softbody(mesh, ground_plane=(0, 1, 0, -2.0)).Start()
        try:
            snaps = []
            for dt, S, read in plan_steps:
                sb.step(dt, S)
                if read:
                    snaps.append(sb.get_positions().copy())
            orders = [sb.plan().order(par) for par in (0, 1)]
            return sb.get_positions().copy(), sb.get_velocities().copy(), snaps, orders
        finally:
            sb.OnDestroy()
    xa, va, sa, orders = run(True)
    xb, vb, sb_, _ = run(False)
    assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and np.array_equal(va.view(np.uint32), vb.view(np.uint32))
    for p, q in zip(sa, sb_):
        assert np.array_equal(p.view(np.uint32), q.view(np.uint32))
    o = make_oracle(oracle_mod, mesh, None, ground_plane=(0, 1, 0, -2.0))
    for par in (0, 1):
        o.set_order(orders[par][0], orders[par][1], parity=par)
    for dt, S, _ in plan_steps:
        o.step(dt, S)
    assert np.array_equal(xa.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(va.view(np.uint32), o.v.view(np.uint32))


def test_tile_packing_is_invisible(oracle_mod, bunny20k, monkeypatch):
    # under-full tiles share a workgroup (csrc/tables.hip build_device); the members keep their own round order, so the
    # bits must not depend on the packing -- irregular mesh with all three constraint types, and the lattice rim
    def run(mesh, pack, **kw):
        if pack:
            monkeypatch.delenv("SB_NO_PACK", raising=False)
        else:
            monkeypatch.setenv("SB_NO_PACK", "1")
        sb = Softbody(mesh, **kw).Start()
        try:
            for _ in range(2):
                sb.step(0.02, 6)
            return sb.get_positions().copy(), sb.get_velocities().copy(), sb.stats()["n_tiles"]
        finally:
            sb.OnDestroy()
    for mesh, kw in ((bunny20k, dict(tile_particles=256, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5)), (jelly_cube(24), dict())):
        xa, va, ta = run(mesh, True, **kw)
        xb, vb, tb = run(mesh, False, **kw)
        assert sum(ta) < sum(tb), (ta, tb)
        assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and np.array_equal(va.view(np.uint32), vb.view(np.uint32))


@pytest.mark.parametrize("switch", ["SB_NO_MIXED_GROUPS", "SB_NO_CLUSTER_LAYERS", "SB_NO_THIRD_LIST", "SB_NO_T2", "SB_NO_PACK", "SB_NO_PALETTE",
                                    "SB_NO_MASS_PALETTE", "SB_NO_UNIFORM_MASS", "SB_NO_BANK_ORDER", "SB_NO_LAZY_TICK", "SB_NO_WAVE_ITEMS",
                                    "SB_NO_COST_ORDER", "SB_STORE_THROUGH_MAX_TILES"])
def test_every_diagnostic_switch_still_matches_the_oracle(oracle_mod, bunny20k, monkeypatch, switch):
    # the A/B switches of DESIGN.md 6 change the plan (then the published order changes with it) or only the device
    # layout: either way the plugin must reproduce the oracle walking the order the plugin publishes
    monkeypatch.setenv(switch, "1")
    comp = (1e-7, 1e-7, 1e-5)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, bunny20k, ticks=2, substeps=7, compliance=comp)
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (switch, rel, mabs)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, jelly_cube(20, pin_top=True), ticks=2, substeps=7, tile_particles=64)
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (switch, rel, mabs)


@pytest.mark.parametrize("mask", ["1", "2", "3"])
def test_write_through_store_masks_match_the_oracle(oracle_mod, bunny20k, monkeypatch, mask):
    # small launches store previous positions (bit 0) and positions (bit 1) through the L2; force each choice separately
    monkeypatch.setenv("SB_STORE_THROUGH_MAX_TILES", "0")
    monkeypatch.setenv("SB_STORE_THROUGH_LARGE", mask)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, jelly_cube(40), ticks=2, substeps=20, ground_plane=None)
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (rel, mabs)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, bunny20k, ticks=2, substeps=6, compliance=(1e-7, 1e-7, 1e-5))
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (rel, mabs)


@pytest.mark.parametrize("lanes", ["128", "256"])
def test_both_workgroup_widths_match_the_oracle(oracle_mod, bunny20k, monkeypatch, lanes):
    # small tiles run as 256-lane workgroups (one constraint per lane and round) or, in launches of >= 10240 tiles,
    # as 128-lane workgroups (two per lane): force each width on meshes the oracle finishes in seconds
    monkeypatch.setenv("SB_TILE_LANES", lanes)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, jelly_cube(40), ticks=2, substeps=20, ground_plane=None)
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (rel, mabs)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, bunny20k, ticks=2, substeps=6, compliance=(1e-7, 1e-7, 1e-5), tile_particles=256)
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (rel, mabs)


@pytest.mark.parametrize("lanes", ["256", "512"])
@pytest.mark.parametrize("tile", [256, 1024])
def test_both_widths_of_tiles_with_tets_and_hinges_match_the_oracle(oracle_mod, bunny20k, monkeypatch, lanes, tile):
    # tiles that hold four-lane constraints run as 4-wave or 8-wave workgroups (SB_QUAD_LANES): the host deals the wave
    # items for that many waves, the bits must not change; 1024-particle tiles use the large-tile instantiations
    monkeypatch.setenv("SB_QUAD_LANES", lanes)
    rel, mabs, bit, x, v, o, st = _run_pair(oracle_mod, bunny20k, ticks=2, substeps=6, compliance=(1e-7, 1e-7, 1e-5), tile_particles=tile)
    assert bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)), (rel, mabs)


def test_bench_configuration_256_properties(monkeypatch):
    # BASELINE.json:9 at full size (16.8 M particles, 50.1 M springs): no oracle run, size-independent properties instead --
    # (1) the rest lattice is a bitwise fixed point without gravity, (2) both workgroup widths and the unpacked launch
    # give the same bits after two ticks of the perturbed cube, (3) the centre of mass stays put, springs relax
    rest = jelly_cube(256, perturb=0.0)
    sb = Softbody(rest, gravity=(0, 0, 0), substeps=20).Start()
    try:
        sb.step(); sb.step()
        assert np.array_equal(sb.get_positions().view(np.uint32), rest.pos.view(np.uint32)) and not sb.get_velocities().any()
        assert sb.stats()["n_tiles"][0] == 32768
    finally:
        sb.OnDestroy()
    del rest
    mesh = jelly_cube(256)
    runs = []
    for env in ({"SB_TILE_LANES": "128"}, {"SB_TILE_LANES": "256"}, {"SB_NO_PACK": "1"}):
        for k in ("SB_TILE_LANES", "SB_NO_PACK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sb = Softbody(mesh, gravity=(0, 0, 0), substeps=20).Start()
        try:
            sb.step(); sb.step()
            runs.append((sb.get_positions().copy(), sb.get_velocities().copy()))
        finally:
            sb.OnDestroy()
    for x, v in runs[1:]:
        assert np.array_equal(x.view(np.uint32), runs[0][0].view(np.uint32)) and np.array_equal(v.view(np.uint32), runs[0][1].view(np.uint32))
    x = runs[0][0]
    assert np.isfinite(x).all()
    assert np.abs(x.astype(np.float64).mean(0) - mesh.pos.astype(np.float64).mean(0)).max() < 1e-4
    ij = mesh.dist_ij[::97]
    d0 = np.linalg.norm(mesh.pos[ij[:, 0]] - mesh.pos[ij[:, 1]], axis=1) - 1.0
    d1 = np.linalg.norm(x[ij[:, 0]] - x[ij[:, 1]], axis=1) - 1.0
    assert np.abs(d1).mean() < 0.5 * np.abs(d0).mean()
