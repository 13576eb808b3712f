"""Kinematic particles (SPEC.md 2, sb_set_kinematic_positions): pinned particles moved by the host between ticks -- the attachment of a
soft body to an animated object. Bit for bit the oracle with the same assignment, under every way the tick boundary can be crossed."""
import numpy as np
import pytest

from helpers import make_oracle
from softbodyunity_amd import Softbody, bunny_surrogate, jelly_cube, native

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("peek_small", [False, True])
@pytest.mark.parametrize("case", ["hanging_cube", "tets_dragged_over_the_ground", "hanging_cube_global_colours_only"])
def test_moved_pins_match_the_oracle(case, peek_small, oracle_mod, monkeypatch):
    # peek_small: position reads between a move and the step peek (side array + the pending targets scattered onto it) instead of
    # completing the tick, also on these small meshes
    if peek_small:
        monkeypatch.setenv("SB_PEEK_MIN_TILES", "0")
    else:
        monkeypatch.delenv("SB_PEEK_MIN_TILES", raising=False)
    if case.startswith("hanging_cube"):
        n = 20
        mesh = jelly_cube(n)
        pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)       # the top layer
        kw = dict(substeps=8, damping=0.05)
        if case.endswith("global_colours_only"):
            kw["tile_particles"] = -1              # tiles without constraints of their own: the MARK step alone carries the targets
        okw = dict(damping=0.05)
    else:
        mesh = bunny_surrogate(target_verts=5000, seed=11)
        pins = np.argsort(mesh.pos[:, 0])[-40:].astype(np.int32)                                    # a handle on one side
        kw = dict(substeps=6, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-4,
                  ground_plane=(0, 1, 0, float(mesh.pos[:, 1].min()) - 0.05))
        okw = dict(compliance=(1e-7, 1e-7, 1e-4), ground_plane=kw["ground_plane"])
    mesh.inv_mass[pins] = 0.0
    rest = mesh.pos[pins].copy()
    sb = Softbody(mesh, **kw).Start()
    try:
        o = make_oracle(oracle_mod, mesh, sb.plan(), **okw)
        for t in range(12):
            # the handle swings; every second tick a position read sits between the move and the step (peek / flush paths), and one
            # tick is stepped without a move (the fused tick boundary comes back)
            if t != 5:
                target = rest + np.array([0.3 * np.sin(0.4 * t), 0.1 * np.cos(0.7 * t) - 0.1, 0.05 * t], np.float32)
                sb.set_kinematic_positions(pins, target)
                o.set_kinematic_positions(pins, target)
            if t & 1:
                assert np.array_equal(_bits(sb.get_positions()), _bits(o.x)), f"after the move of tick {t}"
            sb.step()
            o.step(0.02, kw["substeps"])
        x, v = sb.get_positions(), sb.get_velocities()
        assert np.array_equal(_bits(x), _bits(o.x)) and np.array_equal(_bits(v), _bits(o.v))
        assert np.array_equal(_bits(x[pins]), _bits(target)) and not v[pins].any()              # where they were put, at rest
        st = sb.stats()
        # moves that met a held-back last kernel travelled INSIDE the fused first kernel of the next tick (tile_kernel KIND 5)
        assert st["ticks_fused_kinematic"] >= (9 if peek_small else 4), st
        assert st["ticks_fused"] >= st["ticks_fused_kinematic"]
        moved = np.linalg.norm(x - mesh.pos, axis=1)
        assert moved[np.setdiff1d(np.arange(mesh.n), pins)].max() > 0.05                         # the body followed
    finally:
        sb.OnDestroy()


def test_only_pinned_particles_are_kinematic_and_nothing_changes_on_error():
    mesh = jelly_cube(10)
    mesh.inv_mass[:100] = 0.0
    sb = Softbody(mesh, substeps=4).Start()
    try:
        sb.step()
        before = sb.get_positions().copy()
        with pytest.raises(native.SoftbodyError, match="non-zero inverse mass"):
            sb.set_kinematic_positions([3, 500], np.zeros((2, 3), np.float32))           # 500 is a free particle
        with pytest.raises(native.SoftbodyError, match="out of range"):
            sb.set_kinematic_positions([3, mesh.n], np.zeros((2, 3), np.float32))
        with pytest.raises(native.SoftbodyError, match="NaN"):
            sb.set_kinematic_positions([3], np.full((1, 3), np.nan, np.float32))
        assert np.array_equal(_bits(sb.get_positions()), _bits(before))
        sb.set_kinematic_positions(np.zeros(0, np.int32), np.zeros((0, 3), np.float32))   # an empty list is fine
        # many calls in a row (the ring of host tables wraps), growing lists
        for k in range(1, 12):
            ids = np.arange(min(100, 10 * k), dtype=np.int32)
            sb.set_kinematic_positions(ids, before[ids] + 0.01 * k)
        sb.step()
        got = sb.get_positions()
        assert np.array_equal(_bits(got[:100]), _bits(before[:100] + np.float32(0.01 * 11)))
    finally:
        sb.OnDestroy()


def test_a_rank_of_a_partitioned_solver_moves_the_pins_it_owns(oracle_mod):
    # every rank is handed the whole list; it applies the entries it owns (the ghost copies on its neighbours arrive with the next
    # exchange) -- here 8 ranks of one mesh on the one GPU, the host as the wire (tests/hosted.py), against the oracle
    from hosted import HostedRanks
    mesh = jelly_cube(24)
    pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)
    mesh.inv_mass[pins] = 0.0
    rest = mesh.pos[pins].copy()
    with HostedRanks(mesh, 8, 6, tile_particles=64, damping=0.05) as H:
        o = make_oracle(oracle_mod, mesh, H.ranks[0].plan(), damping=0.05)
        owners = H.ranks[0].owner()
        assert len(np.unique(owners[pins])) >= 4                     # the pinned layer is spread over several ranks
        for t in range(4):
            target = rest + np.array([0.2 * np.sin(0.5 * t), -0.05 * t, 0.1 * t], np.float32)
            for sb in H.ranks:
                sb.set_kinematic_positions(pins, target)
            o.set_kinematic_positions(pins, target)
            H.tick(); o.step(0.02, 6)
        x, v, _ = H.merged_state()
        assert np.array_equal(_bits(x), _bits(o.x)) and np.array_equal(_bits(v), _bits(o.v))
        assert np.array_equal(_bits(x[pins]), _bits(target))
        with pytest.raises(native.SoftbodyError, match="twice"):
            H.ranks[0].set_kinematic_positions([int(pins[0]), int(pins[0])], np.zeros((2, 3), np.float32))
