"""Sharded authoring (include/softbody.h sb_domain): a rank that plans only its WINDOW of the mesh must arrive at exactly what it
gets when it plans the whole mesh -- same owned particles, same ghosts in the same order, same constraints executed in the same
order, same halo lists, same pair hashes -- and the partitioned run on windows must equal the unpartitioned oracle bit for bit.
Pure host code (no GPU)."""
import numpy as np
import pytest

from softbodyunity_amd import native
from softbodyunity_amd.mesh import jelly_cube, jelly_cube_window
from helpers import WindowRankSim, build_plan, make_oracle, run_tick


def _seq(plan, mesh, gid, parity):
    """The constraints this rank executes, in order, as rows (type, whole-mesh ids of the end points)."""
    t, ids = plan.order(parity)
    m = plan.local_order_mask(parity).astype(bool)
    assert (t[m] == 0).all()
    return gid[mesh.dist_ij[ids[m]]]


@pytest.mark.parametrize("n,tile,world,dims", [(32, 64, 8, (0, 0, 0)), (32, 64, 2, (0, 0, 0)), (40, 64, 4, (1, 2, 2)), (24, 27, 8, (2, 2, 2))])
def test_window_plan_equals_whole_plan(n, tile, world, dims):
    whole = jelly_cube(n, pin_top=True)
    ident = np.arange(whole.n)
    for rank in range(world):
        pw = build_plan(whole, rank=rank, world=world, part_dims=dims, tile_particles=tile)
        win = jelly_cube_window(n, rank, world, dims, tile, pin_top=True)
        assert win.n < whole.n or world == 1
        gid = win.global_id.astype(np.int64)
        ps = native.Plan.build(win.rest_pos, win.dist_ij, rank=rank, world=world, part_dims=dims, tile_particles=tile,
                               domain=win.domain, global_id=win.global_id)
        # owned + ghost particles, device order
        lw, ow = pw.local_particles(); ls, os_ = ps.local_particles()
        assert ow == os_ and np.array_equal(lw, gid[ls])
        assert np.array_equal(np.nonzero(pw.owner(whole.n) == rank)[0], np.sort(gid[ps.owner(win.n) == rank]))
        # constraints executed, in order, both parities
        for parity in (0, 1):
            assert np.array_equal(_seq(pw, whole, ident, parity), _seq(ps, win, gid, parity))
            # ... grouped the same way (the kernel runs a group concurrently)
            gw = np.diff(np.concatenate([[0], np.cumsum(pw.local_order_mask(parity))])[pw.groups(parity)])
            gs = np.diff(np.concatenate([[0], np.cumsum(ps.local_order_mask(parity))])[ps.groups(parity)])
            assert np.array_equal(gw[gw > 0], gs[gs > 0])
        # halo lists
        assert pw.halo_slot_count() == ps.halo_slot_count()
        for slot in range(pw.halo_slot_count()):
            hw, hs = pw.halo(slot, world), ps.halo(slot, world)
            assert sorted(hw) == sorted(hs)
            for peer in hw:
                assert np.array_equal(hw[peer][0], gid[hs[peer][0]]) and np.array_equal(hw[peer][1], gid[hs[peer][1]])
        assert np.array_equal(pw.pair_hashes(), ps.pair_hashes())


@pytest.mark.parametrize("seed", range(6))
def test_window_plan_equals_whole_plan_randomised(seed):
    # cube edge, tile size, world and block shape drawn from the seed (incl. non-cubic block grids and worlds that are not powers of two)
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(18, 44))
    tile = int(rng.choice([27, 64, 125, 216, 512]))
    world, dims = [(2, (0, 0, 0)), (3, (3, 1, 1)), (4, (1, 4, 1)), (6, (3, 2, 1)), (8, (0, 0, 0)), (4, (2, 1, 2))][seed]
    whole = jelly_cube(n)
    ident = np.arange(whole.n)
    for rank in range(world):
        # (block partition on both sides: a sharded plan is always blocks, while SB_PARTITION_AUTO on the WHOLE mesh would switch to RCB
        # where the cells do not divide evenly among the blocks -- 7 cells over 2 ranks is 4 : 3)
        pw = build_plan(whole, rank=rank, world=world, part_dims=dims, tile_particles=tile, partition=native.SB_PARTITION_BLOCKS)
        win = jelly_cube_window(n, rank, world, dims, tile)
        gid = win.global_id.astype(np.int64)
        ps = native.Plan.build(win.rest_pos, win.dist_ij, rank=rank, world=world, part_dims=dims, tile_particles=tile,
                               domain=win.domain, global_id=win.global_id)
        lw, ow = pw.local_particles(); ls, os_ = ps.local_particles()
        assert ow == os_ and np.array_equal(lw, gid[ls]), (n, tile, world, dims, rank)
        for parity in (0, 1):
            assert np.array_equal(_seq(pw, whole, ident, parity), _seq(ps, win, gid, parity)), (n, tile, world, dims, rank, parity)
        assert np.array_equal(pw.pair_hashes(), ps.pair_hashes())


def test_pair_hashes_are_symmetric_and_detect_a_differing_neighbour():
    n, tile, world = 32, 64, 8
    P = []
    for rank in range(world):
        win = jelly_cube_window(n, rank, world, (0, 0, 0), tile)
        P.append(native.Plan.build(win.rest_pos, win.dist_ij, rank=rank, world=world, tile_particles=tile, domain=win.domain,
                                   global_id=win.global_id).pair_hashes())
    P = np.stack(P)
    assert np.array_equal(P, P.T) and (P[~np.eye(world, dtype=bool)] != 0).all() and (np.diag(P) == 0).all()
    # a rank that planned with another switch (here: no LDS-bank-aware lane order => other programs for the shared tiles)
    win = jelly_cube_window(n, 3, world, (0, 0, 0), tile)
    odd = native.Plan.build(win.rest_pos, win.dist_ij, rank=3, world=world, tile_particles=tile, domain=win.domain, global_id=win.global_id,
                            plan_flags=native.SB_PLAN_NO_BANK_ORDER).pair_hashes()
    assert (odd[np.arange(world) != 3] != P[3][np.arange(world) != 3]).all()


def test_window_violations_are_rejected():
    n, tile, world = 32, 64, 8
    win = jelly_cube_window(n, 0, world, (0, 0, 0), tile)
    kw = dict(rank=0, world=world, tile_particles=tile, domain=win.domain)
    with pytest.raises(native.SoftbodyError):          # ids not ascending
        native.Plan.build(win.rest_pos, win.dist_ij, global_id=win.global_id[::-1].copy(), **kw)
    with pytest.raises(native.SoftbodyError):          # a particle outside the rank's window
        far = win.rest_pos.copy(); far[5] = (31.0, 31.0, 31.0)
        native.Plan.build(far, win.dist_ij, global_id=win.global_id, **kw)
    with pytest.raises(native.SoftbodyError):          # RCB needs the whole mesh
        native.Plan.build(win.rest_pos, win.dist_ij, global_id=win.global_id, partition=native.SB_PARTITION_RCB, **kw)
    with pytest.raises(native.SoftbodyError):          # domain without ids
        native.Plan.build(win.rest_pos, win.dist_ij, **kw)


@pytest.mark.parametrize("world,dims", [(2, (0, 0, 0)), (8, (0, 0, 0)), (6, (1, 3, 2))])
def test_heterogeneous_window_carries_the_whole_cubes_masses_and_rest_lengths(world, dims):
    # per-particle masses and per-spring rest lengths of jelly_cube(heterogeneous=True), reproduced for a window by seeking in the stream
    n, tile = 24, 64
    whole = jelly_cube(n, heterogeneous=True, pin_top=True)
    key = whole.dist_ij[:, 0].astype(np.int64) * whole.n + whole.dist_ij[:, 1]
    order = np.argsort(key); ks = key[order]
    for rank in range(world):
        win = jelly_cube_window(n, rank, world, dims, tile, heterogeneous=True, pin_top=True)
        g = win.global_id.astype(np.int64)
        assert np.array_equal(win.pos, whole.pos[g]) and np.array_equal(win.inv_mass.view(np.uint32), whole.inv_mass[g].view(np.uint32))
        k = g[win.dist_ij[:, 0]] * whole.n + g[win.dist_ij[:, 1]]
        at = np.searchsorted(ks, k)
        assert np.array_equal(ks[at], k) and np.array_equal(win.dist_rest.view(np.uint32), whole.dist_rest[order[at]].view(np.uint32))
        assert len(np.unique(win.dist_rest)) > 0.9 * len(win.dist_rest)


@pytest.mark.parametrize("het", [False, True])
def test_partitioned_run_on_windows_equals_the_unpartitioned_oracle(oracle_mod, het):
    n, tile, world, S = 32, 64, 8, 6
    whole = jelly_cube(n, pin_top=True, heterogeneous=het)
    ref = make_oracle(oracle_mod, whole, build_plan(whole, tile_particles=tile))
    ranks = [WindowRankSim(oracle_mod, jelly_cube_window(n, r, world, (0, 0, 0), tile, pin_top=True, heterogeneous=het), r, world, (0, 0, 0), tile)
             for r in range(world)]

    def exchange(slot, with_prev):
        staged = []
        for R in ranks:
            if slot >= len(R.halos):
                continue
            for peer, (_, recv) in R.halos[slot].items():
                if len(recv):
                    Q = ranks[peer]
                    send = Q.halos[slot][R.rank][0]
                    assert np.array_equal(Q.gid[send], R.gid[recv]), "send / recv lists of a halo slot differ between the two ranks"
                    staged.append((R, recv, Q.o.x[send].copy(), Q.o.xprev[send].copy() if with_prev else None))
        for R, ids, vals, prev in staged:
            R.o.x[ids] = vals
            if prev is not None:
                R.o.xprev[ids] = prev

    for _ in range(2):
        ref.step(0.02, S)
        run_tick(ranks, ranks[0].o.scalars(0.02, S), S, True, exchange)
    x = np.full_like(ref.x, np.nan); v = np.full_like(ref.v, np.nan)
    for R in ranks:
        x[R.gid[R.owned]] = R.o.x[R.owned]; v[R.gid[R.owned]] = R.o.v[R.owned]
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))


def test_fixed_slice_of_the_window_fuzzer():
    """tests/fuzz/fuzz_windows.py: random lattice boxes (full, with holes, L-shaped), worlds, rank grids, tile sizes -- wherever the whole-mesh plan
    is lattice-type (no leftover layers, no global colours) every rank's WINDOW plan must reproduce it (owned sets, pair hashes, shape)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_windows.py"), "--seed", "0", "--max", "150", "--seconds", "200"],
                       capture_output=True, text=True, timeout=400)
    lines = r.stdout.splitlines()
    summary = [l for l in lines if l.startswith("SUMMARY")]
    bad = [l for l in lines if l.split(" ", 1)[0] in ("MISMATCH", "ERROR", "CRASH")]
    assert summary and not bad and r.returncode == 0, "\n".join(bad[:5] + summary + [r.stderr[-800:]])
    assert "150 scenarios" in summary[0] and sum(l.startswith("OK") for l in lines) >= 120, summary[0]
