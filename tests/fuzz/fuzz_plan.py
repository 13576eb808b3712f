#!/usr/bin/env python3
"""Randomised test of the host-only planner (sb_plan_build; no GPU needed) on meshes nobody would author on purpose: all particles on a point, a
line or a plane, isolated particles, no constraints at all, duplicate constraints, one or two particles, a complete graph, a 1-D chain, a hub,
extreme coordinates, NaN / infinite positions, more ranks than particles, every tile size and partition. The planner must either refuse with a
message (SoftbodyError) or return a plan that passes the invariants the GPU execution relies on (tests/test_plan.py _check_plan: every parity's
order a permutation, groups are matchings, tasks of a phase touch disjoint particles) and whose ranks' halo lists fit together -- never crash,
never hang. usage: python tests/fuzz/fuzz_plan.py [--seconds 120] [--seed 0] [--only SEED] [--max N]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity as fz                                                # noqa: E402  (parent / child harness)

import numpy as np                                                      # noqa: E402
from softbodyunity_amd import native                                    # noqa: E402
from softbodyunity_amd.mesh import SoftbodyMesh                         # noqa: E402
from test_plan import _check_plan                                       # noqa: E402

SHAPES = ["cloud", "point", "line", "plane", "chain", "complete", "hub", "isolated", "no-constraints", "one", "two", "duplicates", "huge", "tiny",
          "nan", "inf", "grid-holes"]


def make_scenario(seed):
    rng = np.random.default_rng(seed)
    shape = str(rng.choice(SHAPES))
    n = int(rng.integers(3, 400))
    pos = rng.uniform(0, max(1.0, n ** (1 / 3)), (n, 3))
    ij = None
    if shape == "point":
        pos[:] = pos[0]
    elif shape == "line":
        pos[:, 1:] = 0.5
    elif shape == "plane":
        pos[:, 1] = 0.25
    elif shape == "chain":
        pos = np.stack([np.arange(n) * 1.0, np.zeros(n), np.zeros(n)], 1)
        ij = np.stack([np.arange(n - 1), np.arange(1, n)], 1)
    elif shape == "complete":
        n = min(n, int(rng.integers(3, 48))); pos = pos[:n]
        ij = np.array([(a, b) for a in range(n) for b in range(a + 1, n)])
    elif shape == "hub":
        ij = np.stack([np.zeros(n - 1, int), np.arange(1, n)], 1)
    elif shape == "one":
        n = 1; pos = pos[:1]; ij = np.zeros((0, 2), int)
    elif shape == "two":
        n = 2; pos = pos[:2]; ij = np.array([[0, 1]])
    elif shape == "no-constraints":
        ij = np.zeros((0, 2), int)
    elif shape == "huge":
        pos *= float(rng.choice([1e6, 1e15, 1e29]))
    elif shape == "tiny":
        pos *= float(rng.choice([1e-6, 1e-20, 1e-37]))
    elif shape == "grid-holes":
        g = int(rng.integers(3, 9)); P = np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(float)
        pos = P[rng.random(len(P)) < 0.6]; n = len(pos)
        if n < 2:
            pos = P[:2]; n = 2
    if ij is None:       # nearest neighbours (ties on degenerate clouds are fine: any pairs will do)
        from scipy.spatial import cKDTree
        k = min(int(rng.integers(2, 7)), n)
        _, nb = cKDTree(pos + rng.normal(0, 1e-9, pos.shape) * (shape not in ("huge", "tiny"))).query(pos, k=k)
        nb = np.asarray(nb).reshape(n, -1)
        ij = np.array(sorted({(min(i, int(j)), max(i, int(j))) for i in range(n) for j in nb[i, 1:] if int(j) != i})).reshape(-1, 2)
    if shape == "isolated" and len(ij) > 4:
        keep = rng.random(len(ij)) < 0.3
        ij = ij[keep]
    if shape == "duplicates" and len(ij):
        ij = np.concatenate([ij, ij[rng.integers(0, len(ij), len(ij) // 2)], ij[:1][:, ::-1]])
    quads = np.zeros((0, 4), int)
    if n >= 8 and rng.random() < 0.5 and shape not in ("one", "two"):
        q = np.array([rng.choice(n, 4, replace=False) for _ in range(int(rng.integers(1, max(2, n // 4))))])
        quads = q
    vol, bend = quads[0::2], quads[1::2]
    pos32 = pos.astype(np.float32)
    if shape == "nan":
        pos32[int(rng.integers(0, n)), int(rng.integers(0, 3))] = np.nan
    if shape == "inf":
        pos32[int(rng.integers(0, n)), int(rng.integers(0, 3))] = np.inf
    with np.errstate(all="ignore"):
        rest = np.linalg.norm(pos32[ij[:, 0]] - pos32[ij[:, 1]], axis=1).astype(np.float32) if len(ij) else np.zeros(0, np.float32)
    mesh = SoftbodyMesh(rest_pos=pos32.copy(), pos=pos32.copy(), vel=np.zeros_like(pos32), inv_mass=np.ones(n, np.float32),
                        dist_ij=ij.astype(np.int32).reshape(-1, 2), dist_rest=rest, vol_ijkl=vol.astype(np.int32).reshape(-1, 4),
                        vol_rest=np.ones(len(vol), np.float32), bend_ijkl=bend.astype(np.int32).reshape(-1, 4),
                        bend_rest=np.tile(np.array([[1.0, 0.0]], np.float32), (len(bend), 1)))
    sc = {"seed": seed, "shape": shape, "n": n, "springs": len(ij), "tets": len(vol), "hinges": len(bend), "_mesh": mesh,
          "tile": int(rng.choice([0, -1, 1, 2, 7, 32, 64, 256, 512])), "world": int(rng.choice([1, 1, 2, 3, 5, 8, 17])),
          "partition": int(rng.choice([0, 1, 2]))}
    return sc


def run(sc):
    mesh, W = sc["_mesh"], sc["world"]
    plans = []
    try:
        for r in range(W):
            plans.append(native.Plan.build(mesh.rest_pos, mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl, rank=r, world=W, tile_particles=sc["tile"],
                                           partition=sc["partition"]))
    except native.SoftbodyError as e:
        return "REFUSED", str(e)[:160]
    try:
        if len(mesh.dist_rest) + len(mesh.vol_rest) + len(mesh.bend_rest):
            _check_plan(mesh, plans[0])
        else:       # nothing to order: both parities empty
            assert all(len(plans[0].order(par)[1]) == 0 for par in (0, 1))
        own = np.zeros(mesh.n, np.int64)
        for r, p in enumerate(plans):
            o = p.owner(mesh.n)
            assert np.array_equal(o, plans[0].owner(mesh.n)), "ranks disagree on the ownership"
            own += (o == r)
            loc, n_owned = p.local_particles()
            assert n_owned == int((o == r).sum()) and len(np.unique(loc)) == len(loc)
        assert np.all(own == 1), "the ranks' owned sets do not partition the particles"
        for slot in range(plans[0].halo_slot_count()):
            H = [p.halo(slot, W) for p in plans]
            for a in range(W):
                for b, (send, recv) in H[a].items():
                    assert np.array_equal(recv, H[b][a][0]), f"halo slot {slot}: what {a} receives from {b} is not what {b} sends"
                    assert np.all(plans[0].owner(mesh.n)[recv] == b) if len(recv) else True
    except AssertionError as e:
        return "MISMATCH", f"invariant: {e}"
    return "OK", ""


if __name__ == "__main__":
    sys.exit(fz.main(make_scenario, run, __file__))
