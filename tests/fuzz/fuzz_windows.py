#!/usr/bin/env python3
"""Randomised test of SHARDED AUTHORING on lattice-like bodies with the host-only planner (no GPU): random boxes nx x ny x nz of structural
springs (jittered, optionally with holes punched out or an L-shape: then NOT lattice-like), random world / part_dims / tile size; every rank
plans its WINDOW (cut as sb_group_finalize cuts it: rest position inside sb_domain_window's box, constraints among them in the whole mesh's
order) and the result is compared with ranks planning the WHOLE mesh: owned sets partition the body, pair hashes symmetric and equal to the
whole-mesh plans', one tick-program shape (halo slot count). Where the whole-mesh plan is LATTICE-TYPE (two tilings, no leftover layers, no
global colours) that must always hold -- the supported case; elsewhere a disagreement is legitimate -- it is what the agreement check exists to
find -- and is only counted.
usage: python tests/fuzz/fuzz_windows.py [--seconds 120] [--seed 0] [--only SEED] [--max N]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity as fz                                                # noqa: E402  (parent / child harness)

import numpy as np                                                      # noqa: E402
from softbodyunity_amd import native                                    # noqa: E402


def lattice_box(nx, ny, nz, rng, jitter):
    idx = np.arange(nx * ny * nz, dtype=np.int64).reshape(nz, ny, nx)
    gz, gy, gx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    rest = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1).astype(np.float32)
    rest = (rest + rng.uniform(-jitter, jitter, rest.shape)).astype(np.float32)
    ij = []
    for lo, hi in ((idx[:, :, :-1], idx[:, :, 1:]), (idx[:, :-1, :], idx[:, 1:, :]), (idx[:-1, :, :], idx[1:, :, :])):
        ij.append(np.stack([lo.ravel(), hi.ravel()], 1))
    return rest, np.concatenate(ij).astype(np.int32)


def make_scenario(seed):
    rng = np.random.default_rng(seed)
    nx, ny, nz = (int(rng.integers(2, 40)) for _ in range(3))
    rest, ij = lattice_box(nx, ny, nz, rng, float(rng.choice([0.0, 0.02, 0.1])))
    shape = str(rng.choice(["full", "full", "full", "holes", "L"]))
    keep = np.ones(len(rest), bool)
    if shape == "holes":
        keep = rng.random(len(rest)) > 0.15
    elif shape == "L":
        keep = ~((rest[:, 0] > 0.5 * nx) & (rest[:, 1] > 0.5 * ny))
    if keep.sum() < 2:
        keep[:] = True
    new = -np.ones(len(rest), np.int64); new[keep] = np.arange(int(keep.sum()))
    rest = rest[keep]
    ij = new[ij[np.all(new[ij] >= 0, axis=1)]].astype(np.int32)
    world = int(rng.choice([2, 3, 4, 6, 8, 12]))
    dims = (0, 0, 0)
    if rng.random() < 0.4:       # explicit rank grids
        opts = [d for d in [(world, 1, 1), (1, world, 1), (1, 1, world), (2, world // 2, 1), (1, 2, world // 2), (2, 2, world // 4)] if d[0] * d[1] * d[2] == world and min(d) >= 1]
        dims = opts[int(rng.integers(0, len(opts)))]
    return {"seed": seed, "box": f"{nx}x{ny}x{nz}", "shape": shape, "n": len(rest), "springs": len(ij), "world": world, "dims": dims,
            "tile": int(rng.choice([16, 27, 64, 100, 128, 256, 512])), "_rest": rest, "_ij": ij}


def run(sc):
    rest, ij, W, tile, dims = sc["_rest"], sc["_ij"], sc["world"], sc["tile"], sc["dims"]
    try:
        dom = native.domain_from_mesh(rest, ij)
        whole = [native.Plan.build(rest, ij, rank=r, world=W, part_dims=dims, tile_particles=tile, partition=native.SB_PARTITION_BLOCKS) for r in range(W)]
        wins, gids = [], []
        for r in range(W):
            lo, hi = native.domain_window(dom, r, W, dims, tile)
            inside = np.all((rest >= np.array(lo)) & (rest < np.array(hi)), axis=1)
            gid = np.nonzero(inside)[0].astype(np.int32)
            if len(gid) == 0:
                return "REFUSED", "a rank's window is empty (more ranks than occupied cells)"
            new = -np.ones(len(rest), np.int64); new[gid] = np.arange(len(gid))
            wij = new[ij[np.all(new[ij] >= 0, axis=1)]].astype(np.int32)
            wins.append(native.Plan.build(rest[gid], wij, rank=r, world=W, part_dims=dims, tile_particles=tile, partition=native.SB_PARTITION_BLOCKS, domain=dom, global_id=gid))
            gids.append(gid)
    except native.SoftbodyError as e:
        return "REFUSED", str(e)[:200]
    why = []
    own = np.zeros(len(rest), np.int32)
    for r in range(W):
        o = wins[r].owner(len(gids[r]))
        own[gids[r][o == r]] += 1
        if not np.array_equal(gids[r][o == r], np.nonzero(whole[0].owner(len(rest)) == r)[0]):
            why.append(f"rank {r}: the window's owned set is not the whole-mesh plan's")
    if not np.all(own == 1):
        why.append("the windows' owned sets do not partition the body")
    slots = {p.halo_slot_count() for p in wins}
    if len(slots) != 1 or slots != {whole[0].halo_slot_count()}:
        why.append(f"tick programs of different shape: halo slots {sorted(p.halo_slot_count() for p in wins)} against the whole plan's {whole[0].halo_slot_count()}")
    for a in range(W):
        pa, wa = list(wins[a].pair_hashes()), list(whole[a].pair_hashes())
        for b in range(W):
            if a != b and pa[b] != list(wins[b].pair_hashes())[a]:
                why.append(f"pair hashes of ranks {a} and {b} differ"); break
        if pa != wa:
            why.append(f"rank {a}: the window's pair hashes are not the whole-mesh plan's")
    if not why:
        return "OK", ""
    # Windows must reproduce the whole-mesh plan where that plan is LATTICE-TYPE: two tilings and nothing else (2 halo slots: no leftover
    # layers, no global colours). A body with holes, or a full box at a tile size that leaves leftovers (100: cells of 4 or 5 springs do not
    # divide the shift evenly), has leftover layers whose cluster tiles are not local to a window: a disagreement there is what the agreement
    # check exists to find (a per-process host gets the error, the group host falls back to the whole mesh) -- counted as REFUSED here.
    return ("MISMATCH" if whole[0].halo_slot_count() == 2 else "REFUSED"), "; ".join(why[:3])


if __name__ == "__main__":
    sys.exit(fz.main(make_scenario, run, __file__))
