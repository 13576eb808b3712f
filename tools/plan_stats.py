#!/usr/bin/env python3
"""Planner statistics of a mesh, host only (no GPU): per phase the tiles, rounds per tile, constraints per round by type,
and the particle degrees that bound the rounds. usage: python tools/plan_stats.py [bunny verts | cube N] [tile]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import native  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    tile = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    mesh = bunny_surrogate(target_verts=size) if kind == "bunny" else jelly_cube(size)
    t0 = time.time()
    plan = native.Plan.build(mesh.rest_pos, mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl, tile_particles=tile)
    print(f"{mesh.label}: {mesh.n} particles, {len(mesh.dist_rest)} springs, {len(mesh.vol_rest)} tets, {len(mesh.bend_rest)} hinges; "
          f"plan {time.time() - t0:.1f} s")
    idx = [mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl]
    for t, name in enumerate(("distance", "volume", "bending")):
        if len(idx[t]):
            deg = np.bincount(np.asarray(idx[t]).ravel(), minlength=mesh.n)
            print(f"  degree {name}: mean {deg.mean():.1f} max {deg.max()}")
    types, ids = plan.order(0)
    tasks, groups = plan.tasks(0), plan.groups(0)
    total_rounds_crit = 0
    for ph in plan.phases(0):
        ob, oe = ph["order_begin"], ph["order_end"]
        tk = tasks[ph["task_begin"]:ph["task_end"] + 1]
        g = groups[(groups >= ob) & (groups <= oe)]
        sizes = np.diff(g)
        gt = types[g[:-1]]
        rounds_per_task = np.array([np.count_nonzero((g[:-1] >= a) & (g[:-1] < b)) for a, b in zip(tk[:-1], tk[1:])])
        line = (f"  phase kind {ph['kind']} tiling {ph['tiling']}: {oe - ob} constraints, {len(tk) - 1} tiles, rounds/tile mean "
                f"{rounds_per_task.mean():.1f} max {rounds_per_task.max()}; ")
        for t, name in enumerate(("dist", "vol", "bend")):
            m = gt == t
            if m.any():
                line += f"{name}: {m.sum()} rounds of mean {sizes[m].mean():.0f}; "
        # degree inside this phase's list
        degs = []
        for t in range(3):
            sel = ids[ob:oe][types[ob:oe] == t]
            if len(sel):
                degs.append(np.bincount(np.asarray(idx[t])[sel].ravel(), minlength=mesh.n))
        if degs:
            d = np.sum(degs, axis=0)
            line += f"combined degree in list: mean {d[d > 0].mean():.1f} max {d.max()}"
        print(line)
        total_rounds_crit += rounds_per_task.max() * (2 if ph["kind"] in (1, 2) else 1)
    print(f"  critical-path rounds per substep (max over tiles, both passes of the two fused lists): {total_rounds_crit // 1}")


if __name__ == "__main__":
    main()
