#!/usr/bin/env python3
"""profiles/<tag>_bunny100k_summary.md: kernel-trace table of the tet surrogate (100 k vertices; BUNNY_VERTS / BUNNY_CACHE as in
tools/bunny_run.py) + planner statistics (groups per tile, lanes busy per group). usage: python tools/bunny_summary.py <rocprof dir> <tag>"""
import collections
import csv
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import native  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402

d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in os.listdir(d):
    if f.endswith("kernel_trace.csv"):
        for r in csv.DictReader(open(os.path.join(d, f))):
            wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            acc[(r["Kernel_Name"][:58], wg, int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
import pickle  # noqa: E402
verts = int(os.environ.get("BUNNY_VERTS", "100000"))       # (same switches as tools/bunny_run.py)
cache = os.environ.get("BUNNY_CACHE")
mesh = pickle.load(open(cache, "rb")) if cache and os.path.exists(cache) else bunny_surrogate(target_verts=verts)
print(f"# tet surrogate ({mesh.n} vertices, {len(mesh.dist_rest)} springs, {len(mesh.vol_rest)} tets, {len(mesh.bend_rest)} hinges"
      f"{'; config 5' if verts == 100000 else ''}), 20 substeps per tick ({tag})\n")
print("## rocprofv3 --kernel-trace: per launch shape\n")
print("| kernel | workgroups | lanes | launches | avg µs | total ms |\n|---|---|---|---|---|---|")
tot = 0
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    v = acc[k]
    tot += sum(v)
    print(f"| `{k[0]}` | {k[1]} | {k[2]} | {len(v)} | {np.mean(v) / 1e3:.1f} | {sum(v) / 1e6:.2f} |")
print(f"\nsum of kernel time over the 25 ticks of the run: {tot / 1e6:.1f} ms = {tot / 25e6:.2f} ms per tick\n")
plan = native.Plan.build(mesh.rest_pos, mesh.dist_ij, mesh.vol_ijkl, mesh.bend_ijkl)      # tile_particles = 0: automatic (256)
types, ids = plan.order(0)
tasks, groups = plan.tasks(0), plan.groups(0)
print("## planner: groups per tile and lanes busy (parity 0; a group = constraints projected concurrently, one barrier)\n")
print("| phase | tiles | constraints | groups per tile mean / max | springs + 4·(tets + hinges) lanes per group, mean | of 256 |\n|---|---|---|---|---|---|")
names = {1: "first list (fused tile kernel)", 2: "second list (fused tile kernel)", 3: "T2 layer (sparse tiles: a balanced list or a cluster layer)", 0: "global colour"}
steps_per_substep = 0
for ph in plan.phases(0):
    ob, oe = ph["order_begin"], ph["order_end"]
    tk = tasks[ph["task_begin"]:ph["task_end"] + 1]
    g = groups[(groups >= ob) & (groups <= oe)]
    per_tile = np.array([np.count_nonzero((g[:-1] >= a) & (g[:-1] < b)) for a, b in zip(tk[:-1], tk[1:])])
    lanes = []
    for a, b in zip(g[:-1], g[1:]):
        t = types[a:b]
        lanes.append(np.count_nonzero(t == 0) + 4 * np.count_nonzero(t != 0))
    print(f"| {names[ph['kind']]} | {len(tk) - 1} | {oe - ob} | {per_tile.mean():.1f} / {per_tile.max()} | {np.mean(lanes):.0f} | {np.mean(lanes) / 256:.0%} |")
    steps_per_substep += int(per_tile.max())
print(f"\ndependent steps per substep (= groups of the longest tile, summed over the lists a substep walks; one workgroup barrier per group): "
      f"**{steps_per_substep}** (round 2: 157 = 41 + 43 + 32 + 41)")
