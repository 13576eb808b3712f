#!/usr/bin/env python3
"""Host-only: wall time and peak resident memory of planning the n^3 cube -- the whole mesh on one rank (what every rank of a
partitioned solver did in round 2) against one rank's WINDOW under sharded authoring (sb_domain). No GPU needed.
usage: plan_shard_timing.py [n=256] [world=8] [rank=7]   (each measurement in its own process: ru_maxrss is per process)"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode, n, world, rank):
    from softbodyunity_amd import native
    from softbodyunity_amd.mesh import jelly_cube, jelly_cube_window
    t0 = time.time()
    if mode == "whole1":
        m = jelly_cube(n); kw = dict(rank=0, world=1)
    elif mode == "whole":
        m = jelly_cube(n); kw = dict(rank=rank, world=world)
    else:
        m = jelly_cube_window(n, rank, world); kw = dict(rank=rank, world=world, domain=m.domain, global_id=m.global_id)
    t1 = time.time()
    p = native.Plan.build(m.rest_pos, m.dist_ij, tile_particles=512, **kw)
    t2 = time.time()
    loc, owned = p.local_particles()
    print(json.dumps({"mode": mode, "n_particles_given": int(m.n), "n_constraints_given": int(len(m.dist_rest)), "mesh_s": round(t1 - t0, 2),
                      "plan_s": round(t2 - t1, 2), "peak_rss_GiB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20, 2),
                      "owned": int(owned), "local": int(len(loc))}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
        world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        rank = int(sys.argv[3]) if len(sys.argv) > 3 else world - 1
        for mode in ("whole1", "whole", "window"):
            out = subprocess.run([sys.executable, __file__, "--child", mode, str(n), str(world), str(rank)], capture_output=True, text=True)
            print(out.stdout.strip() or out.stderr[-500:], flush=True)
