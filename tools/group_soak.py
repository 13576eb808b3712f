#!/usr/bin/env python3
"""Long run of a group (sb_group_*: one process, `world` ranks on the one device) beside a single solver of the same mesh, same script:
kinematic pins moved before every tick, a render-set readback with normals after every tick, a blocking read now and then. The group must stay
bit-identical to the single solver (which the parity tests pin to the oracle) tick after tick -- the exchanges, the peeks, the fused kinematic
kernels and the gathered snapshots of thousands of ticks. usage: python tools/group_soak.py [n=64] [world=8] [ticks=1000] [threads|walk] [peer|rccl-loopback]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np                                                       # noqa: E402
from readback_bench import surface_triangles                             # noqa: E402
from softbodyunity_amd import Softbody, SoftbodyGroup, jelly_cube, native   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
host = sys.argv[4] if len(sys.argv) > 4 else "threads"
dims = {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world, (world, 1, 1))
mesh = jelly_cube(n)
pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)
mesh.inv_mass[pins] = 0.0
rest = mesh.pos[pins].copy()
tri = surface_triangles(n)
tune = native.SbTuning(); native.lib().sb_tuning_default(C.byref(tune)); tune.peek_min_tiles = 0
S = 10
one = Softbody(mesh, substeps=S, damping=0.02, tuning=tune).Start()
grp = SoftbodyGroup(mesh, [0] * world, substeps=S, damping=0.02, partition=native.SB_PARTITION_BLOCKS, part_dims=dims,
                    halo_transport=native.SB_TRANSPORT_PEER, walk=host == "walk", tuning=tune).Start()
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
bad = []
t0 = time.perf_counter()
try:
    for sb in (one, grp):
        sb.set_render_triangles(tri); sb.set_readback_render_set_only(True)
    for t in range(ticks):
        target = rest + np.array([0.3 * np.sin(0.05 * t), 0.1 * np.cos(0.07 * t) - 0.1, 0.2 * np.sin(0.03 * t)], np.float32)
        one.set_kinematic_positions(pins, target); grp.set_kinematic_positions(pins, target)
        one.step(); grp.step()
        one.readback_begin(); grp.readback_begin()
        pa, na = one.readback_end(normals=True); pb, nb = grp.readback_end(normals=True)
        if not (np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(na), bits(nb))):
            bad.append(("snapshot", t))
        if t % 97 == 0 and not np.array_equal(bits(one.get_positions()), bits(grp.get_positions())):
            bad.append(("read", t))
        if len(bad) > 5:
            break
    xa, va, xb, vb = one.get_positions(), one.get_velocities(), grp.get_positions(), grp.get_velocities()
    final = bool(np.array_equal(bits(xa), bits(xb)) and np.array_equal(bits(va), bits(vb)) and np.isfinite(xa).all())
    st = [grp.rank(r).stats() for r in range(world)]
    print(json.dumps({"n": n, "world": world, "ticks": ticks, "host_model": host, "substeps": S, "bit_identical_every_tick": not bad, "first_mismatches": bad[:5],
                      "final_state_identical": final, "seconds": time.perf_counter() - t0,
                      "ranks": {"readback_peeks": [s["readback_peeks"] for s in st], "ticks_fused": [s["ticks_fused"] for s in st],
                                "ticks_fused_kinematic": [s["ticks_fused_kinematic"] for s in st]}}), flush=True)
finally:
    one.OnDestroy(); grp.OnDestroy()
sys.exit(0 if (not bad and final) else 1)
