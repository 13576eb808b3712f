#!/usr/bin/env python3
"""Soak runs: many ticks of the benchmark meshes, finite positions and a stable tick time at the end.
usage: python tools/soak.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube  # noqa: E402

for name, mesh, ticks, kw in (("256^3 cube", jelly_cube(256), 2000, {}), ("64^3 cube", jelly_cube(64), 20000, {}),
                              ("100k tet surrogate", bunny_surrogate(target_verts=100_000), 2000,
                               dict(distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5, ground_plane=(0, 1, 0, -1.5)))):
    sb = Softbody(mesh, substeps=20, **kw).Start()
    for _ in range(5):
        sb.step()
    sb.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        sb.step()
    sb.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / ticks
    x = sb.get_positions()
    print(f"{name}: {ticks} ticks ({ticks * 20} substeps) {ms:.4f} ms per tick, finite {bool(np.isfinite(x).all())}, "
          f"bbox {x.min(0).round(2).tolist()} .. {x.max(0).round(2).tolist()}", flush=True)
    sb.OnDestroy()
