#!/usr/bin/env python3
"""Config 5 cost attribution: ms per tick of the 100k surrogate with all constraint types, without bending, and with
springs only; per-slot launch timing. usage: python tools/bunny_attrib.py [tile]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 512
full = bunny_surrogate(target_verts=100_000)
for name, drop in (("all", ()), ("no bending", ("bend",)), ("springs only", ("bend", "vol"))):
    import copy
    m = copy.copy(full)
    if "bend" in drop:
        m.bend_ijkl = np.zeros((0, 4), np.int32); m.bend_rest = np.zeros((0, 2), np.float32)
    if "vol" in drop:
        m.vol_ijkl = np.zeros((0, 4), np.int32); m.vol_rest = np.zeros(0, np.float32)
    sb = Softbody(m, substeps=20, tile_particles=tile, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
    for _ in range(5):
        sb.step()
    sb.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        sb.step()
    sb.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 50
    sl, cnt = sb.step_profiled()
    st = sb.stats()
    G = st["n_global_colours"]
    print(f"tile {tile} {name:13s}: {ms:.3f} ms/tick | T0 mid {sl[0]:.3f}/{cnt[0]} T1 mid {sl[1]:.3f}/{cnt[1]} T2 {sl[4 + G]:.3f}/{cnt[4 + G]} "
          f"| tiles {st['n_tiles']} t2 layers {st['n_t2_layers']} t2 tiles {st['n_t2_tiles']} globals {G}", flush=True)
    sb.OnDestroy()
