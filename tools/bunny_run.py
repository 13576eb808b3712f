#!/usr/bin/env python3
"""The tet surrogate for a profiler run: 5 warm-up ticks + 20 ticks of 20 substeps (default tile size).
BUNNY_VERTS=<vertices> (default 100 000 = config 5); BUNNY_CACHE=<file>: keep the generated mesh there between passes (a 1 M-vertex mesh
takes a minute to generate)."""
import os
import pickle
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402
verts = int(os.environ.get("BUNNY_VERTS", "100000"))
cache = os.environ.get("BUNNY_CACHE")
if cache and os.path.exists(cache):
    mesh = pickle.load(open(cache, "rb"))
else:
    mesh = bunny_surrogate(target_verts=verts)
    if cache:
        pickle.dump(mesh, open(cache, "wb"), protocol=4)
sb = Softbody(mesh, substeps=20, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
for _ in range(25):
    sb.step()
sb.synchronize()
print(sb.stats())
sb.OnDestroy()
