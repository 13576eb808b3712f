#!/usr/bin/env python3
"""ms per tick of the tet surrogate at BUNNY_VERTS vertices (BUNNY_CACHE keeps the mesh between runs), whatever SB_* switches the
environment holds. usage: python tools/bunny_time.py [label]"""
import os
import pickle
import sys
import time
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
sb = Softbody(mesh, substeps=20, tile_particles=int(os.environ.get("TILE", "0")), distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
for _ in range(5):
    sb.step()
sb.synchronize()
ticks = int(os.environ.get("TICKS", "40"))
t0 = time.perf_counter()
for _ in range(ticks):
    sb.step()
sb.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / ticks
st = sb.stats()
import hashlib
h = hashlib.sha256(sb.get_positions().tobytes()).hexdigest()[:12]
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default'}: {mesh.n} vertices {ms:.4f} ms/tick  tiles {st['n_tiles']} t2 layers {st['n_t2_layers']} t2 tiles {st['n_t2_tiles']} state {h}", flush=True)
sb.OnDestroy()
