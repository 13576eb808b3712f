#!/usr/bin/env python3
"""Config 5 (100 k surrogate): ms per tick against tile_particles (two interleaved rounds) + tiles / T2 layers / T2 tiles.
usage: python tools/bunny_tile_sweep.py"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody
from softbodyunity_amd.mesh import bunny_surrogate
mesh = bunny_surrogate(target_verts=100_000)
res = {}
for rnd in range(2):
    for tile in (128, 160, 192, 224, 256, 320, 384, 512):
        sb = Softbody(mesh, substeps=20, tile_particles=tile, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
        for _ in range(5): sb.step()
        sb.synchronize(); t0 = time.perf_counter()
        for _ in range(60): sb.step()
        sb.synchronize()
        st = sb.stats()
        res.setdefault(tile, []).append(round(1e3*(time.perf_counter()-t0)/60, 4))
        res[str(tile)+"_info"] = [st["n_tiles"], st["n_t2_layers"], st["n_t2_tiles"]]
        sb.OnDestroy()
print(json.dumps(res))
