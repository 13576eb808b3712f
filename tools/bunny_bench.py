#!/usr/bin/env python3
"""Config 5 timing (surrogate tet mesh: springs + tet volumes + surface hinges): ms per tick and workgroup counts,
with and without tile packing and the third tiling. usage: python tools/bunny_bench.py [target_verts]"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    mesh = bunny_surrogate(target_verts=n)
    out = {"verts": mesh.n, "springs": len(mesh.dist_rest), "tets": len(mesh.vol_rest), "hinges": len(mesh.bend_rest)}
    tile = int(os.environ.get("TILE", "0"))
    out["tile_particles"] = tile
    for name, env in (("default", {}), ("unpacked", {"SB_NO_PACK": "1"}), ("no_third_tiling", {"SB_NO_T2": "1"})):
        for k in ("SB_NO_PACK", "SB_NO_T2"):
            os.environ.pop(k, None)
        os.environ.update(env)
        sb = Softbody(mesh, substeps=20, tile_particles=tile, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
        for _ in range(5):
            sb.step()
        sb.synchronize()
        t0 = time.perf_counter()
        ticks = 100
        for _ in range(ticks):
            sb.step()
        sb.synchronize()
        st = sb.stats()
        out[name] = {"ms_per_tick": 1e3 * (time.perf_counter() - t0) / ticks, "workgroups": st["n_tiles"],
                     "global_colours": st["n_global_colours"], "constraints_in_global": st["constraints_in_global"],
                     "t2_layers": st["n_t2_layers"], "t2_workgroups": st["n_t2_tiles"], "t2_constraints": st["t2_constraints"]}
        sb.OnDestroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
