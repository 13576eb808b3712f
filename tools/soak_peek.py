#!/usr/bin/env python3
"""Soak of the peeked reads: many ticks with a render-set readback (+ GPU normals) after EVERY tick, once peeking and once with
SB_NO_PEEK=1 (one process each: the switch is read at sb_create); the two runs must hand out identical snapshots (hash of all of them)
and end in the identical state, and the table validator must still find nothing. usage: python tools/soak_peek.py cube256|cube64|bunny [ticks]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("SB_PEEK_MIN_TILES", "0")     # peek on every mesh here, also where the plugin's default would not (small launches)
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cube64"
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    kw = {}
    if which.startswith("cube"):
        from readback_bench import surface_triangles
        n = int(which[4:])
        mesh, tri = jelly_cube(n), surface_triangles(n)
        kw = dict(ground_plane=(0, 1, 0, -40.0))          # the cube lands on it during the run: the collision path of the peek
    else:
        mesh = bunny_surrogate(target_verts=100_000)
        t = mesh.vol_ijkl.reshape(-1, 4)
        faces = np.concatenate([t[:, [0, 1, 2]], t[:, [0, 1, 3]], t[:, [0, 2, 3]], t[:, [1, 2, 3]]])
        _, idx, cnt = np.unique(np.sort(faces, axis=1), axis=0, return_index=True, return_counts=True)
        tri = faces[idx[cnt == 1]].astype(np.int32)
        kw = dict(distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5, ground_plane=(0, 1, 0, -1.5))
    sb = Softbody(mesh, substeps=20, **kw).Start()
    sb.set_render_triangles(tri); sb.set_readback_render_set_only(True)
    h = hashlib.sha256()
    sb.step(); sb.readback_begin()
    sb.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks - 1):
        sb.step(); sb.readback_begin()
        pos, nrm = sb.readback_end(normals=True)
        h.update(pos.tobytes()); h.update(nrm.tobytes())
    pos, nrm = sb.readback_end(normals=True)
    h.update(pos.tobytes()); h.update(nrm.tobytes())
    ms = 1e3 * (time.perf_counter() - t0) / (ticks - 1)
    x, v = sb.get_positions(), sb.get_velocities()
    st, rep = sb.stats(), sb.validate()
    sb.OnDestroy()
    print(json.dumps({"mesh": which, "particles": int(mesh.n), "ticks": ticks, "peek": not os.environ.get("SB_NO_PEEK"), "ms_per_tick_with_readback": round(ms, 4),
                      "snapshots_sha256": h.hexdigest()[:16], "final_state_sha256": hashlib.sha256(x.tobytes() + v.tobytes()).hexdigest()[:16],
                      "finite": bool(np.isfinite(x).all()), "readback_peeks": st["readback_peeks"], "ticks_fused": st["ticks_fused"],
                      "validator_errors": rep["errors"]}), flush=True)


if __name__ == "__main__":
    main()
