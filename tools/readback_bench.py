#!/usr/bin/env python3
"""PCIe-inclusive tick rates at 256^3: no readback, blocking sb_get_positions every tick, asynchronous
double-buffered readback every tick, the same with GPU vertex normals of the cube's surface triangles."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody, jelly_cube  # noqa: E402


def surface_triangles(n):
    """Two triangles per surface quad of the n^3 lattice, wound so that the normals point outwards."""
    idx = lambda x, y, z: (z * n + y) * n + x
    tris = []
    r = np.arange(n - 1)
    for fixed in (0, n - 1):
        for a in r:
            for b in r:
                for f in (lambda u, v: idx(u, v, fixed), lambda u, v: idx(u, fixed, v), lambda u, v: idx(fixed, u, v)):
                    tris += [(f(a, b), f(a + 1, b), f(a + 1, b + 1)), (f(a, b), f(a + 1, b + 1), f(a, b + 1))]
    tri = np.array(tris, np.int64)
    xyz = np.stack([tri % n, (tri // n) % n, tri // (n * n)], axis=-1).astype(np.float64)      # (m,3 corners,3)
    nrm = np.cross(xyz[:, 1] - xyz[:, 0], xyz[:, 2] - xyz[:, 0])
    inward = np.einsum("ij,ij->i", nrm, xyz.mean(1) - (n - 1) / 2.0) < 0
    tri[inward] = tri[inward][:, [0, 2, 1]]
    return tri.astype(np.int32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ticks = 30
    mesh = jelly_cube(n)
    sb = Softbody(mesh, substeps=20).Start()
    out = {"n": n, "ticks": ticks}
    buf = np.zeros((mesh.n, 3), np.float32)

    def timed(name, body, drain):
        for _ in range(3):
            body()
        drain(); sb.synchronize()
        t0 = time.perf_counter()
        for _ in range(ticks):
            body()
        drain(); sb.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t0) / ticks

    timed("no_readback_ms", lambda: sb.step(), lambda: None)
    timed("blocking_get_positions_ms", lambda: (sb.step(), sb.get_positions(buf)), lambda: None)
    state = {"pending": 0}

    def async_tick():
        sb.step(); sb.readback_begin(); state["pending"] += 1
        if state["pending"] == 2:
            sb.readback_end(); state["pending"] -= 1

    def drain():
        while state["pending"]:
            sb.readback_end(); state["pending"] -= 1

    timed("async_readback_ms", async_tick, drain)
    tri = surface_triangles(n)
    sb.set_render_triangles(tri)
    out["render_triangles"] = int(len(tri))
    timed("async_readback_with_normals_ms", async_tick, drain)
    sb.set_readback_render_set_only(True)
    timed("async_render_set_only_with_normals_ms", async_tick, drain)
    out["render_set_particles"] = int(len(np.unique(tri)))
    st = sb.stats()
    out["peek"] = {"enabled": not os.environ.get("SB_NO_PEEK"), "readback_peeks": st["readback_peeks"], "peek_tiles": st["readback_peek_tiles"],
                   "t0_tiles": st["n_tiles"][0]}
    sb.OnDestroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
