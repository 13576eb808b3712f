#!/usr/bin/env python3
"""Interleaved A/B of one environment switch of the plugin (read at sb_create / sb_finalize) on config 5 (100 k surrogate) and
the 64^3 cube: ms per tick without and with VAR=1 (or VAR=VALUE). usage: python tools/env_ab.py SB_NO_COST_ORDER|SB_QUAD_LANES=512 [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube  # noqa: E402


def run(mesh, ticks, **kw):
    sb = Softbody(mesh, substeps=20, **kw).Start()
    for _ in range(5):
        sb.step()
    sb.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        sb.step()
    sb.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / ticks
    sb.OnDestroy()
    return round(ms, 4)


def main():
    var, _, val = sys.argv[1].partition("=")
    val = val or "1"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    bunny = bunny_surrogate(target_verts=100_000)
    cube = jelly_cube(64)
    res = {"bunny100k": {"unset": [], "set": []}, "cube64": {"unset": [], "set": []}}
    for _ in range(rounds):
        for state in ("unset", "set"):
            os.environ.pop(var, None)
            if state == "set":
                os.environ[var] = val
            res["bunny100k"][state].append(run(bunny, 100, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5))
            res["cube64"][state].append(run(cube, 400))
    print(json.dumps({"switch": f"{var}={val}", **res}))


if __name__ == "__main__":
    main()
