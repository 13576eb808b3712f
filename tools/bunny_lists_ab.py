#!/usr/bin/env python3
"""Config 5 (100 k surrogate): ms per tick against the number of balanced extra lists (SB_BALANCED_LISTS = 1, 2, 3; planner,
plan.cpp static split), interleaved rounds in one process. usage: python tools/bunny_lists_ab.py [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402


def run(mesh, ticks=100):
    sb = Softbody(mesh, substeps=20, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
    for _ in range(5):
        sb.step()
    sb.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        sb.step()
    sb.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / ticks
    st = sb.stats()
    slot_ms, slot_cnt = sb.step_profiled()
    sb.OnDestroy()
    return round(ms, 4), {"t2_layers": st["n_t2_layers"], "t2_tiles": st["n_t2_tiles"], "tiles": st["n_tiles"], "global_colours": st["n_global_colours"],
                          "slot_ms": [round(float(x), 4) for x in slot_ms], "slot_launches": [int(x) for x in slot_cnt]}


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mesh = bunny_surrogate(target_verts=100_000)
    res = {}
    for _ in range(rounds):
        for lists in ("1", "2", "3"):
            os.environ["SB_BALANCED_LISTS"] = lists
            ms, info = run(mesh)
            res.setdefault(lists, {"ms_per_tick": [], "info": info})["ms_per_tick"].append(ms)
    print(json.dumps({"mesh": mesh.label, "balanced_lists": res}))


if __name__ == "__main__":
    main()
