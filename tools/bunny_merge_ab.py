#!/usr/bin/env python3
"""Config 5 (100 k surrogate): ms per tick with and without the tile merge that places the leftover constraints in the balanced
lists (SB_NO_TILE_MERGE=1 keeps the cluster layer), interleaved rounds in one process. usage: python tools/bunny_merge_ab.py [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bunny_lists_ab import run  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mesh = bunny_surrogate(target_verts=100_000)
    os.environ.pop("SB_BALANCED_LISTS", None)
    res = {}
    for _ in range(rounds):
        for name, val in (("merge (default)", None), ("SB_NO_TILE_MERGE=1", "1")):
            os.environ.pop("SB_NO_TILE_MERGE", None)
            if val:
                os.environ["SB_NO_TILE_MERGE"] = val
            ms, info = run(mesh)
            res.setdefault(name, {"ms_per_tick": [], "info": info})["ms_per_tick"].append(ms)
    print(json.dumps({"mesh": mesh.label, "tile_merge": res}))


if __name__ == "__main__":
    main()
