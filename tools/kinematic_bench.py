#!/usr/bin/env python3
"""An attached body: the n^3 cube hangs from its pinned top layer, which the host moves before every tick (sb_set_kinematic_positions).
ms per tick with the targets travelling inside the fused first kernel of the next tick (default) or completing the previous tick
first (SB_NO_KIN_FUSE=1: one more launch per tick). usage: python tools/kinematic_bench.py [n] [substeps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody, jelly_cube  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    mesh = jelly_cube(n)
    pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)
    mesh.inv_mass[pins] = 0.0
    rest = mesh.pos[pins].copy()
    sb = Softbody(mesh, substeps=S).Start()
    ticks = 60

    def run(k0):
        for t in range(ticks):
            sb.set_kinematic_positions(pins, rest + np.float32(0.05 * np.sin(0.1 * (k0 + t))))
            sb.step()
        sb.synchronize()
    run(0)
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter(); run(ticks * (r + 1)); best = min(best, 1e3 * (time.perf_counter() - t0) / ticks)
    st = sb.stats()
    x = sb.get_positions()
    sb.OnDestroy()
    print(json.dumps({"n": n, "substeps": S, "pins": int(len(pins)), "fused": not os.environ.get("SB_NO_KIN_FUSE"), "ms_per_tick": round(best, 4),
                      "ticks_fused_kinematic": st["ticks_fused_kinematic"], "ticks_fused": st["ticks_fused"], "finite": bool(np.isfinite(x).all())}))


if __name__ == "__main__":
    main()
