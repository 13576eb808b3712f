#!/usr/bin/env python3
"""sb_group_* host models on ONE device: what a tick costs the HOST when one process drives 8 ranks of the 256^3 cube -- a thread per rank
(default) against the calling thread walking the tick across the ranks (SB_GROUP_WALK). All ranks share the box's one GPU, so the GPU time
per tick says nothing about scaling; the host's enqueue time per tick does: it must stay below the ~0.7 ms a rank's GPU work takes at
256^3 / 8, or the host is the bottleneck. usage: python tools/group_host_models.py [n=256] [world=8] [ticks=30]
(GPU_MAX_HW_QUEUES >= world for the peer transport's waiting kernels)"""
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np                                                       # noqa: E402
from softbodyunity_amd import SoftbodyGroup, jelly_cube, native          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dims = {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world, (world, 1, 1))
mesh = jelly_cube(n)
rows = []
for transport, tname, dbg in ((native.SB_TRANSPORT_PEER, "peer", 0), (native.SB_TRANSPORT_RCCL, "rccl self-exchange", native.SB_DEBUG_LOOPBACK)):
    for walk in (False, True):
        t0 = time.perf_counter()
        g = SoftbodyGroup(mesh, [0] * world, substeps=20, tile_particles=512, partition=native.SB_PARTITION_BLOCKS, part_dims=dims,
                          halo_transport=transport, debug_flags=dbg, walk=walk).Start()
        setup = time.perf_counter() - t0
        try:
            for _ in range(8):      # (past the seven ticks in which SB_SCHEDULE_AUTO measures the two eager schedules and decides)
                g.step()
            g.synchronize()
            call = []
            t0 = time.perf_counter()
            for _ in range(ticks):
                c0 = time.perf_counter()
                g.step()
                call.append(time.perf_counter() - c0)
            t_enq = time.perf_counter() - t0
            g.synchronize()
            t_all = time.perf_counter() - t0
            x = g.get_positions()
            row = {"host_model": "walk (calling thread)" if walk else "thread per rank", "transport": tname, "n": n, "world": world, "ticks": ticks,
                   "host_ms_per_tick_enqueue": 1e3 * t_enq / ticks, "host_ms_per_call_median": 1e3 * float(np.median(call)),
                   "wall_ms_per_tick_all_ranks_on_one_gpu": 1e3 * t_all / ticks, "finite": bool(np.isfinite(x).all()), "setup_s": setup}
            rows.append(row)
            print(json.dumps(row), flush=True)
        finally:
            g.OnDestroy()
