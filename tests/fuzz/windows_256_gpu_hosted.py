#!/usr/bin/env python3
"""The driver's 8-GPU configuration on ONE GPU, device path included: the 256^3 cube as 8 WINDOWS (sharded authoring, what `bench.py --gpus 8`
hands over), every rank a complete solver on the one device (its own plan, tiles with ghost runs, lane-packed slots, pack kernel, T1 kernels
reading the receive buffer), driven launch by launch with the host as the wire (tests/hosted.py) -- everything of the 8-rank launch except
RCCL and xGMI. The state after each tick of 20 substeps must equal the golden checksum of the unpartitioned CPU oracle.
usage (GPU box): python tests/fuzz/windows_256_gpu_hosted.py [n=256] [world=8] [ticks=2]"""
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                      # noqa: E402
from hosted import HostedRanks                                          # noqa: E402
from softbodyunity_amd import Softbody, native                          # noqa: E402
from softbodyunity_amd.mesh import jelly_cube_window                    # noqa: E402
from softbodyunity_amd.verify import add_checksums, state_checksum      # noqa: E402


class HostedWindowRanks(HostedRanks):
    """HostedRanks whose ranks were handed their WINDOW of the mesh (own numbering, whole-mesh ids in mesh.global_id)."""

    def __init__(self, windows, substeps, dt=0.02, **kw):
        self.mesh, self.world, self.S, self.dt = None, len(windows), substeps, dt
        self.windows = windows
        self.ranks = []
        try:
            for r, w in enumerate(windows):
                self.ranks.append(Softbody(w, substeps=substeps, fixed_delta_time=dt, device=0, rank=r, world=self.world, unique_id=bytes(128),
                                           partition=native.SB_PARTITION_BLOCKS, debug_flags=native.SB_DEBUG_NO_COMM, **kw).Start())
        except Exception:
            self.close()
            raise
        self.L = native.lib()
        st = self.ranks[0].stats()
        self.G, self.n_t2, self.tiling = st["n_global_colours"], st["n_t2_layers"], st["n_tilings"] == 2
        plans = [sb.plan() for sb in self.ranks]
        self.counts = []
        for slot in range(plans[0].halo_slot_count()):
            per_rank = []
            for p in plans:
                sc = np.zeros(self.world, np.int32); rc = np.zeros(self.world, np.int32)
                native.check(self.L.sb_plan_halo_counts(p._h, slot, native.ptr(sc), native.ptr(rc)))
                per_rank.append((sc, rc))
            for a in range(self.world):
                for b in range(self.world):
                    assert per_rank[a][0][b] == per_rank[b][1][a], "send and receive counts of a halo slot differ between two ranks"
            self.counts.append(per_rank)
        self.exchanged_floats = 0

    def checksum(self):
        parts = []
        for r, sb in enumerate(self.ranks):
            own = sb.owner() == r
            parts.append(state_checksum(sb.get_positions()[own], sb.get_velocities()[own], self.windows[r].global_id[own]))
        return add_checksums(parts)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tile, S = 512, 20
t0 = time.time()
wins = [jelly_cube_window(n, r, W, (0, 0, 0), tile) for r in range(W)]
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "state_checksums.json"))).get(f"cube{n}_s{S}_tile{tile}")
ok = True
with HostedWindowRanks(wins, S, tile_particles=tile) as H:
    st = [sb.stats() for sb in H.ranks]
    print(f"{W} ranks on windows of {n}^3: owned {[s['n_particles_owned'] for s in st]}, ghosts {[s['n_particles_local'] - s['n_particles_owned'] for s in st]}, "
          f"T0 tiles {[s['n_tiles'][0] for s in st]}, lane-packed tiles {[s['lane_packed_tiles'] for s in st][:2]} ..., set-up {time.time() - t0:.1f} s", flush=True)
    for t in range(1, ticks + 1):
        H.tick()
        got = H.checksum()
        want = golden["ticks"].get(str(t)) if golden else None
        same = want is not None and int(want, 16) == got
        ok = ok and same
        print(f"tick {t}: checksum 0x{got:016x} golden {want} bitwise {same} ({time.time() - t0:.1f} s)", flush=True)
    val = [sb.validate()["errors"] for sb in H.ranks]
    ok = ok and all(v == [0] * 6 for v in val)
    print("table validator:", val[0], "on every rank" if all(v == val[0] for v in val) else val)
print("HOSTED WINDOWS OK" if ok else "HOSTED WINDOWS MISMATCH")
sys.exit(0 if ok else 1)
