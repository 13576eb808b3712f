#!/usr/bin/env python3
"""Loopback check of the halo schedules (serialised / overlapped, eager / captured in the hipGraph): each combination in its
own process, results compared bit for bit per transport. usage: python tools/lb_combo_test.py [rccl] [peer]"""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
import numpy as np
from softbodyunity_amd import Softbody, comm_unique_id
from softbodyunity_amd.mesh import jelly_cube
mesh = jelly_cube(32)
sb = Softbody(mesh, substeps=8, device=0, rank=0, world=2, tile_particles=64, unique_id=comm_unique_id()).Start()
for _ in range(5):
    sb.step()
sb.synchronize()
x = sb.get_positions()[sb.owner() == 0]
print("HASH", hashlib.sha256(x.tobytes()).hexdigest()[:16], bool(np.isfinite(x).all()))
sb.OnDestroy()
''' % ROOT

TRANSPORTS = sys.argv[1:] or ["rccl"]          # rccl and/or peer (SB_HALO_TRANSPORT=peer: the peer-store transport)
for transport in TRANSPORTS:
    hashes = {}
    for overlap, graph in (("", ""), ("1", ""), ("", "1"), ("1", "1")):
        env = dict(os.environ, SB_TEST_LOOPBACK="1")
        for k, v in (("SB_HALO_OVERLAP", overlap), ("SB_GRAPH_RCCL", graph), ("SB_HALO_TRANSPORT", "peer" if transport == "peer" else "")):
            env.pop(k, None)
            if v:
                env[k] = v
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        print(f"{transport} overlap={overlap or 0} graph={graph or 0}: rc={r.returncode} {line[0] if line else r.stderr[-400:]}", flush=True)
        hashes[(overlap, graph)] = line[0] if line else None
    print(f"{transport} all equal:", len(set(hashes.values())) == 1 and None not in hashes.values())
