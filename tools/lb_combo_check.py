#!/usr/bin/env python3
"""Loopback check of the halo schedules (serialised / overlapped, eager / captured in the hipGraph): each combination in its
own process, results compared bit for bit per transport. usage: python tools/lb_combo_check.py [--torch-first] [rccl] [peer]

--torch-first imports PyTorch before the plugin in every child, as bench.py and the driver's N > 1 launch do: the plugin is then
bound to PyTorch's bundled HIP runtime + RCCL (sb_runtime_info), on which the overlapped + captured schedule must be REFUSED
with SB_ERR_UNSUPPORTED (-7) instead of crashing (HIP < 7.2: unbounded recursion in hipStreamEndCapture)."""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
if os.environ.get("LB_TORCH_FIRST"):
    import torch
import numpy as np
from softbodyunity_amd import Softbody, comm_unique_id, native
from softbodyunity_amd.mesh import jelly_cube
ri = native.runtime_info()
print("BOUND hip", ri["hip_runtime_version"], "rccl", ri["rccl_version"], ri["rccl_library"], flush=True)
mesh = jelly_cube(32)
try:
    sb = Softbody(mesh, substeps=8, device=0, rank=0, world=2, tile_particles=64, unique_id=comm_unique_id()).Start()
except native.SoftbodyError as e:
    print("REFUSED", e.code, flush=True)
    sys.exit(0)
print("SCHEDULE", sb.stats()["halo_schedule"], flush=True)
for _ in range(5):
    sb.step()
sb.synchronize()
x = sb.get_positions()[sb.owner() == 0]
print("HASH", hashlib.sha256(x.tobytes()).hexdigest()[:16], bool(np.isfinite(x).all()))
sb.OnDestroy()
''' % ROOT

args = sys.argv[1:]
torch_first = "--torch-first" in args
TRANSPORTS = [a for a in args if not a.startswith("--")] or ["rccl"]          # rccl and/or peer (the peer-store transport)
for transport in TRANSPORTS:
    hashes = {}
    for overlap, graph in (("", ""), ("1", ""), ("", "1"), ("1", "1")):
        env = dict(os.environ, SB_TEST_LOOPBACK="1")       # (read by the Python harness: sb_desc.debug_flags, halo_schedule, halo_transport)
        for k, v in (("SB_HALO_OVERLAP", overlap), ("SB_GRAPH_RCCL", graph), ("SB_HALO_TRANSPORT", "peer" if transport == "peer" else ""),
                     ("LB_TORCH_FIRST", "1" if torch_first else "")):
            env.pop(k, None)
            if v:
                env[k] = v
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith(("HASH", "REFUSED"))]
        bound = [l for l in r.stdout.splitlines() if l.startswith("BOUND")]
        print(f"{transport} overlap={overlap or 0} graph={graph or 0}: rc={r.returncode} {line[0] if line else r.stderr[-400:]}  [{bound[0] if bound else ''}]", flush=True)
        hashes[(overlap, graph)] = line[0] if line else None
    ran = [h for h in hashes.values() if h and h.startswith("HASH")]
    print(f"{transport} all equal:", len(set(ran)) == 1 and None not in hashes.values(), "refused:", sorted(k for k, h in hashes.items() if h and h.startswith("REFUSED")))
