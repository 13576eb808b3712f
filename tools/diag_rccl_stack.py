#!/usr/bin/env python3
"""One-off diagnosis (VERDICT r2 item 2 / ADVICE r2 item 1): which RCCL / HIP runtime the plugin binds, and where the
captured multi-rank schedules fault. Loopback (SB_TEST_LOOPBACK) on one GPU.
usage: diag_rccl_stack.py [--torch-first] [--overlap] [--graph] [--solvers N] [--together]"""
import argparse
import hashlib
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--torch-first", action="store_true")
ap.add_argument("--overlap", action="store_true")
ap.add_argument("--graph", action="store_true")
ap.add_argument("--solvers", type=int, default=1)
ap.add_argument("--together", action="store_true", help="keep every solver alive until the end (else create/step/destroy one after the other)")
a = ap.parse_args()
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SB_TEST_LOOPBACK"] = "1"
for k, on in (("SB_HALO_OVERLAP", a.overlap), ("SB_GRAPH_RCCL", a.graph)):
    os.environ.pop(k, None)
    if on:
        os.environ[k] = "1"
if a.torch_first:
    import torch  # noqa: F401
    print("torch", torch.__version__, "hip", torch.version.hip, flush=True)
import numpy as np
from softbodyunity_amd import Softbody, comm_unique_id
from softbodyunity_amd.mesh import jelly_cube


def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        if any(s in p for s in ("librccl", "libamdhip64", "libhsa-runtime", "libsoftbody")) and p not in seen:
            seen.add(p)
            print("mapped:", p, flush=True)


mesh = jelly_cube(32)
alive = []
for k in range(a.solvers):
    sb = Softbody(mesh, substeps=8, device=0, rank=0, world=2, tile_particles=64, unique_id=comm_unique_id()).Start()
    if k == 0:
        maps()
    for _ in range(5):
        sb.step()
    sb.synchronize()
    x = sb.get_positions()[sb.owner() == 0]
    print(f"solver {k}: HASH", hashlib.sha256(x.tobytes()).hexdigest()[:16], bool(np.isfinite(x).all()), flush=True)
    if a.together:
        alive.append(sb)
    else:
        sb.OnDestroy()
        print(f"solver {k}: destroyed", flush=True)
for k, sb in enumerate(alive):
    sb.step(); sb.synchronize()
for k, sb in enumerate(alive):
    sb.OnDestroy()
    print(f"solver {k}: destroyed", flush=True)
print("DONE", flush=True)
