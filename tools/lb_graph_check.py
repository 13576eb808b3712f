#!/usr/bin/env python3
"""Loopback check that capturing the RCCL calls in the hipGraph gives the eager bits (run as a script on the GPU box)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SB_TEST_LOOPBACK"] = "1"
from softbodyunity_amd import Softbody, comm_unique_id
from softbodyunity_amd.mesh import jelly_cube
mesh = jelly_cube(32)
outs = []
for g in ("", "1"):
    if g: os.environ["SB_GRAPH_RCCL"] = "1"
    else: os.environ.pop("SB_GRAPH_RCCL", None)
    sb = Softbody(mesh, substeps=8, rank=0, world=2, tile_particles=64, unique_id=comm_unique_id()).Start()
    for _ in range(5): sb.step()
    sb.synchronize()
    x = sb.get_positions()[sb.owner() == 0].copy(); sb.OnDestroy(); outs.append(x)
print("graph == eager:", np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)))
