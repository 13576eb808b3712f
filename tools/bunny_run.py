#!/usr/bin/env python3
"""The 100k surrogate for a profiler run: 5 warm-up ticks + 20 ticks of 20 substeps (default tile size)."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody  # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402
sb = Softbody(bunny_surrogate(target_verts=100_000), substeps=20, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
for _ in range(25):
    sb.step()
sb.synchronize()
print(sb.stats())
sb.OnDestroy()
