#!/usr/bin/env python3
"""The driver's multi-GPU configuration at the PLAN level, without a GPU: the 256^3 cube handed over as windows (sharded authoring, what
`bench.py --gpus N` does) on 2, 4 and 8 ranks -- the owned sets partition the cube, every pair of ranks agrees on what it shares (pair hashes
symmetric), every rank plans a tick program of the same shape, and the windows' pair hashes equal those of ranks planning the WHOLE mesh.
Host-only planner; ~1 minute, ~4 GB. usage: python tests/fuzz/windows_256.py   (profiles/r04zzz_windows_256_plan_check.txt, build container)"""
import os
import sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from softbodyunity_amd import native
from softbodyunity_amd.mesh import jelly_cube_window, jelly_cube
n, tile = 256, 512
for W in (2, 4, 8):
    t0=time.time()
    ph=[]; owned=0; slots=set()
    for r in range(W):
        w=jelly_cube_window(n, r, W, (0,0,0), tile)
        p=native.Plan.build(w.rest_pos, w.dist_ij, rank=r, world=W, tile_particles=tile, partition=1, domain=w.domain, global_id=w.global_id)
        ph.append(p.pair_hashes()); owned+=int((p.owner(w.n)==r).sum()); slots.add(p.halo_slot_count())
        del p, w
    sym=all(ph[a][b]==ph[b][a] for a in range(W) for b in range(W) if a!=b)
    print(f"W={W}: owned total {owned} == {n**3}: {owned==n**3}; pair hashes symmetric: {sym}; halo slot counts {slots}; {time.time()-t0:.1f}s", flush=True)
    if W==8: ph8=ph
# whole-mesh plan of rank 0 and 7 at W=8: pair hashes must equal the windows'
m=jelly_cube(n)
for r in (0,7):
    p=native.Plan.build(m.rest_pos, m.dist_ij, rank=r, world=8, tile_particles=tile, partition=1)
    print("whole-mesh rank",r,"pair hashes equal the window's:", list(p.pair_hashes())==list(ph8[r]), flush=True)
    del p
