#!/usr/bin/env python3
"""The largest lattice the int32 ABI is expected to carry comfortably on one GPU: an n^3 jelly cube (default 512^3 = 134 M particles,
402 M springs) through the plugin, ONE tick of 20 substeps, against the CPU oracle on the same mesh bit for bit (the oracle walks
402 M constraints x 20 substeps: about a minute on the GPU box's host cores), then a short timing run.
usage (GPU box): python tests/fuzz/max_size_check.py [n | bunny:<vertices>] [ticks timed] [noparity] [het]
Checker script: the oracle is used as the checker only (tests/helpers.py), nothing here is product code."""
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from softbodyunity_amd import Softbody, jelly_cube  # noqa: E402
from oracle import oracle  # noqa: E402
from helpers import make_oracle  # noqa: E402

BUNNY = len(sys.argv) > 1 and sys.argv[1].startswith("bunny:")      # bunny:<vertices>: the irregular tet surrogate (springs + volumes + hinges) at that size
n = int(sys.argv[1].split(":")[1]) if BUNNY else (int(sys.argv[1]) if len(sys.argv) > 1 else 512)
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 10
PARITY = "noparity" not in sys.argv[3:]
HET = "het" in sys.argv[3:]          # per-particle masses, per-spring rest lengths: 8-byte constraint slots (bench.py --heterogeneous)
S = 20
out = {"n": n, "heterogeneous": HET, "substeps": S}


def note(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


t0 = time.time()
KW = {}
if BUNNY:
    from softbodyunity_amd.mesh import bunny_surrogate  # noqa: E402
    mesh = bunny_surrogate(target_verts=n)
    KW = dict(distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5)
else:
    mesh = jelly_cube(n, heterogeneous=HET)
out.update(particles=mesh.n, springs=len(mesh.dist_rest), tets=len(mesh.vol_rest), hinges=len(mesh.bend_rest))
out["mesh_seconds"] = round(time.time() - t0, 2)
note(f"mesh {mesh.n} particles, {len(mesh.dist_rest)} springs in {out['mesh_seconds']} s")
t0 = time.time()
sb = Softbody(mesh, substeps=S, **KW).Start()
out["start_seconds"] = round(time.time() - t0, 2)
st = sb.stats()
out["tiles"] = st["n_tiles"]; out["launch_bytes"] = [int(b) for b in st["launch_bytes"][:2]]
note(f"Start() {out['start_seconds']} s, tiles {st['n_tiles']}, bytes per mid-tick launch {out['launch_bytes']}")
try:
    sb.step(); sb.synchronize()
    x = sb.get_positions(); v = sb.get_velocities()
    out["finite"] = bool(np.isfinite(x).all() and np.isfinite(v).all())
    if PARITY:
        note("one tick on the GPU done; oracle ...")
        t0 = time.time()
        o = make_oracle(oracle, mesh, sb.plan(), compliance=(KW.get("distance_compliance", 0.0), KW.get("volume_compliance", 0.0), KW.get("bending_compliance", 0.0)))
        os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
        o.step(0.02, S, parallel=True)        # (task-parallel walk of the same published order: bit-identical to the sequential one, tests/test_oracle_kat.py)
        out["oracle_seconds"] = round(time.time() - t0, 1)
        out["bitwise_positions"] = bool(np.array_equal(x.view(np.uint32), o.x.view(np.uint32)))
        out["bitwise_velocities"] = bool(np.array_equal(v.view(np.uint32), o.v.view(np.uint32)))
        note(f"oracle {out['oracle_seconds']} s: positions {out['bitwise_positions']}, velocities {out['bitwise_velocities']}")
        del o
    del x, v
    for _ in range(2):
        sb.step()
    sb.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed):
        sb.step()
    sb.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / timed
    out["ms_per_tick"] = round(ms, 4)
    out["particle_substeps_per_s"] = mesh.n * S / (ms * 1e-3)
    # compulsory HBM bytes of a mid-tick launch (sb_get_stats: the tables actually uploaded) / launch time, against 8 TB/s
    # SURVEY 8d algorithmic bytes per substep: 88 B per particle + 68 per spring + 132 per tet / hinge
    out["algorithmic_GBps"] = (88.0 * mesh.n + 68.0 * len(mesh.dist_rest) + 132.0 * (len(mesh.vol_rest) + len(mesh.bend_rest))) * S / (ms * 1e-3) / 1e9
    out["plan"] = {k: st[k] for k in ("n_tiles", "n_t2_layers", "n_t2_tiles", "n_global_colours", "constraints_in_global")}
    out["model_GBps"] = 0.5 * (out["launch_bytes"][0] + out["launch_bytes"][1]) / (ms * 1e-3 / S) / 1e9
    out["frac_of_8TBps_model_bytes"] = out["model_GBps"] / 8000.0
    pm, pc = sb.step_profiled()           # HIP-event pair around every launch of one eager tick: T0 and T1 mid-tick launches apart
    out["us_per_launch_T0_T1"] = [round(1e3 * float(pm[k]) / max(int(pc[k]), 1), 1) for k in (0, 1)]
    out["GBps_T0_T1"] = [round(out["launch_bytes"][k] / (out["us_per_launch_T0_T1"][k] * 1e-6) / 1e9) for k in (0, 1)]
    out["finite_after_timing"] = bool(np.isfinite(sb.get_positions()).all())
finally:
    sb.OnDestroy()
out["peak_host_rss_GiB"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20, 2)
print(json.dumps(out))
