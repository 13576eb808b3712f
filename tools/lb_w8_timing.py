#!/usr/bin/env python3
"""Rank 0's share of the 256^3 cube split over 8 ranks, RCCL self-exchange (SB_TEST_LOOPBACK), for the four halo schedules;
no torch in the process (the plugin binds /opt/rocm's RCCL). usage: python tools/lb_w8_timing.py [ticks] [world] [serial]"""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TICKS = int(sys.argv[1]) if len(sys.argv) > 1 else 100     # both transports: RCCL send/recv and the peer-store mailboxes
WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ONLY_SERIAL = len(sys.argv) > 3 and sys.argv[3] == "serial"
ONLY_RCCL = len(sys.argv) > 3 and sys.argv[3] == "rccl"
CHILD = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import numpy as np
from softbodyunity_amd import Softbody, comm_unique_id
from softbodyunity_amd.mesh import jelly_cube
mesh = jelly_cube(256)
from softbodyunity_amd import native
sched = native.halo_schedule_from_env()
if sched == native.SB_SCHEDULE_AUTO and not os.environ.get("LB_SCHEDULE"):
    sched = native.SB_SCHEDULE_SERIAL_EAGER
sb = Softbody(mesh, substeps=20, device=0, rank=0, world=%d, unique_id=comm_unique_id(), halo_schedule=sched).Start()
print("SCHEDULE", sb.stats()["halo_schedule"], flush=True)
for _ in range(10):
    sb.step()
sb.synchronize()
t0 = time.perf_counter()
for _ in range(%d):
    sb.step()
sb.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / %d
x = sb.get_positions()[sb.owner() == 0]
print("RESULT %%.4f ms/tick hash %%s finite %%s" %% (ms, hashlib.sha256(x.tobytes()).hexdigest()[:12], bool(np.isfinite(x).all())))
sb.OnDestroy()
''' % (ROOT, WORLD, TICKS, TICKS)

# (since round 3 SB_SCHEDULE_AUTO picks the overlapped eager schedule when the largest per-peer message is >= 1 MiB: the "overlap=0 graph=0"
# row therefore asks for the serialised eager schedule explicitly, LB_SERIAL_EAGER -> halo_schedule = SB_SCHEDULE_SERIAL_EAGER)
for transport in (("rccl",) if ONLY_RCCL else ("rccl", "peer")):
    for overlap, graph in ((("", ""),) if ONLY_SERIAL else (("", ""), ("", "1"), ("1", ""), ("1", "1"), ("auto", ""))):
        env = dict(os.environ, SB_TEST_LOOPBACK="1")
        for k, v in (("SB_HALO_OVERLAP", "" if overlap == "auto" else overlap), ("SB_GRAPH_RCCL", graph), ("SB_HALO_TRANSPORT", "peer" if transport == "peer" else ""),
                     ("LB_SCHEDULE", "auto" if overlap == "auto" else "")):
            env.pop(k, None)
            if v:
                env[k] = v
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD], env=env, capture_output=True, text=True, timeout=400)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        sch = [l for l in r.stdout.splitlines() if l.startswith("SCHEDULE")]
        print(f"W={WORLD} {transport} overlap={overlap or 0} graph={graph or 0}: rc={r.returncode} {line[0] if line else r.stderr[-300:]} [{sch[0] if sch else ''}]", flush=True)
