#!/usr/bin/env python3
"""The driver's 8-GPU configuration EXECUTED on the CPU: the 256^3 cube as 8 windows (sharded authoring, what `bench.py --gpus 8` hands
over), every rank a partitioned oracle walking ITS window's local program in the kernel order of one tick (tests/helpers.py run_tick: T0 / T1
tile kernels, ghost refresh before every T1 kernel through the ranks' own send / receive lists), halo by memcpy -- the state after one tick of
20 substeps must equal the golden checksum of the UNPARTITIONED oracle (tests/golden/state_checksums.json). No GPU; test infrastructure (the
oracle is the checker of the plan here). ~5 minutes, ~6 GB. usage: python tests/fuzz/windows_256_run.py [n=256] [world=8] [ticks=1] [het]"""
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                      # noqa: E402
from oracle import oracle                                               # noqa: E402
from helpers import WindowRankSim, run_tick                             # noqa: E402
from softbodyunity_amd.mesh import jelly_cube_window                    # noqa: E402
from softbodyunity_amd.verify import add_checksums, state_checksum      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
het = len(sys.argv) > 4 and sys.argv[4] == "het"      # every particle its own mass, every spring its own rest length
tile, S, dt = 512, 20, 0.02
t0 = time.time()
ranks = [WindowRankSim(oracle, jelly_cube_window(n, r, W, (0, 0, 0), tile, heterogeneous=het), r, W, (0, 0, 0), tile) for r in range(W)]
print(f"{W} windows of {n}^3 planned: {[int(R.owned.sum()) for R in ranks]} owned particles, {time.time() - t0:.1f} s", flush=True)


def exchange(slot, with_prev):
    staged = []
    for R in ranks:
        if slot >= len(R.halos):
            continue
        for peer, (_, recv) in R.halos[slot].items():
            if len(recv):
                Q = ranks[peer]
                send = Q.halos[slot][R.rank][0]
                assert np.array_equal(Q.gid[send], R.gid[recv]), "send / recv lists of a halo slot differ between the two ranks"
                staged.append((R, recv, Q.o.x[send].copy(), Q.o.xprev[send].copy() if with_prev else None))
    for R, ids, vals, prev in staged:
        R.o.x[ids] = vals
        if prev is not None:
            R.o.xprev[ids] = prev


golden = json.load(open(os.path.join(ROOT, "tests", "golden", "state_checksums.json"))).get(f"cube{n}{'het' if het else ''}_s{S}_tile{tile}")
ok = True
for t in range(1, ticks + 1):
    s = ranks[0].o.scalars(dt, S)
    run_tick(ranks, s, S, True, exchange)
    parts = [state_checksum(R.o.x[R.owned], R.o.v[R.owned], R.gid[R.owned]) for R in ranks]
    got = add_checksums(parts)
    want = golden["ticks"].get(str(t)) if golden else None
    same = want is not None and int(want, 16) == got
    ok = ok and same
    print(f"tick {t}: checksum 0x{got:016x} golden {want} bitwise {same} ({time.time() - t0:.1f} s)", flush=True)
print("WINDOWS RUN OK" if ok else "WINDOWS RUN MISMATCH")
sys.exit(0 if ok else 1)
