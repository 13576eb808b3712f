"""Generates tests/golden/state_checksums.json from the CPU oracle: order-independent checksums
(softbodyunity_amd/verify.py) of positions + velocities of the benchmark configurations after 1..T ticks.

NOT generated from the reference (/root/reference holds only README.md:1 -- nothing to import or run): these values
pin the HIP path to the oracle at sizes where a live oracle run inside bench.py would take minutes. bench.py compares
the state it timed (after warmup + steps ticks) with the entry for that tick count; tests/test_gpu_parity.py does the
same for a few tick counts. A planner change that alters the published schedule changes `schedule` and the generator
must be re-run:

    python tests/golden/make_checksums.py            # all configurations (256^3 takes a few minutes on 8 cores)
    python tests/golden/make_checksums.py 64         # one cube edge only
    python tests/golden/make_checksums.py het 256    # only the heterogeneous variant of one edge
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from softbodyunity_amd.mesh import jelly_cube  # noqa: E402
from softbodyunity_amd.verify import schedule_hash, state_checksum  # noqa: E402
from helpers import build_plan, make_oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "state_checksums.json")
# (cube edge, substeps, tile_particles, ticks): BASELINE.json:8 (64^3) and :9/:10 (256^3), bench.py defaults
# the last field: heterogeneous masses / rest lengths (bench.py --heterogeneous, the worst-case data layout)
CONFIGS = [(64, 20, 512, 40, False), (256, 20, 512, 40, False), (64, 20, 512, 40, True), (256, 20, 512, 40, True)]


def key(n, substeps, tile, het=False):
    return f"cube{n}{'het' if het else ''}_s{substeps}_tile{tile}"


def main():
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    het_only = "het" in sys.argv[1:]
    table = json.load(open(OUT)) if os.path.exists(OUT) else {}
    os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
    for n, S, tile, ticks, het in CONFIGS:
        if (only and n not in only) or (het_only and not het):
            continue
        t0 = time.time()
        mesh = jelly_cube(n, heterogeneous=het)
        plan = build_plan(mesh, tile_particles=tile)
        o = make_oracle(oracle, mesh, plan)
        entry = {"n_particles": mesh.n, "substeps": S, "dt": 0.02, "tile_particles": tile, "schedule": schedule_hash(plan),
                 "source": "oracle/oracle.c orc_step_tasks (bit-identical to the sequential orc_step, tests/test_plan.py)",
                 "ticks": {}}
        for t in range(1, ticks + 1):
            o.step(0.02, S, parallel=True)
            entry["ticks"][str(t)] = f"0x{state_checksum(o.x, o.v):016x}"
            print(f"{key(n, S, tile, het)} tick {t}: {entry['ticks'][str(t)]}  ({time.time() - t0:.0f} s)", flush=True)
        table[key(n, S, tile, het)] = entry
        with open(OUT, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
