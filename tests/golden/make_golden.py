"""Generates tests/golden/*.npz from the CPU oracle (NOT from the reference: /root/reference holds only
README.md:1, so there is nothing to import or run — these vectors pin regressions, not reference parity).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from softbodyunity_amd.mesh import jelly_cube  # noqa: E402
from helpers import build_plan, make_oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # config 1 (BASELINE.json:7): 8^3 cube, 10 substeps, 5 ticks, default tiling
    mesh = jelly_cube(8)
    plan = build_plan(mesh)
    o = make_oracle(oracle, mesh, plan)
    for _ in range(5):
        o.step(0.02, 10)
    (t0, i0), (t1, i1) = plan.order(0), plan.order(1)
    np.savez_compressed(os.path.join(HERE, "cube8_s10_t5.npz"), x=o.x, v=o.v, order_id0=i0, order_id1=i1)
    # natural-order variant (no planner involved): pins the oracle itself
    o = make_oracle(oracle, mesh, None)
    for _ in range(5):
        o.step(0.02, 10)
    np.savez_compressed(os.path.join(HERE, "cube8_s10_t5_natural.npz"), x=o.x, v=o.v)


if __name__ == "__main__":
    main()
