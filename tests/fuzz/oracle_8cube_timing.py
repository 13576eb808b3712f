import sys, time, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle
from softbodyunity_amd.mesh import jelly_cube
from helpers import build_plan, make_oracle
m = jelly_cube(8); o = make_oracle(oracle, m, build_plan(m))
for _ in range(200): o.step(0.02, 10)
t = time.perf_counter(); n = 20000
for _ in range(n): o.step(0.02, 10)
dt = time.perf_counter() - t
print("8^3 S=10 oracle 1 thread: %.3e particle-substeps/s (%.1f us per tick)" % (512 * 10 * n / dt, 1e6 * dt / n))
