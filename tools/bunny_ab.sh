#!/bin/bash
# Interleaved A/B of plugin variants on config 5 (100 k surrogate): tools/bunny_ab.sh v1 v2 ...  ("product" = the product build)
# One process per measurement (a library is loaded once per process); ROUNDS rounds. Prints ms per tick.
ROUNDS=${ROUNDS:-3}
for round in $(seq 1 $ROUNDS); do
  for name in "$@"; do
    v=$name; [ "$v" = "product" ] && v=""
    SB_LIB_VARIANT=$v python - <<'PY' || echo "$name FAILED"
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from softbodyunity_amd import Softbody
from softbodyunity_amd.mesh import bunny_surrogate
sb = Softbody(bunny_surrogate(target_verts=100_000), substeps=20, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
for _ in range(5): sb.step()
sb.synchronize(); t0 = time.perf_counter()
for _ in range(100): sb.step()
sb.synchronize()
print("%-12s %.4f ms/tick" % (os.environ.get("SB_LIB_VARIANT") or "product", 1e3 * (time.perf_counter() - t0) / 100))
sb.OnDestroy()
PY
  done
done
