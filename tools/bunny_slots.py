import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody
from softbodyunity_amd.mesh import bunny_surrogate
mesh = bunny_surrogate(target_verts=100000)
for env in ({}, {"SB_NO_T2": "1"}):
    os.environ.pop("SB_NO_T2", None); os.environ.update(env)
    sb = Softbody(mesh, substeps=20, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-5).Start()
    for _ in range(3): sb.step()
    ms, cnt = sb.step_profiled()
    G = sb.stats()["n_global_colours"]
    print(env, "T0 mid %.3f/%d  T1 mid %.3f/%d  globals %.3f/%d  first %.3f last %.3f  T2 %.3f/%d" % (ms[0], cnt[0], ms[1], cnt[1], ms[2:2+G].sum(), cnt[2:2+G].sum(), ms[2+G], ms[3+G], ms[4+G], cnt[4+G]))
    sb.OnDestroy()
