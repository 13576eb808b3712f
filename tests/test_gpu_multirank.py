"""Two ranks through the real RCCL halo path. The GPU box has ONE device, and RCCL refuses two ranks on one
device unless told otherwise, so this test runs only when RCCL accepts the duplicate (it is skipped, not
failed, when communicator creation is refused). The schedule itself is covered bit-exactly on CPU by
tests/test_plan.py and tests/test_multirank_gloo.py."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, uid, tile, out_dir):
    sys.path.insert(0, ROOT)
    from softbodyunity_amd import Softbody, native
    from softbodyunity_amd.mesh import jelly_cube
    mesh = jelly_cube(24, pin_top=True)
    try:
        sb = Softbody(mesh, substeps=6, device=0, rank=rank, world=world, tile_particles=tile, unique_id=uid).Start()
    except native.SoftbodyError as e:
        open(os.path.join(out_dir, f"err{rank}.txt"), "w").write(str(e))
        return
    for _ in range(3):
        sb.step()
    x = sb.get_positions(); v = sb.get_velocities()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x, v=v, owned=sb.owner() == rank, st=np.array(sb.stats()["halo_particles_t1"]))
    sb.OnDestroy()


@pytest.mark.parametrize("tile", [64, -1])
def test_two_ranks_one_device_rccl(tmp_path, oracle_mod, tile):
    from softbodyunity_amd import comm_unique_id
    from softbodyunity_amd.mesh import jelly_cube
    from helpers import build_plan, make_oracle
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    uid = comm_unique_id()
    mp.spawn(_worker, args=(2, uid, tile, str(tmp_path)), nprocs=2, join=True)
    errs = [f for f in os.listdir(tmp_path) if f.startswith("err")]
    if errs:
        pytest.skip("RCCL refused two ranks on one device: " + open(tmp_path / errs[0]).read()[:200])
    mesh = jelly_cube(24, pin_top=True)
    ref = make_oracle(oracle_mod, mesh, build_plan(mesh, tile_particles=tile))
    for _ in range(3):
        ref.step(0.02, 6)
    x = np.zeros_like(ref.x); v = np.zeros_like(ref.v)
    for r in range(2):
        d = np.load(tmp_path / f"rank{r}.npz")
        x[d["owned"]] = d["x"][d["owned"]]; v[d["owned"]] = d["v"][d["owned"]]
        if tile > 0:
            assert int(d["st"]) > 0          # ghosts really travelled
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32))
    assert np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))
