"""sb_group_* (include/softbody_group.h): ONE process driving every rank of a partitioned solver -- what a Unity player with several
GPUs does behind one Softbody component. All ranks sit on the box's ONE device here (between two devices nothing has ever run), each
scenario in a process of its own (tests/group_case.py): bit for bit the CPU oracle of the unpartitioned mesh."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*args):
    # several ranks of one process on one device: a hardware queue per rank for the peer transport's waiting kernels
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "group_case.py"), *[str(a) for a in args]], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "GROUP OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    return out.stdout


@pytest.mark.parametrize("host", ["threads", "walk"])
@pytest.mark.parametrize("world,mesh_kind", [(4, "cube"), (3, "bunny"), (4, "blocks")])
def test_one_process_drives_every_rank_through_the_group(world, mesh_kind, host):
    # the peer transport inside one process (mailboxes by plain pointer); "blocks": the group cuts the ranks' windows itself
    out = _run("basic", world, mesh_kind, "peer", host)
    if mesh_kind == "blocks":
        assert "sharded=True" in out


@pytest.mark.parametrize("host", ["threads", "walk"])
@pytest.mark.parametrize("world,mesh_kind", [(4, "cube"), (2, "bunny")])
def test_group_over_rccl_self_exchange(world, mesh_kind, host):
    # W communicators of one process; walk mode: every rank's sends / receives of an exchange inside ONE ncclGroupStart / ncclGroupEnd
    _run("basic", world, mesh_kind, "rccl-loopback", host)


@pytest.mark.parametrize("host", ["threads", "walk"])
def test_group_kinematic_pins_peek_and_render_readback(host):
    _run("features", 4, host)


def test_a_group_of_one_is_a_plain_solver(oracle_mod):
    import numpy as np
    from helpers import build_plan, make_oracle
    from softbodyunity_amd import SoftbodyGroup, jelly_cube
    mesh = jelly_cube(16)
    g = SoftbodyGroup(mesh, [0], substeps=10).Start()
    try:
        ref = make_oracle(oracle_mod, mesh, build_plan(mesh))
        for _ in range(3):
            g.FixedUpdate(); ref.step(0.02, 10)
        assert np.array_equal(g.vertices.view(np.uint32), ref.x.view(np.uint32))
        assert np.array_equal(g.get_velocities().view(np.uint32), ref.v.view(np.uint32))
    finally:
        g.OnDestroy()
