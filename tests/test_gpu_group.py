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
@pytest.mark.parametrize("world,mesh_kind", [(4, "cube"), (2, "bunny"), (2, "blocks")])
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


@pytest.mark.parametrize("walk", [False, True])
def test_a_transport_that_cannot_be_set_up_fails_the_whole_group_cleanly(walk):
    # RCCL does not take two ranks on one device: every rank's communicator set-up fails -- on the ranks' threads, or inside the one RCCL
    # group of walk mode -- and sb_group_finalize reports it (no hang, no half-built group), after which the group can only be destroyed
    import numpy as np
    from softbodyunity_amd import SoftbodyGroup, jelly_cube, native
    g = SoftbodyGroup(jelly_cube(12), [0, 0], substeps=4, tile_particles=64, halo_transport=native.SB_TRANSPORT_RCCL, walk=walk)
    with pytest.raises(native.SoftbodyError) as e:
        g.Start()
    assert e.value.code == native.SB_ERR_RCCL, str(e.value)
    assert g._g is None            # Start() destroyed what it had built


def test_group_calls_out_of_order_are_refused():
    import ctypes as C
    import numpy as np
    from softbodyunity_amd import jelly_cube, native
    L = native.lib()
    d = native.SbDesc(); L.sb_desc_default(C.byref(d))
    g = C.c_void_p()
    native.check(L.sb_group_create(C.byref(d), None, 1, 0, C.byref(g)))
    try:
        assert L.sb_group_finalize(g) == native.SB_ERR_STATE                       # nothing authored
        assert L.sb_group_step(g, 0.02, 4) == native.SB_ERR_STATE
        out = np.zeros((8, 3), np.float32)
        assert L.sb_group_get_positions(g, native.ptr(out), 8) == native.SB_ERR_STATE
        assert L.sb_group_readback_begin(g) == native.SB_ERR_STATE
        m = jelly_cube(8)
        pos = native.f32(m.pos, (-1, 3)); w = native.f32(m.inv_mass, (-1,))
        native.check(L.sb_group_set_particles(g, native.ptr(pos), None, native.ptr(w), m.n))
        ij = native.i32(m.dist_ij, (-1, 2)); r = native.f32(m.dist_rest, (-1,))
        bad = ij.copy(); bad[0, 1] = m.n
        assert L.sb_group_set_distance_constraints(g, native.ptr(bad), native.ptr(r), len(r), 0.0) == native.SB_ERR_INVALID_ARG
        native.check(L.sb_group_set_distance_constraints(g, native.ptr(ij), native.ptr(r), len(r), 0.0))
        native.check(L.sb_group_finalize(g))
        assert L.sb_group_finalize(g) == native.SB_ERR_STATE and L.sb_group_set_particles(g, native.ptr(pos), None, native.ptr(w), m.n) == native.SB_ERR_STATE
        native.check(L.sb_group_step(g, 0.02, 4))
        assert L.sb_group_get_positions(g, native.ptr(out), 8) == native.SB_ERR_INVALID_ARG       # wrong n
        p = C.POINTER(C.c_float)()
        assert L.sb_group_readback_end(g, C.byref(p)) == native.SB_ERR_STATE                        # nothing pending
        native.check(L.sb_group_readback_begin(g)); native.check(L.sb_group_readback_begin(g))
        assert L.sb_group_readback_begin(g) == native.SB_ERR_STATE                                  # two pending already
    finally:
        assert L.sb_group_destroy(g) == native.SB_OK                                                # with snapshots still pending
