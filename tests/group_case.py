"""One sb_group_* scenario in a process of its own (tests/test_gpu_group.py starts it: several ranks of ONE process on ONE device need
GPU_MAX_HW_QUEUES set before the first HIP call for the peer transport). Prints `GROUP OK ...` or `GROUP MISMATCH ...`.

usage: group_case.py basic    <world> <cube|bunny|blocks> <peer|rccl-loopback> <threads|walk>
       group_case.py features <world> <threads|walk>
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import oracle                                              # noqa: E402  (test infrastructure: the checker)
from helpers import build_plan, make_oracle                            # noqa: E402
from softbodyunity_amd import Softbody, SoftbodyGroup, comm_unique_id, native   # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube         # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def basic(world, mesh_kind, transport, host):
    walk = host == "walk"
    if mesh_kind == "bunny":
        mesh, tile, comp, S = bunny_surrogate(target_verts=3000, seed=9), 128, (1e-7, 1e-7, 1e-4), 6
    elif mesh_kind == "blocks":      # the block partition: the group cuts every rank's window itself (sharded authoring)
        mesh, tile, comp, S = jelly_cube(32, pin_top=True), 64, (0.0, 0.0, 0.0), 6
    else:
        mesh, tile, comp, S = jelly_cube(24, pin_top=True), 64, (0.0, 0.0, 0.0), 6
    kw = dict(substeps=S, tile_particles=tile, distance_compliance=comp[0], volume_compliance=comp[1], bending_compliance=comp[2],
              partition=native.SB_PARTITION_BLOCKS if mesh_kind == "blocks" else native.SB_PARTITION_AUTO)
    if transport == "peer":
        tune = None
        if mesh_kind == "bunny":       # ranks of unequal size: a read between two ticks peeks on the ranks with at least this many tiles and
            tune = native.SbTuning(); native.lib().sb_tuning_default(C.byref(tune)); tune.peek_min_tiles = 9      # completes the tick on the others
        g = SoftbodyGroup(mesh, [0] * world, halo_transport=native.SB_TRANSPORT_PEER, walk=walk, tuning=tune, **kw).Start()
        try:
            for t in range(3):
                g.step()
                if t == 1:
                    mid = g.get_positions().copy()
            x, v = g.get_positions(), g.get_velocities()
            stats = [g.rank(r).stats() for r in range(world)]
            val = [g.rank(r).validate() for r in range(world)]
        finally:
            g.OnDestroy()
        ref = make_oracle(oracle, mesh, build_plan(mesh, tile_particles=tile), compliance=comp)
        ok = True
        for t in range(3):
            ref.step(0.02, S)
            if t == 1:
                ok = ok and np.array_equal(bits(mid), bits(ref.x))
        ok = ok and np.array_equal(bits(x), bits(ref.x)) and np.array_equal(bits(v), bits(ref.v))
        ghosts = sum(s["n_particles_local"] - s["n_particles_owned"] for s in stats)
        owned = sum(s["n_particles_owned"] for s in stats)
        ok = ok and ghosts > 0 and owned == mesh.n and all(r["errors"] == [0] * 6 for r in val)
        print(("GROUP OK" if ok else "GROUP MISMATCH"), f"peer world={world} mesh={mesh_kind} host={host} ghosts={ghosts} "
              f"sharded={mesh_kind == 'blocks'} window_particles={[s['n_particles_local'] for s in stats]} "
              f"T0_tiles={[s['n_tiles'][0] for s in stats]} peeks={[s['readback_peeks'] for s in stats]}")
        return ok
    # RCCL refuses two ranks on one device: every rank a size-1 communicator of its own, every peer the rank itself (SB_DEBUG_LOOPBACK).
    # A self-exchange is not the physics of the partitioned mesh, but it IS the call pattern -- W communicators of one process, and in
    # walk mode every rank's sends and receives inside ONE ncclGroupStart / ncclGroupEnd -- and each rank's result must equal what the
    # same rank gives when stepped on its own through sb_step.
    # (cube on 4: the overlapped eager schedule; blocks: SB_SCHEDULE_AUTO, i.e. the first ticks alternate between the two eager schedules under
    # HIP events on every rank's own thread -- in walk mode AUTO stays the serialised schedule; else the serialised eager schedule)
    sched = native.SB_SCHEDULE_OVERLAP_EAGER if (mesh_kind == "cube" and world == 4) else (native.SB_SCHEDULE_AUTO if mesh_kind == "blocks" else native.SB_SCHEDULE_SERIAL_EAGER)
    n_ticks = 9 if mesh_kind == "blocks" else 3
    g = SoftbodyGroup(mesh, [0] * world, halo_transport=native.SB_TRANSPORT_RCCL, halo_schedule=sched, debug_flags=native.SB_DEBUG_LOOPBACK,
                      walk=walk, **kw).Start()
    L = native.lib()
    got = []
    try:
        for _ in range(n_ticks):
            g.step()
        g.synchronize()
        gathered = g.get_positions() if mesh_kind == "blocks" else None      # (sharded ranks number their own windows: read through the group)
        for r in range(world):
            h = g._rank_handle(r)
            out = np.zeros((mesh.n, 3), np.float32)
            if gathered is not None:
                out = gathered
            else:
                native.check(L.sb_get_positions(h, native.ptr(out), mesh.n))
            st = native.SbStats(); native.check(L.sb_get_stats(h, C.byref(st)))
            got.append((out, st.halo_schedule, st.halo_particles_t1, st.halo_auto_state))
    finally:
        g.OnDestroy()
    ok = True
    for r in range(world):
        sb = Softbody(mesh, device=0, rank=r, world=world, unique_id=comm_unique_id(), halo_schedule=sched, debug_flags=native.SB_DEBUG_LOOPBACK,
                      halo_transport=native.SB_TRANSPORT_RCCL, **kw).Start()
        try:
            for _ in range(n_ticks):
                sb.step()
            own = sb.owner() == r
            ok = ok and np.array_equal(bits(sb.get_positions()[own]), bits(got[r][0][own]))
            if sched != native.SB_SCHEDULE_AUTO:
                ok = ok and got[r][1] == sb.stats()["halo_schedule"]
            else:       # measured on every rank's thread and decided (threads); not measured at all when one thread walks the ranks
                ok = ok and got[r][3] == (0 if walk else 2)
            ok = ok and np.isfinite(got[r][0][own]).all() and got[r][2] > 0
        finally:
            sb.OnDestroy()
    print(("GROUP OK" if ok else "GROUP MISMATCH"), f"rccl-loopback world={world} mesh={mesh_kind} host={host} schedule={got[0][1]}")
    return ok


def features(world, host):
    """Kinematic pins, the peek, render readback with GPU normals (whole array and render set) through a group, against the oracle."""
    from readback_bench import surface_triangles
    n = 24
    mesh = jelly_cube(n)
    pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)
    mesh.inv_mass[pins] = 0.0
    rest = mesh.pos[pins].copy()
    tri = surface_triangles(n)
    tune = native.SbTuning(); native.lib().sb_tuning_default(C.byref(tune)); tune.peek_min_tiles = 0       # small launches peek too
    S = 8
    g = SoftbodyGroup(mesh, [0] * world, substeps=S, tile_particles=64, damping=0.05, halo_transport=native.SB_TRANSPORT_PEER,
                      walk=host == "walk", tuning=tune).Start()
    ok = True
    why = []
    try:
        o = make_oracle(oracle, mesh, build_plan(mesh, tile_particles=64), damping=0.05)
        g.set_render_triangles(tri)
        for t in range(10):
            compact = t >= 5
            if t == 5:
                g.set_readback_render_set_only(True)
            if t != 3:       # one tick without a move: the plain fused boundary comes back
                target = rest + np.array([0.3 * np.sin(0.4 * t), 0.1 * np.cos(0.7 * t) - 0.1, 0.05 * t], np.float32)
                g.set_kinematic_positions(pins, target); o.set_kinematic_positions(pins, target)
            if t in (2, 7):  # a blocking read between the move and the step: shows the pending targets, keeps them pending
                if not np.array_equal(bits(g.get_positions()), bits(o.x)):
                    ok = False; why.append(f"read after the move of tick {t}")
            g.step(); o.step(0.02, S)
            g.readback_begin()
            pos, nrm = (a.copy() for a in g.readback_end(normals=True))
            ref_n = oracle.vertex_normals(o.x, tri)
            if compact:
                ids = g.render_set()
                good = np.array_equal(ids, np.unique(tri)) and np.array_equal(bits(pos), bits(o.x[ids])) and np.array_equal(bits(nrm), bits(ref_n[ids]))
            else:
                good = np.array_equal(bits(pos), bits(o.x)) and np.array_equal(bits(nrm), bits(ref_n))
            if not good:
                ok = False; why.append(f"snapshot of tick {t} ({'render set' if compact else 'whole array'})")
        x, v = g.get_positions(), g.get_velocities()
        if not (np.array_equal(bits(x), bits(o.x)) and np.array_equal(bits(v), bits(o.v))):
            ok = False; why.append("final state")
        st = [g.rank(r).stats() for r in range(world)]
        peeks = sum(s["readback_peeks"] for s in st); fused = min(s["ticks_fused"] for s in st); kin = sum(s["ticks_fused_kinematic"] for s in st)
        # every snapshot and the reads between move and step were served without completing the tick, the targets travelled inside the
        # fused first kernel on the ranks that own pins
        if not (peeks >= 10 and fused >= 8 and kin >= 6):
            ok = False; why.append(f"peeks {peeks} fused {fused} kinematic-fused {kin}")
        try:       # an id twice, a free particle: refused, nothing changed
            g.set_kinematic_positions([int(pins[0]), int(pins[0])], np.zeros((2, 3), np.float32)); ok = False; why.append("duplicate id accepted")
        except native.SoftbodyError as e:
            ok = ok and "twice" in str(e)
        try:
            g.set_kinematic_positions([0], np.zeros((1, 3), np.float32)); ok = False; why.append("free particle accepted")
        except native.SoftbodyError as e:
            ok = ok and "non-zero inverse mass" in str(e)
    finally:
        g.OnDestroy()
    print(("GROUP OK" if ok else "GROUP MISMATCH " + "; ".join(why)), f"features world={world} host={host}")
    return ok


if __name__ == "__main__":
    kind = sys.argv[1]
    good = basic(int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]) if kind == "basic" else features(int(sys.argv[2]), sys.argv[3])
    sys.exit(0 if good else 1)
