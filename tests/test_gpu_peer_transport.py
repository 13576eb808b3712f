"""The opt-in peer-store halo transport (SB_HALO_TRANSPORT=peer): ghosts stored straight into the neighbours' mailboxes,
flags instead of ncclSend/ncclRecv. Unlike RCCL it CAN run between several ranks on one GPU -- the mailboxes of the other
processes are mapped through hipIpc handles -- so this is the one multi-rank test that goes through the real sb_step path
(eager launches and the captured hipGraph), ranks as separate processes, and still reproduces the oracle bit for bit.
Between two DEVICES it has never run (1-GPU box): DESIGN.md 7."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, mesh_kind, graph, out_dir):
    import ctypes as C

    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from softbodyunity_amd import Softbody, native
    from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube, jelly_cube_window
    if mesh_kind == "window":       # sharded authoring: this process hands over only its window of the 32^3 cube (sb_set_domain)
        mesh = jelly_cube_window(32, rank, world, (0, 0, 0), 64, pin_top=True)
    else:
        mesh = jelly_cube(24, pin_top=True) if mesh_kind == "cube" else bunny_surrogate(target_verts=3000, seed=9)
    comp = (1e-7, 1e-7, 1e-4) if mesh_kind == "bunny" else (0.0, 0.0, 0.0)
    # sb_desc.halo_transport / halo_schedule set explicitly (graph: the tick, exchange kernels included, captured in the hipGraph)
    sb = Softbody(mesh, substeps=6, device=0, rank=rank, world=world, tile_particles=128 if mesh_kind == "bunny" else 64,
                  distance_compliance=comp[0], volume_compliance=comp[1], bending_compliance=comp[2], halo_transport=native.SB_TRANSPORT_PEER,
                  halo_schedule=native.SB_SCHEDULE_SERIAL_GRAPH if graph else native.SB_SCHEDULE_SERIAL_EAGER).Start()   # no communicator: the host connects
    L = native.lib()
    mine = np.zeros(native.SB_IPC_HANDLE_BYTES, np.uint8)
    native.check(L.sb_peer_mailbox_handle(sb._h, native.ptr(mine)))
    handles = [torch.zeros(native.SB_IPC_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(handles, torch.from_numpy(mine))
    for r in range(world):
        if r != rank:
            h = handles[r].numpy().copy()
            native.check(L.sb_peer_connect(sb._h, r, native.ptr(h), None))
    dist.barrier()
    for _ in range(3):
        sb.step()
    x = sb.get_positions(); v = sb.get_velocities()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x, v=v, owned=sb.owner() == rank,
             gid=mesh.global_id if mesh_kind == "window" else np.arange(mesh.n),
             ghosts=np.array(sb.stats()["n_particles_local"] - sb.stats()["n_particles_owned"]))
    sb.synchronize()
    dist.barrier()              # nobody unmaps a mailbox a neighbour may still be writing to
    sb.OnDestroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mesh_kind,graph", [(2, "cube", False), (4, "cube", True), (3, "bunny", False), (2, "bunny", True),
                                                   (4, "window", False), (2, "window", True)])
def test_ranks_as_processes_through_the_peer_transport(tmp_path, oracle_mod, world, mesh_kind, graph):
    from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube
    from helpers import build_plan, make_oracle
    port = 29300 + (os.getpid() % 1500) + world * 11 + (5 if graph else 0)
    mp.spawn(_worker, args=(world, port, mesh_kind, graph, str(tmp_path)), nprocs=world, join=True)
    mesh = (jelly_cube(32, pin_top=True) if mesh_kind == "window" else jelly_cube(24, pin_top=True)) if mesh_kind != "bunny" else bunny_surrogate(target_verts=3000, seed=9)
    comp = (1e-7, 1e-7, 1e-4) if mesh_kind == "bunny" else (0.0, 0.0, 0.0)
    ref = make_oracle(oracle_mod, mesh, build_plan(mesh, tile_particles=128 if mesh_kind == "bunny" else 64), compliance=comp)
    for _ in range(3):
        ref.step(0.02, 6)
    x = np.zeros_like(ref.x); v = np.zeros_like(ref.v); cover = np.zeros(mesh.n, int); ghosts = 0
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        g = d["gid"][d["owned"]]
        x[g] = d["x"][d["owned"]]; v[g] = d["v"][d["owned"]]; cover[g] += 1; ghosts += int(d["ghosts"])
    assert np.all(cover == 1) and ghosts > 0
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32))
    assert np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))


def _worker_features(rank, world, port, out_dir):
    """A rank of a partitioned solver has what a single rank has: kinematic targets (it applies the pins it owns), position reads and a
    render-set readback that PEEK while its tick's last kernel is held back -- through the real tick path, ranks as processes."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from readback_bench import surface_triangles
    from softbodyunity_amd import Softbody, native
    from softbodyunity_amd.mesh import jelly_cube
    n = 24
    mesh = jelly_cube(n)
    pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)
    mesh.inv_mass[pins] = 0.0
    rest = mesh.pos[pins].copy()
    tune = native.SbTuning(); native.lib().sb_tuning_default(C.byref(tune)); tune.peek_min_tiles = 0
    sb = Softbody(mesh, substeps=6, device=0, rank=rank, world=world, tile_particles=64, damping=0.05, halo_transport=native.SB_TRANSPORT_PEER,
                  halo_schedule=native.SB_SCHEDULE_SERIAL_EAGER, tuning=tune).Start()
    L = native.lib()
    mine = np.zeros(native.SB_IPC_HANDLE_BYTES, np.uint8)
    native.check(L.sb_peer_mailbox_handle(sb._h, native.ptr(mine)))
    handles = [torch.zeros(native.SB_IPC_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(handles, torch.from_numpy(mine))
    for r in range(world):
        if r != rank:
            h = handles[r].numpy().copy()
            native.check(L.sb_peer_connect(sb._h, r, native.ptr(h), None))
    dist.barrier()
    sb.set_render_triangles(surface_triangles(n))
    sb.set_readback_render_set_only(True)
    owned = sb.owner() == rank
    snaps, reads = [], []
    for t in range(6):
        target = rest + np.array([0.3 * np.sin(0.4 * t), 0.1 * np.cos(0.7 * t) - 0.1, 0.05 * t], np.float32)
        sb.set_kinematic_positions(pins, target)                 # the whole list on every rank
        if t & 1:
            reads.append(sb.get_positions()[owned].copy())       # between the move and the step
        sb.step()
        sb.readback_begin()
        snaps.append(sb.readback_end().copy())
    ids = sb.render_set().copy()
    st = sb.stats()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=sb.get_positions(), v=sb.get_velocities(), owned=owned, ids=ids,
             snaps=np.stack(snaps), reads=np.stack(reads), peeks=np.array(st["readback_peeks"]), fused=np.array(st["ticks_fused"]),
             kin=np.array(st["ticks_fused_kinematic"]))
    try:
        sb.readback_begin(); sb.readback_end(normals=True)
        refused = False
    except native.SoftbodyError as e:
        refused = "sb_group_readback_get_normals" in str(e)
    assert refused, "per-rank vertex normals of a partitioned solver must be refused with a pointer to the group API"
    sb.synchronize()
    dist.barrier()
    sb.OnDestroy()
    dist.destroy_process_group()


def test_ranks_as_processes_kinematic_peek_and_render_set(tmp_path, oracle_mod):
    from softbodyunity_amd.mesh import jelly_cube
    from helpers import build_plan, make_oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from readback_bench import surface_triangles
    world = 4
    port = 29300 + (os.getpid() % 1500) + 977
    mp.spawn(_worker_features, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n = 24
    mesh = jelly_cube(n)
    pins = np.nonzero(mesh.pos[:, 1] > mesh.pos[:, 1].max() - 0.5)[0].astype(np.int32)
    mesh.inv_mass[pins] = 0.0
    rest = mesh.pos[pins].copy()
    surf = np.unique(surface_triangles(n))
    ref = make_oracle(oracle_mod, mesh, build_plan(mesh, tile_particles=64), damping=0.05)
    d = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    seen = np.zeros(mesh.n, int)
    for r in range(world):       # a rank's render set = the surface particles it owns
        assert np.array_equal(d[r]["ids"], surf[d[r]["owned"][surf]])
        seen[d[r]["ids"]] += 1
    assert np.array_equal(np.nonzero(seen)[0], surf) and seen.max() == 1
    k_read = 0
    for t in range(6):
        target = rest + np.array([0.3 * np.sin(0.4 * t), 0.1 * np.cos(0.7 * t) - 0.1, 0.05 * t], np.float32)
        ref.set_kinematic_positions(pins, target)
        if t & 1:
            for r in range(world):
                assert np.array_equal(d[r]["reads"][k_read].view(np.uint32), ref.x[d[r]["owned"]].view(np.uint32)), f"read before tick {t}, rank {r}"
            k_read += 1
        ref.step(0.02, 6)
        for r in range(world):
            assert np.array_equal(d[r]["snaps"][t].view(np.uint32), ref.x[d[r]["ids"]].view(np.uint32)), f"render-set snapshot of tick {t}, rank {r}"
    x = np.zeros_like(ref.x); v = np.zeros_like(ref.v)
    for r in range(world):
        x[d[r]["owned"]] = d[r]["x"][d[r]["owned"]]; v[d[r]["owned"]] = d[r]["v"][d[r]["owned"]]
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))
    # nothing completed a tick early: every snapshot and read peeked, every tick after the first fused with the one before, and the
    # ranks that own pins took their targets inside the fused kernel
    assert all(int(q["peeks"]) >= 8 for q in d) and all(int(q["fused"]) >= 5 for q in d) and sum(int(q["kin"]) for q in d) >= 5


def test_peer_transport_loopback_schedules_agree():
    # rank 0's share with every neighbour replaced by itself: serialised / overlapped x eager / captured, one process each
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lb_combo_check.py"), "peer"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("peer overlap=")]
    assert len(lines) == 4 and all("rc=0 HASH" in l and " True " in l for l in lines), out.stdout[-1500:]
    assert "peer all equal: True" in out.stdout, out.stdout[-1500:]


_ONE_PROCESS = r'''
import os, sys
import numpy as np
root = sys.argv[1]; world = int(sys.argv[2]); mesh_kind = sys.argv[3]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle import oracle
from softbodyunity_amd import Softbody, native
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube
from helpers import build_plan, make_oracle
mesh = jelly_cube(24, pin_top=True) if mesh_kind == "cube" else bunny_surrogate(target_verts=3000, seed=9)
comp = (0.0, 0.0, 0.0) if mesh_kind == "cube" else (1e-7, 1e-7, 1e-4)
tile = 64 if mesh_kind == "cube" else 128
L = native.lib()
ranks = [Softbody(mesh, substeps=6, device=0, rank=r, world=world, tile_particles=tile, distance_compliance=comp[0],
                  volume_compliance=comp[1], bending_compliance=comp[2]).Start() for r in range(world)]
for a in range(world):
    for b in range(world):
        if a != b:
            native.check(L.sb_peer_connect(ranks[a]._h, b, None, ranks[b]._h))
for _ in range(3):
    for sb in ranks:
        sb.step()                      # asynchronous: the host never waits for a neighbour
x = np.zeros((mesh.n, 3), np.float32); v = np.zeros_like(x); cover = np.zeros(mesh.n, int)
for r, sb in enumerate(ranks):
    own = sb.owner() == r
    xr = sb.get_positions(); vr = sb.get_velocities()
    x[own] = xr[own]; v[own] = vr[own]; cover += own
for sb in ranks:
    sb.synchronize()
for sb in ranks:
    sb.OnDestroy()
ref = make_oracle(oracle, mesh, build_plan(mesh, tile_particles=tile), compliance=comp)
for _ in range(3):
    ref.step(0.02, 6)
ok = bool(np.all(cover == 1)) and np.array_equal(x.view(np.uint32), ref.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))
print("ONE PROCESS OK" if ok else "ONE PROCESS MISMATCH")
'''


@pytest.mark.parametrize("world,mesh_kind", [(4, "cube"), (3, "bunny")])
def test_one_process_drives_every_rank(world, mesh_kind):
    # What a Unity player would do with several GPUs: ONE process (one thread) owns a handle per rank, connects the mailboxes
    # by pointer (sb_peer_connect with the peer's solver) and calls sb_step on the handles one after the other -- the
    # exchange is kernels and flags only, so nothing blocks on the host. Here every rank sits on the ONE device, which needs
    # one hardware queue per rank (GPU_MAX_HW_QUEUES: HIP multiplexes a process' streams onto 4 queues per device by default,
    # and a waiting kernel holds up the streams behind it in its queue); one rank per device needs nothing of the kind.
    env = dict(os.environ, SB_HALO_TRANSPORT="peer", GPU_MAX_HW_QUEUES="16")
    out = subprocess.run([sys.executable, "-c", _ONE_PROCESS, ROOT, str(world), mesh_kind], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ONE PROCESS OK" in out.stdout, out.stdout[-1500:] + out.stderr[-2500:]
