"""N>1 path on CPU: two real processes (gloo), each plans its own rank with the plugin's host code,
runs the oracle on its partition and exchanges ghosts through torch.distributed using the plugin's
halo schedule. Merged result must equal the unpartitioned oracle bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tile, out_dir, sharded=False, tet=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from softbodyunity_amd.mesh import jelly_cube, jelly_cube_window
    from helpers import RankSim, WindowRankSim, run_tick
    if sharded:      # sharded authoring: this process only ever sees its window of the 32^3 cube
        win = jelly_cube_window(32, rank, world, (0, 0, 0), tile, pin_top=True)
        R = WindowRankSim(oracle, win, rank, world, (0, 0, 0), tile)
        R.o.params.compliance[0] = 1e-7
    elif tet:        # irregular tet mesh (springs + volumes + hinges), cost-weighted RCB ownership, T2 layers with their own exchanges
        from softbodyunity_amd import native
        from softbodyunity_amd.mesh import bunny_surrogate
        mesh = bunny_surrogate(3000)
        R = RankSim(oracle, mesh, rank, world, (0, 0, 0), tile, (0.0, -9.81, 0.0), 0.0, (1e-7, 0.0, 1e-6), partition=native.SB_PARTITION_RCB)
    else:
        mesh = jelly_cube(16, pin_top=True)
        R = RankSim(oracle, mesh, rank, world, (0, 0, 0), tile, (0.0, -9.81, 0.0), 0.0, (1e-7, 0.0, 0.0))
    S, dt = 6, 0.02

    def exchange(slot, with_prev):
        if slot >= len(R.halos):
            return
        ops, recvs = [], []
        for peer, (send_ids, recv_ids) in sorted(R.halos[slot].items()):
            arrs = [R.o.x] + ([R.o.xprev] if with_prev else [])
            for a in arrs:
                if len(send_ids):
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(a[send_ids])), peer))
                if len(recv_ids):
                    buf = torch.empty((len(recv_ids), 3), dtype=torch.float32)
                    recvs.append((a, recv_ids, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for a, ids, buf in recvs:
            a[ids] = buf.numpy()

    for _ in range(2):
        s = R.o.scalars(dt, S)
        run_tick([R], s, S, tile > 0, exchange)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=R.o.x, v=R.o.v, owned=R.owned, gid=getattr(R, "gid", np.arange(len(R.owned))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile", [64, -1])
def test_two_ranks_gloo_equals_single(tmp_path, oracle_mod, tile):
    from softbodyunity_amd.mesh import jelly_cube
    from helpers import build_plan, make_oracle
    world = 2
    port = 29500 + (os.getpid() % 2000) + (0 if tile > 0 else 1)
    mp.spawn(_worker, args=(world, port, tile, str(tmp_path)), nprocs=world, join=True)
    mesh = jelly_cube(16, pin_top=True)
    ref = make_oracle(oracle_mod, mesh, build_plan(mesh, tile_particles=tile), compliance=(1e-7, 0.0, 0.0))
    for _ in range(2):
        ref.step(0.02, 6)
    x = np.zeros_like(ref.x); v = np.zeros_like(ref.v); cover = np.zeros(mesh.n, int)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        x[d["owned"]] = d["x"][d["owned"]]; v[d["owned"]] = d["v"][d["owned"]]; cover += d["owned"]
    assert np.all(cover == 1)
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32))
    assert np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))


def test_four_ranks_gloo_on_their_windows_equal_single(tmp_path, oracle_mod):
    # sharded authoring end to end over a real process boundary: no process holds the whole 32^3 cube
    from softbodyunity_amd.mesh import jelly_cube
    from helpers import build_plan, make_oracle
    world, tile = 4, 64
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_worker, args=(world, port, tile, str(tmp_path), True), nprocs=world, join=True)
    mesh = jelly_cube(32, pin_top=True)
    ref = make_oracle(oracle_mod, mesh, build_plan(mesh, tile_particles=tile), compliance=(1e-7, 0.0, 0.0))
    for _ in range(2):
        ref.step(0.02, 6)
    x = np.zeros_like(ref.x); v = np.zeros_like(ref.v); cover = np.zeros(mesh.n, int)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        g = d["gid"][d["owned"]]
        x[g] = d["x"][d["owned"]]; v[g] = d["v"][d["owned"]]; cover[g] += 1
    assert np.all(cover == 1)
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))


def test_three_ranks_gloo_rcb_tet_mesh_equal_single(tmp_path, oracle_mod):
    # the irregular-mesh partition (cost-weighted RCB, an odd rank count) over a real process boundary: springs, volumes and hinges,
    # T0/T1 tiles plus the sparse T2 layers and their own ghost exchanges
    from softbodyunity_amd import native
    from softbodyunity_amd.mesh import bunny_surrogate
    from helpers import build_plan, make_oracle
    world, tile = 3, 128
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(_worker, args=(world, port, tile, str(tmp_path), False, True), nprocs=world, join=True)
    mesh = bunny_surrogate(3000)
    ref = make_oracle(oracle_mod, mesh, build_plan(mesh, tile_particles=tile), compliance=(1e-7, 0.0, 1e-6))
    for _ in range(2):
        ref.step(0.02, 6)
    x = np.zeros_like(ref.x); v = np.zeros_like(ref.v); cover = np.zeros(mesh.n, int)
    owned = []
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        x[d["owned"]] = d["x"][d["owned"]]; v[d["owned"]] = d["v"][d["owned"]]; cover += d["owned"]; owned.append(int(d["owned"].sum()))
    assert np.all(cover == 1) and min(owned) > 0
    assert np.array_equal(x.view(np.uint32), ref.x.view(np.uint32)) and np.array_equal(v.view(np.uint32), ref.v.view(np.uint32))
