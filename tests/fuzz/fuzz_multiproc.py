#!/usr/bin/env python3
"""Randomised differential test of the ONE-PROCESS-PER-RANK path -- what `bench.py --gpus N` runs, with the peer-store transport standing in for
RCCL (which refuses several ranks on one device): W processes on the box's one GPU, every rank its own solver (sb_desc.rank / world), mailboxes
connected through hipIpc handles that travel over gloo, real ghost exchanges inside sb_step (eager or captured). Every rank draws the same
scenario from the same seed; rank 0 gathers the owned entries and compares them BITWISE with the CPU oracle (test infrastructure: the checker).

What varies: the mesh (cube -- whole on every rank, or only the rank's WINDOW under sharded authoring --, tet blob, cloth), partition
(AUTO / BLOCKS / RCB), tile size, substeps and dt (per tick where the scenario varies them), compliances, damping, ground plane, tuning
switches, the halo schedule (serialised eager / captured), and between ticks: set_state, plane changes, kinematic moves of the pins a rank owns,
blocking reads (a rank peeks or flushes by its own tile count), one invalid call.

usage (GPU box): GPU_MAX_HW_QUEUES=16 python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \\
                 tests/fuzz/fuzz_multiproc.py [--seconds 200] [--seed 0] [--max N]"""
import argparse
import ctypes as C
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))

import numpy as np                                                      # noqa: E402
import torch                                                            # noqa: E402
import torch.distributed as dist                                        # noqa: E402

import fuzz_parity as fz                                                # noqa: E402  (scenario generator, oracle mirror pieces)
from softbodyunity_amd import Softbody, native                          # noqa: E402
from softbodyunity_amd.mesh import jelly_cube_window                    # noqa: E402


def all_ok(ok, msg=""):
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, (bool(ok), msg))
    return all(g[0] for g in got), "; ".join(f"rank {r}: {g[1]}" for r, g in enumerate(got) if not g[0])


def scenario(seed, world):
    """fuzz_parity's scenario, bent to this host model: one process per rank, no render readback, cubes sometimes as windows."""
    sc = fz.make_scenario(seed)
    rng = np.random.default_rng(seed ^ 0x5eed)
    sc["host"], sc["world"] = "processes", world
    sc["partition"] = str(rng.choice(["auto", "blocks", "rcb"]))
    sc["schedule"] = str(rng.choice(["serial-eager", "serial-graph"]))
    if sc["schedule"] == "serial-graph":
        sc["graph"] = True
    if sc["tile"] == -1:
        sc["tile"] = 0
    sc["render"] = "none"
    sc["actions"] = [a.replace("n", "b").replace("p", "b") for a in sc["actions"]]      # (a rank's render readback: the particles it owns, no triangles here)
    sc["pipelined"] = False
    m = sc["mesh"]
    sc["window"] = bool(sc["kind"] == "cube" and "stencil=structural" in m and sc["tile"] > 0 and sc["partition"] != "rcb" and rng.random() < 0.5)
    for k in ("walk", "whole_mesh"):
        sc.pop(k, None)
    return sc


def run_rank(sc, rank, world):
    """This rank's part of the scenario -> (reads, x, v, owned mask, caller ids of the local numbering, validator report, bad-call findings)."""
    mesh, comp = sc["_mesh"], sc["compliance"]
    gid = np.arange(mesh.n)
    if sc["window"]:       # sharded authoring: this process hands over only its window of the cube (positions, masses, springs: the whole cube's)
        n, het, pin, cube_seed = sc["_cube"]
        mesh = jelly_cube_window(n, rank, world, (0, 0, 0), sc["tile"], pin_top=pin, heterogeneous=het, seed=cube_seed)
        gid = mesh.global_id.astype(np.int64)
    sb = Softbody(mesh, substeps=sc["substeps"], fixed_delta_time=sc["dt"], tile_particles=sc["tile"], damping=sc["damping"], distance_compliance=comp[0],
                  volume_compliance=comp[1], bending_compliance=comp[2], ground_plane=sc["plane"], use_graph=sc["graph"], tuning=fz.make_tuning(sc["tuning"]), gravity=sc["gravity"],
                  device=0, rank=rank, world=world, partition=native.SB_PARTITION_BLOCKS if sc["window"] else fz.PART[sc["partition"]],
                  halo_transport=native.SB_TRANSPORT_PEER,
                  halo_schedule=native.SB_SCHEDULE_SERIAL_GRAPH if sc["schedule"] == "serial-graph" else native.SB_SCHEDULE_SERIAL_EAGER)
    started, err = False, ""
    try:
        sb.Start(); started = True
    except native.SoftbodyError as e:
        err = str(e)[:200]
    ok, msg = all_ok(started, err)
    if not ok:
        if started:
            sb.OnDestroy()
        return None, msg
    L = native.lib()
    result, err = None, ""
    try:
        mine = np.zeros(native.SB_IPC_HANDLE_BYTES, np.uint8)
        native.check(L.sb_peer_mailbox_handle(sb._h, native.ptr(mine)))
        handles = [torch.zeros(native.SB_IPC_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(handles, torch.from_numpy(mine))
        for r in range(world):
            if r != rank:
                h = handles[r].numpy().copy()
                native.check(L.sb_peer_connect(sb._h, r, native.ptr(h), None))
        dist.barrier()
        own = sb.owner() == rank
        whole = sc["_mesh"]
        pins_local = np.nonzero((mesh.inv_mass == 0) & own)[0].astype(np.int32)
        # (a host need not sort the pins by owner: a rank skips the ids it does not own -- half of the scenarios hand EVERY pin the rank can see)
        pins_given = np.nonzero(mesh.inv_mass == 0)[0].astype(np.int32) if sc["seed"] % 2 else pins_local
        reads, bad = [], []
        for t in range(sc["ticks"]):
            acts = sc["actions"][t]
            if "x" in acts:
                b = fz.bad_call(dict(sc, _mesh=mesh, _pins=pins_local), sb, False, sc["_bad"][t])
                if b:
                    bad.append(b)
            if "s" in acts:
                sb.set_state(sc["_state"][t][0][gid], sc["_state"][t][1][gid])
            if "g" in acts:
                pl = sc["_planes"][t]
                native.check(L.sb_set_ground_plane(sb._h, *[float(c) for c in pl[:4]], int(pl[4])))
            if "k" in acts and len(pins_given):
                sb.set_kinematic_positions(pins_given, whole.pos[gid[pins_given]] + sc["_move"][t])
            if "r" in acts:
                reads.append(sb.get_positions()[own].copy())
            sb.step(*sc["_per_tick"][t])
            if "b" in acts:
                sb.readback_begin()
                reads.append(np.array(sb.readback_end(), copy=True)[own])
        x, v = sb.get_positions()[own].copy(), sb.get_velocities()[own].copy()
        result = dict(reads=reads, x=x, v=v, ids=gid[own], val=sb.validate()["errors"], bad=bad,
                      ghosts=sb.stats()["n_particles_local"] - sb.stats()["n_particles_owned"], peeks=sb.stats()["readback_peeks"])
    except native.SoftbodyError as e:
        err = str(e)[:300]
    ok, msg = all_ok(result is not None, err)
    try:
        sb.synchronize()
    except native.SoftbodyError:
        pass
    dist.barrier()              # nobody unmaps a mailbox a neighbour may still be writing to
    sb.OnDestroy()
    return (result, "") if ok else (None, msg)


def check(sc, parts):
    """Rank 0: the gathered parts against the oracle walking a plan of its own."""
    from oracle import oracle
    from helpers import build_plan, make_oracle
    mesh = sc["_mesh"]
    why = []
    cover = np.zeros(mesh.n, np.int32)
    for p in parts:
        cover[p["ids"]] += 1
        why += p["bad"]
        if p["val"] != [0] * 6:
            why.append(f"table validator {p['val']}")
    if not np.all(cover == 1):
        return ["the ranks' owned sets do not partition the particles"]
    o = make_oracle(oracle, mesh, build_plan(mesh, tile_particles=sc["tile"]), gravity=sc["gravity"], damping=sc["damping"], compliance=sc["compliance"], ground_plane=sc["plane"])
    k = 0
    for t in range(sc["ticks"]):
        acts = sc["actions"][t]
        if "s" in acts:
            o.x[:] = sc["_state"][t][0]; o.v[:] = sc["_state"][t][1]
        if "g" in acts:
            pl = sc["_planes"][t]
            o.set_ground_plane(pl[:3], pl[3], bool(pl[4]))
        if "k" in acts and len(sc["_pins"]):
            o.set_kinematic_positions(sc["_pins"], mesh.pos[sc["_pins"]] + sc["_move"][t])
        if "r" in acts:
            for r, p in enumerate(parts):
                if not fz.same(p["reads"][k], o.x[p["ids"]]):
                    why.append(f"rank {r}: read before tick {t}")
            k += 1
        o.step(*sc["_per_tick"][t])
        if "b" in acts:
            for r, p in enumerate(parts):
                if not fz.same(p["reads"][k], o.x[p["ids"]]):
                    why.append(f"rank {r}: render readback after tick {t}")
            k += 1
    for r, p in enumerate(parts):
        if not fz.same(p["x"], o.x[p["ids"]]):
            why.append(f"rank {r}: final positions")
        if not fz.same(p["v"], o.v[p["ids"]]):
            why.append(f"rank {r}: final velocities")
    return why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=200.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max", type=int, default=10 ** 9)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    t_end = time.time() + a.seconds
    tally, n = {}, 0
    for seed in range(a.seed, a.seed + a.max):
        go = torch.tensor([1 if time.time() < t_end else 0])
        dist.broadcast(go, src=0)
        if not int(go.item()):
            break
        sc = scenario(seed, world)
        t0 = time.time()
        res, msg = run_rank(sc, rank, world)
        parts = [None] * world
        dist.gather_object(res, parts if rank == 0 else None, dst=0)
        if rank == 0:
            if res is None:
                verdict, detail = ("REFUSED" if "error -7" in msg else "ERROR"), msg
            else:
                why = check(sc, parts)
                verdict, detail = ("MISMATCH", "; ".join(why)) if why else ("OK", f"ghosts {[p['ghosts'] for p in parts]} peeks {[p['peeks'] for p in parts]}")
            tally[verdict] = tally.get(verdict, 0) + 1; n += 1
            print(f"{verdict:8s} {time.time() - t0:5.1f}s  {fz.describe(sc)}\n         -> {detail}", flush=True)
    if rank == 0:
        print(f"SUMMARY {n} scenarios on {world} processes: " + ", ".join(f"{k} {v}" for k, v in sorted(tally.items())), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
