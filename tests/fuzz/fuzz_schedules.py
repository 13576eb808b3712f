#!/usr/bin/env python3
"""Randomised equivalence test of the halo schedules and transports on ONE rank of a partitioned solver with a self-exchange
(SB_DEBUG_LOOPBACK: every peer is the rank itself -- RCCL refuses two ranks on one device). A self-exchange is not the physics of the
partitioned mesh (there the oracle is the checker: tests/fuzz/fuzz_parity.py, hosted ranks and sb_group_*), but it IS every launch, pack, send /
receive, unpack, event and graph of a rank's tick, and every (transport, schedule) pair must leave the SAME BITS as the serialised eager
schedule over RCCL -- reads and kinematic moves between ticks, ticks with their own dt and substeps, SB_SCHEDULE_AUTO's calibration
(which alternates the two eager schedules over its first ticks) included.
usage: python tests/fuzz/fuzz_schedules.py [--seconds 240] [--seed 0] [--only SEED] [--max N]"""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
import fuzz_parity as fz                                                # noqa: E402  (harness: parent / child runner, describe)

import numpy as np                                                      # noqa: E402
from softbodyunity_amd import Softbody, comm_unique_id, native          # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, jelly_cube          # noqa: E402

VARIANTS = [("rccl", "serial-graph", 0), ("rccl", "overlap-eager", 0), ("rccl", "overlap-graph", 0), ("rccl", "auto", 0),
            ("rccl", "auto", native.SB_TUNE_AUTO_PREFER_OVERLAP), ("peer", "serial-eager", 0), ("peer", "serial-graph", 0), ("peer", "overlap-eager", 0)]
SCHED = {"auto": native.SB_SCHEDULE_AUTO, "serial-eager": native.SB_SCHEDULE_SERIAL_EAGER, "serial-graph": native.SB_SCHEDULE_SERIAL_GRAPH,
         "overlap-eager": native.SB_SCHEDULE_OVERLAP_EAGER, "overlap-graph": native.SB_SCHEDULE_OVERLAP_GRAPH}


def make_scenario(seed):
    rng = np.random.default_rng(seed)
    sc = {"seed": seed}
    if rng.random() < 0.8:
        n = int(rng.integers(12, 41))
        pin = bool(rng.random() < 0.5)
        mesh = jelly_cube(n, pin_top=pin, heterogeneous=bool(rng.random() < 0.25), seed=int(rng.integers(1, 10 ** 6)))
        sc["mesh"] = f"cube {n}^3 pin_top={int(pin)}"
        comp = (float(rng.choice([0.0, 1e-7])), 0.0, 0.0)
    else:
        mesh = bunny_surrogate(target_verts=int(rng.choice([2000, 6000])), seed=int(rng.integers(1, 10 ** 6)))
        mesh.inv_mass[mesh.pos[:, 1] > np.quantile(mesh.pos[:, 1], 0.95)] = 0.0
        sc["mesh"] = f"tet blob ({mesh.n} particles): the overlapped schedules fall back to the serialised ones"
        comp = (1e-7, 1e-7, 1e-4)
    sc["_mesh"], sc["compliance"] = mesh, comp
    sc["world"] = int(rng.choice([2, 4, 8]))
    sc["rank"] = int(rng.integers(0, sc["world"]))
    sc["tile"] = int(rng.choice([0, 64, 128, 256]))
    sc["substeps"] = int(rng.choice([1, 2, 3, 4, 6, 8]))
    sc["ticks"] = int(rng.integers(2, 10))       # (SB_SCHEDULE_AUTO decides on its seventh tick)
    sc["graph"] = bool(rng.random() < 0.7)
    sc["damping"] = float(rng.choice([0.0, 0.05]))
    t = {"flags": [f for f in ("NO_FUSED_UNPACK", "NO_LAZY_TICK", "NO_PEEK", "NO_KIN_FUSE", "NO_COST_ORDER", "NO_LANE_PACK", "NO_WIDE_SLOTS") if rng.random() < 0.15]}
    t["peek_min_tiles"] = int(rng.choice([-1, 0, 0, 3]))
    sc["tuning"] = t
    k = int(rng.integers(2, 5))
    sc["variants"] = [VARIANTS[i] for i in sorted(rng.choice(len(VARIANTS), size=k, replace=False))]
    if any(v[1].endswith("graph") for v in sc["variants"]):
        sc["graph"] = True       # (a captured halo schedule needs use_graph = 1: sb_finalize says so)
    sc["_lattice"] = sc["mesh"].startswith("cube")
    vary = rng.random() < 0.3
    sc["_pins"] = np.nonzero(mesh.inv_mass == 0)[0].astype(np.int32)
    acts, per_tick = [], []
    for _ in range(sc["ticks"]):
        a = ""
        if len(sc["_pins"]) and rng.random() < 0.4:
            a += "k"
        if rng.random() < 0.3:
            a += "r"
        acts.append(a)
        per_tick.append((float(rng.choice([0.02, 0.01])), int(rng.choice([1, 2, 4, 5]))) if vary else (0.02, sc["substeps"]))
    sc["actions"], sc["per_tick"], sc["_per_tick"] = acts, (per_tick if vary else "fixed"), per_tick
    sc["_move"] = rng.uniform(-0.2, 0.2, (sc["ticks"], 3)).astype(np.float32)
    return sc


def one(sc, transport, schedule, extra_flags):
    mesh, comp = sc["_mesh"], sc["compliance"]
    tune = native.SbTuning(); native.lib().sb_tuning_default(C.byref(tune))
    for f in sc["tuning"]["flags"]:
        tune.flags |= getattr(native, "SB_TUNE_" + f)
    tune.flags |= extra_flags
    tune.peek_min_tiles = sc["tuning"]["peek_min_tiles"]
    sb = Softbody(mesh, substeps=sc["substeps"], tile_particles=sc["tile"], damping=sc["damping"], distance_compliance=comp[0], volume_compliance=comp[1],
                  bending_compliance=comp[2], use_graph=sc["graph"], device=0, rank=sc["rank"], world=sc["world"], unique_id=comm_unique_id(),
                  halo_transport=native.SB_TRANSPORT_PEER if transport == "peer" else native.SB_TRANSPORT_RCCL, halo_schedule=SCHED[schedule],
                  debug_flags=native.SB_DEBUG_LOOPBACK, tuning=tune).Start()
    try:
        own = sb.owner() == sc["rank"]
        pins = sc["_pins"][own[sc["_pins"]]]
        reads = []
        for t in range(sc["ticks"]):
            a = sc["actions"][t]
            if "k" in a and len(pins):
                sb.set_kinematic_positions(pins, mesh.pos[pins] + sc["_move"][t])
            if "r" in a:
                reads.append(sb.get_positions()[own].copy())
            sb.step(*sc["_per_tick"][t])
        x, v = sb.get_positions()[own].copy(), sb.get_velocities()[own].copy()
        st = sb.stats()
        val = sb.validate()
    finally:
        sb.OnDestroy()
    return reads, x, v, st, val


def run(sc):
    try:
        ref = one(sc, "rccl", "serial-eager", 0)
    except native.SoftbodyError as e:
        return ("REFUSED" if e.code == native.SB_ERR_UNSUPPORTED else "ERROR"), "reference: " + str(e)[:300]
    if not np.isfinite(ref[1]).all() or ref[3]["halo_particles_t1"] + ref[3]["halo_particles_global"] <= 0:
        return "ERROR", f"the reference run is not a partitioned rank with a finite state (halo particles {ref[3]['halo_particles_t1']})"
    why, ran = [], []
    # A lattice's ghost lists are symmetric (as many particles sent to a peer as received from it) and a self-exchange moves all of them, the
    # same way over either transport. An irregular mesh's are not: a self-exchange then moves min(sent, received) entries per peer and what
    # the others hold is a leftover of the transport's own buffers -- such a scenario compares the peer transport's schedules with the
    # peer transport's serialised eager schedule.
    ref_peer = None
    for transport, schedule, flags in sc["variants"]:
        name = f"{transport}/{schedule}" + ("+prefer-overlap" if flags else "")
        if transport == "peer" and not sc["_lattice"]:
            if ref_peer is None:
                try:
                    ref_peer = one(sc, "peer", "serial-eager", 0)
                except native.SoftbodyError as e:
                    why.append(f"peer/serial-eager (reference): {str(e)[:200]}"); break
            base = ref_peer
        else:
            base = ref
        try:
            got = one(sc, transport, schedule, flags)
        except native.SoftbodyError as e:
            if e.code == native.SB_ERR_UNSUPPORTED:
                ran.append(name + "=refused"); continue
            why.append(f"{name}: {str(e)[:200]}"); continue
        ran.append(f"{name}={got[3]['halo_schedule']}")
        if len(got[0]) != len(base[0]) or any(not np.array_equal(fz.bits(a), fz.bits(b)) for a, b in zip(got[0], base[0])):
            why.append(f"{name}: a read between ticks differs")
        if not np.array_equal(fz.bits(got[1]), fz.bits(base[1])):
            why.append(f"{name}: final positions ({int((fz.bits(got[1]) != fz.bits(base[1])).any(axis=1).sum())} of {len(base[1])} owned particles differ)")
        if not np.array_equal(fz.bits(got[2]), fz.bits(base[2])):
            why.append(f"{name}: final velocities")
        if got[4]["errors"] != [0] * 6:
            why.append(f"{name}: table validator {got[4]['errors']}")
    return ("MISMATCH", "; ".join(why)) if why else ("OK", "ran " + " ".join(ran))


if __name__ == "__main__":
    sys.exit(fz.main(make_scenario, run, __file__))
