#!/usr/bin/env python3
"""Randomised differential test: random meshes, settings, tuning switches, host models and mid-run host actions through the plugin, every
result compared BITWISE with the CPU oracle walking the plan the plugin published (test infrastructure: the oracle is the checker).

What varies per scenario (all from one seed, printed, so a failure replays with --only SEED):
  mesh      lattice cube (structural springs; optionally every mass / rest length its own, a pinned top layer), tet blob with volume and
            bending constraints, cloth sheet with bending constraints
  settings  substeps, dt, tile size (incl. automatic and no tiling), compliances, damping, ground plane, hipGraph replay
  tuning    a random subset of the sb_tuning switches and widths (include/softbody_debug.h: "every setting gives the same bits")
  host      one solver | W ranks hosted in this process with the host as the wire (partition AUTO / BLOCKS / RCB) |
            sb_group_* over the peer transport on one device (thread per rank or walked by the calling thread)
  actions   between ticks: set_state, the ground plane moved or switched, kinematic moves of pinned particles, a blocking position read (the
            peek / the flush), a tick with its own dt and substeps, a pipelined render readback (whole array, with GPU normals, render set
            only) -- each mirrored on the oracle
Every scenario also runs the table validator. usage: python tests/fuzz/fuzz_parity.py [--seconds 240] [--seed 0] [--only SEED] [--max N]"""
import argparse
import ctypes as C
import os
import sys
import time
import traceback

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")        # (several ranks of one process on one device: a hardware queue each, before the first HIP call)
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))

import numpy as np                                                      # noqa: E402

from oracle import oracle                                               # noqa: E402  (the checker)
from helpers import build_plan, make_oracle                             # noqa: E402
from hosted import HostedRanks                                          # noqa: E402
from softbodyunity_amd import Softbody, SoftbodyGroup, native           # noqa: E402
from softbodyunity_amd.mesh import bunny_surrogate, from_triangle_mesh, jelly_cube   # noqa: E402
from readback_bench import surface_triangles                            # noqa: E402

# switches that change kernel selection or table layout, never results (AUTO_* / PEER_COARSE concern transports that do not run here)
ODD = os.environ.get("FUZZ_ODD", "1") != "0"     # include the degenerate meshes of tests/fuzz/fuzz_plan.py
FLAGS = ["NO_MASS_PALETTE", "NO_UNIFORM_MASS", "NO_PALETTE", "NO_WAVE_ITEMS", "NO_LANE_PACK", "NO_COST_ORDER", "NO_FUSED_UNPACK", "NO_LAZY_TICK",
         "NO_PACK", "NO_PEEK", "NO_KIN_FUSE", "NO_WIDE_SLOTS"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same(got, want):
    """Bitwise -- except where the oracle itself overflowed (coordinates of 1e29: squares beyond the float range): there the plugin must be
    non-finite in the same places (which NaN an overflow leaves is the one thing x86 and gfx950 need not agree on) and bitwise everywhere else."""
    if got.shape != want.shape:
        return False
    fin = np.isfinite(want)
    if fin.all():
        return np.array_equal(bits(got), bits(want))
    return np.array_equal(np.isfinite(got), fin) and np.array_equal(bits(got)[fin], bits(want)[fin])


def grid_cloth(a, b):
    xs, ys = np.meshgrid(np.arange(a), np.arange(b), indexing="ij")
    V = np.stack([xs.ravel(), np.zeros(a * b), ys.ravel()], axis=1).astype(np.float32)
    F = []
    for i in range(a - 1):
        for j in range(b - 1):
            p, q, r, s = i * b + j, (i + 1) * b + j, (i + 1) * b + j + 1, i * b + j + 1
            F += [(p, q, r), (p, r, s)]
    return V, np.array(F, np.int32)


def make_scenario(seed):
    rng = np.random.default_rng(seed)
    sc = {"seed": seed}
    kind = rng.choice(["cube", "cube", "tets", "cloth"] + (["odd"] if ODD else []))
    sc["kind"] = str(kind)
    if kind == "odd":       # meshes nobody would author on purpose (tests/fuzz/fuzz_plan.py: a point, a line, a plane, a chain, a complete graph, a hub,
        import fuzz_plan    # isolated particles, no constraints, one or two particles, duplicate constraints, extreme scales)
        while True:
            o = fuzz_plan.make_scenario(int(rng.integers(0, 2 ** 31)))
            if o["shape"] not in ("nan", "inf"):
                break
        mesh = o["_mesh"]
        mesh.pos = (mesh.pos + rng.normal(0, 0.01, mesh.pos.shape).astype(np.float32) * np.float32(np.abs(mesh.pos).max() > 0)).astype(np.float32)
        if rng.random() < 0.4:
            mesh.inv_mass[rng.random(mesh.n) < 0.1] = 0.0
        sc["mesh"] = f"odd: {o['shape']} n={o['n']} springs={o['springs']} tets={o['tets']} hinges={o['hinges']}"
        comp = (float(rng.choice([0.0, 1e-6])), float(rng.choice([0.0, 1e-6])), float(rng.choice([0.0, 1e-4])))
    elif kind == "cube":
        n = int(rng.integers(3, 34)) if rng.random() < 0.9 else int(rng.integers(40, 73))      # (the larger ones reach the 8-byte wide-packed slots: > 768 tiles)
        het, pin = bool(rng.random() < 0.3), bool(rng.random() < 0.5)
        full = bool(n <= 12 and rng.random() < 0.25)       # 26-neighbour stencil: many colours, global colours, leftover layers
        cube_seed = int(rng.integers(1, 10 ** 6))
        mesh = jelly_cube(n, pin_top=pin, heterogeneous=het, seed=cube_seed, stencil="full" if full else "structural")
        sc["_cube"] = (n, het, pin, cube_seed)
        sc["mesh"] = f"cube {n}^3 het={int(het)} pin_top={int(pin)} stencil={'full' if full else 'structural'}"
        sc["_tri"] = surface_triangles(n)
        comp = (float(rng.choice([0.0, 0.0, 1e-7, 1e-5])), 0.0, 0.0)
    elif kind == "tets":
        tv = int(rng.choice([300, 800, 2000, 5000, 12000]))
        mesh = bunny_surrogate(target_verts=tv, seed=int(rng.integers(1, 10 ** 6)))
        sc["mesh"] = f"tet blob ~{tv} vertices ({mesh.n} particles, {len(mesh.vol_rest)} tets, {len(mesh.bend_rest)} hinges)"
        comp = (float(rng.choice([0.0, 1e-7])), float(rng.choice([0.0, 1e-7])), float(rng.choice([1e-5, 1e-4, 1e-3])))
        if rng.random() < 0.5:       # pins: the particles highest up
            mesh.inv_mass[mesh.pos[:, 1] > np.quantile(mesh.pos[:, 1], 0.97)] = 0.0
    else:
        a, b = int(rng.integers(4, 50)), int(rng.integers(4, 50))
        V, F = grid_cloth(a, b)
        mesh, pov = from_triangle_mesh(V, F)
        sc["_tri"] = np.ascontiguousarray(pov[F], dtype=np.int32)
        mesh.inv_mass[mesh.rest_pos[:, 0] == 0] = 0.0
        sc["mesh"] = f"cloth {a}x{b} ({mesh.n} particles, {len(mesh.bend_rest)} hinges)"
        comp = (float(rng.choice([0.0, 1e-7])), 0.0, float(rng.choice([1e-4, 1e-3, 1e-2])))
    sc["_mesh"] = mesh
    sc["compliance"] = comp
    sc["substeps"] = int(rng.choice([1, 2, 3, 4, 5, 6, 8]))
    sc["ticks"] = int(rng.integers(1, 5))
    sc["dt"] = float(rng.choice([0.02, 0.01])) if rng.random() < 0.9 else float(rng.choice([0.1, 0.001, 1.0 / 60.0]))
    sc["tile"] = int(rng.choice([0, 0, 64, 128, 256, 512, -1])) if rng.random() < 0.85 else int(rng.choice([16, 48, 100, 300, 700, 1000]))
    sc["gravity"] = (0.0, -9.81, 0.0) if rng.random() < 0.7 else tuple(float(c) for c in rng.choice([(0.0, 0.0, 0.0), (3.0, -4.0, 1.5), (0.0, 9.81, 0.0), (0.0, -100.0, 0.0)]))
    sc["damping"] = float(rng.choice([0.0, 0.0, 0.02, 0.5]))
    ymin = float(mesh.pos[:, 1].min())
    sc["plane"] = (0.0, 1.0, 0.0, ymin - float(rng.choice([0.0, 0.05, 0.5]))) if rng.random() < 0.4 else None
    sc["graph"] = bool(rng.random() < 0.5)
    # tuning
    t = {"flags": [f for f in FLAGS if rng.random() < 0.2]}
    t["tile_lanes"] = int(rng.choice([0, 0, 0, 128, 256, 512]))
    t["quad_lanes"] = int(rng.choice([0, 0, 256]))
    t["narrow_min_tiles"] = int(rng.choice([0, 0, 1, 40]))
    t["store_through_max_tiles"] = int(rng.choice([-1, -1, 0, 10 ** 6]))
    t["store_through_large"] = int(rng.choice([0, 0, 1, 2, 3]))
    t["peek_min_tiles"] = int(rng.choice([-1, 0, 0, 4]))
    t["win_dwords"] = int(rng.choice([0, 0, 1024, 2048]))
    sc["tuning"] = t
    # host model
    host = str(rng.choice(["single", "single", "hosted", "group"]))
    sc["host"] = host
    if host != "single":
        sc["world"] = int(rng.choice([2, 3, 4, 8, 2, 3, 4, 8, 5, 6, 7, 12, 16] if host == "hosted" else [2, 3, 4, 2, 3, 4, 5, 6, 8]))      # (more ranks than cells happens: empty ranks)
        sc["partition"] = str(rng.choice(["auto", "blocks", "rcb"]))
        sc["walk"] = bool(rng.random() < 0.5)
        sc["whole_mesh"] = bool(rng.random() < 0.3)
        if sc["tile"] == -1:
            sc["tile"] = 0
    # host actions per tick, in this order around the tick's step:
    #   x = ONE INVALID CALL (bad_call: must be refused with an error code and leave every later result untouched),
    #   s = set_state (fresh positions and velocities), g = ground plane moved / switched, k = kinematic move, r = blocking read,
    #   [the step, with this tick's own dt and substeps where the scenario varies them], b = pipelined render readback (n: with normals)
    # hosted ranks (the host is the wire, launch by launch) take s and the per-tick dt / substeps only
    pins = np.nonzero(mesh.inv_mass == 0)[0].astype(np.int32)
    sc["_pins"] = pins
    vary = rng.random() < 0.3
    sc["render"] = "none"
    if host != "hosted" and "_tri" in sc and rng.random() < 0.5:
        sc["render"] = str(rng.choice(["triangles", "render-set-only"]))
    acts, per_tick = [], []
    for _ in range(sc["ticks"]):
        a = ""
        if rng.random() < 0.12:
            a += "s"
        if host != "hosted":
            if rng.random() < 0.12:
                a += "g"
            if len(pins) and rng.random() < 0.5:
                a += "k"
            if rng.random() < 0.35:
                a += "r"
            if rng.random() < 0.3:
                a += "n" if sc["render"] != "none" else "b"
            if rng.random() < 0.2:
                a += "x"
        acts.append(a)
        per_tick.append((float(rng.choice([0.02, 0.01, 0.005])), int(rng.choice([1, 2, 3, 5, 7]))) if vary else (sc["dt"], sc["substeps"]))
    # pipelined readbacks: begin after a tick, end one or two ticks LATER (at most two pending, oldest first) -- the snapshot must be the
    # state at ITS begin whatever the host did since (set_state, moves, ticks)
    sc["pipelined"] = bool(host != "hosted" and sc["render"] == "none" and rng.random() < 0.4)
    if sc["pipelined"]:
        acts = [a.replace("b", "p") for a in acts]
    sc["actions"] = acts
    sc["per_tick"] = per_tick if vary else "fixed"
    sc["_per_tick"] = per_tick
    sc["_bad"] = [int(rng.integers(0, 1 << 30)) for _ in acts]
    sc["_move"] = rng.uniform(-0.2, 0.2, (sc["ticks"], 3)).astype(np.float32)
    sc["_state"] = [(mesh.pos + rng.uniform(-0.03, 0.03, mesh.pos.shape).astype(np.float32), rng.uniform(-0.5, 0.5, mesh.pos.shape).astype(np.float32))
                    if "s" in a else None for a in acts]
    sc["_planes"] = [(0.0, 1.0, 0.0, ymin - float(rng.choice([0.0, 0.1, 0.3])), int(rng.random() < 0.8)) if "g" in a else None for a in acts]
    return sc


def describe(sc):
    return ", ".join(f"{k}={v}" for k, v in sc.items() if not k.startswith("_"))


def make_tuning(t):
    tune = native.SbTuning()
    native.lib().sb_tuning_default(C.byref(tune))
    for f in t["flags"]:
        tune.flags |= getattr(native, "SB_TUNE_" + f)
    for k in ("tile_lanes", "quad_lanes", "narrow_min_tiles", "store_through_max_tiles", "store_through_large", "peek_min_tiles", "win_dwords"):
        setattr(tune, k, t[k])
    return tune


PART = {"auto": native.SB_PARTITION_AUTO, "blocks": native.SB_PARTITION_BLOCKS, "rcb": native.SB_PARTITION_RCB}


def run(sc):
    """-> (verdict, detail): verdict in OK / MISMATCH / REFUSED (the plugin declined the combination with a message) / ERROR"""
    mesh, S, dt, comp = sc["_mesh"], sc["substeps"], sc["dt"], sc["compliance"]
    kw = dict(substeps=S, fixed_delta_time=dt, tile_particles=sc["tile"], damping=sc["damping"], distance_compliance=comp[0], volume_compliance=comp[1],
              bending_compliance=comp[2], ground_plane=sc["plane"], use_graph=sc["graph"], tuning=make_tuning(sc["tuning"]), gravity=sc["gravity"])
    checks = []       # (label, got, want) compared bitwise at the end
    why = []

    try:
        if sc["host"] == "hosted":
            with HostedRanks(mesh, sc["world"], S, dt=dt, partition=PART[sc["partition"]], **{k: v for k, v in kw.items() if k not in ("substeps", "fixed_delta_time")}) as H:
                for t in range(sc["ticks"]):
                    if "s" in sc["actions"][t]:
                        for sb in H.ranks:
                            sb.set_state(*sc["_state"][t])
                    H.dt, H.S = sc["_per_tick"][t]
                    H.tick()
                x, v, ghosts = H.merged_state()
                val = [sb.validate() for sb in H.ranks]
            checks_state = (x, v)
        elif sc["host"] == "group":
            g = SoftbodyGroup(mesh, [0] * sc["world"], halo_transport=native.SB_TRANSPORT_PEER, walk=sc["walk"], whole_mesh=sc["whole_mesh"], partition=PART[sc["partition"]], **kw).Start()
            try:
                checks_state, val = drive(sc, g, checks, group=True)
            finally:
                g.OnDestroy()
        else:
            sb = Softbody(mesh, **kw).Start()
            try:
                checks_state, val = drive(sc, sb, checks, group=False)
            finally:
                sb.OnDestroy()
    except native.SoftbodyError as e:
        # (SB_ERR_UNSUPPORTED = the plugin declines a combination by design; anything else a scenario of this generator provokes is a finding)
        return ("REFUSED" if e.code == native.SB_ERR_UNSUPPORTED else "ERROR"), str(e)[:300]
    # ---- the oracle, same sequence ----
    # (a plan of its own from the host-only planner, same mesh and tile size: the published order does not depend on the partition, the
    # host model or any tuning switch -- which is part of what is being tested)
    ref_plan = build_plan(mesh, tile_particles=sc["tile"])
    o = make_oracle(oracle, mesh, ref_plan, gravity=sc["gravity"], damping=sc["damping"], compliance=comp, ground_plane=sc["plane"])
    k = 0
    opend = []
    tri = sc.get("_tri")

    def compare(want):
        nonlocal k
        label, got = checks[k]; k += 1
        if not same(got, want):
            why.append(label)

    for label, got in [c for c in checks if c[1] is None]:
        why.append(label)
    checks[:] = [c for c in checks if c[1] is not None]
    for t in range(sc["ticks"]):
        acts = sc["actions"][t]
        if "s" in acts:
            o.x[:] = sc["_state"][t][0]; o.v[:] = sc["_state"][t][1]
        if "g" in acts:
            pl = sc["_planes"][t]
            o.set_ground_plane(pl[:3], pl[3], bool(pl[4]))
        if "k" in acts:
            o.set_kinematic_positions(sc["_pins"], mesh.pos[sc["_pins"]] + sc["_move"][t])
        if "r" in acts:
            compare(o.x.copy())
        o.step(*sc["_per_tick"][t])
        if "p" in acts:
            if len(opend) == 2:
                compare(opend.pop(0))
            opend.append(o.x.copy())
        if "b" in acts or "n" in acts:
            rs = np.unique(tri) if (sc["render"] == "render-set-only") else None
            compare(o.x.copy() if rs is None else o.x[rs].copy())
            if "n" in acts:
                nrm = oracle.vertex_normals(o.x, tri)
                compare(nrm if rs is None else nrm[rs])
    while opend:
        compare(opend.pop(0))
    x, v = checks_state
    if not same(x, o.x):
        why.append(f"final positions ({int((bits(x) != bits(o.x)).any(axis=1).sum())} of {mesh.n} particles differ)")
    if not same(v, o.v):
        why.append("final velocities")
    if not all(r["errors"] == [0] * 6 for r in val):
        why.append(f"table validator: {[r['errors'] for r in val]}")
    overflow = not (np.isfinite(o.x).all() and np.isfinite(o.v).all())
    if why and overflow:
        why.append("(the oracle's own state is not finite)")
    return ("MISMATCH", "; ".join(why)) if why else ("OK", "the oracle overflowed: non-finite in the same places, bitwise elsewhere" if overflow else "")


def bad_call(sc, sb, group, which):
    """One call a host must not make, chosen by `which`; -> None when the plugin refused it with an error code, else what went wrong."""
    L, mesh, n = native.lib(), sc["_mesh"], sc["_mesh"].n
    h = sb._g if group else sb._h
    f = (lambda name: getattr(L, ("sb_group_" if group else "sb_") + name))
    buf = np.zeros((n + 1, 3), np.float32)
    free = np.nonzero(mesh.inv_mass != 0)[0].astype(np.int32)
    pins = sc["_pins"]
    one = np.zeros((1, 3), np.float32)
    nan3 = np.full((1, 3), np.nan, np.float32)
    p = native.ptr
    calls = [
        ("step with dt = 0", lambda: f("step")(h, 0.0, 3)),
        ("step with dt = NaN", lambda: f("step")(h, float("nan"), 3)),
        ("step with 0 substeps", lambda: f("step")(h, 0.02, 0)),
        ("step with -1 substeps", lambda: f("step")(h, 0.02, -1)),
        ("get_positions with n + 1", lambda: f("get_positions")(h, p(buf), n + 1)),
        ("get_positions into NULL", lambda: f("get_positions")(h, None, n)),
        ("get_velocities with n - 1", lambda: f("get_velocities")(h, p(buf), n - 1)),
        ("set_state with n + 1", lambda: f("set_state")(h, p(buf), p(buf), n + 1)),
        ("set_state with NULL velocities", lambda: f("set_state")(h, p(buf), None, n)),
        ("set_particles after finalize", lambda: f("set_particles")(h, p(buf), p(buf), p(np.ones(n, np.float32)), n)),
    ] + ([] if sc.get("pipelined") else [       # (with pipelined readbacks one may be pending: the call would be valid)
        ("readback_end without a begin", lambda: f("readback_end")(h, C.byref(C.POINTER(C.c_float)()))),
    ]) + [
        ("render triangles out of range", lambda: f("set_render_triangles")(h, np.array([0, 1, n], np.int32).ctypes.data_as(C.POINTER(C.c_int32)), 1)),
        ("kinematic target: id out of range", lambda: f("set_kinematic_positions")(h, p(np.array([n], np.int32)), p(one), 1)),
        ("kinematic target: negative count", lambda: f("set_kinematic_positions")(h, p(np.array([0], np.int32)), p(one), -1)),
    ]
    if len(free):
        calls.append(("kinematic target on a free particle", lambda: f("set_kinematic_positions")(h, p(free[:1].copy()), p(one), 1)))
    if len(pins):
        calls.append(("kinematic target NaN", lambda: f("set_kinematic_positions")(h, p(pins[:1].copy()), p(nan3), 1)))
        calls.append(("kinematic target: an id twice", lambda: f("set_kinematic_positions")(h, p(np.array([pins[0], pins[0]], np.int32)), p(np.zeros((2, 3), np.float32)), 2)))
    if not group:
        tune = native.SbTuning(); L.sb_tuning_default(C.byref(tune))
        calls.append(("set_tuning after finalize", lambda: L.sb_set_tuning(h, C.byref(tune))))
        calls.append(("finalize twice", lambda: L.sb_finalize(h)))
    name, fn = calls[which % len(calls)]
    rc = fn()
    return None if rc < 0 else f"invalid call accepted (rc {rc}): {name}"


def drive(sc, sb, checks, group):
    mesh = sc["_mesh"]
    L = native.lib()
    if sc["render"] != "none":
        sb.set_render_triangles(sc["_tri"])
        if sc["render"] == "render-set-only":
            sb.set_readback_render_set_only(True)
    pend = []
    for t in range(sc["ticks"]):
        acts = sc["actions"][t]
        if "x" in acts:
            bad = bad_call(sc, sb, group, sc["_bad"][t])
            if bad:
                checks.append((bad, None))
        if "s" in acts:
            sb.set_state(*sc["_state"][t])
        if "g" in acts:
            pl = sc["_planes"][t]
            native.check((L.sb_group_set_ground_plane if group else L.sb_set_ground_plane)(sb._g if group else sb._h, *[float(c) for c in pl[:4]], int(pl[4])))
        if "k" in acts:
            sb.set_kinematic_positions(sc["_pins"], mesh.pos[sc["_pins"]] + sc["_move"][t])
        if "r" in acts:
            checks.append((f"read before tick {t}", sb.get_positions().copy()))
        sb.step(*sc["_per_tick"][t])
        if "p" in acts:
            if len(pend) == 2:
                checks.append((f"pipelined readback begun after tick {pend.pop(0)}", np.array(sb.readback_end(), copy=True)))
            sb.readback_begin(); pend.append(t)
        if "b" in acts or "n" in acts:
            sb.readback_begin()
            if "n" in acts:
                pos, nrm = sb.readback_end(normals=True)
                checks.append((f"render readback after tick {t}", np.array(pos, copy=True)))
                checks.append((f"render normals after tick {t}", np.array(nrm, copy=True)))
            else:
                checks.append((f"render readback after tick {t}", np.array(sb.readback_end(), copy=True)))
    while pend:
        checks.append((f"pipelined readback begun after tick {pend.pop(0)}", np.array(sb.readback_end(), copy=True)))
    x, v = sb.get_positions().copy(), sb.get_velocities().copy()
    if group:
        val = [sb.rank(r).validate() for r in range(sc["world"])]
    else:
        val = [sb.validate()]
    return (x, v), val


def child(a, make_scenario, run):
    """Scenarios in THIS process until the time is up; one line before and one after each (the parent tells a crash from the missing second)."""
    t_end = time.time() + a.seconds
    seeds = [a.only] if a.only is not None else range(a.seed, a.seed + a.max)
    for seed in seeds:
        if time.time() > t_end:
            break
        try:
            sc = make_scenario(seed)
        except Exception as e:       # a bug of the generator: report the seed, go on
            print(f"START {seed} seed={seed} (the generator failed)\nDONE {seed} ERROR 0.0s generator: {type(e).__name__}: {e}", flush=True)
            continue
        print(f"START {seed} {describe(sc)}", flush=True)
        t0 = time.time()
        try:
            verdict, detail = run(sc)
        except Exception as e:       # a bug of the harness or an assertion of the test infrastructure: report, go on
            verdict, detail = "ERROR", f"{type(e).__name__}: {e} | {traceback.format_exc().splitlines()[-3:]}"
        print(f"DONE {seed} {verdict} {time.time() - t0:.1f}s {detail}", flush=True)
    print("END", flush=True)


def main(make_scenario=make_scenario, run=run, script=None):
    """(tests/fuzz/fuzz_schedules.py runs its own generator through the same parent / child harness)"""
    script = os.path.abspath(script or __file__)
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--max", type=int, default=10 ** 9)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a, make_scenario, run)
    # the scenarios run in a child process: a crash (the plugin's, or a GPU fault's abort) costs one scenario, not the run
    import subprocess
    t_end = time.time() + a.seconds
    tally, n, seed = {}, 0, (a.only if a.only is not None else a.seed)
    last = seed + (1 if a.only is not None else a.max)
    while seed < last and time.time() < t_end:
        cmd = [sys.executable, "-X", "faulthandler", script, "--child", "--seed", str(seed), "--max", str(last - seed),
               "--seconds", str(max(1.0, t_end - time.time()))]
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        started, desc, ended = None, "", False
        for line in p.stdout:
            line = line.rstrip("\n")
            if line.startswith("START "):
                _, sd, desc = line.split(" ", 2); started = int(sd)
            elif line.startswith("DONE "):
                _, sd, verdict, secs, *rest = line.split(" ", 4)
                detail = rest[0] if rest else ""
                tally[verdict] = tally.get(verdict, 0) + 1; n += 1
                print(f"{verdict:8s} {secs:>6s}  {desc}" + (f"\n         -> {detail}" if detail else ""), flush=True)
                seed = int(sd) + 1; started = None
            elif line == "END":
                ended = True
        err = p.stderr.read()
        rc = p.wait()
        if started is not None:      # died inside a scenario
            tally["CRASH"] = tally.get("CRASH", 0) + 1; n += 1
            print(f"CRASH    rc={rc}  {desc}\n         -> {err[-1500:]}", flush=True)
            seed = started + 1
        elif ended:
            break
        elif rc != 0:
            print(f"fuzz_parity: the child ended with rc={rc} outside a scenario: {err[-800:]}", flush=True)
            break
    print(f"SUMMARY {n} scenarios: " + ", ".join(f"{k} {v}" for k, v in sorted(tally.items())), flush=True)
    return 1 if (tally.get("MISMATCH", 0) or tally.get("ERROR", 0) or tally.get("CRASH", 0)) else 0


if __name__ == "__main__":
    sys.exit(main())
