#!/usr/bin/env python3
"""bench.py — particle-substeps/s of the soft-body hot path on MI355X (BASELINE.json:2).

A "step" is one tick (= one FixedUpdate = `--substeps` substeps) over the whole synthetic jelly cube.
Workload: the 256^3 structural lattice x 20 substeps (BASELINE.json:9, the configuration the metric's
"achieved HBM GB/s vs 8 TB/s" and the >=10M-particle target are quoted on; it fits one GPU). With
--gpus N > 1 the SAME 256^3 mesh is split spatially over N ranks (BASELINE.json:10) => strong scaling.
`--n 64` runs config[1] instead.

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W` — one process per GPU. torch.distributed (gloo) carries only the
control plane (unique-id broadcast, barriers, max-over-ranks); the ghost exchange itself is RCCL
send/recv inside the plugin on its own HIP stream.

N > 1 (the one run a multi-GPU node makes): the line's `value` is the DEFAULT variant's figure (--transport / --schedule, default RCCL with
SB_SCHEDULE_AUTO, which measures the two eager schedules over its first ticks -- run here before the warm-up -- and keeps the faster; the
serialised eager variant is measured FIRST and its line registered as the fallback); inside the same launch every other admitted
(transport, schedule) pair is timed on the same inputs with a fresh solver -- RCCL serial-eager / overlap-eager / serial-graph / overlap-graph (where sb_runtime_info admits them), the peer-store
transport eager and captured -- each followed by the golden-checksum check, with per-exchange HIP-event times (pack, transport, exposed
wait) and per-rank owned / ghost counts: `config.schedule_ab`. A variant that fails is reported there, never fatal; the line as it
stands is registered with the plugin (sb_debug_last_words) before every further variant and written by a watchdog should one hang, so
ONE line comes out whatever a variant does. --no-ab skips the extra variants.

Sustained figure: after the timed region and the parity legs the state is reset and the same tick runs for >= 2 s (`sustained_ms_per_step`,
`config.sustained`, with the card's clocks and power sampled over THAT window); `value` stays the driver-parameterised region.

Self-verification (outside the timed region, `config.parity`):
  golden : an order-independent checksum of the positions + velocities the timed run ended with (after warmup + steps
           ticks; ranks add their partial sums) against tests/golden/state_checksums.json, which
           tests/golden/make_checksums.py generated from the CPU oracle for 1..40 ticks of the 64^3 and 256^3 cubes;
  small  : the same number of ticks on a 40^3 cube through the plugin against the oracle run live, bit for bit;
  live   : (N = 1, with the CPU baseline) the solver is reset to the initial state and advanced as many ticks as the
           cpu_baseline leg advanced the oracle on the SAME mesh; positions and velocities compared bit for bit.
Roofline block: `achieved`/`frac` are HBM traffic per launch / kernel time against the 8 TB/s peak (kernel time = HIP events over
the timed region divided by the launches in it when the region is launches of one kernel only, as on the lattice workloads;
else HIP-event pairs around every launch of extra eager ticks), with the traffic
taken from the PMC counters (profiles/hbm_traffic.json) after checking it against the compulsory-bytes model of the
tables actually uploaded (sb_get_stats launch_bytes); the SURVEY 8d algorithmic-bytes figure is kept beside it as
`frac_algorithmic` with the on-chip reuse factor.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md chip-level parameters


def b_alg(n_particles, n_constraints):
    """SURVEY.md §8d algorithmic bytes per substep: 52 integrate + 68/constraint + 36 velocity."""
    return 52.0 * n_particles + 68.0 * n_constraints + 36.0 * n_particles


def _golden_entry(n, substeps, tile, het=False):
    path = os.path.join(ROOT, "tests", "golden", "state_checksums.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(f"cube{n}{'het' if het else ''}_s{substeps}_tile{tile}")


class GpuStateSampler:
    """What the amdgpu driver shows an ordinary user about THIS rank's GPU while the workload runs: shader / memory / fabric clock,
    package power and its cap (sysfs, no child process). The headline launch sits at the package power cap on MI355X (measured:
    1 391 W of 1 400 W), so boxes whose silicon needs more voltage hold a lower shader clock -- 2.0-2.15 GHz against 2.4 GHz --
    and run the same launch in 161 us instead of 148 us (tools/clock_probe.py, profiles/r03s2_clock_probe.json). The line carries
    the numbers so that a roofline fraction can be read against the state of the box that produced it."""

    def __init__(self, pci_bus_id):
        import glob
        self.dev = None
        cards = sorted(glob.glob("/sys/class/drm/card*/device"))
        for d in cards:
            if pci_bus_id and os.path.realpath(d).lower().endswith(pci_bus_id.lower()):
                self.dev = d
        self.cards = cards
        self.samples = []
        self._stop = None
        self._th = None

    @staticmethod
    def _active(path):
        try:
            for line in open(path):
                if line.rstrip().endswith("*"):
                    digits = "".join(ch for ch in line.split(":", 1)[1] if ch.isdigit() or ch == ".")
                    return float(digits) if digits else None
        except OSError:
            pass
        return None

    def _sample(self, d):
        import glob
        out = {}
        for name in ("sclk", "mclk", "fclk"):
            v = self._active(os.path.join(d, f"pp_dpm_{name}"))
            if v is not None:
                out[name + "_MHz"] = v
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for key, name in (("power1_input", "power_W"), ("power1_average", "power_W"), ("power1_cap", "power_cap_W")):
                try:
                    out[name] = float(open(os.path.join(hw, key)).read()) * 1e-6
                except (OSError, ValueError):
                    pass
        try:
            out["busy_percent"] = float(open(os.path.join(d, "gpu_busy_percent")).read())
        except (OSError, ValueError):
            pass
        return out

    def start(self):
        import threading
        if not self.cards:
            return self
        self._stop = threading.Event()

        def watch():
            while not self._stop.is_set():
                # the device's PCI address is known: one card; else every card, the busiest one is picked afterwards
                self.samples.append({d: self._sample(d) for d in ([self.dev] if self.dev else self.cards)})
                self._stop.wait(0.01)
        self._th = threading.Thread(target=watch, daemon=True)
        self._th.start()
        return self

    def stop(self):
        if self._th is None:
            return None
        self._stop.set(); self._th.join()
        if not self.samples:
            return None
        dev = self.dev
        if dev is None:      # (a launcher that hides the PCI address) the card that was busiest while we ran
            dev = max(self.cards, key=lambda d: sum(smp[d].get("busy_percent", 0.0) for smp in self.samples))
        keys = sorted({k for smp in self.samples for k in smp[dev]})
        rng = {k: [min(smp[dev][k] for smp in self.samples if k in smp[dev]), max(smp[dev][k] for smp in self.samples if k in smp[dev])] for k in keys}
        out = {"source": dev + (" (matched by PCI address)" if self.dev else " (busiest card)"), "samples": len(self.samples),
               "window": "warm-up + timed region", "min_max": rng}
        pw, cap = rng.get("power_W"), rng.get("power_cap_W")
        if pw and cap and cap[1] > 0:
            out["at_power_cap"] = bool(pw[1] >= 0.97 * cap[1])
        return out


def _pci_bus_id(torch, device):
    try:
        p = torch.cuda.get_device_properties(device)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None



SCHEDULES = {"auto": 0, "serial-eager": 1, "serial-graph": 2, "overlap-eager": 3, "overlap-graph": 4}
SCHEDULE_NAMES = {1: "serial-eager", 2: "serial-graph", 3: "overlap-eager", 4: "overlap-graph"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--cube-edge", dest="n", type=int, default=256,
                    help="cube edge (particles = n^3); under torch.distributed.run spell it --cube-edge (its parser takes --n for an abbreviation of its own options)")
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--tile", type=int, default=512, help="target particles per LDS tile, -1 = global colours only")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the self-verification legs (profiling runs)")
    ap.add_argument("--no-gpu-state", action="store_true", help="do not sample the card's clocks and power (sysfs) while measuring")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained leg behind the timed region")
    ap.add_argument("--sustained-seconds", type=float, default=2.0)
    ap.add_argument("--cpu-sample-n", type=int, default=0, help="cube edge for the CPU baseline sample (0 = the workload itself)")
    ap.add_argument("--strict-traffic", action="store_true",
                    help="fail (instead of falling back to the compulsory-bytes model and saying so) when profiles/hbm_traffic.json is "
                         "missing or disagrees with the model of this build's tables by more than 3 %%")
    ap.add_argument("--allow-stale-traffic", action="store_true", help="(accepted for old scripts; stale entries never fail without --strict-traffic)")
    ap.add_argument("--heterogeneous", action="store_true",
                    help="the data-layout worst case beside the headline's best case: per-particle masses and per-spring rest lengths "
                         "(4-byte inverse masses, 8-byte constraint slots; mesh.jelly_cube(heterogeneous=True))")
    ap.add_argument("--transport", choices=["rccl", "peer"], default="rccl",
                    help="ghost exchange of the DEFAULT variant for --gpus N > 1: RCCL send/recv (default) or the opt-in peer-store mailboxes "
                         "(DESIGN.md 7; never run between two devices)")
    ap.add_argument("--whole-mesh", action="store_true", help="N > 1: every rank generates and plans the whole mesh (round-2 behaviour) instead of its window")
    ap.add_argument("--schedule", choices=list(SCHEDULES), default="auto",
                    help="halo schedule of the DEFAULT variant for N > 1 (sb_desc.halo_schedule); auto = serial-eager")
    ap.add_argument("--no-ab", action="store_true", help="N > 1: time the default (transport, schedule) only, not every admitted pair (config.schedule_ab)")
    ap.add_argument("--ab-steps", type=int, default=0, help="N > 1: timed ticks of the A/B variants (0 = --steps)")
    ap.add_argument("--debug-golden-mismatch-variant", default="", help=argparse.SUPPRESS)
    ap.add_argument("--variant-timeout", type=float, default=150.0, help="N > 1: seconds one variant may take before the watchdog writes the line as it stands and ends the run")
    ap.add_argument("--rccl-stand-in", choices=["peer"], default=None,
                    help="TEST AID for one-GPU boxes, never a measurement: the variants NAMED rccl/* run over the peer-store transport (RCCL refuses two "
                         "ranks on one device), so that the control flow of the N > 1 launch -- fallback line first, default, A/B block -- executes end to end")
    ap.add_argument("--debug-hang-variant", default=None,
                    help="TEST AID: the named A/B variant never returns (every rank sleeps), so that the watchdog's path -- write the line as it stands, "
                         "end the run with a non-zero code -- can be exercised")
    ap.add_argument("--loopback-world", type=int, default=0,
                    help="diagnostic: run as rank 0 of this many ranks with SB_TEST_LOOPBACK (RCCL self-exchange on one GPU); "
                         "the reported value counts only the particles this rank owns")
    return ap.parse_args()


class LastWords:
    """ONE line leaves this process whatever happens (rank 0 of an N > 1 run): the line as it stands is registered with the plugin, whose
    signal handler writes it should the process die (a GPU fault ends in abort(); a launcher tearing the job down sends SIGTERM), and a
    watchdog thread writes it should a variant hang."""

    def __init__(self, fd, native):
        self.fd, self.native, self.line = fd, native, None
        self.deadline, self.what = None, ""
        self._th = None

    def stash(self, line_dict):
        self.line = (json.dumps(line_dict) + "\n").encode()
        self.native.lib().sb_debug_last_words(self.fd, self.line, len(self.line), 70)

    def clear(self):
        self.deadline = None
        self.native.lib().sb_debug_last_words(self.fd, None, 0, 0)

    def arm(self, seconds, what):
        import threading
        self.deadline, self.what = time.time() + seconds, what
        if self._th is None:
            def watch():
                while True:
                    time.sleep(1.0)
                    d = self.deadline
                    if d is not None and time.time() > d:
                        print(f"bench.py: '{self.what}' exceeded its time limit: writing the line as it stands and ending the run", file=sys.stderr, flush=True)
                        if self.line:
                            os.write(self.fd, self.line)
                        os._exit(71)
            self._th = threading.Thread(target=watch, daemon=True)
            self._th.start()

    def disarm(self):
        self.deadline = None


def main():
    args = parse_args()

    # stdout carries exactly one JSON line: native libraries (RCCL prints a version banner on communicator
    # creation) write to fd 1, so park fd 1 on stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # multi-process GPU work on this image needs dmabuf IPC (RCCL's and hipIpc's handle exchange fail with the legacy mode); the
    # launch environment exports it already -- keep it if a launcher scrubbed the environment. Must precede the first HIP call.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))
    # a launcher may expose all GPUs to every rank (use LOCAL_RANK) or exactly one per rank (use 0)
    n_dev = torch.cuda.device_count()
    device = local_rank if (n_dev == 0 or local_rank < n_dev) else local_rank % n_dev
    if n_dev > 0:
        torch.cuda.set_device(device)      # torch.cuda.synchronize() in barrier() then waits for THIS rank's device, not for device 0 on every rank

    # torch is imported BEFORE the plugin on purpose and always: the plugin then binds the HIP runtime and the RCCL torch brought
    # (one ROCm stack per process); which ones is recorded in the line (config.runtime, from sb_runtime_info)
    from softbodyunity_amd import jelly_cube, native
    ctx = dict(args=args, rank=rank, world=world, device=device, n_dev=n_dev, torch=torch, dist=dist, native=native)

    t_setup = time.time()
    # N > 1: sharded authoring -- every rank generates, hands over and plans only ITS WINDOW of the cube (its block + two cells
    # of margin, include/softbody.h sb_domain); the whole 256^3 mesh exists nowhere. (--tile -1 and --whole-mesh keep the whole mesh
    # on every rank.)
    part_world = args.loopback_world if args.loopback_world > 1 else world        # (the loopback diagnostic is rank 0 of that many ranks)
    sharded = part_world > 1 and args.tile > 0 and not args.whole_mesh
    if sharded:
        from softbodyunity_amd.mesh import jelly_cube_window
        mesh = jelly_cube_window(args.n, rank, part_world, _dims(part_world), args.tile, heterogeneous=args.heterogeneous)
    else:
        mesh = jelly_cube(args.n, heterogeneous=args.heterogeneous)
    ctx.update(mesh=mesh, mesh_s=time.time() - t_setup, sharded=sharded, t_setup=t_setup)

    if world > 1:
        out = multi_rank(ctx, real_stdout)
    else:
        out = single_rank(ctx)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


def make_solver(ctx, transport="rccl", schedule="auto", sb_world=None, debug_flags=0):
    """A fresh solver of the workload's mesh for this rank (+ the transport's set-up across the ranks)."""
    from softbodyunity_amd import Softbody, comm_unique_id
    args, rank, world, dist, torch, native = ctx["args"], ctx["rank"], ctx["world"], ctx["dist"], ctx["torch"], ctx["native"]
    peer = transport == "peer" or (transport == "rccl" and args.rccl_stand_in == "peer" and world > 1)
    uid = None
    if world > 1 and not peer:       # (the peer transport needs no RCCL communicator: the mailbox handles travel over gloo below)
        buf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = torch.tensor(list(comm_unique_id()), dtype=torch.uint8)
        dist.broadcast(buf, src=0)
        uid = bytes(buf.tolist())
    if sb_world is None:
        sb_world = world
    elif debug_flags & native.SB_DEBUG_LOOPBACK:
        uid = comm_unique_id()
    sb = Softbody(ctx["mesh"], substeps=args.substeps, fixed_delta_time=0.02, device=ctx["device"], rank=rank, world=sb_world,
                  tile_particles=args.tile, use_graph=not args.no_graph, unique_id=uid, plan_flags=0, debug_flags=debug_flags,
                  halo_transport=native.SB_TRANSPORT_PEER if peer else native.SB_TRANSPORT_RCCL, halo_schedule=SCHEDULES[schedule])
    sb.Start()
    return sb


def connect_peers(ctx, sb):
    torch, dist, rank, world = ctx["torch"], ctx["dist"], ctx["rank"], ctx["world"]
    handles = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(handles, torch.from_numpy(sb.peer_mailbox_handle().copy()))
    for r in range(world):
        if r != rank:
            sb.peer_connect(r, handles[r].numpy())
    dist.barrier()


def make_barrier(ctx, sb):
    torch, dist = ctx["torch"], ctx["dist"]

    def barrier():
        sb.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
    return barrier


def calibration_ticks(sb):
    """SB_SCHEDULE_AUTO measures the two eager schedules over its first ticks (sb_stats.halo_auto_*): run them BEFORE the warm-up, so that
    neither the alternating ticks nor the one blocking all-gather of the decision falls into the timed region. They are ordinary ticks (same
    bits in either schedule) and count towards the golden checksum's tick number."""
    n = 0
    while n < 12 and sb.stats()["halo_auto_state"] == 1:
        sb.step(); n += 1
    return n


def timed_region(ctx, sb, warmup, steps, sampler=None):
    """W untimed ticks, then exactly K ticks between barriers (+ synchronize on both sides); max over the ranks."""
    torch, dist = ctx["torch"], ctx["dist"]
    barrier = make_barrier(ctx, sb)
    for _ in range(warmup):
        sb.step()
    barrier()
    t0 = time.perf_counter()
    sb.profile_begin()
    for _ in range(steps):
        sb.step()
    ev_ms = sb.profile_end()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, ev_ms


def golden_leg(ctx, sb, total_ticks):
    """The state the run ended with against the oracle's golden checksum (ranks add their partial sums)."""
    from softbodyunity_amd.verify import add_checksums, schedule_hash, state_checksum
    args, rank, world, dist, mesh = ctx["args"], ctx["rank"], ctx["world"], ctx["dist"], ctx["mesh"]
    gid = mesh.global_id if ctx["sharded"] else None
    owned_mask = sb.owner() == rank
    ids = np.nonzero(owned_mask)[0]
    x_end = sb.get_positions(); v_end = sb.get_velocities()
    finite = bool(np.isfinite(x_end[ids]).all())
    part = state_checksum(x_end[ids], v_end[ids], ids if gid is None else gid[ids])
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, part)
        fin = [None] * world
        dist.all_gather_object(fin, finite)
        finite = all(fin)
    else:
        parts = [part]
    checksum = add_checksums(parts)
    golden = _golden_entry(args.n, args.substeps, args.tile, args.heterogeneous)
    want = golden["ticks"].get(str(total_ticks)) if golden else None
    out = {"n": args.n ** 3, "checksum": f"0x{checksum:016x}", "expected": want, "bitwise": (int(want, 16) == checksum) if want else None,
           "source": "tests/golden/state_checksums.json (CPU oracle, tests/golden/make_checksums.py)"}
    if golden and world == 1 and rank == 0:
        out["schedule_matches"] = schedule_hash(sb.plan()) == golden["schedule"]
    return out, finite


def sustained_leg(ctx, sb, ms_per_step_hint):
    """The same tick for >= --sustained-seconds from the initial state, the card's clocks and power sampled over THAT window: the
    headline region is a fraction of a second, too short for the package power cap to bite (256^3: 20 ticks = 58 ms)."""
    args, rank, mesh, torch = ctx["args"], ctx["rank"], ctx["mesh"], ctx["torch"]
    ticks = int(min(max(np.ceil(1e3 * args.sustained_seconds / max(ms_per_step_hint, 1e-3)), args.steps), 20000))
    if ctx["dist"] is not None:      # every rank the same count
        t = torch.tensor([ticks], dtype=torch.int64)
        ctx["dist"].broadcast(t, src=0)
        ticks = int(t.item())
    for attempt in range(3):      # (the hint comes from a short region that may hold a graph capture: lengthen until the window is long enough)
        sb.set_state(mesh.pos, mesh.vel)
        sampler = None
        if rank == 0 and not args.no_gpu_state:
            sampler = GpuStateSampler(_pci_bus_id(torch, ctx["device"]) if ctx["n_dev"] > 0 else None).start()
        elapsed, ev_ms = timed_region(ctx, sb, 2, ticks)
        state = sampler.stop() if sampler is not None else None
        if state:
            state["window"] = "the sustained leg"
        if elapsed >= 0.97 * args.sustained_seconds or ticks >= 20000:
            break
        ticks = int(min(np.ceil(1.15 * ticks * args.sustained_seconds / max(elapsed, 1e-6)), 20000))       # (elapsed is the max over the ranks: same count everywhere)
    return {"ticks": ticks, "seconds": elapsed, "ms_per_step": 1e3 * elapsed / ticks, "ms_per_step_hip_events": ev_ms / ticks, "gpu_state": state,
            "note": "state reset to the initial one, 2 untimed ticks, then `ticks` ticks between barriers"}


def exchange_leg(ctx, sb, ticks=3):
    """Per-exchange HIP-event times of `ticks` extra eager ticks (outside every timed region): pack kernel, transport, the whole exchange,
    and what the compute stream waited for it."""
    st = sb.stats()
    if st["halo_schedule"] not in (1, 3):       # captured schedules: events inside a capture have no host-visible time
        return None
    sb.exchange_timing(True)
    for _ in range(ticks):
        sb.step()
    t = sb.exchange_timing_read()
    sb.exchange_timing(False)
    n = max(t["exchanges"], 1)
    return {"exchanges_timed": t["exchanges"], "exchanges_per_tick": t["exchanges"] / ticks, "ticks": ticks,
            "per_exchange_us": {"pack": 1e3 * t["pack_ms"] / n, "transport": 1e3 * t["transport_ms"] / n, "total": 1e3 * t["total_ms"] / n,
                                "exposed_wait": 1e3 * t["exposed_wait_ms"] / n}}


def teardown(ctx, sb):
    # teardown order: every rank drains its own stream, then all ranks meet, THEN the solvers go -- under the peer transport
    # sb_destroy frees the mailbox the neighbours' kernels store into (include/softbody.h)
    try:
        sb.synchronize()
    except Exception as e:      # (report, but still meet the other ranks and free the solver)
        print(f"bench.py: sb_synchronize at teardown: {e}", file=sys.stderr)
    if ctx["dist"] is not None:
        ctx["dist"].barrier()
    sb.OnDestroy()


def all_ok(ctx, ok, msg=""):
    """Every rank learns whether every rank got through a phase (a variant that fails on one rank is abandoned by all, together)."""
    dist = ctx["dist"]
    if dist is None:
        return ok, [msg] if msg else []
    got = [None] * ctx["world"]
    dist.all_gather_object(got, (bool(ok), msg))
    return all(g[0] for g in got), [f"rank {r}: {g[1]}" for r, g in enumerate(got) if not g[0]]


def run_variant(ctx, name, transport, schedule, steps, warmup, is_default):
    """One (transport, schedule) pair on a fresh solver of the same inputs: timed region, golden checksum, exchange timing."""
    args, rank, world, native = ctx["args"], ctx["rank"], ctx["world"], ctx["native"]
    rec = {"name": name, "transport": transport, "schedule_requested": schedule}
    sb, err = None, ""
    t0 = time.time()
    if args.debug_hang_variant == name:
        time.sleep(10 ** 6)
    try:
        sb = make_solver(ctx, transport, schedule)
    except Exception as e:       # refused (SB_ERR_UNSUPPORTED), RCCL could not form the communicator, ...
        err = f"{type(e).__name__}: {e}"
    ok, errs = all_ok(ctx, sb is not None, err)
    if not ok:
        if sb is not None:
            sb.OnDestroy()
        rec["error"] = "; ".join(errs)[:600]
        return rec, None
    result = None
    try:
        if transport == "peer" or (transport == "rccl" and args.rccl_stand_in == "peer"):
            connect_peers(ctx, sb)
        stats = sb.stats()
        rec["schedule"] = SCHEDULE_NAMES.get(stats["halo_schedule"])
        rec["setup_seconds"] = time.time() - t0
        calib = calibration_ticks(sb)
        stats = sb.stats()
        rec["schedule"] = SCHEDULE_NAMES.get(stats["halo_schedule"])
        if stats["halo_auto_state"] == 2:
            rec["auto_calibration"] = {"ticks_before_warmup": calib, "timed_ticks": stats["halo_auto_ticks"],
                                       "ms_per_tick_slowest_rank": {"serial-eager": stats["halo_auto_ms"][0], "overlap-eager": stats["halo_auto_ms"][1]},
                                       "kept": rec["schedule"], "rule": "overlap-eager only where more than 3 % faster"}
        sampler = None
        if is_default and rank == 0 and not args.no_gpu_state:
            sampler = GpuStateSampler(_pci_bus_id(ctx["torch"], ctx["device"]) if ctx["n_dev"] > 0 else None).start()
        elapsed, ev_ms = timed_region(ctx, sb, warmup, steps)
        gpu_state = sampler.stop() if sampler is not None else None
        N = args.n ** 3
        rec.update({"steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "value": N * args.substeps * steps / elapsed,
                    "tick_ms_hip_events_rank0": ev_ms / steps})
        if not args.no_parity:
            g, finite = golden_leg(ctx, sb, calib + warmup + steps)
            if args.debug_golden_mismatch_variant == name and g["bitwise"] is True:       # (test hook: what a variant with wrong bits looks like to the run)
                g = dict(g, checksum="0x" + "0" * 16, bitwise=False)
            rec["golden"] = {"checksum": g["checksum"], "expected": g["expected"], "bitwise": g["bitwise"]}
            rec["finite"] = finite
        else:
            g, finite = None, None
        ex = exchange_leg(ctx, sb)
        if ctx["dist"] is not None:
            allx = [None] * world
            ctx["dist"].all_gather_object(allx, ex)
            if ex is not None:
                keys = ("pack", "transport", "total", "exposed_wait")
                ex = dict(ex, per_exchange_us_max_over_ranks={k: max(a["per_exchange_us"][k] for a in allx if a) for k in keys},
                          per_exchange_us_rank0=ex["per_exchange_us"])
                ex.pop("per_exchange_us")
        rec["exchange"] = ex
        result = dict(sb=sb, stats=stats, elapsed=elapsed, ev_ms=ev_ms, golden=g, finite=finite, gpu_state=gpu_state, calib=calib)
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    ok, errs = all_ok(ctx, result is not None, err)
    if not ok:
        rec["error"] = "; ".join(errs)[:600]
        try:
            teardown(ctx, sb)
        except Exception as e:
            print(f"bench.py: teardown of variant {name}: {e}", file=sys.stderr)
        return rec, None
    return rec, result


def full_variant(ctx, name, transport, schedule, with_sustained, runtime):
    """One variant measured the way the line's own figure is: timed region + golden leg (run_variant), per-launch event timing, the table
    validator, per-rank counts, optionally the sustained leg -> (record, line or None on rank != 0 / failure, ok)."""
    args, rank, world, dist = ctx["args"], ctx["rank"], ctx["world"], ctx["dist"]
    N = args.n ** 3
    M = 3 * args.n * args.n * (args.n - 1)
    rec, res = run_variant(ctx, name, transport, schedule, args.steps, args.warmup, True)
    if res is None:
        return rec, None, False
    sb, stats = res["sb"], res["stats"]
    setup_s = time.time() - ctx["t_setup"]
    # per-kernel HIP-event timing on the solver's stream: two extra eager ticks, outside the timed region
    slot_ms, slot_cnt = profiled_ticks(sb, 2)
    parity = {"ticks": res["calib"] + args.warmup + args.steps}
    if res["golden"] is not None:
        parity["golden"] = res["golden"]
    if not args.no_parity:
        rep = sb.validate()
        parity["tables"] = {"constraints_checked": rep["constraints_checked"], "tiles_checked": rep["tiles_checked"], "errors": rep["errors"],
                            "clean": rep["errors"] == [0] * 6}
    finite = res["finite"] if res["finite"] is not None else bool(np.isfinite(sb.get_positions()[sb.owner() == rank]).all())
    parity["finite"] = finite
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "owned": stats["n_particles_owned"], "ghosts": stats["n_particles_local"] - stats["n_particles_owned"],
                                      "halo_peers": stats["halo_peers"], "halo_particles_sent_per_exchange": stats["halo_particles_t1"],
                                      "halo_particles_recv_per_exchange": stats["halo_particles_recv"], "tiles": stats["n_tiles"],
                                      "lane_packed_tiles": stats["lane_packed_tiles"], "device_bytes": stats["device_bytes"]})
    sustained = sustained_leg(ctx, sb, rec["ms_per_step"]) if with_sustained else None
    out = None
    if rank == 0:
        out = build_line(ctx, sb, stats, res["elapsed"], res["ev_ms"], slot_ms, slot_cnt, parity, finite, res["gpu_state"], runtime, setup_s, False, N, M)
        out["config"]["halo_transport"] = transport
        if args.rccl_stand_in:
            out["config"]["rccl_stand_in"] = f"TEST RUN: the variants named rccl/* ran over the {args.rccl_stand_in} transport (one-GPU box); not a measurement of RCCL"
        if sustained:
            out["sustained_ms_per_step"] = sustained["ms_per_step"]
            out["config"]["sustained"] = sustained
        out["config"]["schedule_ab"] = {"default": None, "variants": [rec], "per_rank": per_rank,
                                        "note": "every variant: fresh solver, same mesh windows, same warmup/steps protocol (barrier + synchronize on both sides, "
                                                "max over ranks), then the golden-checksum check; per-exchange times are HIP events of 3 extra eager ticks"}
    teardown(ctx, sb)
    return rec, out, True


def multi_rank(ctx, real_stdout):
    """--gpus N > 1: the default variant gives `value`; every other admitted (transport, schedule) pair follows in the same launch."""
    args, rank, world, dist, native, mesh = ctx["args"], ctx["rank"], ctx["world"], ctx["dist"], ctx["native"], ctx["mesh"]
    runtime = native.runtime_info()
    words = LastWords(real_stdout, native) if rank == 0 else None
    default_name = f"{args.transport}/{args.schedule}"
    measured = {}
    # The default rccl/auto MEASURES the two eager schedules on the devices at hand over its first ticks (sb_stats.halo_auto_*), i.e. it runs
    # ticks of the overlapped schedule, which has never run between two devices. So the serialised eager schedule -- what the default falls
    # back to anyway -- goes first and its line is registered: whatever the default's calibration then does, ONE line comes out.
    safe_rec, safe_out = None, None
    if args.transport == "rccl" and args.schedule == "auto":
        if words:
            words.arm(4 * args.variant_timeout, "the serialised eager variant (set-up included)")
        safe_rec, safe_out, ok = full_variant(ctx, "rccl/serial-eager", "rccl", "serial-eager", False, runtime)
        if ok:
            measured["rccl/serial-eager"] = safe_rec
            if rank == 0:
                safe_out["config"]["schedule_ab"].update({"default": default_name, "value_from": "rccl/serial-eager: the default variant (rccl/auto: the two eager "
                                                          "schedules measured over the first ticks, the faster kept) had not completed when this line was registered"})
                words.stash(safe_out)
    if words:
        words.arm(4 * args.variant_timeout, "the default variant (set-up included)")
    rec, out, ok = full_variant(ctx, default_name, args.transport, args.schedule, not args.no_sustained, runtime)
    # A default whose state does not end on the golden checksum is a FAILED variant, whatever it timed: with a verified serialised eager line
    # in hand that line stands (every rank sees the same gathered checksum, so all take the same branch).
    if ok and (rec.get("golden") or {}).get("bitwise") is False and safe_rec is not None and (safe_rec.get("golden") or {}).get("bitwise") is True:
        rec["error"] = f"the run did not end on the golden checksum ({rec['golden']['checksum']} against {rec['golden']['expected']}): not used for `value`"
        ok = False
    if not ok:
        if safe_rec is None or "value" not in safe_rec:
            raise SystemExit(f"bench.py: the default variant {default_name} failed: {rec.get('error')}")
        out = safe_out      # (rank 0) the serialised eager variant's line stands, the default's failure is recorded beside it
        if rank == 0:
            out["config"]["schedule_ab"]["variants"].append(rec)
            out["config"]["schedule_ab"]["value_from"] = f"rccl/serial-eager: the default variant ({default_name}) FAILED, see its record in `variants`"
    else:
        measured[default_name] = rec
        if rank == 0:
            out["config"]["schedule_ab"]["default"] = default_name
            if safe_rec is not None:
                out["config"]["schedule_ab"]["variants"].append(safe_rec)
            words.stash(out)
    if not args.no_ab:
        # safest first: what only differs in launch order, then captured RCCL calls, last the transport that has never run between two devices
        plan = [("rccl", "serial-eager"), ("rccl", "overlap-eager")]
        if runtime["capture_serial_ok"] and not args.no_graph:
            plan.append(("rccl", "serial-graph"))
        if runtime["capture_overlap_ok"] and not args.no_graph:
            plan.append(("rccl", "overlap-graph"))
        plan += [("peer", "serial-eager")] + ([("peer", "serial-graph")] if not args.no_graph else [])
        resolved_default = (args.transport, rec.get("schedule")) if (ok and "auto_calibration" not in rec) else None
        ab_steps = args.ab_steps or args.steps
        for transport, schedule in plan:
            name = f"{transport}/{schedule}"
            if name in measured:
                continue
            if (transport, schedule) == resolved_default:      # (what an uncalibrated AUTO resolves to is not timed twice)
                if rank == 0:
                    out["config"]["schedule_ab"]["variants"].append({"name": name, "same_as": default_name})
                continue
            if words:
                words.arm(args.variant_timeout, f"variant {name}")
            vrec, vres = run_variant(ctx, name, transport, schedule, ab_steps, args.warmup, False)
            if vres is not None:
                teardown(ctx, vres["sb"])
            if rank == 0:
                out["config"]["schedule_ab"]["variants"].append(vrec)
                good = [v for v in out["config"]["schedule_ab"]["variants"] if "value" in v and (v.get("golden") or {}).get("bitwise") is not False]
                best = max(good, key=lambda v: v["value"])
                out["config"]["schedule_ab"]["fastest_verified"] = {"name": best["name"], "ms_per_step": best["ms_per_step"], "value": best["value"]}
                words.stash(out)
        if words:
            words.disarm()
    if words:
        words.clear()
    return out


def profiled_ticks(sb, prof_ticks=2):
    slot_ms = None
    for _ in range(prof_ticks):
        ms, cnt = sb.step_profiled()
        slot_ms = ms.astype(np.float64) if slot_ms is None else slot_ms + ms
        slot_cnt = cnt
    return slot_ms / prof_ticks, slot_cnt


def single_rank(ctx):
    args, rank, native, mesh, torch = ctx["args"], ctx["rank"], ctx["native"], ctx["mesh"], ctx["torch"]
    loopback = args.loopback_world > 1
    runtime = native.runtime_info() if loopback else None
    N = args.n ** 3
    M = 3 * args.n * args.n * (args.n - 1)
    if loopback:
        sb = make_solver(ctx, args.transport, args.schedule, sb_world=args.loopback_world, debug_flags=native.SB_DEBUG_LOOPBACK)
    else:
        sb = make_solver(ctx)
    out = None
    try:
        stats = sb.stats()
        setup_s = time.time() - ctx["t_setup"]
        calib = calibration_ticks(sb)      # (loopback diagnostic with SB_SCHEDULE_AUTO: the schedule measurement stays out of the timed region)
        if calib:
            stats = sb.stats()
        sampler = None
        if not args.no_gpu_state:
            sampler = GpuStateSampler(_pci_bus_id(torch, ctx["device"]) if ctx["n_dev"] > 0 else None).start()
        elapsed, ev_ms = timed_region(ctx, sb, args.warmup, args.steps)
        gpu_state = sampler.stop() if sampler is not None else None

        # ---- self-verification, leg 1: the state the timed run ended with, against the oracle's golden checksum ------
        total_ticks = args.warmup + args.steps
        parity = {"ticks": total_ticks}
        if not args.no_parity and not loopback:
            parity["golden"], finite = golden_leg(ctx, sb, total_ticks)
        else:
            finite = bool(np.isfinite(sb.get_positions()[sb.owner() == rank]).all())
        # the tables the timed launches read, re-read by the validator kernel (sb_debug_validate): a group or a launch that touched a
        # particle twice would be a race, whatever the state looks like
        if not args.no_parity:
            rep = sb.validate()
            parity["tables"] = {"constraints_checked": rep["constraints_checked"], "tiles_checked": rep["tiles_checked"], "errors": rep["errors"],
                                "clean": rep["errors"] == [0] * 6}
        slot_ms, slot_cnt = profiled_ticks(sb, 2)
        if loopback:
            N = int(stats["n_particles_owned"])      # diagnostic mode: only this rank's share is simulated
        out = build_line(ctx, sb, stats, elapsed, ev_ms, slot_ms, slot_cnt, parity, finite, gpu_state, runtime, setup_s, loopback, N, M)
        if loopback:
            out["config"]["halo_transport"] = args.transport
            out["config"]["exchange"] = exchange_leg(ctx, sb)
            if stats["halo_auto_state"] == 2:
                out["config"]["auto_calibration"] = {"ticks_before_warmup": calib, "ms_per_tick": {"serial-eager": stats["halo_auto_ms"][0], "overlap-eager": stats["halo_auto_ms"][1]}}
        if not loopback and not args.no_parity:
            parity["small"] = small_parity(total_ticks, args, ctx["device"])
        if not loopback and not args.no_cpu_baseline:
            out["cpu_baseline"], live = cpu_baseline(mesh, sb, args)
            if live is not None:
                parity["live"] = live
        if not args.no_sustained and not loopback:
            sus = sustained_leg(ctx, sb, out["ms_per_step"])
            out["sustained_ms_per_step"] = sus["ms_per_step"]
            out["config"]["sustained"] = sus
    finally:
        teardown(ctx, sb)
    return out


def build_line(ctx, sb, stats, elapsed, ev_ms, slot_ms, slot_cnt, parity, finite, gpu_state, runtime, setup_s, loopback, N, M):
    """The JSON line of the contract from one measured solver (rank 0)."""
    args, world, sharded = ctx["args"], ctx["world"], ctx["sharded"]
    value = N * args.substeps * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    G = stats["n_global_colours"]
    owned = stats["n_particles_owned"]
    names = ["tile_kernel<1> on T0 (rounds + collide/velocity/integrate + rounds)",
             "tile_kernel<1> on T1 (rounds + collide/velocity/integrate + rounds)"]
    names += [f"global colour {c}" for c in range(G)]
    names += ["tile_kernel<0> (first kernel of a tick)", "tile_kernel<2> (last kernel of a tick)",
              "tile_kernel<3> on the T2 layers (constraints inside neither T0 nor T1)"]
    # ALGORITHMIC bytes per launch: a mid-tick tile kernel does one velocity update (36 B)
    # + one integrate (52 B) per particle and projects every constraint it stores TWICE (before and after
    # the MARK step: the tail of one substep and the head of the next), 68 B each time
    alg_bytes = [88.0 * owned + 2 * 68.0 * stats["tile_constraints"][t] for t in (0, 1)]
    # compulsory HBM bytes per launch from the tables actually uploaded (sb_get_stats): the model the PMC figure is checked against
    lb = stats["launch_bytes"]
    model_bytes = [float(lb[0]), float(lb[1])]
    mask0 = None
    for c in range(G):
        if mask0 is None:
            plan = sb.plan(); mask0 = plan.local_order_mask(0).astype(bool); ph0 = [p for p in plan.phases(0) if p["kind"] == 0]
        cnt_c = int(mask0[ph0[c]["order_begin"]:ph0[c]["order_end"]].sum())
        alg_bytes.append(68.0 * cnt_c)
        model_bytes.append(68.0 * cnt_c)      # global colours gather/scatter straight on HBM: no on-chip reuse to model
    last = (args.substeps & 1) if stats["n_tilings"] == 2 else 0
    alg_bytes += [52.0 * owned + 68.0 * stats["tile_constraints"][0], 36.0 * owned + 68.0 * stats["tile_constraints"][last]]
    alg_bytes.append(68.0 * stats["t2_constraints"])
    model_bytes += [float(lb[2]), float(lb[3]) if last == 0 else float(lb[1] - (lb[0] - lb[3])), float(lb[4])]
    k_dom = int(np.argmax(slot_ms))
    launches = max(int(slot_cnt[k_dom]), 1)
    dom_ms = float(slot_ms[k_dom]) / launches          # HIP-event pair around every launch of an eager tick
    dom_ms_pairs = dom_ms
    # When the timed region consists of launches of ONE kernel only -- the lattice workloads: tile_kernel<1>, T0 and T1
    # launches alternating, `substeps` of them per tick once the lazy tick boundary has fused the first and last kernels --
    # its average launch duration is the HIP-event time of the timed region itself divided by the launches in it (the
    # event pairs of the eager ticks add ~4 % of dispatch gap per launch).
    region_avg = (world == 1 and not loopback and not args.no_graph and G == 0 and stats["n_t2_layers"] == 0
                  and stats["n_tilings"] == 2 and args.substeps % 2 == 0 and k_dom in (0, 1) and args.steps >= 2)
    if region_avg:
        dom_ms = ev_ms / (args.steps * args.substeps)
        launches = args.substeps
        for arr in (alg_bytes, model_bytes):
            arr[0] = arr[1] = 0.5 * (arr[0] + arr[1])
        dom_name = "tile_kernel<1>, mid-tick (rounds + collide/velocity/integrate + rounds), T0 and T1 launches alternating"
    else:
        dom_name = names[k_dom]
    # HBM traffic per launch of the dominant kernel: PMC counters (profiles/hbm_traffic.json, produced by
    # tools/prof_summary.py from separate FETCH_SIZE / WRITE_SIZE passes) -- accepted only when within 3 % of the
    # compulsory-bytes model of THIS build's tables, so the file cannot go stale unnoticed
    traffic, traffic_src, traffic_meta = None, None, None
    key = f"n{args.n}{'het' if args.heterogeneous else ''}_tile{args.tile}_gpus{world}"
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    problem = None
    if not loopback and key in tj and str(k_dom) in tj[key]:
        traffic = float(tj[key][str(k_dom)])
        if region_avg and "0" in tj[key] and "1" in tj[key]:
            traffic = 0.5 * (float(tj[key]["0"]) + float(tj[key]["1"]))     # the two launch shapes alternate
        traffic_meta = tj[key].get("meta")
        dev = abs(traffic - model_bytes[k_dom]) / model_bytes[k_dom]
        if dev > 0.03:    # the entry was measured on another kernel / data layout: say so, use the model
            problem = (f"profiles/hbm_traffic.json[{key}][{k_dom}] = {traffic:.4g} B disagrees with the compulsory-bytes model "
                       f"{model_bytes[k_dom]:.4g} B of this build by {100 * dev:.1f} %: re-measure (tools/profile_round.sh)")
            traffic_src = f"stale ({100 * dev:.1f} % off the model of this build's tables): achieved/frac use the compulsory-bytes model"
            traffic = None
        else:
            traffic_src = "pmc"
    elif args.n == 256 and args.tile == 512 and world == 1 and not loopback:
        problem = f"profiles/hbm_traffic.json has no PMC entry for {key} slot {k_dom}"
    if problem:
        if args.strict_traffic:
            raise RuntimeError("bench.py: " + problem)
        print("WARNING: " + problem, file=sys.stderr)
    hbm_bytes = traffic if traffic is not None else model_bytes[k_dom]
    achieved = hbm_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    alg_rate = alg_bytes[k_dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    job_alg = b_alg(N, M) * value / N   # algorithmic B/s of the whole job
    parity["finite"] = finite
    mesh_s = ctx["mesh_s"]
    return {
        "metric": "particle-substeps/sec", "value": value, "unit": "particle-substeps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.n}^3 jelly cube, structural springs (N={N}, M={M}), {args.substeps} substeps/tick, "
                               f"dt=0.02, explicit index-array graph, tile_particles={args.tile}" +
                               (", HETEROGENEOUS masses and rest lengths (4-byte inverse masses, 8-byte constraint slots)" if args.heterogeneous else ""),
                   "partition": "x".join(str(d) for d in _dims(world)), "graph_replay": (not args.no_graph) and (world == 1 or stats["halo_schedule"] in (2, 4)),
                   "authoring": ("sharded: each rank hands over and plans its window only (sb_set_domain)" if sharded else "whole mesh on every rank"),
                   "halo_transport": None,
                   "halo_schedule": SCHEDULE_NAMES.get(stats["halo_schedule"]) if (world > 1 or loopback) else None,
                   "runtime": runtime, "gpu_state": gpu_state,
                   "finite": finite, "parity": parity},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic,
                     "traffic_source": traffic_src or "none for this configuration: achieved/frac use the compulsory-bytes model",
                     "traffic_measured_on": traffic_meta,
                     "model_bytes_per_launch": model_bytes[k_dom],
                     "traffic_over_model": (traffic / model_bytes[k_dom]) if traffic else None,
                     "kernel": dom_name, "kernel_avg_ms": dom_ms, "kernel_launches_per_tick": launches,
                     "kernel_avg_ms_source": ("HIP events over the timed region / launches in it" if region_avg
                                              else "HIP-event pair around every launch of an eager tick"),
                     "kernel_avg_ms_event_pairs": dom_ms_pairs,
                     "algorithmic_bytes_per_launch": alg_bytes[k_dom],
                     "algorithmic_GBps": alg_rate, "frac_algorithmic": alg_rate / (HBM_PEAK / 1e9),
                     "reuse_factor": alg_bytes[k_dom] / hbm_bytes,
                     "note": "achieved/frac = HBM bytes per launch (PMC FETCH_SIZE/WRITE_SIZE where measured for this "
                             "configuration, else the compulsory-bytes model of the uploaded tables) / kernel time, against "
                             "8 TB/s. frac_algorithmic uses the SURVEY 8d figure (88 B/particle + 68 B per projected "
                             "constraint per mid-tick launch); it exceeds 1 because the tile kernel serves those accesses "
                             "from LDS (reuse_factor = algorithmic / HBM bytes)",
                     "job_algorithmic_GBps": job_alg / 1e9, "job_frac_algorithmic": job_alg / (HBM_PEAK * world),
                     "B_alg_per_particle_substep": b_alg(N, M) / N,
                     "tick_ms_hip_events": ev_ms / args.steps,
                     "per_slot_ms_per_tick": {names[k]: float(slot_ms[k]) for k in range(len(names))},
                     "per_slot_launches_per_tick": {names[k]: int(slot_cnt[k]) for k in range(len(names))}},
        "setup_seconds": setup_s, "setup_breakdown": {"mesh_generation": mesh_s, "Start (author + plan + upload)": setup_s - mesh_s},
        "plan": stats,
    }


def _dims(world):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world, (world, 1, 1))


def _oracle_tools():
    from oracle import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import build_plan, make_oracle
    return oracle, build_plan, make_oracle


def small_parity(ticks, args, device):
    """Checker leg: the same number of ticks on a 40^3 cube, plugin vs the CPU oracle run live, bit for bit."""
    from softbodyunity_amd import Softbody, jelly_cube
    oracle, _, make_oracle = _oracle_tools()
    m = jelly_cube(40, heterogeneous=args.heterogeneous)
    sb = Softbody(m, substeps=args.substeps, device=device, tile_particles=args.tile, use_graph=not args.no_graph).Start()
    try:
        o = make_oracle(oracle, m, sb.plan())
        for _ in range(ticks):
            sb.step()
            o.step(0.02, args.substeps)
        x = sb.get_positions(); v = sb.get_velocities()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle.parity_error(x, o.x, m.pos)
    return {"n": m.n, "ticks": ticks, "rel": rel, "max_abs": mabs,
            "bitwise": bool(bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32)))}


def cpu_baseline(mesh, sb, args):
    """The CPU oracle (kind "port": the reference has no CPU path to time, /root/reference/README.md:1), on this box's
    host cores, over a bounded sample of the same workload: by default the workload's own mesh, one sequential tick
    (Unity FixedUpdate semantics) and then task-parallel ticks. When the sample IS the workload the oracle's final
    state doubles as the checker of a live parity leg: the solver is reset and advanced the same number of ticks."""
    oracle, build_plan, make_oracle = _oracle_tools()
    from softbodyunity_amd import jelly_cube
    n_s = args.cpu_sample_n or args.n
    same = n_s == args.n
    m = mesh if same else jelly_cube(n_s, heterogeneous=args.heterogeneous)
    plan = sb.plan() if same else build_plan(m, tile_particles=args.tile)
    o = make_oracle(oracle, m, plan)
    S = args.substeps
    t0 = time.perf_counter(); ticks = 0
    while True:
        o.step(0.02, S); ticks += 1
        if time.perf_counter() - t0 > 8.0 or ticks >= 50:
            break
    t1 = time.perf_counter() - t0
    v1 = m.n * S * ticks / t1
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    t0 = time.perf_counter(); pt = 0
    while True:
        o.step(0.02, S, parallel=True); pt += 1
        if time.perf_counter() - t0 > 5.0 or pt >= 200:
            break
    tp = time.perf_counter() - t0
    vp = m.n * S * pt / tp
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    live = None
    if same and not args.no_parity:
        sb.set_state(m.pos, m.vel)
        for _ in range(ticks + pt):
            sb.step()
        x = sb.get_positions(); v = sb.get_velocities()
        rel, mabs, bit = oracle.parity_error(x, o.x, m.pos)
        live = {"n": m.n, "ticks": ticks + pt, "rel": rel, "max_abs": mabs,
                "bitwise": bool(bit and np.array_equal(v.view(np.uint32), o.v.view(np.uint32))),
                "checker": "oracle/oracle.c on the workload's own mesh (the cpu_baseline sample), same initial state"}
    return {"value": v1, "unit": "particle-substeps/s", "cores": 1, "kind": "port",
            "sample": f"{ticks} tick(s) x {S} substeps of the {n_s}^3 cube, sequential oracle (Unity FixedUpdate semantics), "
                      f"same published schedule",
            "all_cores": {"value": vp, "cores": cores, "sample": f"{pt} tick(s), task-parallel oracle (OpenMP)"},
            "cpu_model": model}, live


if __name__ == "__main__":
    main()
