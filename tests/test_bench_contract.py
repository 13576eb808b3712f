"""bench.py prints exactly one JSON line with the driver's contract keys (run on a small cube so it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "32", "--steps", "3", "--warmup", "1", "--cpu-sample-n", "16"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line"
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 32 ** 3 * 20 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # frac is a physical fraction of the HBM peak (traffic / time / peak); the algorithmic-bytes figure lives beside it
    assert 0.0 < r["frac"] <= 1.0 and r["frac_algorithmic"] > 0 and r["reuse_factor"] > 1.0
    # the lattice run is launches of one kernel only: its duration is the timed region's HIP-event time / launches in it,
    # and must sit at or below the eager event-pair figure (which carries the dispatch gap) but nowhere far from it
    assert r["kernel_avg_ms_source"].startswith("HIP events over the timed region") and r["kernel_launches_per_tick"] == 20
    assert abs(r["kernel_avg_ms"] - r["tick_ms_hip_events"] / 20) < 1e-9
    assert 0.5 * r["kernel_avg_ms_event_pairs"] < r["kernel_avg_ms"] < 1.25 * r["kernel_avg_ms_event_pairs"]
    assert r["model_bytes_per_launch"] > 0 and (r["traffic"] is None or abs(r["traffic"] / r["model_bytes_per_launch"] - 1) <= 0.03)
    p = j["config"]["parity"]
    assert p["ticks"] == 4 and p["finite"] is True
    assert p["small"]["bitwise"] is True and p["small"]["rel"] == 0.0 and p["small"]["ticks"] == 4
    assert p["golden"]["expected"] is None          # no golden entry for a 32^3 cube
    # the sustained figure: the same tick for >= 2 s from the initial state, the card's state sampled over that window
    assert j["sustained_ms_per_step"] > 0 and j["config"]["sustained"]["seconds"] >= 1.9 and j["config"]["sustained"]["ticks"] >= 3
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and j["config"]["finite"] is True


@pytest.mark.gpu
def test_bench_checks_itself_against_the_golden_checksums_and_the_live_oracle():
    # config 2 (BASELINE.json:8): the state the timed run ends with must hash to the oracle's golden checksum for that
    # tick count, and the live leg (solver reset, oracle on the same mesh) must agree bit for bit
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "64", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    p = j["config"]["parity"]
    assert p["golden"]["bitwise"] is True and p["golden"]["schedule_matches"] is True and p["golden"]["n"] == 64 ** 3
    assert p["live"]["bitwise"] is True and p["live"]["n"] == 64 ** 3 and p["live"]["ticks"] >= 2
    assert p["small"]["bitwise"] is True
    assert "64^3" in j["cpu_baseline"]["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("het", [False, True])
def test_bench_multi_rank_launch_on_one_gpu_through_the_peer_transport(het):
    # the driver's N > 1 launch (torch.distributed.run, one process per rank) on the ONE GPU of the box: the ranks share the device
    # and exchange ghosts through the peer-store mailboxes (RCCL refuses several ranks on one device). Everything of the multi-rank
    # bench except RCCL / xGMI runs: gloo control plane, per-rank plan of the split mesh, state checksum summed over the ranks
    # against the oracle's golden value, max-over-ranks timing, one JSON line from rank 0.
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    port = 29700 + os.getpid() % 200 + (200 if het else 0)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "peer", "--cube-edge", "64",
                          "--steps", "4", "--warmup", "2", "--no-ab"] + (["--heterogeneous"] if het else []), capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["partition"] == "2x1x1" and j["config"]["halo_transport"] == "peer"
    assert j["config"]["authoring"].startswith("sharded") and ("HETEROGENEOUS" in j["config"]["workload"]) == het     # windows in both layouts
    g = j["config"]["parity"]["golden"]
    assert g["n"] == 64 ** 3 and g["bitwise"] is True and g["expected"] is not None
    assert j["config"]["finite"] is True and j["value"] > 0 and "cpu_baseline" not in j


@pytest.mark.gpu
def test_bench_multi_rank_times_every_admitted_schedule_in_one_launch():
    # The one launch a multi-GPU node makes must be decisive: `value` is the default variant's figure, and config.schedule_ab carries every
    # other admitted (transport, schedule) pair, each on a fresh solver of the same inputs, each verified on the golden checksum, with
    # per-exchange HIP-event times and per-rank owned / ghost counts. A variant that cannot run is reported, never fatal: on this box's ONE
    # GPU RCCL refuses two ranks on a device, so the RCCL variants carry an error and the peer-store ones the numbers.
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    port = 29950 + os.getpid() % 40
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "peer", "--cube-edge", "64",
                          "--steps", "4", "--warmup", "2", "--sustained-seconds", "0.3"], capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    ab = j["config"]["schedule_ab"]
    v = {x["name"]: x for x in ab["variants"]}
    assert ab["default"] == "peer/auto" and ab["variants"][0]["name"] == "peer/auto"
    assert {"rccl/serial-eager", "rccl/overlap-eager", "peer/serial-eager", "peer/serial-graph"} <= set(v)
    d = v["peer/auto"]
    assert d["schedule"] == "serial-eager" and abs(d["value"] - j["value"]) / j["value"] < 1e-9 and abs(d["ms_per_step"] - j["ms_per_step"]) < 1e-9
    assert v["peer/serial-eager"].get("same_as") == "peer/auto"             # what AUTO resolves to is not timed twice
    measured = [x for x in ab["variants"] if "value" in x]
    assert len(measured) >= 2 and "peer/serial-graph" in [x["name"] for x in measured]
    for x in measured:
        assert x["golden"]["bitwise"] is True and x["golden"]["expected"] is not None and x["finite"] is True and x["steps"] == 4 and x["warmup"] == 2
        if x["schedule"] in ("serial-eager", "overlap-eager"):
            e = x["exchange"]
            assert e["exchanges_per_tick"] == 10 and e["per_exchange_us_rank0"]["total"] > 0
            assert set(e["per_exchange_us_max_over_ranks"]) == {"pack", "transport", "total", "exposed_wait"}
        else:
            assert x["exchange"] is None
    for name in ("rccl/serial-eager", "rccl/overlap-eager"):               # reported, never fatal
        assert "value" in v[name] or ("error" in v[name] and "rank" in v[name]["error"])
    assert ab["fastest_verified"]["name"] in [x["name"] for x in measured]
    pr = ab["per_rank"]
    assert [q["rank"] for q in pr] == [0, 1] and sum(q["owned"] for q in pr) == 64 ** 3 and all(q["ghosts"] > 0 and q["halo_peers"] == 1 for q in pr)
    assert j["sustained_ms_per_step"] > 0 and j["config"]["sustained"]["seconds"] >= 0.3 and j["config"]["sustained"]["ticks"] >= 4
    assert j["config"]["parity"]["golden"]["bitwise"] is True and "cpu_baseline" not in j


@pytest.mark.gpu
def test_bench_multi_rank_default_rccl_auto_control_flow_with_a_stand_in_transport():
    # What the driver launches on a multi-GPU node is `bench.py --gpus N` with the DEFAULT transport and schedule (rccl/auto), which this box's
    # one GPU cannot run (RCCL refuses two ranks on a device). The control flow of that launch -- the serialised eager variant first and its
    # line registered as the fallback, then the default (whose figure is `value`), then every other admitted pair -- must not meet its first
    # execution there: --rccl-stand-in peer runs the variants NAMED rccl/* over the peer transport (a test aid, flagged in the line).
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    port = 29890 + os.getpid() % 40
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cube-edge", "64", "--steps", "4", "--warmup", "2",
                          "--sustained-seconds", "0.2", "--rccl-stand-in", "peer"], capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert "TEST RUN" in j["config"]["rccl_stand_in"] and j["config"]["halo_transport"] == "rccl"
    ab = j["config"]["schedule_ab"]
    names = [x["name"] for x in ab["variants"]]
    assert ab["default"] == "rccl/auto" and names[0] == "rccl/auto" and names[1] == "rccl/serial-eager"      # the default's record first, the fallback's second
    assert names.count("rccl/serial-eager") == 1 and {"rccl/overlap-eager", "rccl/serial-graph", "peer/serial-eager", "peer/serial-graph"} <= set(names)
    assert "value_from" not in ab                                                                             # the default completed: the line is its own
    v = {x["name"]: x for x in ab["variants"]}
    assert abs(v["rccl/auto"]["value"] - j["value"]) / j["value"] < 1e-9 and j["sustained_ms_per_step"] > 0
    measured = [x for x in ab["variants"] if "value" in x]
    assert len(measured) >= 5 and all(x["golden"]["bitwise"] is True for x in measured), [(x["name"], x.get("golden")) for x in measured]
    assert ab["fastest_verified"]["name"] in names and j["config"]["parity"]["golden"]["bitwise"] is True


@pytest.mark.gpu
def test_bench_default_that_misses_the_golden_checksum_gives_way_to_the_verified_serialised_line():
    # The default rccl/auto runs ticks of the overlapped schedule while it calibrates -- a schedule that has never run between two devices. Should
    # its state NOT end on the golden checksum there, its figure must not become `value`: the serialised eager variant measured first (and
    # verified) stands, the default is recorded as failed. (--debug-golden-mismatch-variant plants the mismatch.)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    port = 29810 + os.getpid() % 40
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cube-edge", "64", "--steps", "4", "--warmup", "2",
                          "--no-sustained", "--no-ab", "--rccl-stand-in", "peer", "--debug-golden-mismatch-variant", "rccl/auto"],
                         capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    ab = j["config"]["schedule_ab"]
    v = {x["name"]: x for x in ab["variants"]}
    assert "FAILED" in ab["value_from"] and "golden checksum" in v["rccl/auto"]["error"]
    assert abs(v["rccl/serial-eager"]["value"] - j["value"]) / j["value"] < 1e-9 and j["config"]["parity"]["golden"]["bitwise"] is True


@pytest.mark.gpu
def test_bench_line_survives_a_variant_that_hangs():
    # A later A/B variant that never returns (a collective one rank never joins, a kernel that never finishes) must not cost the run its line:
    # rank 0's watchdog writes the line as it stands -- the default's figure, the variants measured so far -- and ends the run with a non-zero
    # code (which is what makes the launcher tear the other ranks down).
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    port = 29850 + os.getpid() % 40
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "peer", "--cube-edge", "64", "--steps", "4",
                          "--warmup", "2", "--no-sustained", "--variant-timeout", "6", "--debug-hang-variant", "peer/serial-graph"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[0])
    ab = j["config"]["schedule_ab"]
    assert j["value"] > 0 and ab["default"] == "peer/auto" and ab["variants"][0]["golden"]["bitwise"] is True
    assert "peer/serial-graph" not in [v["name"] for v in ab["variants"] if "value" in v]          # the hanging one never reported
    assert "exceeded its time limit" in out.stderr


def test_traffic_file_entries_match_the_compulsory_model_of_a_host_built_plan():
    # profiles/hbm_traffic.json must not go stale: every entry is within 3 % of 49 B per particle + the tile streams
    # (4 B per dictionary-coded slot -- or 16 B (heterogeneous: 40 B) per lane of a 128-lane workgroup where the slots are lane-packed: single-rank spring
    # meshes whose launches are narrow, i.e. at least 10 240 tiles) + 128 B per tile of the plan the host-only planner builds
    import re
    from softbodyunity_amd import native
    from softbodyunity_amd.mesh import jelly_cube
    tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for key, ent in tj.items():
        m = re.match(r"n(\d+)(het)?_tile(-?\d+)_gpus(\d+)", key)
        n, het, tile, gpus = int(m.group(1)), bool(m.group(2)), int(m.group(3)), int(m.group(4))
        if gpus != 1:
            continue
        mesh = jelly_cube(n)
        plan = native.Plan.build(mesh.rest_pos, mesh.dist_ij, tile_particles=tile)
        for slot in ("0", "1"):
            if slot not in ent:
                continue
            ph = [p for p in plan.phases(int(slot)) if p["kind"] == 1][0]
            slots = ph["order_end"] - ph["order_begin"]
            tiles = ph["task_end"] - ph["task_begin"]
            # headline layout: 48 B of particle state + ~1 B, 4-byte dictionary-coded slots; heterogeneous layout (bench.py
            # --heterogeneous): + a 4-byte inverse mass per particle, 8-byte slots
            lane_packed = tiles >= 10240                      # kernel_types.hpp kLanePack*, tables.hip build_device
            stream = (40.0 if het else 16.0) * 128 * tiles if lane_packed else (8.0 if het else 4.0) * slots
            model = (52.0 if het else 49.0) * mesh.n + stream + 128.0 * tiles
            assert abs(ent[slot] / model - 1) <= 0.03, (key, slot, ent[slot], model)


@pytest.mark.gpu
def test_bench_heterogeneous_variant_checks_itself_too():
    # the data-layout worst case (per-particle masses, per-spring rest lengths): own golden checksums, same self-verification
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "64", "--heterogeneous", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    p = j["config"]["parity"]
    assert "HETEROGENEOUS" in j["config"]["workload"]
    assert p["golden"]["bitwise"] is True and p["golden"]["schedule_matches"] is True
    assert p["live"]["bitwise"] is True and p["small"]["bitwise"] is True
    # 8-byte slots and 4-byte inverse masses: more compulsory bytes per launch than the headline layout of the same size
    base = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity"],
                          capture_output=True, text=True, timeout=600)
    jb = json.loads([l for l in base.stdout.splitlines() if l.strip()][0])
    assert j["roofline"]["model_bytes_per_launch"] > 1.1 * jb["roofline"]["model_bytes_per_launch"]
