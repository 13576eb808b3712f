#!/usr/bin/env python3
"""Why do boxes differ (headline launch 148 us on one, 161 us on another, stable within a box)? Samples what the amdgpu driver
exposes to an ordinary user -- current sclk / mclk / fclk / socclk levels, average power and its cap, temperatures -- every 20 ms
while the 256^3 headline workload runs, and prints the ranges beside the measured ms per tick. usage: python tools/clock_probe.py [n]"""
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from softbodyunity_amd import Softbody, jelly_cube  # noqa: E402


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def _active_level(text):
    """pp_dpm_* files list the levels, the active one carries a '*': '1: 2400Mhz *' -> 2400"""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            digits = "".join(ch for ch in line.split(":", 1)[1] if ch.isdigit() or ch == ".")
            return float(digits) if digits else None
    return None


def sample(dev):
    out = {}
    for name in ("sclk", "mclk", "fclk", "socclk"):
        v = _active_level(_read(os.path.join(dev, f"pp_dpm_{name}")))
        if v is not None:
            out[name + "_MHz"] = v
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for key, scale in (("power1_average", 1e-6), ("power1_input", 1e-6), ("power1_cap", 1e-6), ("temp1_input", 1e-3), ("temp2_input", 1e-3),
                           ("temp3_input", 1e-3), ("freq1_input", 1e-6), ("freq2_input", 1e-6)):
            t = _read(os.path.join(hw, key))
            if t and t.lstrip("-").isdigit():
                out[key] = float(t) * scale
    busy = _read(os.path.join(dev, "gpu_busy_percent"))
    if busy and busy.isdigit():
        out["gpu_busy_percent"] = float(busy)
    return out


def own_card(devs):
    """The sysfs node of HIP device 0, by PCI address (the runtime the plugin loaded is asked through ctypes)."""
    import ctypes
    for name in ("libamdhip64.so.7", "libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
                addr = buf.value.decode().lower()
                for d in devs:
                    if os.path.realpath(d).lower().endswith(addr):
                        return d
        except OSError:
            pass
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    devs = sorted(d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")) or glob.glob(os.path.join(d, "hwmon", "hwmon*")))
    report = {"n": n, "devices_seen": devs}
    sb = Softbody(jelly_cube(n), substeps=20).Start()
    samples = {d: [] for d in devs}
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            for d in devs:
                samples[d].append(sample(d))
            time.sleep(0.02)
    idle = {d: sample(d) for d in devs}
    th = threading.Thread(target=watch, daemon=True)
    for _ in range(5):
        sb.step()
    sb.synchronize()
    th.start()
    per = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(60):
            sb.step()
        sb.synchronize()
        per.append(1e3 * (time.perf_counter() - t0) / 60)
    stop.set(); th.join()
    sb.OnDestroy()
    report["ms_per_tick_5x60"] = per
    report["own_card"] = own_card(devs)
    report["library_variant"] = os.environ.get("SB_LIB_VARIANT") or "product"
    report["idle"] = idle
    rng = {}
    for d in devs:
        keys = sorted({k for s in samples[d] for k in s})
        rng[d] = {k: [min(s[k] for s in samples[d] if k in s), max(s[k] for s in samples[d] if k in s)] for k in keys}
        rng[d]["samples"] = len(samples[d])
    report["under_load_min_max"] = rng
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
