"""The C-ABI library loads without a GPU and exports exactly what include/softbody.h declares;
the ctypes table (== the C# [DllImport] list) covers every declared function. No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from softbodyunity_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("softbody.h", "softbody_group.h", "softbody_plan.h", "softbody_debug.h")


def _declared(headers=HEADERS):
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(sb_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_the_product_header_stays_thin():
    # what a Unity maintainer reads: the solver handle's product surface; test hooks, the validator, per-launch timing and the tuning
    # switches live in softbody_debug.h, the planner inspection in softbody_plan.h, the single-process multi-device host in softbody_group.h
    core = _declared(("softbody.h",))
    assert len(core) <= 40, core
    assert not [n for n in core if n.startswith(("sb_debug_", "sb_plan_", "sb_group_", "sb_profile_"))]
    assert not [n for n in _declared(("softbody_group.h",)) if not n.startswith("sb_group_")]
    # the plugin reads no environment variable that selects kernels or table layouts (sb_tuning replaces them): what is left prints
    csrc = os.path.join(ROOT, "softbodyunity_amd", "csrc")
    left = []
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")) and not f.startswith("_old"):
            left += re.findall(r'getenv\("(\w+)"\)', open(os.path.join(csrc, f)).read())
    assert sorted(set(left)) == ["SB_PLAN_TIMING", "SB_PRINT_ALLOC"], left


def test_every_declared_symbol_is_exported_and_bound():
    decl = _declared()
    assert len(decl) >= 35
    out = subprocess.check_output(["nm", "-D", "--defined-only", native.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for name in decl:
        assert name in exported, f"{name} declared in softbody.h but not exported"
        assert name in native.SIGNATURES, f"{name} has no ctypes/[DllImport] binding"
    extra = {e for e in exported if e.startswith("sb_")} - set(decl)
    assert not extra, f"exported but undeclared: {extra}"


def test_csharp_binding_lists_the_same_entry_points():
    cs = open(os.path.join(ROOT, "csharp", "SoftbodyNative.cs")).read()
    bound = set(re.findall(r"static extern \w[\w\.\[\]]*\s+(sb_[a-z0-9_]+)\s*\(", cs))
    assert bound == set(_declared()), (set(_declared()) - bound, bound - set(_declared()))
    assert 'DllImport("softbody_mi355x"' in cs or 'const string Lib = "softbody_mi355x"' in cs


def test_struct_layouts_match_header():
    # sb_desc: 3 + 3 ints, 3 + 1 floats, 2 ints (ABI 5) + partition, plan_flags, transport, schedule, debug_flags, 3 reserved = 80
    # bytes; sb_phase_info has int64 alignment
    assert C.sizeof(native.SbDesc) == 80
    assert C.sizeof(native.SbPlanOpts) == 48 and C.sizeof(native.SbDomain) == 8 + 8 * 8 + 8
    assert C.sizeof(native.SbPhaseInfo) == 48
    assert C.sizeof(native.SbStats) == 8 * 2 + 8 * 3 + 4 * 2 + 8 * 4 + 8 * 5 + 8 * 3 + 8 * 5 + (4 * 2 + 8 * 3 + 8 + 8 + 4 * 2) + 8 * 6 + 4 * 2 + 8 * 2
    assert C.sizeof(native.SbRuntimeInfo) == 8 * 4 + 2 * 256


def test_header_structs_compile_to_the_same_sizes(tmp_path):
    # the ctypes twin against the header itself, through the C compiler
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "softbody.h"\n#include "softbody_group.h"\n#include "softbody_plan.h"\n#include "softbody_debug.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sb_desc), sizeof(sb_plan_opts), '
                   'sizeof(sb_phase_info), sizeof(sb_stats), sizeof(sb_runtime_info_t), sizeof(sb_domain), sizeof(sb_validate_report), sizeof(sb_tuning), '
                   'sizeof(sb_exchange_timing)); return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert got == [C.sizeof(native.SbDesc), C.sizeof(native.SbPlanOpts), C.sizeof(native.SbPhaseInfo), C.sizeof(native.SbStats),
                   C.sizeof(native.SbRuntimeInfo), C.sizeof(native.SbDomain), C.sizeof(native.SbValidateReport), C.sizeof(native.SbTuning),
                   C.sizeof(native.SbExchangeTiming)]


def test_runtime_info_names_the_bound_libraries():
    # no GPU needed: which HIP runtime / RCCL the plugin resolved to in THIS process (pytest imported torch first when a test
    # module needed torch.multiprocessing: then both are PyTorch's bundled ones, else the system's)
    ri = native.runtime_info()
    assert ri["hip_runtime_version"] >= 70000000 and os.path.exists(ri["hip_library"])
    assert ri["rccl_version"] >= 22606 and os.path.exists(ri["rccl_library"])
    assert ri["capture_overlap_ok"] == (ri["hip_runtime_version"] >= 70200000)
    import sys
    if "torch" in sys.modules:
        assert ri["rccl_was_resident"] and "torch" in ri["rccl_library"] and "torch" in ri["hip_library"]


def test_loads_without_gpu_and_fails_loudly():
    L = native.lib()
    assert L.sb_abi_version() == 8
    d = native.SbDesc()
    L.sb_desc_default(C.byref(d))
    assert d.world == 1 and d.tile_particles == 0 and d.use_graph == 1 and abs(d.gravity[1] + 9.81) < 1e-6
    h = C.c_void_p()
    rc = L.sb_create(C.byref(d), C.byref(h))
    if rc == native.SB_OK:               # a gfx950 device is present (GPU box): nothing to fail on
        assert L.sb_destroy(h) == native.SB_OK
        return
    assert rc == native.SB_ERR_NO_DEVICE and b"no CPU path" in L.sb_last_error()
    with pytest.raises(native.SoftbodyError):
        from softbodyunity_amd import Softbody, jelly_cube
        Softbody(jelly_cube(4)).Start()


def test_null_and_bad_arguments_are_rejected():
    L = native.lib()
    assert L.sb_create(None, None) == native.SB_ERR_INVALID_ARG
    assert L.sb_destroy(None) == native.SB_ERR_INVALID_ARG
    assert L.sb_step(None, 0.02, 10) == native.SB_ERR_INVALID_ARG
    assert L.sb_plan_order_count(None) == -1
    h = C.c_void_p()
    rest = np.zeros((2, 3), np.float32)
    assert L.sb_plan_build(native.ptr(rest), 0, None, 0, None, 0, None, 0, None, C.byref(h)) == native.SB_ERR_INVALID_ARG


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "softbodyunity_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_only_tests_smoke_and_the_cpu_baseline_leg_touch_the_oracle():
    # tools/ (measurement and A/B scripts) never use the checker: the scripts that do -- the fuzzers, the size checks -- live under tests/fuzz/
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f)).read()
            assert "from oracle" not in txt and "import oracle" not in txt and "liboracle" not in txt, f
    # bench.py: one place (_oracle_tools), reached from the parity / cpu_baseline legs after the timed region
    txt = open(os.path.join(ROOT, "bench.py")).read()
    import re
    assert len(re.findall(r"^\s*(?:from oracle\b|import oracle\b)", txt, re.M)) == 1 and "def _oracle_tools" in txt


def test_last_words_survive_a_fatal_signal(tmp_path):
    # sb_debug_last_words (softbody_debug.h): bench.py --gpus N registers its line as it stands before every further A/B variant; should the
    # process then die of a fatal signal -- or be told to stop by a launcher tearing the job down -- the plugin's handler writes the line
    import signal
    import sys
    code = r"""
import ctypes, os, signal, sys
sys.path.insert(0, %r)
from softbodyunity_amd import native
L = native.lib()
line = b'{"metric": "particle-substeps/sec", "value": 1.0}\n'
assert L.sb_debug_last_words(1, b"an older line\n", 14, 70) == 0
assert L.sb_debug_last_words(1, line, len(line), 70) == 0          # the newer registration replaces the older one
if sys.argv[1] == "clear":
    assert L.sb_debug_last_words(1, None, 0, 0) == 0
    os.kill(os.getpid(), signal.SIGTERM)                              # default disposition again: no line
os.kill(os.getpid(), getattr(signal, sys.argv[1]))
""" % ROOT
    for sig in ("SIGSEGV", "SIGABRT", "SIGTERM"):
        out = subprocess.run([sys.executable, "-c", code, sig], capture_output=True, text=True, timeout=120)
        assert out.returncode == 70 and out.stdout == '{"metric": "particle-substeps/sec", "value": 1.0}\n', (sig, out.returncode, out.stdout, out.stderr[-500:])
    out = subprocess.run([sys.executable, "-c", code, "clear"], capture_output=True, text=True, timeout=120)
    assert out.returncode == -signal.SIGTERM and out.stdout == ""


def test_group_entry_points_reject_bad_arguments_without_a_gpu():
    L = native.lib()
    g = C.c_void_p()
    d = native.SbDesc(); L.sb_desc_default(C.byref(d))
    assert L.sb_group_create(None, None, 1, 0, C.byref(g)) == native.SB_ERR_INVALID_ARG
    assert L.sb_group_create(C.byref(d), None, 0, 0, C.byref(g)) == native.SB_ERR_INVALID_ARG
    assert L.sb_group_create(C.byref(d), None, 2, 8, C.byref(g)) == native.SB_ERR_INVALID_ARG and b"flags" in L.sb_last_error()
    d.halo_schedule = native.SB_SCHEDULE_SERIAL_GRAPH
    assert L.sb_group_create(C.byref(d), None, 2, native.SB_GROUP_WALK, C.byref(g)) == native.SB_ERR_UNSUPPORTED      # walk mode: eager schedules only
    assert L.sb_group_step(None, 0.02, 10) == native.SB_ERR_INVALID_ARG and L.sb_group_rank_count(None) == -1
    assert L.sb_group_destroy(None) == native.SB_ERR_INVALID_ARG
    t = native.SbTuning(); L.sb_tuning_default(C.byref(t))
    assert t.store_through_max_tiles == -1 and t.peek_min_tiles == -1 and t.flags == 0
    assert L.sb_set_tuning(None, C.byref(t)) == native.SB_ERR_INVALID_ARG
    d.halo_schedule = 0
    rc = L.sb_group_create(C.byref(d), None, 1, 0, C.byref(g))
    if rc == native.SB_OK:             # a gfx950 device is present (GPU box)
        assert L.sb_group_step(g, 0.02, 10) == native.SB_ERR_STATE and L.sb_group_destroy(g) == native.SB_OK
    else:
        assert rc == native.SB_ERR_NO_DEVICE
