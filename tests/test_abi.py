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


def _declared():
    src = open(os.path.join(ROOT, "include", "softbody.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(sb_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


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
    # sb_desc: 3 + 3 ints, 3 + 1 floats, 2 ints = 48 bytes; sb_phase_info has int64 alignment
    assert C.sizeof(native.SbDesc) == 48
    assert C.sizeof(native.SbPlanOpts) == 24
    assert C.sizeof(native.SbPhaseInfo) == 48
    assert C.sizeof(native.SbStats) == 8 * 2 + 8 * 3 + 4 * 2 + 8 * 4 + 8 * 5 + 8 * 3 + 8 * 5


def test_loads_without_gpu_and_fails_loudly():
    L = native.lib()
    assert L.sb_abi_version() == 5
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
