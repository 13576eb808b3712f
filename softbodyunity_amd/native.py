"""ctypes binding of libsoftbody_mi355x.so — the Python twin of csharp/SoftbodyNative.cs.

Every function below is one [DllImport] in the C# file, same name, same argument order
(include/softbody.h). No reference binding exists to mirror (/root/reference/README.md:1 is the
whole reference tree). This module never imports the oracle and has no CPU execution path: if the
shared library is missing it raises, and sb_create fails without a gfx950 device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SB_LIB_VARIANT=name selects libsoftbody_mi355x_name.so (A/B timing builds, see csrc/Makefile); default = the product
_VARIANT = os.environ.get("SB_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, f"libsoftbody_mi355x{'_' + _VARIANT if _VARIANT else ''}.so")

SB_UNIQUE_ID_BYTES = 128
SB_IPC_HANDLE_BYTES = 64
SB_OK = 0
SB_ERR_INVALID_ARG, SB_ERR_STATE, SB_ERR_NO_DEVICE, SB_ERR_HIP, SB_ERR_RCCL, SB_ERR_NOMEM, SB_ERR_UNSUPPORTED = \
    -1, -2, -3, -4, -5, -6, -7


class SbDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("part_dims", C.c_int32 * 3),
                ("gravity", C.c_float * 3), ("damping", C.c_float), ("tile_particles", C.c_int32),
                ("use_graph", C.c_int32)]


class SbStats(C.Structure):
    _fields_ = [("n_particles_owned", C.c_int64), ("n_particles_local", C.c_int64),
                ("n_constraints_local", C.c_int64 * 3), ("n_tilings", C.c_int32), ("n_global_colours", C.c_int32),
                ("n_tiles", C.c_int64 * 2), ("tile_constraints", C.c_int64 * 2), ("constraints_in_tiles", C.c_int64),
                ("constraints_in_global", C.c_int64), ("halo_particles_t1", C.c_int64),
                ("halo_particles_global", C.c_int64), ("device_bytes", C.c_int64),
                ("n_t2_layers", C.c_int64), ("n_t2_tiles", C.c_int64), ("t2_constraints", C.c_int64),
                ("launch_bytes", C.c_int64 * 5)]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class SbPlanOpts(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("part_dims", C.c_int32 * 3), ("tile_particles", C.c_int32)]


class SbPhaseInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("type", C.c_int32), ("tiling", C.c_int32), ("halo_slot", C.c_int32),
                ("order_begin", C.c_int64), ("order_end", C.c_int64), ("task_begin", C.c_int64), ("task_end", C.c_int64)]


# name -> (restype, argtypes); this table is what tests/test_abi.py checks against include/softbody.h
_P = C.c_void_p
SIGNATURES = {
    "sb_desc_default": (None, [C.POINTER(SbDesc)]),
    "sb_create": (C.c_int, [C.POINTER(SbDesc), C.POINTER(_P)]),
    "sb_destroy": (C.c_int, [_P]),
    "sb_set_particles": (C.c_int, [_P, _P, _P, _P, C.c_int32]),
    "sb_set_rest_positions": (C.c_int, [_P, _P, C.c_int32]),
    "sb_set_distance_constraints": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float]),
    "sb_set_volume_constraints": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float]),
    "sb_set_bending_constraints": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float]),
    "sb_set_ground_plane": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32]),
    "sb_finalize": (C.c_int, [_P]),
    "sb_comm_unique_id": (C.c_int, [_P]),
    "sb_comm_init": (C.c_int, [_P, _P]),
    "sb_peer_mailbox_handle": (C.c_int, [_P, _P]),
    "sb_peer_connect": (C.c_int, [_P, C.c_int32, _P, _P]),
    "sb_step": (C.c_int, [_P, C.c_float, C.c_int32]),
    "sb_get_positions": (C.c_int, [_P, _P, C.c_int32]),
    "sb_get_velocities": (C.c_int, [_P, _P, C.c_int32]),
    "sb_set_state": (C.c_int, [_P, _P, _P, C.c_int32]),
    "sb_readback_begin": (C.c_int, [_P]),
    "sb_readback_end": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float))]),
    "sb_set_render_triangles": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32]),
    "sb_readback_get_normals": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float))]),
    "sb_set_readback_render_set_only": (C.c_int, [_P, C.c_int32]),
    "sb_readback_get_render_set": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]),
    "sb_get_owner": (C.c_int, [_P, _P, C.c_int32]),
    "sb_profile_begin": (C.c_int, [_P]),
    "sb_profile_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "sb_synchronize": (C.c_int, [_P]),
    "sb_step_profiled": (C.c_int, [_P, C.c_float, C.c_int32, _P, _P, C.c_int32]),
    "sb_debug_launch": (C.c_int, [_P, C.c_float, C.c_int32, C.c_int32, C.c_int32]),
    "sb_debug_halo_pack": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "sb_debug_halo_unpack": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "sb_get_stats": (C.c_int, [_P, C.POINTER(SbStats)]),
    "sb_plan_build": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.POINTER(SbPlanOpts),
                                C.POINTER(_P)]),
    "sb_plan_destroy": (C.c_int, [_P]),
    "sb_get_plan": (C.c_int, [_P, C.POINTER(_P)]),
    "sb_plan_order_count": (C.c_int64, [_P]),
    "sb_plan_get_order": (C.c_int, [_P, C.c_int32, _P, _P]),
    "sb_plan_phase_count": (C.c_int32, [_P, C.c_int32]),
    "sb_plan_get_phases": (C.c_int, [_P, C.c_int32, C.POINTER(SbPhaseInfo)]),
    "sb_plan_task_count": (C.c_int64, [_P, C.c_int32]),
    "sb_plan_get_tasks": (C.c_int, [_P, C.c_int32, _P]),
    "sb_plan_group_count": (C.c_int64, [_P, C.c_int32]),
    "sb_plan_get_groups": (C.c_int, [_P, C.c_int32, _P]),
    "sb_plan_get_owner": (C.c_int, [_P, _P]),
    "sb_plan_local_count": (C.c_int64, [_P, C.POINTER(C.c_int64)]),
    "sb_plan_get_local_particles": (C.c_int, [_P, _P]),
    "sb_plan_halo_slot_count": (C.c_int32, [_P]),
    "sb_plan_halo_counts": (C.c_int, [_P, C.c_int32, _P, _P]),
    "sb_plan_get_halo": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "sb_plan_get_local_order_mask": (C.c_int, [_P, C.c_int32, _P]),
    "sb_last_error": (C.c_char_p, []),
    "sb_abi_version": (C.c_int, []),
}

_lib = None


class SoftbodyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"softbody_mi355x error {code}: {msg}")
        self.code = code


def lib():
    """Load the plugin. Raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(softbodyunity_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != SB_OK:
        raise SoftbodyError(rc, lib().sb_last_error().decode("utf-8", "replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def i32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a if shape is None else a.reshape(shape)


class Plan:
    """Read-only view of a planner result (sb_plan_*). Owns the handle unless borrowed from a solver."""

    def __init__(self, handle, owned):
        self._h = C.c_void_p(handle)
        self._owned = owned

    @classmethod
    def build(cls, rest_pos, dist_ij=None, vol_ijkl=None, bend_ijkl=None, rank=0, world=1, part_dims=(0, 0, 0),
              tile_particles=0):
        L = lib()
        rest = f32(rest_pos, (-1, 3))
        d = i32(dist_ij if dist_ij is not None else np.zeros((0, 2)), (-1, 2))
        v = i32(vol_ijkl if vol_ijkl is not None else np.zeros((0, 4)), (-1, 4))
        b = i32(bend_ijkl if bend_ijkl is not None else np.zeros((0, 4)), (-1, 4))
        o = SbPlanOpts(rank, world, (C.c_int32 * 3)(*part_dims), tile_particles)
        h = C.c_void_p()
        check(L.sb_plan_build(ptr(rest), rest.shape[0], ptr(d), d.shape[0], ptr(v), v.shape[0], ptr(b), b.shape[0],
                              C.byref(o), C.byref(h)))
        p = cls(h.value, True)
        p.n = rest.shape[0]
        p.world = world
        return p

    def close(self):
        if self._owned and self._h:
            lib().sb_plan_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def order(self, parity=0):
        L = lib()
        m = L.sb_plan_order_count(self._h)
        t = np.zeros(m, np.uint8); ids = np.zeros(m, np.int32)
        check(L.sb_plan_get_order(self._h, parity, ptr(t), ptr(ids)))
        return t, ids

    def phases(self, parity=0):
        L = lib()
        k = L.sb_plan_phase_count(self._h, parity)
        arr = (SbPhaseInfo * k)()
        check(L.sb_plan_get_phases(self._h, parity, arr))
        return [dict(kind=a.kind, type=a.type, tiling=a.tiling, halo_slot=a.halo_slot, order_begin=a.order_begin,
                     order_end=a.order_end, task_begin=a.task_begin, task_end=a.task_end) for a in arr]

    def tasks(self, parity=0):
        L = lib()
        k = L.sb_plan_task_count(self._h, parity)
        out = np.zeros(k + 1, np.int64)
        check(L.sb_plan_get_tasks(self._h, parity, ptr(out)))
        return out

    def groups(self, parity=0):
        L = lib()
        k = L.sb_plan_group_count(self._h, parity)
        out = np.zeros(k + 1, np.int64)
        check(L.sb_plan_get_groups(self._h, parity, ptr(out)))
        return out

    def phase_task_offsets(self, parity=0):
        ph = self.phases(parity)
        if not ph:
            return np.zeros(1, np.int64)
        return np.array([p["task_begin"] for p in ph] + [ph[-1]["task_end"]], np.int64)

    def owner(self, n):
        out = np.zeros(n, np.int32)
        check(lib().sb_plan_get_owner(self._h, ptr(out)))
        return out

    def local_particles(self):
        L = lib()
        owned = C.c_int64()
        k = L.sb_plan_local_count(self._h, C.byref(owned))
        out = np.zeros(k, np.int32)
        check(L.sb_plan_get_local_particles(self._h, ptr(out)))
        return out, owned.value

    def halo_slot_count(self):
        return lib().sb_plan_halo_slot_count(self._h)

    def halo(self, slot, world):
        """-> {peer: (send_ids, recv_ids)} in caller particle numbering."""
        L = lib()
        sc = np.zeros(world, np.int32); rc = np.zeros(world, np.int32)
        check(L.sb_plan_halo_counts(self._h, slot, ptr(sc), ptr(rc)))
        out = {}
        for peer in range(world):
            if sc[peer] == 0 and rc[peer] == 0:
                continue
            s = np.zeros(sc[peer], np.int32); r = np.zeros(rc[peer], np.int32)
            check(L.sb_plan_get_halo(self._h, slot, peer, ptr(s), ptr(r)))
            out[peer] = (s, r)
        return out

    def local_order_mask(self, parity=0):
        L = lib()
        m = L.sb_plan_order_count(self._h)
        out = np.zeros(m, np.uint8)
        check(L.sb_plan_get_local_order_mask(self._h, parity, ptr(out)))
        return out
