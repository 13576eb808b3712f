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
SB_PARTITION_AUTO, SB_PARTITION_BLOCKS, SB_PARTITION_RCB = 0, 1, 2
SB_PLAN_NO_T2, SB_PLAN_NO_THIRD_LIST, SB_PLAN_NO_CLUSTER_LAYERS, SB_PLAN_NO_MIXED_GROUPS, SB_PLAN_NO_BANK_ORDER, SB_PLAN_NO_TILE_MERGE = 1, 2, 4, 8, 16, 32
SB_TRANSPORT_RCCL, SB_TRANSPORT_PEER = 0, 1
SB_SCHEDULE_AUTO, SB_SCHEDULE_SERIAL_EAGER, SB_SCHEDULE_SERIAL_GRAPH, SB_SCHEDULE_OVERLAP_EAGER, SB_SCHEDULE_OVERLAP_GRAPH = 0, 1, 2, 3, 4
SB_DEBUG_NO_COMM, SB_DEBUG_LOOPBACK = 1, 2
SB_GROUP_WALK = 1
SB_GROUP_WHOLE_MESH = 2
# sb_tuning.flags (include/softbody_debug.h): A/B measurement switches, same bits for every setting
(SB_TUNE_NO_MASS_PALETTE, SB_TUNE_NO_UNIFORM_MASS, SB_TUNE_NO_PALETTE, SB_TUNE_NO_WAVE_ITEMS, SB_TUNE_NO_LANE_PACK, SB_TUNE_NO_COST_ORDER,
 SB_TUNE_NO_FUSED_UNPACK, SB_TUNE_PEER_COARSE, SB_TUNE_NO_LAZY_TICK, SB_TUNE_NO_PACK, SB_TUNE_NO_PEEK, SB_TUNE_NO_KIN_FUSE,
 SB_TUNE_NO_WIDE_SLOTS, SB_TUNE_AUTO_PREFER_OVERLAP, SB_TUNE_NO_AUTO_CALIBRATION) = (1 << k for k in range(15))
SB_ERR_INVALID_ARG, SB_ERR_STATE, SB_ERR_NO_DEVICE, SB_ERR_HIP, SB_ERR_RCCL, SB_ERR_NOMEM, SB_ERR_UNSUPPORTED = \
    -1, -2, -3, -4, -5, -6, -7


class SbDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("part_dims", C.c_int32 * 3),
                ("gravity", C.c_float * 3), ("damping", C.c_float), ("tile_particles", C.c_int32),
                ("use_graph", C.c_int32), ("partition", C.c_int32), ("plan_flags", C.c_uint32),
                ("halo_transport", C.c_int32), ("halo_schedule", C.c_int32), ("debug_flags", C.c_uint32),
                ("reserved", C.c_int32 * 3)]


class SbTuning(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("tile_lanes", C.c_int32), ("quad_lanes", C.c_int32), ("narrow_min_tiles", C.c_int32),
                ("store_through_max_tiles", C.c_int32), ("store_through_large", C.c_int32), ("peek_min_tiles", C.c_int32),
                ("lds_pad_bytes", C.c_int32), ("win_dwords", C.c_int32), ("prev_offset_bytes", C.c_int32), ("reserved", C.c_int32 * 6)]


class SbExchangeTiming(C.Structure):
    _fields_ = [("exchanges", C.c_int64), ("pack_ms", C.c_double), ("transport_ms", C.c_double), ("total_ms", C.c_double), ("exposed_wait_ms", C.c_double)]


class SbStats(C.Structure):
    _fields_ = [("n_particles_owned", C.c_int64), ("n_particles_local", C.c_int64),
                ("n_constraints_local", C.c_int64 * 3), ("n_tilings", C.c_int32), ("n_global_colours", C.c_int32),
                ("n_tiles", C.c_int64 * 2), ("tile_constraints", C.c_int64 * 2), ("constraints_in_tiles", C.c_int64),
                ("constraints_in_global", C.c_int64), ("halo_particles_t1", C.c_int64),
                ("halo_particles_global", C.c_int64), ("device_bytes", C.c_int64),
                ("n_t2_layers", C.c_int64), ("n_t2_tiles", C.c_int64), ("t2_constraints", C.c_int64),
                ("launch_bytes", C.c_int64 * 5), ("partition", C.c_int32), ("halo_peers", C.c_int32),
                ("partition_cost", C.c_int64), ("partition_cost_max", C.c_int64), ("partition_cost_total", C.c_int64),
                ("halo_particles_recv", C.c_int64), ("plan_hash", C.c_uint64), ("halo_schedule", C.c_int32),
                ("halo_unpack_fused", C.c_int32), ("readback_peeks", C.c_int64), ("readback_peek_tiles", C.c_int64), ("ticks_fused", C.c_int64), ("ticks_fused_kinematic", C.c_int64), ("lane_packed_tiles", C.c_int64 * 2),
                ("halo_auto_state", C.c_int32), ("halo_auto_ticks", C.c_int32), ("halo_auto_ms", C.c_double * 2)]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class SbValidateReport(C.Structure):
    _fields_ = [("tiles_checked", C.c_int64), ("groups_checked", C.c_int64), ("constraints_checked", C.c_int64), ("errors", C.c_int64 * 6),
                ("first_stage", C.c_int32), ("first_tile", C.c_int32), ("first_group", C.c_int32), ("first_kind", C.c_int32)]


class SbDomain(C.Structure):
    _fields_ = [("n_global", C.c_int64), ("lo", C.c_double * 3), ("hi", C.c_double * 3), ("spacing", C.c_double), ("fill", C.c_double),
                ("four_vertex_constraints", C.c_int32), ("reserved", C.c_int32)]


class SbPlanOpts(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("part_dims", C.c_int32 * 3), ("tile_particles", C.c_int32),
                ("partition", C.c_int32), ("plan_flags", C.c_uint32), ("domain", C.POINTER(SbDomain)), ("global_id", C.c_void_p)]


class SbRuntimeInfo(C.Structure):
    _fields_ = [("hip_runtime_version", C.c_int32), ("hip_driver_version", C.c_int32), ("rccl_version", C.c_int32),
                ("rccl_header_version", C.c_int32), ("rccl_was_resident", C.c_int32), ("capture_serial_ok", C.c_int32),
                ("capture_overlap_ok", C.c_int32), ("reserved", C.c_int32), ("hip_library", C.c_char * 256),
                ("rccl_library", C.c_char * 256)]


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
    "sb_set_domain": (C.c_int, [_P, C.POINTER(SbDomain), _P, C.c_int32]),
    "sb_domain_from_mesh": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.POINTER(SbDomain)]),
    "sb_domain_window": (C.c_int, [C.POINTER(SbDomain), C.POINTER(SbPlanOpts), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sb_finalize": (C.c_int, [_P]),
    "sb_comm_unique_id": (C.c_int, [_P]),
    "sb_comm_init": (C.c_int, [_P, _P]),
    "sb_peer_mailbox_handle": (C.c_int, [_P, _P]),
    "sb_peer_connect": (C.c_int, [_P, C.c_int32, _P, _P]),
    "sb_step": (C.c_int, [_P, C.c_float, C.c_int32]),
    "sb_get_positions": (C.c_int, [_P, _P, C.c_int32]),
    "sb_get_velocities": (C.c_int, [_P, _P, C.c_int32]),
    "sb_set_kinematic_positions": (C.c_int, [_P, _P, _P, C.c_int32]),
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
    "sb_debug_validate": (C.c_int, [_P, C.c_int32, C.POINTER(SbValidateReport)]),
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
    "sb_plan_get_pair_hashes": (C.c_int, [_P, _P]),
    "sb_plan_get_local_order_mask": (C.c_int, [_P, C.c_int32, _P]),
    "sb_runtime_info": (C.c_int, [C.POINTER(SbRuntimeInfo)]),
    "sb_tuning_default": (None, [C.POINTER(SbTuning)]),
    "sb_set_tuning": (C.c_int, [_P, C.POINTER(SbTuning)]),
    "sb_debug_exchange_timing": (C.c_int, [_P, C.c_int32]),
    "sb_debug_exchange_timing_read": (C.c_int, [_P, C.POINTER(SbExchangeTiming)]),
    "sb_debug_last_words": (C.c_int, [C.c_int32, C.c_char_p, C.c_int64, C.c_int32]),
    # one process driving several devices (include/softbody_group.h)
    "sb_group_create": (C.c_int, [C.POINTER(SbDesc), _P, C.c_int32, C.c_uint32, C.POINTER(_P)]),
    "sb_group_destroy": (C.c_int, [_P]),
    "sb_group_set_particles": (C.c_int, [_P, _P, _P, _P, C.c_int32]),
    "sb_group_set_rest_positions": (C.c_int, [_P, _P, C.c_int32]),
    "sb_group_set_distance_constraints": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float]),
    "sb_group_set_volume_constraints": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float]),
    "sb_group_set_bending_constraints": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float]),
    "sb_group_set_ground_plane": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32]),
    "sb_group_finalize": (C.c_int, [_P]),
    "sb_group_step": (C.c_int, [_P, C.c_float, C.c_int32]),
    "sb_group_get_positions": (C.c_int, [_P, _P, C.c_int32]),
    "sb_group_get_velocities": (C.c_int, [_P, _P, C.c_int32]),
    "sb_group_set_state": (C.c_int, [_P, _P, _P, C.c_int32]),
    "sb_group_set_kinematic_positions": (C.c_int, [_P, _P, _P, C.c_int32]),
    "sb_group_set_render_triangles": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32]),
    "sb_group_set_readback_render_set_only": (C.c_int, [_P, C.c_int32]),
    "sb_group_readback_begin": (C.c_int, [_P]),
    "sb_group_readback_end": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float))]),
    "sb_group_readback_get_normals": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float))]),
    "sb_group_readback_get_render_set": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]),
    "sb_group_synchronize": (C.c_int, [_P]),
    "sb_group_rank_count": (C.c_int32, [_P]),
    "sb_group_get_rank": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
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


# ---- harness conveniences: the A/B switches of tools/ and tests/ as environment variables. The PLUGIN reads none of them;
# they are translated here into the sb_desc / sb_plan_opts fields a host would set (include/softbody.h). --------------------
_PLAN_FLAG_ENV = (("SB_NO_T2", SB_PLAN_NO_T2), ("SB_NO_THIRD_LIST", SB_PLAN_NO_THIRD_LIST),
                  ("SB_NO_CLUSTER_LAYERS", SB_PLAN_NO_CLUSTER_LAYERS), ("SB_NO_MIXED_GROUPS", SB_PLAN_NO_MIXED_GROUPS),
                  ("SB_NO_BANK_ORDER", SB_PLAN_NO_BANK_ORDER), ("SB_NO_TILE_MERGE", SB_PLAN_NO_TILE_MERGE))


def plan_flags_from_env():
    lists = int(os.environ.get("SB_BALANCED_LISTS", "0") or 0) & 3       # irregular meshes: 1..3 balanced extra lists (0 = the default, 2)
    return sum(bit for name, bit in _PLAN_FLAG_ENV if os.environ.get(name)) | (lists << 8)


def halo_transport_from_env():
    return SB_TRANSPORT_PEER if os.environ.get("SB_HALO_TRANSPORT") == "peer" else SB_TRANSPORT_RCCL


def halo_schedule_from_env():
    overlap, graph = bool(os.environ.get("SB_HALO_OVERLAP")), bool(os.environ.get("SB_GRAPH_RCCL"))
    if not overlap and not graph:
        return SB_SCHEDULE_AUTO
    return (SB_SCHEDULE_OVERLAP_GRAPH if graph else SB_SCHEDULE_OVERLAP_EAGER) if overlap else SB_SCHEDULE_SERIAL_GRAPH


def debug_flags_from_env():
    return (SB_DEBUG_NO_COMM if os.environ.get("SB_TEST_NO_COMM") else 0) | (SB_DEBUG_LOOPBACK if os.environ.get("SB_TEST_LOOPBACK") else 0)


# tuning switches (sb_tuning): the SB_* names the A/B scripts of tools/ have always used; the plugin itself reads no environment variable
_TUNE_FLAG_ENV = (("SB_NO_MASS_PALETTE", SB_TUNE_NO_MASS_PALETTE), ("SB_NO_UNIFORM_MASS", SB_TUNE_NO_UNIFORM_MASS), ("SB_NO_PALETTE", SB_TUNE_NO_PALETTE),
                  ("SB_NO_WAVE_ITEMS", SB_TUNE_NO_WAVE_ITEMS), ("SB_NO_LANE_PACK", SB_TUNE_NO_LANE_PACK), ("SB_NO_COST_ORDER", SB_TUNE_NO_COST_ORDER),
                  ("SB_NO_FUSED_UNPACK", SB_TUNE_NO_FUSED_UNPACK), ("SB_PEER_COARSE", SB_TUNE_PEER_COARSE), ("SB_NO_LAZY_TICK", SB_TUNE_NO_LAZY_TICK),
                  ("SB_NO_PACK", SB_TUNE_NO_PACK), ("SB_NO_PEEK", SB_TUNE_NO_PEEK), ("SB_NO_KIN_FUSE", SB_TUNE_NO_KIN_FUSE),
                  ("SB_NO_WIDE_SLOTS", SB_TUNE_NO_WIDE_SLOTS), ("SB_AUTO_PREFER_OVERLAP", SB_TUNE_AUTO_PREFER_OVERLAP),
                  ("SB_NO_AUTO_CALIBRATION", SB_TUNE_NO_AUTO_CALIBRATION))
_TUNE_INT_ENV = (("SB_TILE_LANES", "tile_lanes"), ("SB_QUAD_LANES", "quad_lanes"), ("SB_NARROW_MIN_TILES", "narrow_min_tiles"),
                 ("SB_STORE_THROUGH_MAX_TILES", "store_through_max_tiles"), ("SB_STORE_THROUGH_LARGE", "store_through_large"),
                 ("SB_PEEK_MIN_TILES", "peek_min_tiles"), ("SB_LDS_PAD", "lds_pad_bytes"), ("SB_WIN_DWORDS", "win_dwords"),
                 ("SB_PREV_OFFSET", "prev_offset_bytes"))


def tuning_from_env():
    """sb_tuning filled from the harness' SB_* environment switches, or None when none of them is set (sb_set_tuning is then not called)."""
    t = SbTuning()
    lib().sb_tuning_default(C.byref(t))
    any_set = False
    for name, bit in _TUNE_FLAG_ENV:
        if os.environ.get(name):
            t.flags |= bit; any_set = True
    for name, field in _TUNE_INT_ENV:
        v = os.environ.get(name)
        if v not in (None, ""):
            setattr(t, field, int(v)); any_set = True
    return t if any_set else None


def runtime_info():
    """What the plugin is bound to: HIP runtime + RCCL versions and library files, admitted captured schedules."""
    ri = SbRuntimeInfo()
    check(lib().sb_runtime_info(C.byref(ri)))
    return {"hip_runtime_version": ri.hip_runtime_version, "hip_driver_version": ri.hip_driver_version,
            "rccl_version": ri.rccl_version, "rccl_header_version": ri.rccl_header_version,
            "rccl_was_resident": bool(ri.rccl_was_resident), "capture_serial_ok": bool(ri.capture_serial_ok),
            "capture_overlap_ok": bool(ri.capture_overlap_ok), "hip_library": ri.hip_library.decode(),
            "rccl_library": ri.rccl_library.decode()}


def domain_from_mesh(rest_pos, dist_ij=None, vol_ijkl=None, bend_ijkl=None):
    """sb_domain of a whole mesh (the frame a sharded solver's ranks all pass)."""
    rest = f32(rest_pos, (-1, 3))
    d = i32(dist_ij if dist_ij is not None else np.zeros((0, 2)), (-1, 2))
    v = i32(vol_ijkl if vol_ijkl is not None else np.zeros((0, 4)), (-1, 4))
    b = i32(bend_ijkl if bend_ijkl is not None else np.zeros((0, 4)), (-1, 4))
    out = SbDomain()
    check(lib().sb_domain_from_mesh(ptr(rest), rest.shape[0], ptr(d), d.shape[0], ptr(v), v.shape[0], ptr(b), b.shape[0], C.byref(out)))
    return out


def make_domain(n_global, lo, hi, spacing, four_vertex_constraints=False):
    d = SbDomain()
    d.n_global = int(n_global); d.lo[:] = [float(c) for c in lo]; d.hi[:] = [float(c) for c in hi]
    d.spacing = float(spacing); d.fill = 1.0; d.four_vertex_constraints = 1 if four_vertex_constraints else 0
    return d


def domain_window(domain, rank, world, part_dims=(0, 0, 0), tile_particles=0):
    """(lo, hi) of the box of rest positions rank must hand over under sharded authoring (lo inclusive, hi exclusive)."""
    o = SbPlanOpts(rank, world, (C.c_int32 * 3)(*part_dims), tile_particles, SB_PARTITION_BLOCKS, 0, None, None)
    lo = (C.c_double * 3)(); hi = (C.c_double * 3)()
    check(lib().sb_domain_window(C.byref(domain), C.byref(o), lo, hi))
    return np.array(lo[:]), np.array(hi[:])


class Plan:
    """Read-only view of a planner result (sb_plan_*). Owns the handle unless borrowed from a solver."""

    def __init__(self, handle, owned):
        self._h = C.c_void_p(handle)
        self._owned = owned

    @classmethod
    def build(cls, rest_pos, dist_ij=None, vol_ijkl=None, bend_ijkl=None, rank=0, world=1, part_dims=(0, 0, 0),
              tile_particles=0, partition=SB_PARTITION_AUTO, plan_flags=None, domain=None, global_id=None):
        L = lib()
        rest = f32(rest_pos, (-1, 3))
        d = i32(dist_ij if dist_ij is not None else np.zeros((0, 2)), (-1, 2))
        v = i32(vol_ijkl if vol_ijkl is not None else np.zeros((0, 4)), (-1, 4))
        b = i32(bend_ijkl if bend_ijkl is not None else np.zeros((0, 4)), (-1, 4))
        gid = None if global_id is None else i32(global_id, (-1,))
        o = SbPlanOpts(rank, world, (C.c_int32 * 3)(*part_dims), tile_particles, partition,
                       plan_flags_from_env() if plan_flags is None else plan_flags,
                       C.pointer(domain) if domain is not None else None, ptr(gid))
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

    def pair_hashes(self):
        """(world,) uint64: what this rank and each peer must agree on (sb_plan_get_pair_hashes)."""
        out = np.zeros(self.world, np.uint64)
        check(lib().sb_plan_get_pair_hashes(self._h, ptr(out)))
        return out

    def local_order_mask(self, parity=0):
        L = lib()
        m = L.sb_plan_order_count(self._h)
        out = np.zeros(m, np.uint8)
        check(L.sb_plan_get_local_order_mask(self._h, parity, ptr(out)))
        return out
