"""ctypes wrapper around oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg. The product
package (softbodyunity_amd) must never import this module. PARITY UNPINNED (see oracle.c header):
the reference tree is /root/reference/README.md:1 only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class OrcParams(C.Structure):
    _fields_ = [("gravity", C.c_float * 3), ("damping", C.c_float), ("compliance", C.c_float * 3),
                ("plane", C.c_float * 4), ("plane_on", C.c_int32)]


class OrcScalars(C.Structure):
    _fields_ = [("h", C.c_float), ("inv_h", C.c_float), ("hg", C.c_float * 3), ("kd", C.c_float),
                ("at_d", C.c_float), ("at_v", C.c_float), ("at_b", C.c_float)]


class OrcConstraints(C.Structure):
    _fields_ = [("dist_ij", C.c_void_p), ("dist_rest", C.c_void_p), ("m_d", C.c_int32),
                ("vol_ijkl", C.c_void_p), ("vol_rest6", C.c_void_p), ("m_v", C.c_int32),
                ("bend_ijkl", C.c_void_p), ("bend_rest", C.c_void_p), ("m_b", C.c_int32)]


class OrcSchedule(C.Structure):
    _fields_ = [("order_type", C.c_void_p), ("order_id", C.c_void_p), ("phase_task_off", C.c_void_p),
                ("n_phases", C.c_int32), ("task_off", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_step.restype = None
        _lib.orc_step_tasks.restype = None
        _lib.orc_project_range.restype = None
        _lib.orc_integrate.restype = None
        _lib.orc_velocity.restype = None
        _lib.orc_scalars_for.restype = None
        _lib.orc_collide.restype = None
        _lib.orc_vertex_normals.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    """Sequential fp32 restatement of SPEC.md. State lives in numpy arrays owned by this object."""

    def __init__(self, pos, vel, inv_mass, gravity=(0.0, -9.81, 0.0), damping=0.0):
        self.x = _f32(pos, (-1, 3)).copy()
        self.n = self.x.shape[0]
        self.v = (np.zeros_like(self.x) if vel is None else _f32(vel, (-1, 3)).copy())
        self.w = _f32(inv_mass, (-1,)).copy()
        assert self.v.shape == self.x.shape and self.w.shape[0] == self.n
        self.xprev = np.zeros_like(self.x)
        self.params = OrcParams()
        self.params.gravity[:] = [float(g) for g in gravity]
        self.params.damping = float(damping)
        self.params.compliance[:] = [0.0, 0.0, 0.0]
        self.dist_ij = np.zeros((0, 2), np.int32); self.dist_rest = np.zeros(0, np.float32)
        self.vol_ijkl = np.zeros((0, 4), np.int32); self.vol_rest6 = np.zeros(0, np.float32)
        self.bend_ijkl = np.zeros((0, 4), np.int32); self.bend_rest = np.zeros((0, 2), np.float32)
        self.order_type = None
        self.order_id = None
        self.sched = [(None, None, None), (None, None, None)]

    def set_distance(self, ij, rest, compliance=0.0):
        self.dist_ij = _i32(ij).reshape(-1, 2).copy(); self.dist_rest = _f32(rest, (-1,)).copy()
        self.params.compliance[0] = float(compliance)

    def set_volume(self, ijkl, rest_vol, compliance=0.0):
        self.vol_ijkl = _i32(ijkl).reshape(-1, 4).copy()
        # SPEC.md §5: stored rest value R6 = 6*V0 as one f32 product
        self.vol_rest6 = (np.float32(6.0) * _f32(rest_vol, (-1,))).astype(np.float32)
        self.params.compliance[1] = float(compliance)

    def set_bending(self, ijkl, rest_cs, compliance=0.0):
        """rest_cs: (m,2) = (cos phi0, sin phi0) per hinge (SPEC.md §6)."""
        self.bend_ijkl = _i32(ijkl).reshape(-1, 4).copy(); self.bend_rest = _f32(rest_cs, (-1, 2)).copy()
        self.params.compliance[2] = float(compliance)

    def set_ground_plane(self, normal, d, enabled=True):
        """SPEC.md §2 step 2b: n.x >= d."""
        self.params.plane[:] = [float(normal[0]), float(normal[1]), float(normal[2]), float(d)]
        self.params.plane_on = 1 if enabled else 0

    def set_kinematic_positions(self, ids, pos):
        """SPEC.md §2, kinematic particles: between two ticks x <- target for particles with w = 0 (others are refused)."""
        ids = np.asarray(ids, np.int64)
        if (self.w[ids] != 0).any():
            raise ValueError("only particles with inverse mass 0 are kinematic")
        self.x[ids] = np.asarray(pos, np.float32).reshape(-1, 3)

    def collide(self):
        lib().orc_collide(_p(self.x), _p(self.w), C.c_int(self.n), C.byref(self.params))

    def set_order(self, order_type, order_id, phase_task_off=None, task_off=None, parity=None):
        """Schedule published by the planner (SPEC.md §3). parity=None sets both parities to the same order."""
        ot = np.ascontiguousarray(order_type, dtype=np.uint8)
        oi = _i32(order_id)
        total = len(self.dist_rest) + len(self.vol_rest6) + len(self.bend_rest)
        assert ot.shape[0] == total and oi.shape[0] == total
        tasks = None
        if phase_task_off is not None:
            tasks = (np.ascontiguousarray(phase_task_off, dtype=np.int64), np.ascontiguousarray(task_off, dtype=np.int64))
        for p in ((0, 1) if parity is None else (parity,)):
            self.sched[p] = (ot, oi, tasks)
        # single-order view used by project_range
        self.order_type, self.order_id = self.sched[0][0], self.sched[0][1]

    def use_parity(self, parity):
        """Select which parity's order project_range() indexes."""
        self.order_type, self.order_id = self.sched[parity][0], self.sched[parity][1]

    def _sched(self):
        arr = (OrcSchedule * 2)()
        for p in range(2):
            ot, oi, tasks = self.sched[p]
            arr[p].order_type = _p(ot); arr[p].order_id = _p(oi)
            if tasks is not None:
                arr[p].phase_task_off = _p(tasks[0]); arr[p].n_phases = len(tasks[0]) - 1; arr[p].task_off = _p(tasks[1])
        return arr

    def _cons(self):
        c = OrcConstraints()
        c.dist_ij = _p(self.dist_ij); c.dist_rest = _p(self.dist_rest); c.m_d = len(self.dist_rest)
        c.vol_ijkl = _p(self.vol_ijkl); c.vol_rest6 = _p(self.vol_rest6); c.m_v = len(self.vol_rest6)
        c.bend_ijkl = _p(self.bend_ijkl); c.bend_rest = _p(self.bend_rest); c.m_b = len(self.bend_rest)
        return c

    def scalars(self, dt, substeps):
        s = OrcScalars()
        lib().orc_scalars_for(C.byref(self.params), C.c_float(dt), C.c_int(substeps), C.byref(s))
        return s

    def step(self, dt, substeps, parallel=False):
        c = self._cons()
        sch = self._sched()
        if parallel:
            assert self.sched[0][2] is not None and self.sched[1][2] is not None, "set_order(..., phase_task_off, task_off) first"
            fn = lib().orc_step_tasks
        else:
            fn = lib().orc_step
        fn(_p(self.x), _p(self.v), _p(self.w), _p(self.xprev), C.c_int(self.n), C.byref(c), sch,
           C.byref(self.params), C.c_float(dt), C.c_int(substeps))

    # fine-grained entry points used by the partitioned-oracle (halo) tests
    def integrate(self, s):
        lib().orc_integrate(_p(self.x), _p(self.xprev), _p(self.v), _p(self.w), C.c_int(self.n), C.byref(s))

    def velocity(self, s):
        lib().orc_velocity(_p(self.x), _p(self.xprev), _p(self.v), C.c_int(self.n), C.byref(s))

    def project_range(self, s, begin, end):
        c = self._cons()
        lib().orc_project_range(_p(self.x), _p(self.w), C.byref(c), _p(self.order_type), _p(self.order_id),
                                C.c_int64(begin), C.c_int64(end), C.byref(s))


def vertex_normals(x, tri):
    """SPEC.md §6a: (N,3) area-weighted vertex normals of the (M,3) triangle list on positions x."""
    x = _f32(x, (-1, 3)); tri = _i32(tri).reshape(-1, 3)
    out = np.zeros_like(x)
    lib().orc_vertex_normals(_p(x), C.c_int(x.shape[0]), _p(tri), C.c_int64(tri.shape[0]), _p(out))
    return out


def parity_error(x, x_ref, x0):
    """SURVEY §8c metric: max_i ||x_i - xref_i||_2 / diag(bbox(x0)). Returns (rel, max_abs, bitwise)."""
    x = np.asarray(x, np.float64).reshape(-1, 3); r = np.asarray(x_ref, np.float64).reshape(-1, 3)
    x0 = np.asarray(x0, np.float64).reshape(-1, 3)
    diag = float(np.linalg.norm(x0.max(0) - x0.min(0)))
    err = np.linalg.norm(x - r, axis=1)
    bitwise = bool(np.array_equal(np.asarray(x, np.float32).view(np.uint32), np.asarray(x_ref, np.float32).view(np.uint32)))
    return float(err.max() / max(diag, 1e-30)), float(np.abs(x - r).max()), bitwise
