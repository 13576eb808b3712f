"""Host-side mirror of the Unity `Softbody : MonoBehaviour` component (csharp/Softbody.cs).

Same member names and call order as the C# component the north star asks for (BASELINE.json:5):
Start() -> sb_create + sb_set_* + sb_finalize, FixedUpdate() -> sb_step + sb_get_positions,
OnDestroy() -> sb_destroy. No reference component exists to copy (/root/reference/README.md:1 is
the whole reference tree). All compute happens in the HIP plugin; there is no CPU path here.
"""
import ctypes as C
import os

import numpy as np

from . import native
from .native import check, f32, i32, ptr


class Softbody:
    # [SerializeField] block of csharp/Softbody.cs
    def __init__(self, mesh, substeps=20, fixed_delta_time=0.02, gravity=(0.0, -9.81, 0.0), damping=0.0,
                 distance_compliance=0.0, volume_compliance=0.0, bending_compliance=0.0, device=0, rank=0, world=1,
                 part_dims=(0, 0, 0), tile_particles=0, use_graph=True, unique_id=None, ground_plane=None, use_gpu=True,
                 partition=native.SB_PARTITION_AUTO, plan_flags=None, halo_transport=None, halo_schedule=None, debug_flags=None, tuning=None):
        self.mesh = mesh
        self.substeps = int(substeps)
        self.fixed_delta_time = float(fixed_delta_time)
        self.gravity = tuple(float(g) for g in gravity)
        self.damping = float(damping)
        self.compliance = (float(distance_compliance), float(volume_compliance), float(bending_compliance))
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        self.part_dims = tuple(int(d) for d in part_dims)
        self.tile_particles = int(tile_particles)
        self.use_graph = bool(use_graph)
        # sb_desc fields a host sets explicitly; None = taken from the harness' environment switches (native.*_from_env:
        # the plugin itself reads no environment variable that changes the plan, the transport or the schedule)
        self.partition = int(partition)
        self.plan_flags = native.plan_flags_from_env() if plan_flags is None else int(plan_flags)
        self.halo_transport = native.halo_transport_from_env() if halo_transport is None else int(halo_transport)
        self.halo_schedule = native.halo_schedule_from_env() if halo_schedule is None else int(halo_schedule)
        self.debug_flags = native.debug_flags_from_env() if debug_flags is None else int(debug_flags)
        self.unique_id = unique_id
        self.tuning = tuning               # native.SbTuning or None = from the harness' SB_* environment switches (A/B runs)
        self.ground_plane = ground_plane   # None or (nx, ny, nz, d): n.x >= d
        # use_gpu=False mirrors the C# component's CPU branch (csharp/Softbody.cs): no solver handle, no device; Start()
        # only fetches the schedule from the host-only planner. The CPU tick itself is C# (SoftbodyCpuSolver.cs); this
        # package has no CPU execution path, so FixedUpdate() refuses -- tests drive the oracle with cpu_schedule().
        self.use_gpu = bool(use_gpu)
        self._cpu_plan = None
        self._h = None
        self._render_set_only = False
        self.vertices = None  # what the C# component assigns to mesh.vertices after each FixedUpdate

    # ---- MonoBehaviour surface ------------------------------------------------------------------
    def Start(self):
        L = native.lib()
        if not self.use_gpu:
            m = self.mesh
            rest = m.rest_pos if m.rest_pos is not None else m.pos
            self._cpu_plan = native.Plan.build(rest, m.dist_ij, m.vol_ijkl, m.bend_ijkl, rank=0, world=1,
                                               tile_particles=self.tile_particles, plan_flags=self.plan_flags)     # sb_plan_build: host only
            self.n = f32(m.pos, (-1, 3)).shape[0]
            self.vertices = f32(m.pos, (-1, 3)).copy()
            return self
        d = native.SbDesc()
        L.sb_desc_default(C.byref(d))
        d.device, d.rank, d.world = self.device, self.rank, self.world
        d.part_dims[:] = self.part_dims
        d.gravity[:] = self.gravity
        d.damping = self.damping
        d.tile_particles = self.tile_particles
        d.use_graph = 1 if self.use_graph else 0
        d.partition, d.plan_flags = self.partition, self.plan_flags
        d.halo_transport, d.halo_schedule, d.debug_flags = self.halo_transport, self.halo_schedule, self.debug_flags
        h = C.c_void_p()
        check(L.sb_create(C.byref(d), C.byref(h)))
        self._h = h
        try:
            tune = native.tuning_from_env() if self.tuning is None else self.tuning
            if tune is not None:      # A/B measurement switches (include/softbody_debug.h); a product host never calls this
                check(L.sb_set_tuning(h, C.byref(tune)))
            self._author(L, h)
        except Exception:
            self.OnDestroy()      # do not leak the handle when authoring / finalize fails
            raise
        return self

    def _author(self, L, h):
        m = self.mesh
        pos = f32(m.pos, (-1, 3)); vel = f32(m.vel, (-1, 3)); w = f32(m.inv_mass, (-1,))
        self.n = pos.shape[0]
        check(L.sb_set_particles(h, ptr(pos), ptr(vel), ptr(w), self.n))
        if m.rest_pos is not None:
            rest = f32(m.rest_pos, (-1, 3))
            check(L.sb_set_rest_positions(h, ptr(rest), self.n))
        if len(m.dist_rest):
            ij = i32(m.dist_ij, (-1, 2)); r = f32(m.dist_rest, (-1,))
            check(L.sb_set_distance_constraints(h, ptr(ij), ptr(r), r.shape[0], self.compliance[0]))
        if len(m.vol_rest):
            q = i32(m.vol_ijkl, (-1, 4)); r = f32(m.vol_rest, (-1,))
            check(L.sb_set_volume_constraints(h, ptr(q), ptr(r), r.shape[0], self.compliance[1]))
        if len(m.bend_rest):
            q = i32(m.bend_ijkl, (-1, 4)); r = f32(m.bend_rest, (-1, 2))
            check(L.sb_set_bending_constraints(h, ptr(q), ptr(r), r.shape[0], self.compliance[2]))
        peer_only = self.halo_transport == native.SB_TRANSPORT_PEER and self.unique_id is None    # the host connects the mailboxes itself
        if self.world > 1 and not (self.debug_flags & native.SB_DEBUG_NO_COMM) and not peer_only:
            assert self.unique_id is not None and len(self.unique_id) == native.SB_UNIQUE_ID_BYTES
            buf = (C.c_uint8 * native.SB_UNIQUE_ID_BYTES)(*self.unique_id)
            check(L.sb_comm_init(h, buf))
        if self.ground_plane is not None:
            check(L.sb_set_ground_plane(h, *[float(c) for c in self.ground_plane], 1))
        if getattr(m, "domain", None) is not None:      # sharded authoring: `m` is this rank's window of a larger mesh
            gid = i32(m.global_id, (-1,))
            check(L.sb_set_domain(h, C.byref(m.domain), ptr(gid), self.n))
        check(L.sb_finalize(h))
        self.vertices = pos.copy()

    def cpu_schedule(self):
        """use_gpu=False: the published order per substep parity, [(types, ids), (types, ids)] -- what the C# component
        hands to SoftbodyCpuSolver (sb_plan_get_order)."""
        if self._cpu_plan is None:
            raise RuntimeError("cpu_schedule() needs use_gpu=False and Start()")
        return [self._cpu_plan.order(parity) for parity in (0, 1)]

    def FixedUpdate(self, readback=True):
        if not self.use_gpu:
            raise RuntimeError("softbodyunity_amd has no CPU execution path: the CPU tick is csharp/SoftbodyCpuSolver.cs "
                               "(tests drive the oracle with cpu_schedule())")
        check(native.lib().sb_step(self._h, self.fixed_delta_time, self.substeps))
        if readback:
            self.get_positions(self.vertices)
        return self.vertices

    def OnDestroy(self):
        if self._cpu_plan is not None:
            self._cpu_plan.close()      # sb_plan_destroy
            self._cpu_plan = None
        if self._h is not None:
            native.lib().sb_destroy(self._h)
            self._h = None

    # ---- plumbing ---------------------------------------------------------------------------------
    def __enter__(self):
        return self.Start()

    def __exit__(self, *a):
        self.OnDestroy()

    def step(self, dt=None, substeps=None):
        check(native.lib().sb_step(self._h, self.fixed_delta_time if dt is None else dt,
                                   self.substeps if substeps is None else substeps))

    def synchronize(self):
        check(native.lib().sb_synchronize(self._h))

    # peer-store halo transport (SB_HALO_TRANSPORT=peer): the host carries the 64-byte mailbox handles between
    # the ranks with whatever channel it has (bench.py: a gloo all_gather) and connects every other rank's mailbox before the first tick
    def peer_mailbox_handle(self):
        out = np.zeros(native.SB_IPC_HANDLE_BYTES, np.uint8)
        check(native.lib().sb_peer_mailbox_handle(self._h, ptr(out)))
        return out

    def peer_connect(self, rank, handle):
        h = np.ascontiguousarray(handle, np.uint8)
        assert h.shape == (native.SB_IPC_HANDLE_BYTES,)
        check(native.lib().sb_peer_connect(self._h, int(rank), ptr(h), None))

    def get_positions(self, out=None):
        out = np.zeros((self.n, 3), np.float32) if out is None else out
        check(native.lib().sb_get_positions(self._h, ptr(out), self.n))
        return out

    def get_velocities(self, out=None):
        out = np.zeros((self.n, 3), np.float32) if out is None else out
        check(native.lib().sb_get_velocities(self._h, ptr(out), self.n))
        return out

    def readback_begin(self):
        check(native.lib().sb_readback_begin(self._h))

    def readback_end(self, normals=False):
        """-> (N,3) float32 view of the plugin's pinned snapshot (valid until the second readback_begin after it);
        with normals=True -> (positions, vertex normals) -- needs set_render_triangles (SPEC.md 6a). In render-set-only
        mode both arrays are compact, (count,3), entry k belonging to particle render_set()[k]."""
        p = C.POINTER(C.c_float)()
        check(native.lib().sb_readback_end(self._h, C.byref(p)))
        rows = self.n
        if self._render_set_only:
            rows = len(self.render_set())
        pos = np.ctypeslib.as_array(p, shape=(rows, 3))
        if not normals:
            return pos
        q = C.POINTER(C.c_float)()
        check(native.lib().sb_readback_get_normals(self._h, C.byref(q)))
        return pos, np.ctypeslib.as_array(q, shape=(rows, 3))

    def set_readback_render_set_only(self, on=True):
        """Readbacks bring only the particles the render triangles use (compact arrays)."""
        check(native.lib().sb_set_readback_render_set_only(self._h, 1 if on else 0))
        self._render_set_only = bool(on)

    def render_set(self):
        """Particle ids (ascending) of the compact readback entries; valid after a finished readback."""
        ids = C.POINTER(C.c_int32)(); cnt = C.c_int32()
        check(native.lib().sb_readback_get_render_set(self._h, C.byref(ids), C.byref(cnt)))
        return np.ctypeslib.as_array(ids, shape=(cnt.value,)) if cnt.value else np.zeros(0, np.int32)

    def set_render_triangles(self, tri):
        """Render triangles (M,3) particle indices: every later readback also brings area-weighted vertex normals."""
        tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
        check(native.lib().sb_set_render_triangles(self._h, tri.ctypes.data_as(C.POINTER(C.c_int32)), tri.shape[0]))

    def set_kinematic_positions(self, ids, pos):
        """Move pinned particles (inverse mass 0) to new positions between two ticks (SPEC.md 2, attachments)."""
        ids = i32(ids); pos = f32(pos, (-1, 3))
        assert pos.shape[0] == ids.shape[0]
        check(native.lib().sb_set_kinematic_positions(self._h, ptr(ids), ptr(pos), int(ids.shape[0])))

    def set_state(self, pos, vel):
        pos = f32(pos, (-1, 3)); vel = f32(vel, (-1, 3))
        check(native.lib().sb_set_state(self._h, ptr(pos), ptr(vel), self.n))

    def owner(self):
        out = np.zeros(self.n, np.int32)
        check(native.lib().sb_get_owner(self._h, ptr(out), self.n))
        return out

    def validate(self, inject_fault=0):
        """Table validator (sb_debug_validate): a GPU kernel re-reads every table the tile kernels read -> report dict."""
        rep = native.SbValidateReport()
        check(native.lib().sb_debug_validate(self._h, int(inject_fault), C.byref(rep)))
        return {"tiles_checked": rep.tiles_checked, "groups_checked": rep.groups_checked, "constraints_checked": rep.constraints_checked,
                "errors": list(rep.errors), "first_stage": rep.first_stage, "first_tile": rep.first_tile, "first_group": rep.first_group,
                "first_kind": rep.first_kind}

    def stats(self):
        st = native.SbStats()
        check(native.lib().sb_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def plan(self):
        h = C.c_void_p()
        check(native.lib().sb_get_plan(self._h, C.byref(h)))
        p = native.Plan(h.value, False)
        p.n = self.n
        p.world = self.world
        return p

    def step_profiled(self, dt=None, substeps=None):
        """One eager tick with HIP events around every launch -> (ms per slot, launches per slot)."""
        k = self.stats()["n_global_colours"] + 5
        ms = np.zeros(k, np.float32); cnt = np.zeros(k, np.int32)
        check(native.lib().sb_step_profiled(self._h, self.fixed_delta_time if dt is None else dt,
                                            self.substeps if substeps is None else substeps, ptr(ms), ptr(cnt), k))
        return ms, cnt

    def profile_begin(self):
        check(native.lib().sb_profile_begin(self._h))

    def exchange_timing(self, enabled):
        """HIP events around every ghost exchange of the eager schedules (sb_debug_exchange_timing); read with exchange_timing_read()."""
        check(native.lib().sb_debug_exchange_timing(self._h, 1 if enabled else 0))

    def exchange_timing_read(self):
        t = native.SbExchangeTiming()
        check(native.lib().sb_debug_exchange_timing_read(self._h, C.byref(t)))
        return {"exchanges": t.exchanges, "pack_ms": t.pack_ms, "transport_ms": t.transport_ms, "total_ms": t.total_ms, "exposed_wait_ms": t.exposed_wait_ms}

    def profile_end(self):
        ms = C.c_float()
        check(native.lib().sb_profile_end(self._h, C.byref(ms)))
        return ms.value


def comm_unique_id():
    buf = (C.c_uint8 * native.SB_UNIQUE_ID_BYTES)()
    check(native.lib().sb_comm_unique_id(buf))
    return bytes(buf)


class SoftbodyGroup:
    """ONE process driving several devices behind one component (include/softbody_group.h; csharp/Softbody.cs with deviceCount > 1):
    the whole mesh authored once, one call per tick, state gathered in the caller's numbering. Same member names as Softbody."""

    def __init__(self, mesh, devices, substeps=20, fixed_delta_time=0.02, gravity=(0.0, -9.81, 0.0), damping=0.0,
                 distance_compliance=0.0, volume_compliance=0.0, bending_compliance=0.0, part_dims=(0, 0, 0), tile_particles=0,
                 use_graph=True, ground_plane=None, partition=native.SB_PARTITION_AUTO, plan_flags=None,
                 halo_transport=native.SB_TRANSPORT_RCCL, halo_schedule=native.SB_SCHEDULE_AUTO, debug_flags=0, walk=False, tuning=None, whole_mesh=False):
        self.mesh = mesh
        self.whole_mesh = bool(whole_mesh)
        self.devices = [int(d) for d in devices]
        self.substeps, self.fixed_delta_time = int(substeps), float(fixed_delta_time)
        self.gravity, self.damping = tuple(float(g) for g in gravity), float(damping)
        self.compliance = (float(distance_compliance), float(volume_compliance), float(bending_compliance))
        self.part_dims, self.tile_particles, self.use_graph = tuple(int(d) for d in part_dims), int(tile_particles), bool(use_graph)
        self.ground_plane = ground_plane
        self.partition = int(partition)
        self.plan_flags = native.plan_flags_from_env() if plan_flags is None else int(plan_flags)
        self.halo_transport, self.halo_schedule, self.debug_flags = int(halo_transport), int(halo_schedule), int(debug_flags)
        self.walk = bool(walk)
        self.tuning = tuning
        self._g = None
        self._render_set_only = False
        self.vertices = None

    def Start(self):
        L = native.lib()
        d = native.SbDesc()
        L.sb_desc_default(C.byref(d))
        d.part_dims[:] = self.part_dims
        d.gravity[:] = self.gravity
        d.damping, d.tile_particles, d.use_graph = self.damping, self.tile_particles, 1 if self.use_graph else 0
        d.partition, d.plan_flags = self.partition, self.plan_flags
        d.halo_transport, d.halo_schedule, d.debug_flags = self.halo_transport, self.halo_schedule, self.debug_flags
        devs = (C.c_int32 * len(self.devices))(*self.devices)
        g = C.c_void_p()
        check(L.sb_group_create(C.byref(d), devs, len(self.devices), (native.SB_GROUP_WALK if self.walk else 0) | (native.SB_GROUP_WHOLE_MESH if self.whole_mesh else 0), C.byref(g)))
        self._g = g
        try:
            tune = native.tuning_from_env() if self.tuning is None else self.tuning
            if tune is not None:
                for r in range(len(self.devices)):
                    check(L.sb_set_tuning(self._rank_handle(r), C.byref(tune)))
            m = self.mesh
            pos = f32(m.pos, (-1, 3)); vel = f32(m.vel, (-1, 3)); w = f32(m.inv_mass, (-1,))
            self.n = pos.shape[0]
            check(L.sb_group_set_particles(g, ptr(pos), ptr(vel), ptr(w), self.n))
            if m.rest_pos is not None:
                rest = f32(m.rest_pos, (-1, 3))
                check(L.sb_group_set_rest_positions(g, ptr(rest), self.n))
            if len(m.dist_rest):
                ij = i32(m.dist_ij, (-1, 2)); r = f32(m.dist_rest, (-1,))
                check(L.sb_group_set_distance_constraints(g, ptr(ij), ptr(r), r.shape[0], self.compliance[0]))
            if len(m.vol_rest):
                q = i32(m.vol_ijkl, (-1, 4)); r = f32(m.vol_rest, (-1,))
                check(L.sb_group_set_volume_constraints(g, ptr(q), ptr(r), r.shape[0], self.compliance[1]))
            if len(m.bend_rest):
                q = i32(m.bend_ijkl, (-1, 4)); r = f32(m.bend_rest, (-1, 2))
                check(L.sb_group_set_bending_constraints(g, ptr(q), ptr(r), r.shape[0], self.compliance[2]))
            if self.ground_plane is not None:
                check(L.sb_group_set_ground_plane(g, *[float(c) for c in self.ground_plane], 1))
            check(L.sb_group_finalize(g))
            self.vertices = pos.copy()
        except Exception:
            self.OnDestroy()
            raise
        return self

    def FixedUpdate(self, readback=True):
        check(native.lib().sb_group_step(self._g, self.fixed_delta_time, self.substeps))
        if readback:
            self.get_positions(self.vertices)
        return self.vertices

    def OnDestroy(self):
        if self._g is not None:
            native.lib().sb_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self.Start()

    def __exit__(self, *a):
        self.OnDestroy()

    def step(self, dt=None, substeps=None):
        check(native.lib().sb_group_step(self._g, self.fixed_delta_time if dt is None else dt, self.substeps if substeps is None else substeps))

    def synchronize(self):
        check(native.lib().sb_group_synchronize(self._g))

    def get_positions(self, out=None):
        out = np.zeros((self.n, 3), np.float32) if out is None else out
        check(native.lib().sb_group_get_positions(self._g, ptr(out), self.n))
        return out

    def get_velocities(self, out=None):
        out = np.zeros((self.n, 3), np.float32) if out is None else out
        check(native.lib().sb_group_get_velocities(self._g, ptr(out), self.n))
        return out

    def set_state(self, pos, vel):
        pos = f32(pos, (-1, 3)); vel = f32(vel, (-1, 3))
        check(native.lib().sb_group_set_state(self._g, ptr(pos), ptr(vel), self.n))

    def set_kinematic_positions(self, ids, pos):
        ids = i32(ids); pos = f32(pos, (-1, 3))
        assert pos.shape[0] == ids.shape[0]
        check(native.lib().sb_group_set_kinematic_positions(self._g, ptr(ids), ptr(pos), int(ids.shape[0])))

    def set_render_triangles(self, tri):
        tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
        check(native.lib().sb_group_set_render_triangles(self._g, tri.ctypes.data_as(C.POINTER(C.c_int32)), tri.shape[0]))

    def set_readback_render_set_only(self, on=True):
        check(native.lib().sb_group_set_readback_render_set_only(self._g, 1 if on else 0))
        self._render_set_only = bool(on)

    def readback_begin(self):
        check(native.lib().sb_group_readback_begin(self._g))

    def render_set(self):
        ids = C.POINTER(C.c_int32)(); cnt = C.c_int32()
        check(native.lib().sb_group_readback_get_render_set(self._g, C.byref(ids), C.byref(cnt)))
        return np.ctypeslib.as_array(ids, shape=(cnt.value,)) if cnt.value else np.zeros(0, np.int32)

    def readback_end(self, normals=False):
        p = C.POINTER(C.c_float)()
        check(native.lib().sb_group_readback_end(self._g, C.byref(p)))
        rows = len(self.render_set()) if self._render_set_only else self.n
        pos = np.ctypeslib.as_array(p, shape=(rows, 3))
        if not normals:
            return pos
        q = C.POINTER(C.c_float)()
        check(native.lib().sb_group_readback_get_normals(self._g, C.byref(q)))
        return pos, np.ctypeslib.as_array(q, shape=(rows, 3))

    def _rank_handle(self, r):
        h = C.c_void_p()
        check(native.lib().sb_group_get_rank(self._g, int(r), C.byref(h)))
        return h

    def rank(self, r):
        """Borrowed Softbody view of rank r (inspection only: stats, owner, validate, plan)."""
        sb = Softbody.__new__(Softbody)
        sb._h = self._rank_handle(r)
        sb.world, sb.rank = len(self.devices), int(r)
        st = native.SbStats()
        check(native.lib().sb_get_stats(sb._h, C.byref(st)))
        sb.n = None       # (the rank's own numbering may be its window's: use stats / validate, which need no n)
        sb._cpu_plan = None
        return sb
