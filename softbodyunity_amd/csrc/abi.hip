// abi.hip — lifecycle, authoring, communicator / mailbox set-up, sb_finalize, synchronisation and statistics
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"

using namespace sbi;

extern "C" {

void sb_desc_default(sb_desc *d) {
    if (!d) return;
    std::memset(d, 0, sizeof(*d));
    d->world = 1;
    d->gravity[1] = -9.81f;
    d->tile_particles = 0;         // automatic: 512, or 256 when the mesh has 4-vertex constraints (sb_finalize)
    d->use_graph = 1;
}

int sb_create(const sb_desc *desc, sb_solver **out) {
    if (!desc || !out) return fail(SB_ERR_INVALID_ARG, "sb_create: null argument");
    *out = nullptr;
    return guarded([&]() -> int {
        sb_desc d = *desc;
        if (d.world <= 0) d.world = 1;
        if (d.rank < 0 || d.rank >= d.world) return fail(SB_ERR_INVALID_ARG, "sb_create: rank out of range");
        if (!(d.damping >= 0.0f)) return fail(SB_ERR_INVALID_ARG, "sb_create: damping must be >= 0");
        if (d.tile_particles > sbp::kMaxTileLocal) return fail(SB_ERR_INVALID_ARG, "sb_create: tile_particles too large");
        if (d.partition < SB_PARTITION_AUTO || d.partition > SB_PARTITION_RCB) return fail(SB_ERR_INVALID_ARG, "sb_create: partition must be SB_PARTITION_AUTO, _BLOCKS or _RCB");
        if (d.plan_flags & ~kPlanFlagsAll) return fail(SB_ERR_INVALID_ARG, "sb_create: unknown bit in plan_flags");
        if (d.halo_transport != SB_TRANSPORT_RCCL && d.halo_transport != SB_TRANSPORT_PEER) return fail(SB_ERR_INVALID_ARG, "sb_create: halo_transport must be SB_TRANSPORT_RCCL or _PEER");
        if (d.halo_schedule < SB_SCHEDULE_AUTO || d.halo_schedule > SB_SCHEDULE_OVERLAP_GRAPH) return fail(SB_ERR_INVALID_ARG, "sb_create: halo_schedule must be one of SB_SCHEDULE_*");
        if (d.debug_flags & ~(SB_DEBUG_NO_COMM | SB_DEBUG_LOOPBACK)) return fail(SB_ERR_INVALID_ARG, "sb_create: unknown bit in debug_flags");
        if (d.reserved[0] || d.reserved[1] || d.reserved[2]) return fail(SB_ERR_INVALID_ARG, "sb_create: reserved fields must be 0 (zero-initialise sb_desc or call sb_desc_default)");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            return fail(SB_ERR_NO_DEVICE, "sb_create: no HIP device (this plugin has no CPU path)");
        if (d.device < 0 || d.device >= ndev) return fail(SB_ERR_NO_DEVICE, "sb_create: device ordinal out of range");
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, d.device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SB_ERR_NO_DEVICE, std::string("sb_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        HIP_CHECK(hipSetDevice(d.device));
        auto s = std::make_unique<sb_solver>();
        s->desc = d;
        s->peer.enabled = d.world > 1 && d.halo_transport == SB_TRANSPORT_PEER;
        s->loopback = d.world > 1 && (d.debug_flags & SB_DEBUG_LOOPBACK) != 0;
        HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreate(&s->ev0));
        HIP_CHECK(hipEventCreate(&s->ev1));
        *out = s.release();
        return SB_OK;
    });
}

void sb_tuning_default(sb_tuning *t) {
    if (!t) return;
    std::memset(t, 0, sizeof(*t));
    t->store_through_max_tiles = -1;
    t->peek_min_tiles = -1;
}

// A/B measurements only (softbody_debug.h): what used to be 23 environment variables of the plugin. Same bits for every setting.
int sb_set_tuning(sb_solver *s, const sb_tuning *t) {
    if (!s || !t) return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: null argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_tuning after sb_finalize");
    constexpr uint32_t kAll = (SB_TUNE_NO_AUTO_CALIBRATION << 1) - 1u;
    if (t->flags & ~kAll) return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: unknown bit in flags");
    for (int32_t r : t->reserved) if (r) return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: reserved fields must be 0 (sb_tuning_default)");
    if (t->tile_lanes != 0 && t->tile_lanes != 128 && t->tile_lanes != 256 && t->tile_lanes != 512) return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: tile_lanes must be 0, 128, 256 or 512");
    if (t->quad_lanes != 0 && t->quad_lanes != 256 && t->quad_lanes != 512) return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: quad_lanes must be 0, 256 or 512");
    if (t->prev_offset_bytes < 0 || (t->prev_offset_bytes & 15)) return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: prev_offset_bytes must be a non-negative multiple of 16");
    if (t->narrow_min_tiles < 0 || t->store_through_max_tiles < -1 || t->peek_min_tiles < -1 || t->lds_pad_bytes < 0 || t->win_dwords < 0 || (t->store_through_large & ~3))
        return fail(SB_ERR_INVALID_ARG, "sb_set_tuning: value out of range");
    s->tune_flags = t->flags;
    s->lazy_tick = !(t->flags & SB_TUNE_NO_LAZY_TICK);
    s->pack_tiles = !(t->flags & SB_TUNE_NO_PACK);
    s->peek_enabled = !(t->flags & SB_TUNE_NO_PEEK);
    s->kin_fuse = !(t->flags & SB_TUNE_NO_KIN_FUSE);
    s->tile_lanes = t->tile_lanes;
    s->quad_lanes = t->quad_lanes ? t->quad_lanes : 512;
    if (t->narrow_min_tiles > 0) s->narrow_min_tiles = t->narrow_min_tiles;
    if (t->store_through_max_tiles >= 0) s->store_through_max_tiles = t->store_through_max_tiles;
    s->store_through_large = t->store_through_large;
    if (t->peek_min_tiles >= 0) s->peek_min_tiles = t->peek_min_tiles;
    s->lds_pad = (size_t)t->lds_pad_bytes;
    s->win_dwords_cap = t->win_dwords;
    s->prev_offset_bytes = t->prev_offset_bytes;
    return SB_OK;
}

int sb_destroy(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_destroy: null handle");
    (void)hipSetDevice(s->desc.device);
    delete s;          // (the destructor drains the solver's streams first)
    return SB_OK;
}

int sb_set_particles(sb_solver *s, const float *pos, const float *vel, const float *inv_mass, int32_t n) {
    if (!s || !pos || !inv_mass || n <= 0) return fail(SB_ERR_INVALID_ARG, "sb_set_particles: bad argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_particles after sb_finalize");
    return guarded([&]() -> int {
        for (int32_t p = 0; p < n; ++p) if (!(inv_mass[p] >= 0.0f)) return fail(SB_ERR_INVALID_ARG, "sb_set_particles: inverse mass must be >= 0");
        s->n = n;
        s->pos.assign(pos, pos + 3 * (size_t)n);
        if (vel) s->vel.assign(vel, vel + 3 * (size_t)n); else s->vel.assign(3 * (size_t)n, 0.0f);
        s->invm.assign(inv_mass, inv_mass + n);
        return SB_OK;
    });
}

int sb_set_rest_positions(sb_solver *s, const float *rest, int32_t n) {
    if (!s || !rest) return fail(SB_ERR_INVALID_ARG, "sb_set_rest_positions: bad argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_rest_positions after sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_rest_positions: n differs from sb_set_particles");
    return guarded([&]() -> int { s->rest.assign(rest, rest + 3 * (size_t)n); return SB_OK; });
}

static int set_cons(sb_solver *s, const char *who, const int32_t *idx, const float *rest, int32_t m, float compliance,
                    int type, int nv, int nrest) {
    if (!s || m < 0 || (m > 0 && (!idx || !rest))) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": bad argument");
    if (s->finalized) return fail(SB_ERR_STATE, std::string(who) + " after sb_finalize");
    if (s->n <= 0) return fail(SB_ERR_STATE, std::string(who) + " before sb_set_particles");
    if (!(compliance >= 0.0f)) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": compliance must be >= 0");
    return guarded([&]() -> int {
        std::atomic<bool> bad{false};
        sbp::parallel_for_chunks((int64_t)m * nv, 1 << 22, [&](int64_t, int64_t kb, int64_t ke) {
            for (int64_t k = kb; k < ke; ++k) if (idx[k] < 0 || idx[k] >= s->n) { bad = true; return; }
        });
        if (bad) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": particle index out of range");
        std::vector<int32_t> &I = type == 0 ? s->dist_ij : (type == 1 ? s->vol_ijkl : s->bend_ijkl);
        std::vector<float> &R = type == 0 ? s->dist_rest : (type == 1 ? s->vol_rest : s->bend_rest);
        I.assign(idx, idx + (size_t)m * nv);
        R.assign(rest, rest + (size_t)m * nrest);
        s->compliance[type] = compliance;
        return SB_OK;
    });
}

int sb_set_distance_constraints(sb_solver *s, const int32_t *ij, const float *rest_len, int32_t m, float compliance) {
    return set_cons(s, "sb_set_distance_constraints", ij, rest_len, m, compliance, 0, 2, 1);
}
int sb_set_volume_constraints(sb_solver *s, const int32_t *ijkl, const float *rest_vol, int32_t m, float compliance) {
    return set_cons(s, "sb_set_volume_constraints", ijkl, rest_vol, m, compliance, 1, 4, 1);
}
int sb_set_bending_constraints(sb_solver *s, const int32_t *ijkl, const float *rest_cs, int32_t m, float compliance) {
    return set_cons(s, "sb_set_bending_constraints", ijkl, rest_cs, m, compliance, 2, 4, 2);
}

int sb_set_ground_plane(sb_solver *s, float nx, float ny, float nz, float d, int32_t enabled) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_set_ground_plane: null handle");
    if (!(nx == nx) || !(ny == ny) || !(nz == nz) || !(d == d)) return fail(SB_ERR_INVALID_ARG, "sb_set_ground_plane: NaN");
    s->plane[0] = nx; s->plane[1] = ny; s->plane[2] = nz; s->plane[3] = d;
    s->plane_on = enabled ? 1 : 0;
    return SB_OK;   // picked up by the next sb_step (tick parameters are re-uploaded when they change)
}

int sb_comm_unique_id(uint8_t out_id[SB_UNIQUE_ID_BYTES]) {
    if (!out_id) return fail(SB_ERR_INVALID_ARG, "sb_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) <= SB_UNIQUE_ID_BYTES, "unique id size");
    return guarded([&]() -> int {
        ncclUniqueId id;
        NCCL_CHECK(rccl().GetUniqueId(&id));
        std::memset(out_id, 0, SB_UNIQUE_ID_BYTES);
        std::memcpy(out_id, &id, sizeof(id));
        return SB_OK;
    });
}

int sb_comm_init(sb_solver *s, const uint8_t id_bytes[SB_UNIQUE_ID_BYTES]) {
    if (!s || !id_bytes) return fail(SB_ERR_INVALID_ARG, "sb_comm_init: null");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_comm_init after sb_finalize");
    if (s->comm) return fail(SB_ERR_STATE, "sb_comm_init called twice");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        ncclUniqueId id;
        std::memcpy(&id, id_bytes, sizeof(id));
        if (s->loopback) {
            // pipeline test on one GPU (SB_DEBUG_LOOPBACK): a communicator of size 1, every peer replaced by this rank itself
            NCCL_CHECK(rccl().CommInitRank(&s->comm, 1, id, 0));
        } else {
            NCCL_CHECK(rccl().CommInitRank(&s->comm, s->desc.world, id, s->desc.rank));
        }
        return SB_OK;
    });
}

int sb_peer_mailbox_handle(sb_solver *s, uint8_t out_handle[SB_IPC_HANDLE_BYTES]) {
    if (!s || !out_handle) return fail(SB_ERR_INVALID_ARG, "sb_peer_mailbox_handle: null argument");
    if (!s->finalized || !s->peer.mailbox) return fail(SB_ERR_STATE, "sb_peer_mailbox_handle: needs a finalized world > 1 solver with SB_HALO_TRANSPORT=peer");
    static_assert(sizeof(hipIpcMemHandle_t) <= SB_IPC_HANDLE_BYTES, "ipc handle size");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        hipIpcMemHandle_t h;
        HIP_CHECK(hipIpcGetMemHandle(&h, s->peer.mailbox));
        std::memset(out_handle, 0, SB_IPC_HANDLE_BYTES);
        std::memcpy(out_handle, &h, sizeof(h));
        return SB_OK;
    });
}

int sb_peer_connect(sb_solver *s, int32_t rank, const uint8_t handle[SB_IPC_HANDLE_BYTES], sb_solver *same_process_peer) {
    if (!s || (!handle && !same_process_peer)) return fail(SB_ERR_INVALID_ARG, "sb_peer_connect: null argument");
    if (!s->finalized || !s->peer.mailbox) return fail(SB_ERR_STATE, "sb_peer_connect: needs a finalized world > 1 solver with SB_HALO_TRANSPORT=peer");
    if (rank < 0 || rank >= s->desc.world || rank == s->desc.rank) return fail(SB_ERR_INVALID_ARG, "sb_peer_connect: bad rank");
    if (s->peer.remote[(size_t)rank]) return fail(SB_ERR_STATE, "sb_peer_connect: rank already connected");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        if (same_process_peer) {             // the peer's handle lives in this process: its pointer is directly usable
            if (!same_process_peer->peer.mailbox || same_process_peer->desc.rank != rank) return fail(SB_ERR_INVALID_ARG, "sb_peer_connect: the given solver is not that rank");
            s->peer.remote[(size_t)rank] = same_process_peer->peer.mailbox;
        } else {
            hipIpcMemHandle_t h;
            std::memcpy(&h, handle, sizeof(h));
            void *p = nullptr;
            HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            s->peer.remote[(size_t)rank] = (uint32_t *)p; s->peer.opened[(size_t)rank] = 1;
        }
        s->peer.linked = false;
        return SB_OK;
    });
}

int sb_set_domain(sb_solver *s, const sb_domain *domain, const int32_t *global_id, int32_t n) {
    if (!s || !domain || !global_id) return fail(SB_ERR_INVALID_ARG, "sb_set_domain: null argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_domain after sb_finalize");
    if (n != s->n || n <= 0) return fail(SB_ERR_INVALID_ARG, "sb_set_domain: n differs from sb_set_particles");
    if (domain->n_global < n || !(domain->spacing > 0) || domain->reserved != 0) return fail(SB_ERR_INVALID_ARG, "sb_set_domain: bad domain");
    return guarded([&]() -> int {
        s->domain = *domain;
        s->global_id.assign(global_id, global_id + n);
        s->sharded = true;
        return SB_OK;
    });
}

int sb_domain_from_mesh(const float *rest, int32_t n, const int32_t *dist_ij, int32_t m_d, const int32_t *vol, int32_t m_v,
                        const int32_t *bend, int32_t m_b, sb_domain *out) {
    if (!rest || !out || n <= 0 || m_d < 0 || m_v < 0 || m_b < 0) return fail(SB_ERR_INVALID_ARG, "sb_domain_from_mesh: bad argument");
    return guarded([&]() -> int {
        sbp::Domain D;
        sbp::compute_domain(make_input(rest, n, dist_ij, m_d, vol, m_v, bend, m_b), D);
        std::memset(out, 0, sizeof(*out));
        out->n_global = D.n_global; out->spacing = D.ell; out->fill = D.fill;
        for (int a = 0; a < 3; ++a) { out->lo[a] = D.lo[a]; out->hi[a] = D.hi[a]; }
        out->four_vertex_constraints = (m_v + m_b > 0) ? 1 : 0;
        return SB_OK;
    });
}

int sb_domain_window(const sb_domain *domain, const sb_plan_opts *opts, double lo_out[3], double hi_out[3]) {
    if (!domain || !opts || !lo_out || !hi_out) return fail(SB_ERR_INVALID_ARG, "sb_domain_window: null argument");
    if (opts->world < 1 || opts->rank < 0 || opts->rank >= opts->world) return fail(SB_ERR_INVALID_ARG, "sb_domain_window: bad rank / world");
    return guarded([&]() -> int {
        const sbp::Opts o = plan_opts(opts->rank, opts->world, opts->part_dims, opts->tile_particles, SB_PARTITION_BLOCKS, 0u, 0, 0, domain);
        int clo[3], chi[3];
        sbp::rank_window(o.domain, o, clo, chi, lo_out, hi_out);
        return SB_OK;
    });
}

}  // extern "C"  (the phases of sb_finalize are shared with group.hip)

namespace sbi {

// Phase A of sb_finalize: everything a rank does ALONE -- resolve the schedule and plan (finalize_plan: host work only), then build and
// upload the device tables (finalize_device). No collective, no look at a neighbour. Throws / returns an error code like any guarded
// body; the caller decides what a failure means for the other ranks (finalize_agree). The group host (group.hip) runs the two halves
// apart: it compares the ranks' plans between them, and plans again from the whole mesh where windows did not reproduce it.
int finalize_local(sb_solver *s) {
    const int rc = finalize_plan(s);
    return rc ? rc : finalize_device(s);
}

// A rank handed a window goes back to the state before authoring (the group host: the windows' plans did not fit together).
void reset_authoring(sb_solver *s) {
    s->sharded = false;
    std::vector<int32_t>().swap(s->global_id);
    s->domain = sb_domain{};
    s->plan.reset();
    s->plan_hash = 0;
}

int finalize_plan(sb_solver *s) {
    // SB_DEBUG_NO_COMM: the hosted-halo test hooks (sb_debug_*) drive world > 1 without any transport
    const bool no_comm = (s->desc.debug_flags & SB_DEBUG_NO_COMM) != 0;
    int rc = set_device(s); if (rc) return rc;
    // ---- which schedule (world > 1): decided before any work, from what the process is actually bound to ----
    // SB_SCHEDULE_AUTO = SB_SCHEDULE_SERIAL_EAGER, on every rank alike: it asks nothing of the bound runtime beyond send/recv and it is
    // the fastest EAGER schedule in every measurement that exists (one-device loopback shares of 256^3 / 8: 0.796 ms per tick against
    // 0.906 overlapped, profiles/r03s2w_loopback_w8_schedules.txt). Round 3 switched to the overlapped schedule above 1 MiB per peer on
    // the strength of a link-bandwidth model; no schedule has run between two devices yet, so the model is not evidence: the overlapped
    // and the captured schedules stay opt-in and bench.py --gpus N times every admitted one in the same launch (config.schedule_ab).
    int sched = s->desc.world > 1 ? s->desc.halo_schedule : SB_SCHEDULE_SERIAL_EAGER;
    const bool sched_auto = sched == SB_SCHEDULE_AUTO;
    if (sched_auto) sched = SB_SCHEDULE_SERIAL_EAGER;
    if (s->desc.world > 1 && !no_comm) {
        const bool graph = sched == SB_SCHEDULE_SERIAL_GRAPH || sched == SB_SCHEDULE_OVERLAP_GRAPH;
        if (graph && !s->desc.use_graph) return fail(SB_ERR_INVALID_ARG, "sb_finalize: a captured halo schedule needs use_graph = 1");
        if (graph && !s->peer.enabled && rccl().version < 22606)
            return fail(SB_ERR_UNSUPPORTED, "sb_finalize: capturing ncclSend/ncclRecv in a hipGraph is verified on RCCL >= 2.26.6 only; this process is bound to RCCL " +
                        std::to_string(rccl().version) + " (" + rccl().path + "): use SB_SCHEDULE_SERIAL_EAGER");
        if (sched == SB_SCHEDULE_OVERLAP_GRAPH && !capture_overlap_ok())
            return fail(SB_ERR_UNSUPPORTED, "sb_finalize: SB_SCHEDULE_OVERLAP_GRAPH faults on HIP runtimes older than 7.2 (hipStreamEndCapture recurses over the forked "
                        "exchange stream); this process is bound to HIP runtime " + std::to_string(hip_runtime_version()) +
                        " (a host that loaded PyTorch first runs on PyTorch's bundled runtime): use SB_SCHEDULE_SERIAL_GRAPH or an eager schedule");
    }
    const std::vector<float> &rest = s->rest.empty() ? s->pos : s->rest;
    sbp::Input in = make_input(rest.data(), s->n, s->dist_ij.data(), (int64_t)s->dist_rest.size(), s->vol_ijkl.data(),
                               (int64_t)s->vol_rest.size(), s->bend_ijkl.data(), (int64_t)s->bend_rest.size() / 2);
    if (s->sharded) {
        if (s->desc.world < 2) return fail(SB_ERR_INVALID_ARG, "sb_finalize: sharded authoring (sb_set_domain) is for world > 1");
        if (s->desc.partition == SB_PARTITION_RCB) return fail(SB_ERR_UNSUPPORTED, "sb_finalize: the RCB partition needs the whole mesh on every rank (no sb_set_domain)");
        in.global_id = s->global_id.data();
    }
    sbp::Opts o = plan_opts(s->desc.rank, s->desc.world, s->desc.part_dims, s->desc.tile_particles, s->desc.partition, s->desc.plan_flags,
                            in.m_v, in.m_b, s->sharded ? &s->domain : nullptr);
    s->plan = std::make_unique<sb_plan>();
    const bool timing = std::getenv("SB_PLAN_TIMING") != nullptr;       // (printing only)
    auto t0 = std::chrono::steady_clock::now();
    sbp::build_plan(in, o, s->plan->plan);
    sbp::extract_local(s->plan->plan, in, o.rank, s->plan->local);
    s->plan_hash = hash_plan(s->plan->plan);
    auto t1 = std::chrono::steady_clock::now();
    {   // the overlapped schedules apply to plans whose only exchange is the one before the T1 kernels (lattices)
        const sbp::Plan &P = s->plan->plan;
        const sbp::LocalPlan &L = s->plan->local;
        bool t1_halo = false;
        if (L.halo.size() > 1) for (int r = 0; r < L.world; ++r) t1_halo |= !L.halo[1].send_idx[(size_t)r].empty() || !L.halo[1].recv_idx[(size_t)r].empty();
        const bool want_overlap = sched == SB_SCHEDULE_OVERLAP_EAGER || sched == SB_SCHEDULE_OVERLAP_GRAPH;
        s->overlap_halo = want_overlap && s->desc.world > 1 && (s->comm || s->group_walk) && P.tiling && P.gcolours.empty() && P.t2_layers.empty() && t1_halo;
        if (want_overlap && !s->overlap_halo)      // T2 layers / global colours (irregular mesh) or no T1 halo: the serialised form
            sched = sched == SB_SCHEDULE_OVERLAP_GRAPH ? SB_SCHEDULE_SERIAL_GRAPH : SB_SCHEDULE_SERIAL_EAGER;
        // SB_SCHEDULE_AUTO, measured rather than modelled: where the overlapped eager schedule applies (RCCL between the ranks of a
        // lattice-type plan) the first ticks alternate between the two eager schedules -- same bits either way -- under HIP events, and the
        // ranks then keep the one whose slowest rank was faster (schedule.hip calibrate_*). Not for a group walked by one thread (its
        // exchanges are issued across the ranks), not with SB_TUNE_NO_AUTO_CALIBRATION.
        s->calib = sb_solver::AutoSchedule{};
        if (sched_auto && s->desc.world > 1 && s->comm && !s->peer.enabled && !s->group_walk && P.tiling && P.gcolours.empty() && P.t2_layers.empty() && t1_halo &&
            !(s->tune_flags & SB_TUNE_NO_AUTO_CALIBRATION))
            s->calib.state = 1;
    }
    s->schedule = sched;
    s->graph_rccl = sched == SB_SCHEDULE_SERIAL_GRAPH || sched == SB_SCHEDULE_OVERLAP_GRAPH;
    if (timing) std::fprintf(stderr, "[finalize] plan %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    return SB_OK;
}

int finalize_device(sb_solver *s) {
    int rc = set_device(s); if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    build_device(s);
    if (std::getenv("SB_PLAN_TIMING") != nullptr)       // (printing only)
        std::fprintf(stderr, "[finalize] build_device + upload %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    // opt in to the LDS size the largest tile needs
    for (int tl = 0; tl < 3; ++tl)
        if (s->tiling[tl].lds_bytes > 64 * 1024) throw std::runtime_error("internal: tile LDS budget exceeded");
    if (s->overlap_halo || s->calib.state == 1) {
        // opt-in: on the one measurement available (RCCL loopback on one GPU, 8-rank share of 256^3) splitting the T0
        // launch and running the exchange beside the interior tiles pays only inside a captured graph
        // (an ordinary stream: a highest-priority one -- meant to keep the pack kernel and RCCL's few workgroups from queueing behind
        // the interior launch -- made the eager overlapped tick FOUR TIMES slower on this runtime, 0.93 -> 3.35 ms in the W = 8
        // loopback, profiles/r03r_loopback_w8_priority_stream_not_kept.txt)
        HIP_CHECK(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_boundary, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming));
    }
    return SB_OK;
}

// What a rank tells the others about its plan: [status, plan hash, pair hash with rank 0 .. W-1]; status bit 0 = sharded authoring,
// bit 1 = this rank FAILED before it got here (the hashes are then 0).
// The SHAPE of the tick program every rank must share: a rank whose WINDOW holds none of the leftover constraints plans no T2 layer
// (or one global colour less) -- fewer launches, fewer exchanges, fewer halo slots -- while its pair hashes still agree with its
// neighbours' (tests/fuzz/fuzz_parity.py seed 2005569: particles on a line, four windows, one of them with a leftover layer).
uint32_t plan_shape(const sb_solver *s) {
    const sbp::Plan &P = s->plan->plan;
    return 1u + (P.tiling ? 1u : 0u) + 4u * (uint32_t)std::min<size_t>(P.t2_layers.size(), 0xfffu) + 0x4000u * (uint32_t)std::min<size_t>(P.gcolours.size(), 0xffffu);
}

std::vector<uint64_t> agreement_record(const sb_solver *s, bool failed) {
    const int W = s->desc.world;
    std::vector<uint64_t> mine((size_t)W + 2, 0);
    mine[0] = (s->sharded ? 1u : 0u) | (failed ? 2u : 0u) | (failed ? 0u : (uint64_t)plan_shape(s) << 8);
    if (!failed) {
        mine[1] = s->plan_hash;
        for (int r = 0; r < W; ++r) mine[2 + (size_t)r] = s->plan->local.pair_hash[(size_t)r];
    }
    return mine;
}

// Every rank holds the whole table and checks EVERY pair, so that all ranks fail together (a rank that went on alone would wait for
// its neighbours' first exchange forever). A rank that failed on its own (planning, upload, an incomplete window ...) is named to all.
// Whole-mesh ranks must hold the identical plan; any two ranks must agree on what they share (ghost lists both ways, programs of the
// tiles both run) -- the only check a sharded rank, which sees just its window, can make.
int check_agreement(const std::vector<uint64_t> &all, int W, int me) {
    const size_t rec = (size_t)W + 2;
    char msg[400];
    for (int a = 0; a < W; ++a)
        if (all[(size_t)a * rec] & 2u) {
            if (a == me) return SB_ERR_STATE;       // (its own message is already in place)
            std::snprintf(msg, sizeof msg, "sb_finalize: rank %d failed while planning / uploading (see its own sb_last_error); every rank of the solver gives up with it", a);
            return fail(SB_ERR_STATE, msg);
        }
    for (int a = 0; a < W; ++a)
        for (int b = a + 1; b < W; ++b) {
            const uint64_t *ra = all.data() + (size_t)a * rec, *rb = all.data() + (size_t)b * rec;
            if ((ra[0] >> 8) != (rb[0] >> 8)) {
                std::snprintf(msg, sizeof msg, "sb_finalize: ranks %d and %d planned tick programs of different shape (leftover layers / global colours: %llx vs %llx): "
                              "a window that does not reproduce the whole-mesh plan (sharded authoring is for lattice-like meshes) or different meshes on the ranks",
                              a, b, (unsigned long long)(ra[0] >> 8), (unsigned long long)(rb[0] >> 8));
                return fail(SB_ERR_STATE, msg);
            }
            if (!(ra[0] & 1u) && !(rb[0] & 1u) && ra[1] != rb[1]) {
                std::snprintf(msg, sizeof msg, "sb_finalize: ranks %d and %d planned different schedules (plan hash %016llx vs %016llx): "
                              "every rank must pass the same mesh, tile_particles, partition and plan_flags", a, b,
                              (unsigned long long)ra[1], (unsigned long long)rb[1]);
                return fail(SB_ERR_STATE, msg);
            }
            if (ra[2 + (size_t)b] != rb[2 + (size_t)a]) {
                std::snprintf(msg, sizeof msg, "sb_finalize: ranks %d and %d disagree on what they share (ghost lists / programs of the tiles both run: pair hash "
                              "%016llx vs %016llx): same mesh, domain, tile_particles, partition and plan_flags on every rank? window complete (sb_domain_window)?",
                              a, b, (unsigned long long)ra[2 + (size_t)b], (unsigned long long)rb[2 + (size_t)a]);
                return fail(SB_ERR_STATE, msg);
            }
        }
    return SB_OK;
}

// Phase B (RCCL across processes / threads): one all-gather of the agreement records. Entered by a rank that failed in phase A too,
// with the failure marker, so that nobody waits in the collective for a rank that has already returned an error.
int finalize_agree(sb_solver *s, int local_rc) {
    const int W = s->desc.world;
    const size_t rec = (size_t)W + 2;
    const std::string my_error = local_rc ? std::string(last_error_text()) : std::string();
    std::vector<uint64_t> mine = agreement_record(s, local_rc != SB_OK);
    std::vector<uint64_t> all((size_t)W * rec);
    const int rc = guarded([&]() -> int {
        int r0 = set_device(s); if (r0) return r0;
        DevBuf<uint64_t> d_all; int64_t acct = 0;
        d_all.alloc((size_t)W * rec, acct);
        HIP_CHECK(hipMemcpy(d_all.p + (size_t)s->desc.rank * rec, mine.data(), rec * sizeof(uint64_t), hipMemcpyHostToDevice));
        NCCL_CHECK(rccl().AllGather(d_all.p + (size_t)s->desc.rank * rec, d_all.p, rec * sizeof(uint64_t), ncclUint8, s->comm, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        HIP_CHECK(hipMemcpy(all.data(), d_all.p, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        return SB_OK;
    });
    if (local_rc) return fail(local_rc, my_error);       // this rank's own failure is what its host sees
    if (rc) return rc;
    return check_agreement(all, W, s->desc.rank);
}

// Phase C: the peer transport's mailboxes over the communicator the host already set up (setup time only), then the bookkeeping.
int finalize_link(sb_solver *s) {
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        if (s->peer.enabled && s->desc.world > 1) {
            if (s->loopback) {
                peer_link(s);                      // every neighbour is this rank itself
            } else if (s->comm) {
                const int W = s->desc.world;
                hipIpcMemHandle_t mine;
                HIP_CHECK(hipIpcGetMemHandle(&mine, s->peer.mailbox));
                DevBuf<uint8_t> d_all; int64_t acct = 0;
                d_all.alloc((size_t)W * sizeof(mine), acct);
                HIP_CHECK(hipMemcpy(d_all.p + (size_t)s->desc.rank * sizeof(mine), &mine, sizeof(mine), hipMemcpyHostToDevice));
                NCCL_CHECK(rccl().AllGather(d_all.p + (size_t)s->desc.rank * sizeof(mine), d_all.p, sizeof(mine), ncclUint8, s->comm, s->stream));
                HIP_CHECK(hipStreamSynchronize(s->stream));
                std::vector<hipIpcMemHandle_t> all((size_t)W);
                HIP_CHECK(hipMemcpy(all.data(), d_all.p, (size_t)W * sizeof(mine), hipMemcpyDeviceToHost));
                for (int r = 0; r < W; ++r) {
                    if (r == s->desc.rank) continue;
                    bool needed = false;
                    for (const auto &H : s->halos) for (int pr : H->peers) needed |= pr == r;
                    if (!needed) continue;
                    void *p = nullptr;
                    HIP_CHECK(hipIpcOpenMemHandle(&p, all[(size_t)r], hipIpcMemLazyEnablePeerAccess));
                    s->peer.remote[(size_t)r] = (uint32_t *)p; s->peer.opened[(size_t)r] = 1;
                }
                peer_link(s);
            }
            // otherwise (no communicator) the host connects the mailboxes: sb_peer_mailbox_handle / sb_peer_connect, or the group does
        }
        if (std::getenv("SB_PRINT_ALLOC"))       // diagnosis (printing only): where the arrays landed (run-to-run timing modes)
            std::fprintf(stderr, "[alloc] pos3 %p prev %p vel %p wf %p w8 %p T0.stream %p T1.stream %p T0.tiles %p T1.tiles %p\n", (void *)s->d_pos3.p, (void *)s->d_prev.p,
                         (void *)s->d_vel.p, (void *)s->d_wf.p, (void *)s->d_w8.p, (void *)s->tiling[0].stream.p, (void *)s->tiling[1].stream.p,
                         (void *)s->tiling[0].tiles.p, (void *)s->tiling[1].tiles.p);
        HIP_CHECK(hipDeviceSynchronize());
        // authoring copies are no longer needed (keep rest values out of memory for 50M-constraint meshes)
        std::vector<float>().swap(s->pos); std::vector<float>().swap(s->vel); std::vector<float>().swap(s->rest);
        std::vector<int32_t>().swap(s->dist_ij); std::vector<int32_t>().swap(s->vol_ijkl); std::vector<int32_t>().swap(s->bend_ijkl);
        std::vector<float>().swap(s->dist_rest); std::vector<float>().swap(s->vol_rest); std::vector<float>().swap(s->bend_rest);
        s->finalized = true;
        return SB_OK;
    });
}

}  // namespace sbi

extern "C" {

int sb_finalize(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_finalize: null handle");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_finalize called twice");
    if (s->n <= 0) return fail(SB_ERR_STATE, "sb_finalize before sb_set_particles");
    const bool no_comm = (s->desc.debug_flags & SB_DEBUG_NO_COMM) != 0;
    if (s->desc.world > 1 && !s->comm && !s->peer.enabled && !no_comm)
        return fail(SB_ERR_STATE, "sb_finalize: world > 1 needs sb_comm_init first (or halo_transport = SB_TRANSPORT_PEER with sb_peer_connect)");
    int rc = guarded([&]() -> int { return finalize_local(s); });
    // Ranks plan independently: before the first exchange make sure they all arrived at the same plan (same published order, ownership,
    // halo slots, plan options) -- and that they all ARRIVED: a rank that failed above still enters the all-gather, with a failure
    // marker, so the others fail with it instead of waiting for it forever. The peer transport without a communicator compares the
    // hashes when the mailboxes are linked (peer_link).
    if (s->desc.world > 1 && s->comm && !s->loopback) rc = finalize_agree(s, rc);
    if (rc) return rc;
    return finalize_link(s);
}

int sb_synchronize(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_synchronize: null handle");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        flush_deferred(s);
        HIP_CHECK(hipStreamSynchronize(s->stream));
        check_peer_error(s);
        return SB_OK;
    });
}

int sb_get_owner(sb_solver *s, int32_t *owner, int32_t n) {
    if (!s || !owner) return fail(SB_ERR_INVALID_ARG, "sb_get_owner: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_owner before sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_get_owner: n mismatch");
    std::memcpy(owner, s->plan->plan.owner_of_old.data(), (size_t)n * sizeof(int32_t));
    return SB_OK;
}

int sb_profile_begin(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_profile_begin: null handle");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        HIP_CHECK(hipEventRecord(s->ev0, s->stream));
        return SB_OK;
    });
}
int sb_profile_end(sb_solver *s, float *ms) {
    if (!s || !ms) return fail(SB_ERR_INVALID_ARG, "sb_profile_end: null argument");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        HIP_CHECK(hipEventRecord(s->ev1, s->stream));
        HIP_CHECK(hipEventSynchronize(s->ev1));
        HIP_CHECK(hipEventElapsedTime(ms, s->ev0, s->ev1));
        return SB_OK;
    });
}

int sb_get_stats(sb_solver *s, sb_stats *out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_get_stats: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_stats before sb_finalize");
    std::memset(out, 0, sizeof(*out));
    const sbp::Plan &P = s->plan->plan;
    const sbp::LocalPlan &L = s->plan->local;
    out->n_particles_owned = s->n_owned;
    out->n_particles_local = s->n_local;
    for (size_t k = 0; k < L.order_mask[0].size(); ++k) if (L.order_mask[0][k]) ++out->n_constraints_local[P.order_type[0][k]];
    out->n_tilings = P.tiling ? 2 : 1;
    out->n_global_colours = (int32_t)P.gcolours.size();
    for (int tl = 0; tl < 2; ++tl) { out->n_tiles[tl] = s->tiling[tl].n_tiles; out->tile_constraints[tl] = s->tiling[tl].n_slots; }
    out->n_t2_layers = (int64_t)s->t2_layer_range.size();
    out->n_t2_tiles = s->tiling[2].n_tiles;
    out->t2_constraints = s->tiling[2].n_slots;
    out->constraints_in_tiles = P.cons_in_tiles;
    out->constraints_in_global = P.cons_in_global;
    for (size_t slot = 0; slot < s->halos.size(); ++slot) {
        const int64_t cnt = s->halos[slot]->send_off.back();
        if (slot == 1) out->halo_particles_t1 = cnt; else out->halo_particles_global += cnt;   // global colours and T2 layers
    }
    out->device_bytes = s->dev_bytes;
    {   // compulsory bytes per launch (see softbody.h): particle state + the tables a launch reads
        const int64_t mb = s->w_uniform ? 0 : (s->w_palette ? 1 : 4);     // inverse mass: nothing (uniform), palette index, or float
        auto tables = [&](const DevTiling &D) { return D.stream_bytes + (int64_t)D.n_tiles * (int64_t)sizeof(sbk::TileDesc) + (int64_t)D.runs_overflow.count * 8; };
        for (int tl = 0; tl < 2; ++tl) {
            const DevTiling &D = s->tiling[tl];
            if (!D.n_tiles) continue;
            out->launch_bytes[tl] = D.staged_particles * (12 + mb + 12 + 12 + 12) + tables(D);   // x, w, xprev in; x, xprev out
        }
        const DevTiling &D0 = s->tiling[0];
        if (D0.n_tiles) {
            out->launch_bytes[2] = D0.staged_particles * (12 + mb + 12 + 12 + 12) + tables(D0);  // x, w, v in; x, xprev out
            out->launch_bytes[3] = D0.staged_particles * (12 + mb + 12 + 12 + 12) + tables(D0);  // x, w, xprev in; x, v out
        }
        const DevTiling &D2 = s->tiling[2];
        if (D2.n_tiles) out->launch_bytes[4] = D2.staged_particles * (4 + 12 + mb + 12) + tables(D2);
    }
    out->partition = P.partition;
    out->halo_schedule = s->schedule;
    out->halo_unpack_fused = s->fused_unpack ? 1 : 0;
    out->halo_auto_state = s->calib.state;
    out->halo_auto_ticks = s->calib.n[0] + s->calib.n[1];
    out->halo_auto_ms[0] = s->calib.decided_ms[0]; out->halo_auto_ms[1] = s->calib.decided_ms[1];
    out->readback_peeks = s->n_peeks;
    out->readback_peek_tiles = s->n_peek_tiles;
    out->ticks_fused = s->n_fused;
    out->ticks_fused_kinematic = s->n_kin_fused;
    for (int tl = 0; tl < 2; ++tl) out->lane_packed_tiles[tl] = s->tiling[tl].n_packed_tiles;
    out->plan_hash = s->plan_hash;
    if (!P.rank_cost.empty()) {
        out->partition_cost = P.rank_cost[(size_t)s->desc.rank];
        for (int64_t c : P.rank_cost) { out->partition_cost_total += c; out->partition_cost_max = std::max(out->partition_cost_max, c); }
    }
    {
        std::vector<uint8_t> is_peer((size_t)s->desc.world, 0);
        for (const auto &H : s->halos) {
            for (int pr : H->peers) is_peer[(size_t)pr] = 1;
            out->halo_particles_recv += H->recv_off.back();
        }
        for (uint8_t b : is_peer) out->halo_peers += b;
    }
    return SB_OK;
}

}  // extern "C"
