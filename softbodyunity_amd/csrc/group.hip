// group.hip — sb_group_*: ONE process (a Unity player) driving several MI355X behind one Softbody component (include/softbody_group.h).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); [BUILDER-DEFINED] from BASELINE.json:5 and
// SURVEY.md §1 L1' ("one process x 8 devices"), §8b (`device_count`).
//
// A group is N ordinary solvers (sb_desc.rank = r, world = N, device = devices[r]) plus what a single host needs around them: the whole
// mesh authored once (cut into windows here when the partition is the block grid), the transport connected inside the process, one
// call per tick, state gathered in the caller's numbering, one render snapshot on one device. Two host models (softbody_group.h):
// a thread per rank (default) or the calling thread walking the tick program step by step across the ranks (SB_GROUP_WALK).
#include "solver_internal.hpp"

#include <condition_variable>
#include <thread>

using namespace sbi;

namespace {

// One host thread per rank: run(f) executes f(r) on every rank's thread and returns when all are done.
class RankThreads {
public:
    explicit RankThreads(int n) : results_((size_t)n, 0), errors_((size_t)n) {
        for (int r = 0; r < n; ++r) threads_.emplace_back([this, r] { loop(r); });
    }
    ~RankThreads() {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++generation_; }
        cv_go_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // returns the first non-zero status (its message becomes the caller's sb_last_error)
    int run(const std::function<int(int)> &f) {
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = &f; pending_ = (int)threads_.size(); ++generation_;
        }
        cv_go_.notify_all();
        std::unique_lock<std::mutex> l(mu_);
        cv_done_.wait(l, [this] { return pending_ == 0; });
        job_ = nullptr;
        for (size_t r = 0; r < results_.size(); ++r)
            if (results_[r] != SB_OK) return fail(results_[r], "rank " + std::to_string(r) + ": " + errors_[r]);
        return SB_OK;
    }
private:
    void loop(int r) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(int)> *job;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_go_.wait(l, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                job = job_;
            }
            int rc = SB_OK;
            try { rc = (*job)(r); }
            catch (const std::exception &e) { rc = fail(SB_ERR_INVALID_ARG, e.what()); }
            {
                std::lock_guard<std::mutex> l(mu_);
                results_[(size_t)r] = rc;
                errors_[(size_t)r] = rc ? last_error_text() : "";       // (the error text is thread-local to this worker)
                if (--pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<int(int)> *job_ = nullptr;
    uint64_t generation_ = 0;
    int pending_ = 0;
    bool stop_ = false;
    std::vector<int> results_;
    std::vector<std::string> errors_;
};

}  // namespace

struct sb_group {
    sb_desc desc{};
    uint32_t flags = 0;
    int W = 0;
    std::vector<int> devices;
    std::vector<sb_solver *> ranks;
    std::unique_ptr<RankThreads> threads;       // null in walk mode (and for a single rank)
    bool finalized = false, sharded = false;
    // the mesh, authored once
    int32_t n = 0;
    std::vector<float> pos, vel, invm, rest;
    std::vector<int32_t> dist_ij, vol_ijkl, bend_ijkl;
    std::vector<float> dist_rest, vol_rest, bend_rest;
    float compliance[3] = {0, 0, 0};
    float plane[4] = {0, 1, 0, 0};
    int32_t plane_on = 0;
    bool plane_set = false;
    std::vector<uint8_t> pinned;                 // [n] inverse mass == 0 (kept for sb_group_set_kinematic_positions)
    // per rank: the rank's numbering -> the caller's (sharded authoring: window index -> whole-mesh id; empty = identity)
    std::vector<std::vector<int32_t>> gid;
    // kinematic scratch per rank (ids in the rank's numbering)
    std::vector<std::vector<int32_t>> kin_ids;
    std::vector<std::vector<float>> kin_pos;
    std::vector<int32_t> owner;                  // [n] caller numbering -> owning rank (after finalize)
    std::vector<int32_t> index_in_rank;          // [n] caller numbering -> index in the owner's numbering
    // ---- render readback, gathered on the render device (rank 0's) ----
    static constexpr int kSnapSlots = 3;
    std::vector<int32_t> render_tri, render_set;
    bool render_dirty = false, render_set_only = false;
    hipStream_t copy_stream = nullptr;
    DevBuf<int32_t> d_tri, d_adj_off, d_adj_tri, d_render_set;
    DevBuf<float> d_gather[kSnapSlots], d_nrm[kSnapSlots], d_cpos[kSnapSlots];
    float *h_pos[kSnapSlots] = {nullptr, nullptr, nullptr}, *h_nrm[kSnapSlots] = {nullptr, nullptr, nullptr}, *h_cpos[kSnapSlots] = {nullptr, nullptr, nullptr};
    size_t h_nrm_cap = 0, h_cpos_cap = 0;
    hipEvent_t ev_copied[kSnapSlots] = {nullptr, nullptr, nullptr};
    bool snap_compact[kSnapSlots] = {false, false, false}, snap_has_normals[kSnapSlots] = {false, false, false};
    int snap_head = 0, snap_pending = 0, snap_last_ended = -1;
    struct RankRender {                          // per rank, on the rank's device
        DevBuf<int32_t> d_target_of_local;       // owned particle l -> caller id (full snapshots)
        DevBuf<int32_t> d_rs_ids, d_rs_local;    // the render particles the rank owns: caller id, device index
        std::vector<int32_t> rs_local;
        hipEvent_t ev_snap[kSnapSlots] = {nullptr, nullptr, nullptr};
        int64_t acct = 0;
    };
    std::vector<RankRender> rr;
    int64_t dev_bytes = 0;

    int device_of(int r) const { return devices[(size_t)r]; }
    // f(r) for every rank: on the ranks' threads, or one after the other on the calling thread
    int for_ranks(const std::function<int(int)> &f) {
        if (threads) return threads->run(f);
        for (int r = 0; r < W; ++r) { const int rc = f(r); if (rc) return fail(rc, "rank " + std::to_string(r) + ": " + last_error_text()); }
        return SB_OK;
    }
    ~sb_group() {
        threads.reset();
        if (!ranks.empty()) {
            // every rank quiescent before any mailbox / buffer a neighbour writes into goes away
            for (sb_solver *s : ranks) if (s && s->finalized) (void)sb_synchronize(s);
        }
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
        for (int k = 0; k < kSnapSlots; ++k) {
            if (h_pos[k]) (void)hipHostFree(h_pos[k]);
            if (h_nrm[k]) (void)hipHostFree(h_nrm[k]);
            if (h_cpos[k]) (void)hipHostFree(h_cpos[k]);
            if (ev_copied[k]) (void)hipEventDestroy(ev_copied[k]);
            d_gather[k].free(); d_nrm[k].free(); d_cpos[k].free();
        }
        d_tri.free(); d_adj_off.free(); d_adj_tri.free(); d_render_set.free();
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (size_t r = 0; r < rr.size(); ++r) {
            (void)hipSetDevice(devices[r]);
            for (int k = 0; k < kSnapSlots; ++k) if (rr[r].ev_snap[k]) (void)hipEventDestroy(rr[r].ev_snap[k]);
            rr[r].d_target_of_local.free(); rr[r].d_rs_ids.free(); rr[r].d_rs_local.free();
        }
        for (sb_solver *s : ranks) if (s) (void)sb_destroy(s);
    }
};

namespace {

bool walk_mode(const sb_group *g) { return (g->flags & SB_GROUP_WALK) != 0 || g->W == 1; }

// The window of the whole mesh rank r hands over under sharded authoring: the particles whose rest position lies in its box, the
// constraints among them in the whole mesh's order, the particles' whole-mesh ids (ascending).
struct Window {
    std::vector<int32_t> gid;
    std::vector<float> pos, vel, invm, rest;
    std::vector<int32_t> idx[3];
    std::vector<float> restv[3];
};
void cut_window(const sb_group *g, const sb_domain &dom, int r, Window &w) {
    sb_plan_opts o{};
    o.rank = r; o.world = g->W;
    for (int a = 0; a < 3; ++a) o.part_dims[a] = g->desc.part_dims[a];
    // (the automatic tile size must be resolved the way plan_opts resolves it for the whole mesh)
    o.tile_particles = g->desc.tile_particles;
    double lo[3], hi[3];
    if (sb_domain_window(&dom, &o, lo, hi) != SB_OK) throw std::runtime_error(std::string("sb_group_finalize: ") + last_error_text());
    const std::vector<float> &rp = g->rest.empty() ? g->pos : g->rest;
    std::vector<int32_t> new_of((size_t)g->n, -1);
    for (int32_t p = 0; p < g->n; ++p) {
        bool in = true;
        for (int a = 0; a < 3; ++a) { const double c = (double)rp[3 * (size_t)p + a]; in = in && c >= lo[a] && c < hi[a]; }
        if (in) { new_of[(size_t)p] = (int32_t)w.gid.size(); w.gid.push_back(p); }
    }
    const size_t nw = w.gid.size();
    w.pos.resize(3 * nw); w.vel.resize(3 * nw); w.invm.resize(nw);
    if (!g->rest.empty()) w.rest.resize(3 * nw);
    for (size_t k = 0; k < nw; ++k) {
        const size_t p = (size_t)w.gid[k];
        for (int c = 0; c < 3; ++c) {
            w.pos[3 * k + c] = g->pos[3 * p + c]; w.vel[3 * k + c] = g->vel[3 * p + c];
            if (!g->rest.empty()) w.rest[3 * k + c] = g->rest[3 * p + c];
        }
        w.invm[k] = g->invm[p];
    }
    const std::vector<int32_t> *I[3] = {&g->dist_ij, &g->vol_ijkl, &g->bend_ijkl};
    const std::vector<float> *R[3] = {&g->dist_rest, &g->vol_rest, &g->bend_rest};
    const int nv[3] = {2, 4, 4}, nr[3] = {1, 1, 2};
    for (int t = 0; t < 3; ++t) {
        const size_t m = R[t]->size() / (size_t)nr[t];
        for (size_t k = 0; k < m; ++k) {
            bool in = true;
            for (int q = 0; q < nv[t]; ++q) in = in && new_of[(size_t)(*I[t])[nv[t] * k + q]] >= 0;
            if (!in) continue;
            for (int q = 0; q < nv[t]; ++q) w.idx[t].push_back(new_of[(size_t)(*I[t])[nv[t] * k + q]]);
            for (int q = 0; q < nr[t]; ++q) w.restv[t].push_back((*R[t])[nr[t] * k + q]);
        }
    }
}

int author_rank(sb_group *g, int r, const sb_domain *dom) {
    sb_solver *s = g->ranks[(size_t)r];
    int rc;
    if (dom) {
        Window w;
        cut_window(g, *dom, r, w);
        if (w.gid.empty()) return fail(SB_ERR_INVALID_ARG, "sb_group_finalize: a rank's window of the mesh is empty (fewer occupied cells than ranks?)");
        const int32_t nw = (int32_t)w.gid.size();
        if ((rc = sb_set_particles(s, w.pos.data(), w.vel.data(), w.invm.data(), nw))) return rc;
        if (!w.rest.empty() && (rc = sb_set_rest_positions(s, w.rest.data(), nw))) return rc;
        if (!g->dist_rest.empty() && (rc = sb_set_distance_constraints(s, w.idx[0].data(), w.restv[0].data(), (int32_t)w.restv[0].size(), g->compliance[0]))) return rc;
        if (!g->vol_rest.empty() && (rc = sb_set_volume_constraints(s, w.idx[1].data(), w.restv[1].data(), (int32_t)w.restv[1].size(), g->compliance[1]))) return rc;
        if (!g->bend_rest.empty() && (rc = sb_set_bending_constraints(s, w.idx[2].data(), w.restv[2].data(), (int32_t)(w.restv[2].size() / 2), g->compliance[2]))) return rc;
        if ((rc = sb_set_domain(s, dom, w.gid.data(), nw))) return rc;
        g->gid[(size_t)r] = std::move(w.gid);
    } else {
        if ((rc = sb_set_particles(s, g->pos.data(), g->vel.data(), g->invm.data(), g->n))) return rc;
        if (!g->rest.empty() && (rc = sb_set_rest_positions(s, g->rest.data(), g->n))) return rc;
        if (!g->dist_rest.empty() && (rc = sb_set_distance_constraints(s, g->dist_ij.data(), g->dist_rest.data(), (int32_t)g->dist_rest.size(), g->compliance[0]))) return rc;
        if (!g->vol_rest.empty() && (rc = sb_set_volume_constraints(s, g->vol_ijkl.data(), g->vol_rest.data(), (int32_t)g->vol_rest.size(), g->compliance[1]))) return rc;
        if (!g->bend_rest.empty() && (rc = sb_set_bending_constraints(s, g->bend_ijkl.data(), g->bend_rest.data(), (int32_t)(g->bend_rest.size() / 2), g->compliance[2]))) return rc;
    }
    if (g->plane_set && (rc = sb_set_ground_plane(s, g->plane[0], g->plane[1], g->plane[2], g->plane[3], g->plane_on))) return rc;
    return SB_OK;
}

const int32_t *id_map(const sb_group *g, int r) { return g->gid[(size_t)r].empty() ? nullptr : g->gid[(size_t)r].data(); }

// ---- walk mode: the calling thread takes the tick apart --------------------------------------------------------------------------------
// One exchange step for every rank: pack kernels, then the sends and receives of ALL ranks inside one RCCL group, then the unpack
// kernels. `fork`: the overlapped schedule's form -- the exchange travels on each rank's second stream between two events.
void walk_exchange(sb_group *g, const TickStep &st) {
    const bool fork = st.kind == StepKind::ForkExchange;
    bool rccl_needed = false;
    for (int r = 0; r < g->W; ++r) {
        sb_solver *s = g->ranks[(size_t)r];
        HIP_CHECK(hipSetDevice(g->device_of(r)));
        hipStream_t stream = fork ? s->comm_stream : s->stream;
        if (fork) {
            HIP_CHECK(hipEventRecord(s->ev_boundary, s->stream));
            HIP_CHECK(hipStreamWaitEvent(s->comm_stream, s->ev_boundary, 0));
        }
        if (s->xtimer.enabled) s->xtimer.mark(stream);
        if (!s->peer.enabled) halo_exchange_pre(s, st.index, stream);
        if (s->xtimer.enabled) s->xtimer.mark(stream);
        if (s->peer.enabled) halo_exchange_pre(s, st.index, stream);
        rccl_needed = rccl_needed || (!s->peer.enabled && halo_slot_active(s, st.index));
    }
    if (rccl_needed) {
        NCCL_CHECK(rccl().GroupStart());
        try {
            for (int r = 0; r < g->W; ++r) {
                sb_solver *s = g->ranks[(size_t)r];
                HIP_CHECK(hipSetDevice(g->device_of(r)));
                halo_exchange_calls(s, st.index, fork ? s->comm_stream : s->stream);
            }
        } catch (...) {
            (void)rccl().GroupEnd();      // never leave the group open behind an error
            throw;
        }
        NCCL_CHECK(rccl().GroupEnd());
    }
    for (int r = 0; r < g->W; ++r) {
        sb_solver *s = g->ranks[(size_t)r];
        HIP_CHECK(hipSetDevice(g->device_of(r)));
        hipStream_t stream = fork ? s->comm_stream : s->stream;
        halo_exchange_post(s, st.index, stream);
        if (s->xtimer.enabled) s->xtimer.mark(stream);
        if (fork) HIP_CHECK(hipEventRecord(s->ev_halo, s->comm_stream));
    }
}

int walk_step(sb_group *g, float dt, int substeps) {
    if (g->W == 1) return sb_step(g->ranks[0], dt, substeps);
    return guarded([&]() -> int {
        std::vector<TickShape> shape((size_t)g->W);
        for (int r = 0; r < g->W; ++r) {
            HIP_CHECK(hipSetDevice(g->device_of(r)));
            shape[(size_t)r] = begin_tick(g->ranks[(size_t)r], dt, substeps);
        }
        // Every rank writes down its own program. They have the same steps in the same order -- the phases are a property of the plan, the
        // schedule is the same on every rank -- and differ only in what the FIRST tile step is: a rank whose previous tick still holds its
        // last kernel back starts with the fused mid-tick kernel (with or without kinematic targets: a rank that owns no pin has none), a
        // rank whose tick was completed by a read (ranks peek from a tile count on: one may have peeked where another flushed) starts with
        // the tick's plain first kernel. Whether the tick's own last kernel is held back depends on the arguments only, hence agrees.
        std::vector<std::vector<TickStep>> prog((size_t)g->W);
        for (int r = 0; r < g->W; ++r) {
            const TickShape &t = shape[(size_t)r];
            prog[(size_t)r] = tick_program(g->ranks[(size_t)r], substeps, t.fuse, t.defer_last, t.kin);
            bool same = prog[(size_t)r].size() == prog[0].size();
            for (size_t i = 0; same && i < prog[0].size(); ++i) same = prog[(size_t)r][i].kind == prog[0][i].kind && prog[(size_t)r][i].index == prog[0][i].index;
            if (!same) throw HipError(SB_ERR_STATE, "sb_group_step: the ranks' tick programs differ (different schedules or plans on the ranks of one group?)");
        }
        for (size_t i = 0; i < prog[0].size(); ++i) {
            const TickStep &st0 = prog[0][i];
            if (st0.kind == StepKind::Exchange || st0.kind == StepKind::ForkExchange) { walk_exchange(g, st0); continue; }
            for (int r = 0; r < g->W; ++r) {
                HIP_CHECK(hipSetDevice(g->device_of(r)));
                run_step(g->ranks[(size_t)r], prog[(size_t)r][i], nullptr);
            }
        }
        for (int r = 0; r < g->W; ++r) {
            HIP_CHECK(hipSetDevice(g->device_of(r)));
            HIP_CHECK(hipGetLastError());
            end_tick(g->ranks[(size_t)r], shape[(size_t)r]);
        }
        return SB_OK;
    });
}

// The communicators of a walk-mode group: all ranks of one thread, created together inside one RCCL group.
int walk_comm_init(sb_group *g) {
    return guarded([&]() -> int {
        const bool loopback = (g->desc.debug_flags & SB_DEBUG_LOOPBACK) != 0;
        std::vector<ncclUniqueId> ids((size_t)(loopback ? g->W : 1));
        for (auto &id : ids) NCCL_CHECK(rccl().GetUniqueId(&id));
        NCCL_CHECK(rccl().GroupStart());
        try {
            for (int r = 0; r < g->W; ++r) {
                sb_solver *s = g->ranks[(size_t)r];
                HIP_CHECK(hipSetDevice(g->device_of(r)));
                // SB_DEBUG_LOOPBACK (one-device pipeline tests): every rank a communicator of size 1 of its own, every peer the rank itself
                if (loopback) NCCL_CHECK(rccl().CommInitRank(&s->comm, 1, ids[(size_t)r], 0));
                else NCCL_CHECK(rccl().CommInitRank(&s->comm, g->W, ids[0], r));
            }
        } catch (...) {
            (void)rccl().GroupEnd();
            for (sb_solver *s : g->ranks) s->comm = nullptr;      // (whatever the failed group left behind is not a communicator to destroy)
            throw;
        }
        const ncclResult_t r = rccl().GroupEnd();
        if (r != ncclSuccess) {
            for (sb_solver *s : g->ranks) s->comm = nullptr;
            throw HipError(SB_ERR_RCCL, std::string("ncclGroupEnd (communicators of the group's ranks): ") + rccl().GetErrorString(r));
        }
        return SB_OK;
    });
}

// Peer access between the ranks' devices (pointers of one rank dereferenced by kernels of another: mailboxes, the render gather).
void enable_peer_access(sb_group *g) {
    for (int a = 0; a < g->W; ++a)
        for (int b = 0; b < g->W; ++b) {
            const int da = g->device_of(a), db = g->device_of(b);
            if (da == db) continue;
            HIP_CHECK(hipSetDevice(da));
            int can = 0;
            HIP_CHECK(hipDeviceCanAccessPeer(&can, da, db));
            if (!can) throw HipError(SB_ERR_UNSUPPORTED, "sb_group_finalize: device " + std::to_string(da) + " cannot access device " + std::to_string(db) + " (no peer access)");
            const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw HipError(SB_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
}

int check_group(const sb_group *g, bool finalized, const char *who) {
    if (!g) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": null group");
    if (finalized && !g->finalized) return fail(SB_ERR_STATE, std::string(who) + " before sb_group_finalize");
    if (!finalized && g->finalized) return fail(SB_ERR_STATE, std::string(who) + " after sb_group_finalize");
    return SB_OK;
}

}  // namespace

extern "C" {

int sb_group_create(const sb_desc *desc, const int32_t *devices, int32_t n_devices, uint32_t flags, sb_group **out) {
    if (!desc || !out) return fail(SB_ERR_INVALID_ARG, "sb_group_create: null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(SB_ERR_INVALID_ARG, "sb_group_create: n_devices must be 1 .. 64");
    if (flags & ~(SB_GROUP_WALK | SB_GROUP_WHOLE_MESH)) return fail(SB_ERR_INVALID_ARG, "sb_group_create: unknown bit in flags");
    if ((flags & SB_GROUP_WALK) && n_devices > 1) {
        if (desc->halo_schedule == SB_SCHEDULE_SERIAL_GRAPH || desc->halo_schedule == SB_SCHEDULE_OVERLAP_GRAPH)
            return fail(SB_ERR_UNSUPPORTED, "sb_group_create: SB_GROUP_WALK runs the eager schedules only (a capture spanning several devices' streams is not built)");
    }
    return guarded([&]() -> int {
        auto g = std::make_unique<sb_group>();
        g->desc = *desc; g->flags = flags; g->W = n_devices;
        for (int r = 0; r < n_devices; ++r) g->devices.push_back(devices ? devices[r] : r);
        g->ranks.assign((size_t)n_devices, nullptr);
        g->gid.resize((size_t)n_devices); g->kin_ids.resize((size_t)n_devices); g->kin_pos.resize((size_t)n_devices);
        for (int r = 0; r < n_devices; ++r) {
            sb_desc d = *desc;
            d.device = g->devices[(size_t)r]; d.rank = r; d.world = n_devices;
            if (n_devices == 1) { d.halo_transport = SB_TRANSPORT_RCCL; d.halo_schedule = SB_SCHEDULE_AUTO; d.debug_flags = 0; }
            const int rc = sb_create(&d, &g->ranks[(size_t)r]);
            if (rc) return rc;
            g->ranks[(size_t)r]->group_walk = walk_mode(g.get()) && n_devices > 1;
        }
        if (!walk_mode(g.get())) g->threads = std::make_unique<RankThreads>(n_devices);
        *out = g.release();
        return SB_OK;
    });
}

int sb_group_destroy(sb_group *g) {
    if (!g) return fail(SB_ERR_INVALID_ARG, "sb_group_destroy: null group");
    delete g;
    return SB_OK;
}

int sb_group_set_particles(sb_group *g, const float *pos, const float *vel, const float *inv_mass, int32_t n) {
    if (int rc = check_group(g, false, "sb_group_set_particles")) return rc;
    if (!pos || !inv_mass || n <= 0) return fail(SB_ERR_INVALID_ARG, "sb_group_set_particles: bad argument");
    return guarded([&]() -> int {
        for (int32_t p = 0; p < n; ++p) if (!(inv_mass[p] >= 0.0f)) return fail(SB_ERR_INVALID_ARG, "sb_group_set_particles: inverse mass must be >= 0");
        g->n = n;
        g->pos.assign(pos, pos + 3 * (size_t)n);
        if (vel) g->vel.assign(vel, vel + 3 * (size_t)n); else g->vel.assign(3 * (size_t)n, 0.0f);
        g->invm.assign(inv_mass, inv_mass + n);
        return SB_OK;
    });
}

int sb_group_set_rest_positions(sb_group *g, const float *rest, int32_t n) {
    if (int rc = check_group(g, false, "sb_group_set_rest_positions")) return rc;
    if (!rest || n != g->n) return fail(SB_ERR_INVALID_ARG, "sb_group_set_rest_positions: bad argument (n differs from sb_group_set_particles?)");
    return guarded([&]() -> int { g->rest.assign(rest, rest + 3 * (size_t)n); return SB_OK; });
}

static int group_set_cons(sb_group *g, const char *who, const int32_t *idx, const float *rest, int32_t m, float compliance, int type, int nv, int nrest) {
    if (int rc = check_group(g, false, who)) return rc;
    if (m < 0 || (m > 0 && (!idx || !rest))) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": bad argument");
    if (g->n <= 0) return fail(SB_ERR_STATE, std::string(who) + " before sb_group_set_particles");
    if (!(compliance >= 0.0f)) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": compliance must be >= 0");
    return guarded([&]() -> int {
        for (int64_t k = 0; k < (int64_t)m * nv; ++k) if (idx[k] < 0 || idx[k] >= g->n) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": particle index out of range");
        std::vector<int32_t> &I = type == 0 ? g->dist_ij : (type == 1 ? g->vol_ijkl : g->bend_ijkl);
        std::vector<float> &R = type == 0 ? g->dist_rest : (type == 1 ? g->vol_rest : g->bend_rest);
        I.assign(idx, idx + (size_t)m * nv);
        R.assign(rest, rest + (size_t)m * nrest);
        g->compliance[type] = compliance;
        return SB_OK;
    });
}
int sb_group_set_distance_constraints(sb_group *g, const int32_t *ij, const float *rest_len, int32_t m, float compliance) {
    return group_set_cons(g, "sb_group_set_distance_constraints", ij, rest_len, m, compliance, 0, 2, 1);
}
int sb_group_set_volume_constraints(sb_group *g, const int32_t *ijkl, const float *rest_vol, int32_t m, float compliance) {
    return group_set_cons(g, "sb_group_set_volume_constraints", ijkl, rest_vol, m, compliance, 1, 4, 1);
}
int sb_group_set_bending_constraints(sb_group *g, const int32_t *ijkl, const float *rest_cs, int32_t m, float compliance) {
    return group_set_cons(g, "sb_group_set_bending_constraints", ijkl, rest_cs, m, compliance, 2, 4, 2);
}

int sb_group_set_ground_plane(sb_group *g, float nx, float ny, float nz, float d, int32_t enabled) {
    if (!g) return fail(SB_ERR_INVALID_ARG, "sb_group_set_ground_plane: null group");
    if (!(nx == nx) || !(ny == ny) || !(nz == nz) || !(d == d)) return fail(SB_ERR_INVALID_ARG, "sb_group_set_ground_plane: NaN");
    g->plane[0] = nx; g->plane[1] = ny; g->plane[2] = nz; g->plane[3] = d; g->plane_on = enabled ? 1 : 0; g->plane_set = true;
    if (g->finalized) for (sb_solver *s : g->ranks) { const int rc = sb_set_ground_plane(s, nx, ny, nz, d, enabled); if (rc) return rc; }
    return SB_OK;
}

int sb_group_finalize(sb_group *g) {
    if (int rc = check_group(g, false, "sb_group_finalize")) return rc;
    if (g->n <= 0) return fail(SB_ERR_STATE, "sb_group_finalize before sb_group_set_particles");
    const int W = g->W;
    const bool peer = W > 1 && g->desc.halo_transport == SB_TRANSPORT_PEER;
    const bool no_comm = (g->desc.debug_flags & SB_DEBUG_NO_COMM) != 0;
    const bool use_rccl = W > 1 && !peer && !no_comm;
    int rc = guarded([&]() -> int {
        // ---- the mesh to every rank: its window under the block partition (sharded authoring), else the whole mesh ----
        sb_domain dom{};
        // Windows (sharded authoring) where they are known to reproduce the whole-mesh plan: a lattice-like body (distance constraints only,
        // filling its bounding box -- colours, tiles and leftovers of such a mesh are local to a window; a tet mesh's or a cloth's are not:
        // tests/fuzz/fuzz_parity.py found every such mesh refused by the ranks' agreement check) under the block partition -- asked for
        // (SB_PARTITION_BLOCKS), or SB_PARTITION_AUTO on a LARGE mesh (the block grid is what AUTO takes for it anyway, and eight whole-mesh
        // plans of 16.8 M particles are 26 GB of host memory and 8 x 4.9 s of planning against 8 x 0.6 s on windows).
        constexpr int32_t kAutoShardParticles = 1 << 21;
        const bool may_shard = W > 1 && !(g->flags & SB_GROUP_WHOLE_MESH) && g->desc.tile_particles >= 0 && g->vol_rest.empty() && g->bend_rest.empty() &&
                               (g->desc.partition == SB_PARTITION_BLOCKS || (g->desc.partition == SB_PARTITION_AUTO && g->n >= kAutoShardParticles));
        g->sharded = false;
        if (may_shard) {
            const std::vector<float> &rp = g->rest.empty() ? g->pos : g->rest;
            const int rc0 = sb_domain_from_mesh(rp.data(), g->n, g->dist_ij.data(), (int32_t)g->dist_rest.size(), g->vol_ijkl.data(), (int32_t)g->vol_rest.size(),
                                                g->bend_ijkl.data(), (int32_t)(g->bend_rest.size() / 2), &dom);
            if (rc0) return rc0;
            g->sharded = dom.fill >= 1.0;
        }
        g->pinned.assign((size_t)g->n, 0);
        for (int32_t p = 0; p < g->n; ++p) g->pinned[(size_t)p] = g->invm[(size_t)p] == 0.0f;
        if (W > 1) enable_peer_access(g);
        int rc1 = g->for_ranks([&](int r) { return guarded([&]() -> int { return author_rank(g, r, g->sharded ? &dom : nullptr); }); });
        if (rc1) return rc1;
        // ---- the transport ----
        if (use_rccl) {
            if (walk_mode(g)) { if ((rc1 = walk_comm_init(g))) return rc1; }
            else {
                const bool loopback = (g->desc.debug_flags & SB_DEBUG_LOOPBACK) != 0;
                std::vector<std::array<uint8_t, SB_UNIQUE_ID_BYTES>> ids((size_t)(loopback ? W : 1));
                for (auto &id : ids) if ((rc1 = sb_comm_unique_id(id.data()))) return rc1;
                // every rank's thread joins the communicator (the rendezvous of ncclCommInitRank blocks until all have called)
                if ((rc1 = g->for_ranks([&](int r) { return sb_comm_init(g->ranks[(size_t)r], ids[(size_t)(loopback ? r : 0)].data()); }))) return rc1;
            }
        }
        // ---- plan on every rank (host work, side by side), agreement, tables on every device ----
        // All ranks live in this process: their plans are compared right here (the records sb_finalize all-gathers across processes).
        // Windows that do not reproduce the whole-mesh plan -- the fill / constraint-type rule above lets through e.g. a cloud of particles on
        // a line joined by nearest-neighbour springs -- are found by exactly that comparison: the ranks are then handed the whole mesh and plan again.
        // (ranks' own threads where the group has them, else helper threads)
        auto on_every_rank = [&](const std::function<int(int)> &f) -> int {
            if (g->threads) return g->for_ranks(f);
            std::vector<int> lrc((size_t)W, 0);
            std::vector<std::string> lerr((size_t)W);
            {
                std::vector<std::thread> th;
                for (int r = 0; r < W; ++r) th.emplace_back([&, r] {
                    lrc[(size_t)r] = guarded([&]() -> int { return f(r); });
                    if (lrc[(size_t)r]) lerr[(size_t)r] = last_error_text();
                });
                for (auto &t : th) t.join();
            }
            for (int r = 0; r < W; ++r) if (lrc[(size_t)r]) return fail(lrc[(size_t)r], "rank " + std::to_string(r) + ": " + lerr[(size_t)r]);
            return SB_OK;
        };
        auto plan_and_compare = [&]() -> int {
            int rcp = on_every_rank([&](int r) { return guarded([&]() -> int { return finalize_plan(g->ranks[(size_t)r]); }); });
            if (rcp) return rcp;
            if (W > 1 && !(g->desc.debug_flags & SB_DEBUG_LOOPBACK)) {
                std::vector<uint64_t> all;
                for (int r = 0; r < W; ++r) { const auto rec = agreement_record(g->ranks[(size_t)r], false); all.insert(all.end(), rec.begin(), rec.end()); }
                return check_agreement(all, W, -1);
            }
            return SB_OK;
        };
        rc1 = plan_and_compare();
        if (rc1 == SB_ERR_STATE && g->sharded) {
            g->sharded = false;
            for (int r = 0; r < W; ++r) { reset_authoring(g->ranks[(size_t)r]); g->gid[(size_t)r].clear(); }
            if ((rc1 = g->for_ranks([&](int r) { return guarded([&]() -> int { return author_rank(g, r, nullptr); }); }))) return rc1;
            rc1 = plan_and_compare();
        }
        if (rc1) return rc1;
        if ((rc1 = on_every_rank([&](int r) { return guarded([&]() -> int { return finalize_device(g->ranks[(size_t)r]); }); }))) return rc1;
        if ((rc1 = on_every_rank([&](int r) { return finalize_link(g->ranks[(size_t)r]); }))) return rc1;
        // ---- peer transport inside one process: the mailboxes by plain pointer ----
        if (peer && !(g->desc.debug_flags & SB_DEBUG_LOOPBACK)) {
            for (int a = 0; a < W; ++a)
                for (int b = 0; b < W; ++b) {
                    if (a == b) continue;
                    bool needed = false;
                    for (const auto &H : g->ranks[(size_t)a]->halos) for (int pr : H->peers) needed |= pr == b;
                    if (!needed || g->ranks[(size_t)a]->peer.remote[(size_t)b]) continue;
                    if ((rc1 = sb_peer_connect(g->ranks[(size_t)a], b, nullptr, g->ranks[(size_t)b]))) return rc1;
                }
            // (links -- and with them the neighbours' plan / pair hashes -- are checked now, not at the first tick)
            if ((rc1 = g->for_ranks([&](int r) { return guarded([&]() -> int { int rcd = set_device(g->ranks[(size_t)r]); if (rcd) return rcd; peer_link(g->ranks[(size_t)r]); return SB_OK; }); }))) return rc1;
        }
        // ---- who owns what, in the caller's numbering ----
        g->owner.assign((size_t)g->n, -1); g->index_in_rank.assign((size_t)g->n, -1);
        for (int r = 0; r < W; ++r) {
            sb_solver *s = g->ranks[(size_t)r];
            const std::vector<int32_t> &own = s->plan->plan.owner_of_old;
            const int32_t *map = id_map(g, r);
            for (int32_t o = 0; o < s->n; ++o)
                if (own[(size_t)o] == r) {
                    const int32_t c = map ? map[o] : o;
                    if (g->owner[(size_t)c] >= 0) return fail(SB_ERR_STATE, "sb_group_finalize: internal: a particle is owned by two ranks");
                    g->owner[(size_t)c] = r; g->index_in_rank[(size_t)c] = o;
                }
        }
        for (int32_t c = 0; c < g->n; ++c) if (g->owner[(size_t)c] < 0) return fail(SB_ERR_STATE, "sb_group_finalize: internal: a particle is owned by no rank");
        // the authoring copy is no longer needed
        std::vector<float>().swap(g->pos); std::vector<float>().swap(g->vel); std::vector<float>().swap(g->invm); std::vector<float>().swap(g->rest);
        std::vector<int32_t>().swap(g->dist_ij); std::vector<int32_t>().swap(g->vol_ijkl); std::vector<int32_t>().swap(g->bend_ijkl);
        std::vector<float>().swap(g->dist_rest); std::vector<float>().swap(g->vol_rest); std::vector<float>().swap(g->bend_rest);
        g->finalized = true;
        return SB_OK;
    });
    return rc;
}

int sb_group_step(sb_group *g, float dt, int32_t substeps) {
    if (int rc = check_group(g, true, "sb_group_step")) return rc;
    if (!(dt > 0.0f) || substeps <= 0) return fail(SB_ERR_INVALID_ARG, "sb_group_step: dt and substeps must be positive");
    if (walk_mode(g)) return walk_step(g, dt, substeps);
    return g->for_ranks([&](int r) { return sb_step(g->ranks[(size_t)r], dt, substeps); });
}

static int group_get_state(sb_group *g, float *out, int32_t n, bool velocity, const char *who) {
    if (int rc = check_group(g, true, who)) return rc;
    if (!out || n != g->n) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": bad argument (n differs from sb_group_set_particles?)");
    if (g->W == 1) return velocity ? sb_get_velocities(g->ranks[0], out, n) : sb_get_positions(g->ranks[0], out, n);      // (the permutation runs on the GPU)
    // every rank writes the entries it owns (disjoint), the ranks side by side
    return g->for_ranks([&](int r) { return guarded([&]() -> int { return get_state_owned(g->ranks[(size_t)r], out, velocity, id_map(g, r)); }); });
}
int sb_group_get_positions(sb_group *g, float *out, int32_t n) { return group_get_state(g, out, n, false, "sb_group_get_positions"); }
int sb_group_get_velocities(sb_group *g, float *out, int32_t n) { return group_get_state(g, out, n, true, "sb_group_get_velocities"); }

int sb_group_set_state(sb_group *g, const float *pos, const float *vel, int32_t n) {
    if (int rc = check_group(g, true, "sb_group_set_state")) return rc;
    if (!pos || !vel || n != g->n) return fail(SB_ERR_INVALID_ARG, "sb_group_set_state: bad argument");
    return g->for_ranks([&](int r) { return guarded([&]() -> int { return set_state_from(g->ranks[(size_t)r], pos, vel, id_map(g, r)); }); });
}

int sb_group_set_kinematic_positions(sb_group *g, const int32_t *ids, const float *pos, int32_t count) {
    if (int rc = check_group(g, true, "sb_group_set_kinematic_positions")) return rc;
    if (count < 0 || (count > 0 && (!ids || !pos))) return fail(SB_ERR_INVALID_ARG, "sb_group_set_kinematic_positions: bad argument");
    if (count == 0) return SB_OK;
    return guarded([&]() -> int {
        // validate the whole call first (nothing is changed on an error), then hand every rank the entries it owns, in its numbering
        for (auto &v : g->kin_ids) v.clear();
        for (auto &v : g->kin_pos) v.clear();
        for (int32_t k = 0; k < count; ++k) {
            if (ids[k] < 0 || ids[k] >= g->n) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle index out of range");
            if (!g->pinned[(size_t)ids[k]])
                return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle " + std::to_string(ids[k]) + " has a non-zero inverse mass (only pinned particles are kinematic)");
            for (int c = 0; c < 3; ++c) if (!(pos[3 * (size_t)k + c] == pos[3 * (size_t)k + c])) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: NaN");
            const int r = g->owner[(size_t)ids[k]];
            g->kin_ids[(size_t)r].push_back(g->index_in_rank[(size_t)ids[k]]);
            for (int c = 0; c < 3; ++c) g->kin_pos[(size_t)r].push_back(pos[3 * (size_t)k + c]);
        }
        // two moves without a tick between them: the earlier one takes effect first, which completes the held-back tick -- on EVERY rank
        // alike, so that the ranks stay in the same tick state (an id given twice in one call is found by the rank that owns it)
        bool pending = false;
        for (sb_solver *s : g->ranks) pending = pending || s->kin_pending >= 0;
        return g->for_ranks([&](int r) {
            sb_solver *s = g->ranks[(size_t)r];
            if (pending) { const int rcf = guarded([&]() -> int { int rcd = set_device(s); if (rcd) return rcd; flush_deferred(s); return SB_OK; }); if (rcf) return rcf; }
            if (g->kin_ids[(size_t)r].empty()) return (int)SB_OK;
            return guarded([&]() -> int { return set_kinematic(s, g->kin_ids[(size_t)r].data(), g->kin_pos[(size_t)r].data(), (int32_t)g->kin_ids[(size_t)r].size()); });
        });
    });
}

/* ---- render readback, gathered on the render device ------------------------------------------------------------------------------------- */

int sb_group_set_render_triangles(sb_group *g, const int32_t *tri, int32_t m) {
    if (!g || m < 0 || (m > 0 && !tri)) return fail(SB_ERR_INVALID_ARG, "sb_group_set_render_triangles: bad argument");
    if (g->n <= 0) return fail(SB_ERR_STATE, "sb_group_set_render_triangles before sb_group_set_particles");
    if (g->snap_pending) return fail(SB_ERR_STATE, "sb_group_set_render_triangles while a readback is pending");
    return guarded([&]() -> int {
        for (int64_t c = 0; c < 3 * (int64_t)m; ++c) if (tri[c] < 0 || tri[c] >= g->n) return fail(SB_ERR_INVALID_ARG, "sb_group_set_render_triangles: particle index out of range");
        g->render_tri.assign(tri, tri + 3 * (size_t)m);
        g->render_dirty = true;
        if (m == 0) g->render_set_only = false;
        for (bool &b : g->snap_has_normals) b = false;
        return SB_OK;
    });
}

int sb_group_set_readback_render_set_only(sb_group *g, int32_t on) {
    if (!g) return fail(SB_ERR_INVALID_ARG, "sb_group_set_readback_render_set_only: null group");
    if (g->snap_pending) return fail(SB_ERR_STATE, "sb_group_set_readback_render_set_only while a readback is pending");
    if (on && g->render_tri.empty()) return fail(SB_ERR_STATE, "sb_group_set_readback_render_set_only: set the render triangles first");
    g->render_set_only = on != 0;
    return SB_OK;
}

int sb_group_readback_begin(sb_group *g) {
    if (int rc = check_group(g, true, "sb_group_readback_begin")) return rc;
    if (g->snap_pending == 2) return fail(SB_ERR_STATE, "sb_group_readback_begin: two snapshots already pending, call sb_group_readback_end");
    return guarded([&]() -> int {
        const int W = g->W, dev0 = g->device_of(0);
        const size_t n3 = (size_t)g->n * 3;
        HIP_CHECK(hipSetDevice(dev0));
        if (!g->copy_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
            g->rr.resize((size_t)W);
            for (int k = 0; k < sb_group::kSnapSlots; ++k) {
                g->d_gather[k].alloc(n3, g->dev_bytes);
                HIP_CHECK(hipMemset(g->d_gather[k].p, 0, n3 * sizeof(float)));
                HIP_CHECK(hipEventCreateWithFlags(&g->ev_copied[k], hipEventDisableTiming));
            }
            for (int r = 0; r < W; ++r) {      // owned particle -> caller id, on the rank's device
                sb_solver *s = g->ranks[(size_t)r];
                HIP_CHECK(hipSetDevice(g->device_of(r)));
                const sbp::LocalPlan &L = s->plan->local;
                const int32_t *map = id_map(g, r);
                std::vector<int32_t> target((size_t)s->n_owned);
                for (int64_t l = 0; l < s->n_owned; ++l) { const int32_t o = L.local_to_old[(size_t)l]; target[(size_t)l] = map ? map[o] : o; }
                g->rr[(size_t)r].d_target_of_local.upload(target, g->rr[(size_t)r].acct);
                for (int k = 0; k < sb_group::kSnapSlots; ++k) HIP_CHECK(hipEventCreateWithFlags(&g->rr[(size_t)r].ev_snap[k], hipEventDisableTiming));
            }
            HIP_CHECK(hipSetDevice(dev0));
        }
        const bool compact = g->render_set_only && !g->render_tri.empty();
        if (!g->render_tri.empty() && g->render_dirty) {     // incident-triangle lists (triangle ids ascending per particle) + who owns which render particle
            HIP_CHECK(hipStreamSynchronize(g->copy_stream));
            const int64_t m = (int64_t)g->render_tri.size() / 3;
            std::vector<int32_t> off((size_t)g->n + 1, 0), adj((size_t)3 * m);
            for (int64_t c = 0; c < 3 * m; ++c) ++off[(size_t)g->render_tri[c] + 1];
            g->render_set.clear();
            for (int32_t v = 0; v < g->n; ++v) { if (off[(size_t)v + 1]) g->render_set.push_back(v); off[(size_t)v + 1] += off[v]; }
            std::vector<int32_t> cur(off.begin(), off.end() - 1);
            for (int64_t t = 0; t < m; ++t)
                for (int j = 0; j < 3; ++j) adj[(size_t)cur[g->render_tri[3 * t + j]]++] = (int32_t)t;
            g->d_tri.upload(g->render_tri, g->dev_bytes); g->d_adj_off.upload(off, g->dev_bytes); g->d_adj_tri.upload(adj, g->dev_bytes);
            g->d_render_set.upload(g->render_set, g->dev_bytes);
            std::vector<std::vector<int32_t>> ids((size_t)W), loc((size_t)W);
            for (int32_t c : g->render_set) {
                const int r = g->owner[(size_t)c];
                ids[(size_t)r].push_back(c);
                loc[(size_t)r].push_back(local_of_old(g->ranks[(size_t)r])[(size_t)g->index_in_rank[(size_t)c]]);
            }
            for (int r = 0; r < W; ++r) {
                HIP_CHECK(hipSetDevice(g->device_of(r)));
                auto &R = g->rr[(size_t)r];
                R.d_rs_ids.upload(ids[(size_t)r], R.acct); R.d_rs_local.upload(loc[(size_t)r], R.acct);
                R.rs_local = loc[(size_t)r];
                g->ranks[(size_t)r]->n_peek_tiles = -1;        // the peek's tile subset follows the render set
            }
            HIP_CHECK(hipSetDevice(dev0));
            g->render_dirty = false;
        }
        const int k = (g->snap_head + g->snap_pending) % sb_group::kSnapSlots;
        // host buffers of this slot, by what it will carry
        if (!compact && !g->h_pos[k]) HIP_CHECK(hipHostMalloc((void **)&g->h_pos[k], n3 * sizeof(float), hipHostMallocDefault));
        if (!g->render_tri.empty()) {
            const size_t cnt3 = (compact ? g->render_set.size() : (size_t)g->n) * 3;
            if (g->d_nrm[k].count < cnt3) g->d_nrm[k].alloc(cnt3, g->dev_bytes);
            if (compact && g->d_cpos[k].count < cnt3) g->d_cpos[k].alloc(cnt3, g->dev_bytes);
            // (the pinned buffers of all slots share one capacity: a larger need -- render set changed, or a switch to full snapshots --
            // frees them all; no readback is pending at such a change)
            if (g->h_nrm_cap < cnt3) {
                for (int q = 0; q < sb_group::kSnapSlots; ++q) { if (g->h_nrm[q]) (void)hipHostFree(g->h_nrm[q]); g->h_nrm[q] = nullptr; }
                g->h_nrm_cap = cnt3;
            }
            if (!g->h_nrm[k]) HIP_CHECK(hipHostMalloc((void **)&g->h_nrm[k], std::max<size_t>(g->h_nrm_cap, 3) * sizeof(float), hipHostMallocDefault));
            if (compact) {
                if (g->h_cpos_cap < cnt3) {
                    for (int q = 0; q < sb_group::kSnapSlots; ++q) { if (g->h_cpos[q]) (void)hipHostFree(g->h_cpos[q]); g->h_cpos[q] = nullptr; }
                    g->h_cpos_cap = cnt3;
                }
                if (!g->h_cpos[k]) HIP_CHECK(hipHostMalloc((void **)&g->h_cpos[k], std::max<size_t>(g->h_cpos_cap, 3) * sizeof(float), hipHostMallocDefault));
            }
        }
        // every rank: tick-end positions (peek or completed tick) of what it owns, straight into the gather buffer on the render device
        float *dst = g->d_gather[k].p;
        int rc = g->for_ranks([&](int r) {
            return guarded([&]() -> int {
                sb_solver *s = g->ranks[(size_t)r];
                int rcd = set_device(s); if (rcd) return rcd;
                auto &R = g->rr[(size_t)r];
                const float *src = render_source(s, compact, R.rs_local);
                if (compact) launch_snapshot_subset(s, src, R.d_rs_ids.p, R.d_rs_local.p, (int)R.rs_local.size(), dst);
                else launch_snapshot_all(s, src, R.d_target_of_local.p, dst);
                HIP_CHECK(hipEventRecord(R.ev_snap[k], s->stream));
                return SB_OK;
            });
        });
        if (rc) return rc;
        HIP_CHECK(hipSetDevice(dev0));
        for (int r = 0; r < W; ++r) HIP_CHECK(hipStreamWaitEvent(g->copy_stream, g->rr[(size_t)r].ev_snap[k], 0));
        if (!compact) HIP_CHECK(hipMemcpyAsync(g->h_pos[k], g->d_gather[k].p, n3 * sizeof(float), hipMemcpyDeviceToHost, g->copy_stream));
        g->snap_has_normals[k] = false;
        g->snap_compact[k] = compact;
        if (!g->render_tri.empty()) {
            const int count = compact ? (int)g->render_set.size() : (int)g->n;
            launch_normals(g->copy_stream, g->d_gather[k].p, g->d_adj_off.p, g->d_adj_tri.p, g->d_tri.p, g->d_nrm[k].p, count,
                           compact ? g->d_render_set.p : (const int32_t *)nullptr, compact ? g->d_cpos[k].p : (float *)nullptr);
            HIP_CHECK(hipMemcpyAsync(g->h_nrm[k], g->d_nrm[k].p, (size_t)count * 3 * sizeof(float), hipMemcpyDeviceToHost, g->copy_stream));
            if (compact) HIP_CHECK(hipMemcpyAsync(g->h_cpos[k], g->d_cpos[k].p, (size_t)count * 3 * sizeof(float), hipMemcpyDeviceToHost, g->copy_stream));
            g->snap_has_normals[k] = true;
        }
        HIP_CHECK(hipEventRecord(g->ev_copied[k], g->copy_stream));
        ++g->snap_pending;
        return SB_OK;
    });
}

int sb_group_readback_end(sb_group *g, const float **pos_xyz_out) {
    if (!g || !pos_xyz_out) return fail(SB_ERR_INVALID_ARG, "sb_group_readback_end: null argument");
    if (g->snap_pending == 0) return fail(SB_ERR_STATE, "sb_group_readback_end without a pending sb_group_readback_begin");
    return guarded([&]() -> int {
        HIP_CHECK(hipSetDevice(g->device_of(0)));
        const int k = g->snap_head;
        HIP_CHECK(hipEventSynchronize(g->ev_copied[k]));
        for (sb_solver *s : g->ranks) check_peer_error(s);
        *pos_xyz_out = g->snap_compact[k] ? g->h_cpos[k] : g->h_pos[k];
        g->snap_last_ended = k;
        g->snap_head = (g->snap_head + 1) % sb_group::kSnapSlots; --g->snap_pending;
        return SB_OK;
    });
}

int sb_group_readback_get_normals(sb_group *g, const float **out) {
    if (!g || !out) return fail(SB_ERR_INVALID_ARG, "sb_group_readback_get_normals: null argument");
    if (g->snap_last_ended < 0 || !g->snap_has_normals[g->snap_last_ended])
        return fail(SB_ERR_STATE, "sb_group_readback_get_normals: no finished readback with render triangles set");
    *out = g->h_nrm[g->snap_last_ended];
    return SB_OK;
}

int sb_group_readback_get_render_set(sb_group *g, const int32_t **ids, int32_t *count) {
    if (!g || !ids || !count) return fail(SB_ERR_INVALID_ARG, "sb_group_readback_get_render_set: null argument");
    if (g->snap_last_ended < 0 || !g->snap_has_normals[g->snap_last_ended])
        return fail(SB_ERR_STATE, "sb_group_readback_get_render_set: no finished readback with render triangles set");
    *ids = g->render_set.data();
    *count = (int32_t)g->render_set.size();
    return SB_OK;
}

int sb_group_synchronize(sb_group *g) {
    if (int rc = check_group(g, true, "sb_group_synchronize")) return rc;
    return g->for_ranks([&](int r) { return sb_synchronize(g->ranks[(size_t)r]); });
}

int32_t sb_group_rank_count(const sb_group *g) { return g ? g->W : -1; }

int sb_group_get_rank(sb_group *g, int32_t rank, sb_solver **out) {
    if (!g || !out) return fail(SB_ERR_INVALID_ARG, "sb_group_get_rank: null argument");
    if (rank < 0 || rank >= g->W) return fail(SB_ERR_INVALID_ARG, "sb_group_get_rank: no such rank");
    *out = g->ranks[(size_t)rank];
    return SB_OK;
}

}  // extern "C"
