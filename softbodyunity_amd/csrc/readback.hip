// readback.hip — state reads and writes, kinematic targets, the asynchronous render readback with GPU vertex normals
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"
#include "readback_kernels.hip.hpp"

using namespace sbi;

namespace sbi {

// Owned particle l of a rank -> its index in the array a read is delivered in: the rank's own numbering (the caller's, or its window's
// under sharded authoring), or -- for a group that gathers several ranks into one array -- the whole mesh's (id_map[window index]).
static void scatter_owned(const sb_solver *s, const float *staged, float *out, const int32_t *id_map) {
    const sbp::LocalPlan &L = s->plan->local;
    sbp::parallel_for_chunks(s->n_owned, 1 << 18, [&](int64_t, int64_t lb, int64_t le) {     // owned particles have distinct caller ids
        for (int64_t l = lb; l < le; ++l) {
            int32_t o = L.local_to_old[l];
            if (id_map) o = id_map[o];
            for (int c = 0; c < 3; ++c) out[3 * (size_t)o + c] = staged[3 * (size_t)l + c];
        }
    });
}

// Positions (or velocities) of the particles this rank OWNS, written into `out` at their caller index (id_map: see scatter_owned);
// entries of other ranks' particles are left alone. Position reads PEEK while the tick's last kernel is held back (the tick stays
// fusable with the next one), on every rank of a partitioned solver too: the held-back kernel runs on T0 tiles, which hold owned
// particles only and need no ghost.
int get_state_owned(sb_solver *s, float *out, bool velocity, const int32_t *id_map) {
    int rc = set_device(s); if (rc) return rc;
    const bool peek = !velocity && can_peek(s);
    if (peek) { peek_positions(s, false); if (s->kin_pending >= 0) scatter_kinematic(s, s->d_peek.p); }     // (pending targets show in what is read; they stay pending)
    else flush_deferred(s);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    check_peer_error(s);
    s->h_stage.resize((size_t)s->n_owned * 3);
    const float *src = velocity ? s->d_vel.p : (peek ? s->d_peek.p : s->d_pos3.p);
    if (s->n_owned) HIP_CHECK(hipMemcpy(s->h_stage.data(), src, (size_t)s->n_owned * 3 * sizeof(float), hipMemcpyDeviceToHost));
    scatter_owned(s, s->h_stage.data(), out, id_map);
    return SB_OK;
}

}  // namespace sbi

static int get_state(sb_solver *s, float *out, int32_t n, bool velocity) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_get_*: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_* before sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_get_*: n differs from sb_set_particles");
    return guarded([&]() -> int {
        if (s->desc.world != 1) return get_state_owned(s, out, velocity, nullptr);       // only the entries this rank owns (the caller merges the ranks' arrays)
        int rc = set_device(s); if (rc) return rc;
        const sbp::LocalPlan &L = s->plan->local;
        // positions while the tick's last kernel is deferred: peek instead of completing the tick (the next sb_step keeps its fusion)
        const bool peek = !velocity && can_peek(s);
        if (peek) { peek_positions(s, false); if (s->kin_pending >= 0) scatter_kinematic(s, s->d_peek.p); }     // (pending targets show in what is read; they stay pending)
        else flush_deferred(s);
        // single rank: every entry is ours, so the permutation to caller numbering runs on the GPU and one copy
        // lands in the caller's array (a host-side scatter costs 25 ms for 16.7 M particles)
        if (!s->d_local_to_old.p) s->d_local_to_old.upload(L.local_to_old, s->dev_bytes);
        if (!s->d_get_scratch.p) s->d_get_scratch.alloc((size_t)s->n * 3, s->dev_bytes);
        sbk::PosView src = s->pos_view();
        if (peek) src.xyz = s->d_peek.p;
        if (velocity) src.xyz = s->d_vel.p;
        hipLaunchKernelGGL(sbk::snapshot_kernel, dim3((unsigned)((s->n_owned + 255) / 256)), dim3(256), 0, s->stream, src,
                           s->d_local_to_old.p, s->d_get_scratch.p, (int)s->n_owned);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, s->d_get_scratch.p, (size_t)s->n * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        return SB_OK;
    });
}

extern "C" {

int sb_get_positions(sb_solver *s, float *out, int32_t n) { return get_state(s, out, n, false); }
int sb_get_velocities(sb_solver *s, float *out, int32_t n) { return get_state(s, out, n, true); }

}  // extern "C"

namespace sbi {

// The rank's numbering (caller's, or its window's) -> device numbering, -1 for a particle this rank does not hold; built at first use.
const std::vector<int32_t> &local_of_old(sb_solver *s) {
    if (s->local_of_old.empty()) {
        const sbp::LocalPlan &L = s->plan->local;
        s->local_of_old.assign((size_t)s->n, -1);
        for (size_t l = 0; l < L.local_to_old.size(); ++l) s->local_of_old[(size_t)L.local_to_old[l]] = (int32_t)l;
    }
    return s->local_of_old;
}

// Where a render snapshot of this rank reads the tick-end positions from, made valid on the solver's stream: while the tick's last kernel
// is held back that is a PEEK into the side array (compact: only the T0 tiles that hold a particle of `wanted_local`; the subset is built
// once per render set), else the state itself after the tick has been completed. Pending kinematic targets show in a peek.
const float *render_source(sb_solver *s, bool compact, const std::vector<int32_t> &wanted_local) {
    if (!can_peek(s)) { flush_deferred(s); return s->d_pos3.p; }
    if (compact && s->n_peek_tiles < 0) build_peek_subset(s, wanted_local);
    peek_positions(s, compact);
    if (s->kin_pending >= 0) scatter_kinematic(s, s->d_peek.p);
    return s->d_peek.p;
}

// Launch helpers for a host that gathers several ranks' snapshots on one device (group.hip): dst may live on another device.
void launch_snapshot_all(sb_solver *s, const float *src_xyz, const int32_t *d_target_of_local, float *dst_xyz) {
    if (!s->n_owned) return;
    sbk::PosView src = s->pos_view();
    src.xyz = const_cast<float *>(src_xyz);
    hipLaunchKernelGGL(sbk::snapshot_kernel, dim3((unsigned)((s->n_owned + 255) / 256)), dim3(256), 0, s->stream, src, d_target_of_local, dst_xyz, (int)s->n_owned);
    HIP_CHECK(hipGetLastError());
}
void launch_snapshot_subset(sb_solver *s, const float *src_xyz, const int32_t *d_ids, const int32_t *d_local, int count, float *dst_xyz) {
    if (count <= 0) return;
    sbk::PosView src = s->pos_view();
    src.xyz = const_cast<float *>(src_xyz);
    hipLaunchKernelGGL(sbk::snapshot_subset_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s->stream, src, d_ids, d_local, dst_xyz, count);
    HIP_CHECK(hipGetLastError());
}
void launch_normals(hipStream_t st, const float *snap_xyz, const int32_t *adj_off, const int32_t *adj_tri, const int32_t *tri, float *nrm_xyz, int count,
                    const int32_t *subset, float *subset_pos_xyz) {
    if (count <= 0) return;
    hipLaunchKernelGGL(sbk::normals_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, snap_xyz, adj_off, adj_tri, tri, nrm_xyz, count, subset, subset_pos_xyz);
    HIP_CHECK(hipGetLastError());
}

// Replace positions and velocities of every particle this rank holds (owned and ghost); id_map as in get_state_owned.
int set_state_from(sb_solver *s, const float *pos, const float *vel, const int32_t *id_map) {
    int rc = set_device(s); if (rc) return rc;
    const sbp::LocalPlan &L = s->plan->local;
    flush_deferred(s);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    std::vector<float> hp((size_t)s->n_local * 3), hv((size_t)s->n_local * 3);
    for (int64_t l = 0; l < s->n_local; ++l) {
        int32_t o = L.local_to_old[l];
        if (id_map) o = id_map[o];
        for (int c = 0; c < 3; ++c) { hp[3 * (size_t)l + c] = pos[3 * (size_t)o + c]; hv[3 * (size_t)l + c] = vel[3 * (size_t)o + c]; }
    }
    HIP_CHECK(hipMemcpy(s->d_pos3.p, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(s->d_vel.p, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice));
    return SB_OK;
}

// Kinematic targets of the pinned particles THIS RANK OWNS among `count` entries (ids in the rank's numbering; an id this rank does not
// own is skipped: its owner applies it and the ghost copy arrives with the next exchange). Validation (range, inverse mass 0, no NaN, no
// id twice) covers every entry the rank can see. See sb_set_kinematic_positions.
int set_kinematic(sb_solver *s, const int32_t *ids, const float *pos, int32_t count) {
    int rc = set_device(s); if (rc) return rc;
    const std::vector<int32_t> &lof = local_of_old(s);
    if (s->kin_seen.size() != (size_t)s->n) s->kin_seen.assign((size_t)s->n, 0);
    const uint32_t stamp = ++s->kin_stamp;
    int32_t n_mine = 0;
    for (int32_t k = 0; k < count; ++k) {
        if (ids[k] < 0 || ids[k] >= s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle index out of range");
        if (s->invm[(size_t)ids[k]] != 0.0f)
            return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle " + std::to_string(ids[k]) + " has a non-zero inverse mass (only pinned particles are kinematic)");
        for (int c = 0; c < 3; ++c) if (!(pos[3 * (size_t)k + c] == pos[3 * (size_t)k + c])) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: NaN");
        // each id at most once: twice would be an order-dependent result and, in the fused path, a write race on one target slot
        if (s->kin_seen[(size_t)ids[k]] == stamp) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle " + std::to_string(ids[k]) + " appears twice");
        s->kin_seen[(size_t)ids[k]] = stamp;
        const int32_t l = lof[(size_t)ids[k]];
        if (l >= 0 && l < s->n_owned) ++n_mine;
    }
    if (n_mine == 0) return SB_OK;
    if (s->kin_pending >= 0) flush_deferred(s);       // two moves without a tick between them: the earlier one takes effect first
    const int q = s->kin_next;
    s->kin_next = (q + 1) % sb_solver::kKinSlots;
    if (!s->ev_kin[q]) HIP_CHECK(hipEventCreateWithFlags(&s->ev_kin[q], hipEventDisableTiming));
    else HIP_CHECK(hipEventSynchronize(s->ev_kin[q]));       // the kernel that read this table (four calls ago) is done
    if (s->kin_cap[q] < (size_t)n_mine) {
        if (s->h_kin_idx[q]) { (void)hipHostFree(s->h_kin_idx[q]); s->h_kin_idx[q] = nullptr; }
        if (s->h_kin_pos[q]) { (void)hipHostFree(s->h_kin_pos[q]); s->h_kin_pos[q] = nullptr; }
        const size_t cap = std::max<size_t>(256, (size_t)n_mine * 2);
        // mapped pinned memory: the kernels read the tables in place, through the device-side alias of the allocation
        HIP_CHECK(hipHostMalloc((void **)&s->h_kin_idx[q], cap * sizeof(int32_t), hipHostMallocMapped));
        HIP_CHECK(hipHostMalloc((void **)&s->h_kin_pos[q], cap * 3 * sizeof(float), hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer((void **)&s->d_kin_idx[q], s->h_kin_idx[q], 0));
        HIP_CHECK(hipHostGetDevicePointer((void **)&s->d_kin_pos[q], s->h_kin_pos[q], 0));
        s->kin_cap[q] = cap;
    }
    int32_t w = 0;
    for (int32_t k = 0; k < count; ++k) {
        const int32_t l = lof[(size_t)ids[k]];
        if (l < 0 || l >= s->n_owned) continue;
        s->h_kin_idx[q][w] = l;
        for (int c = 0; c < 3; ++c) s->h_kin_pos[q][3 * (size_t)w + c] = pos[3 * (size_t)k + c];
        ++w;
    }
    // PENDING until the next tick starts (the previous tick's held-back last kernel still reads the old positions of these
    // particles): a fused first kernel takes them along, every other way across the tick boundary scatters them (flush_deferred)
    s->kin_pending = q; s->kin_pending_count = n_mine;
    return SB_OK;
}

}  // namespace sbi

extern "C" {

int sb_set_state(sb_solver *s, const float *pos, const float *vel, int32_t n) {
    if (!s || !pos || !vel) return fail(SB_ERR_INVALID_ARG, "sb_set_state: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_set_state before sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_state: n differs from sb_set_particles");
    return guarded([&]() -> int { return set_state_from(s, pos, vel, nullptr); });
}

int sb_set_kinematic_positions(sb_solver *s, const int32_t *ids, const float *pos, int32_t count) {
    if (!s || count < 0 || (count > 0 && (!ids || !pos))) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: bad argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_set_kinematic_positions before sb_finalize");
    if (count == 0) return SB_OK;
    return guarded([&]() -> int { return set_kinematic(s, ids, pos, count); });
}

/* ---- asynchronous render readback -------------------------------------------- */

int sb_readback_begin(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_readback_begin: null handle");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_readback_begin before sb_finalize");
    if (s->snap_pending == 2) return fail(SB_ERR_STATE, "sb_readback_begin: two snapshots already pending, call sb_readback_end");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        if (!s->copy_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
            if (!s->d_local_to_old.p) s->d_local_to_old.upload(s->plan->local.local_to_old, s->dev_bytes);
            for (int k = 0; k < sb_solver::kSnapSlots; ++k) {
                s->d_snap[k].alloc((size_t)s->n * 3, s->dev_bytes);
                HIP_CHECK(hipMemset(s->d_snap[k].p, 0, (size_t)s->n * 3 * sizeof(float)));
                HIP_CHECK(hipHostMalloc((void **)&s->h_snap[k], (size_t)s->n * 3 * sizeof(float), hipHostMallocDefault));
                std::memset(s->h_snap[k], 0, (size_t)s->n * 3 * sizeof(float));
                HIP_CHECK(hipEventCreateWithFlags(&s->ev_snap[k], hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&s->ev_copied[k], hipEventDisableTiming));
            }
        }
        const int k = (s->snap_head + s->snap_pending) % sb_solver::kSnapSlots;
        // snapshot on the compute stream (ordered after every tick enqueued so far, before the next one) ...
        const bool compact = s->render_set_only && !s->render_tri.empty();
        // a rank of a partitioned solver serves the render particles it OWNS; vertex normals need the neighbours' particles too and are
        // computed on the gathered snapshot (sb_group_readback_*), not per rank
        const bool single = s->desc.world == 1;
        if (!s->render_tri.empty() && s->render_dirty) {     // (re)build the incident-triangle lists: triangle ids ascending per particle
            HIP_CHECK(hipStreamSynchronize(s->copy_stream));
            const int64_t m = (int64_t)s->render_tri.size() / 3;
            std::vector<int32_t> off((size_t)s->n + 1, 0), adj((size_t)3 * m);
            for (int64_t c = 0; c < 3 * m; ++c) ++off[(size_t)s->render_tri[c] + 1];
            s->render_set.clear();
            const std::vector<int32_t> &lof = local_of_old(s);
            for (int32_t v = 0; v < s->n; ++v) {
                if (off[(size_t)v + 1] && lof[(size_t)v] >= 0 && lof[(size_t)v] < s->n_owned) s->render_set.push_back(v);
                off[(size_t)v + 1] += off[v];
            }
            std::vector<int32_t> cur(off.begin(), off.end() - 1);
            for (int64_t t = 0; t < m; ++t)
                for (int j = 0; j < 3; ++j) adj[(size_t)cur[s->render_tri[3 * t + j]]++] = (int32_t)t;
            std::vector<int32_t> local_of(s->render_set.size());
            for (size_t q = 0; q < local_of.size(); ++q) local_of[q] = lof[(size_t)s->render_set[q]];
            s->render_local = local_of;
            s->d_tri.upload(s->render_tri, s->dev_bytes);
            s->d_adj_off.upload(off, s->dev_bytes);
            s->d_adj_tri.upload(adj, s->dev_bytes);
            s->d_render_set.upload(s->render_set, s->dev_bytes);
            s->d_render_local.upload(local_of, s->dev_bytes);
            for (int q = 0; q < sb_solver::kSnapSlots; ++q) {
                if (single && !s->h_nrm[q]) {
                    s->d_nrm[q].alloc((size_t)s->n * 3, s->dev_bytes);
                    HIP_CHECK(hipHostMalloc((void **)&s->h_nrm[q], (size_t)s->n * 3 * sizeof(float), hipHostMallocDefault));
                }
                s->d_cpos[q].alloc(s->render_set.size() * 3, s->dev_bytes);
                if (s->h_cpos[q]) { (void)hipHostFree(s->h_cpos[q]); s->h_cpos[q] = nullptr; }
                HIP_CHECK(hipHostMalloc((void **)&s->h_cpos[q], std::max<size_t>(s->render_set.size(), 1) * 3 * sizeof(float), hipHostMallocDefault));
            }
            s->render_dirty = false;
            s->n_peek_tiles = -1;
        }
        sbk::PosView src = s->pos_view();
        // the tick's last kernel is deferred: snapshot a peek and leave it deferred
        src.xyz = const_cast<float *>(render_source(s, compact, s->render_local));
        if (compact) {      // only the render set leaves the device: snapshot just those particles
            const int cnt = (int)s->render_set.size();
            if (cnt && single)      // (into the caller-numbered array: the normals kernel gathers neighbours by caller id and emits the compact arrays)
                hipLaunchKernelGGL(sbk::snapshot_subset_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, src,
                                   s->d_render_set.p, s->d_render_local.p, s->d_snap[k].p, cnt);
            else if (cnt)           // (no normals here: straight into the compact array)
                hipLaunchKernelGGL(sbk::snapshot_compact_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, src,
                                   s->d_render_local.p, s->d_cpos[k].p, cnt);
        } else if (s->n_owned)
            hipLaunchKernelGGL(sbk::snapshot_kernel, dim3((unsigned)((s->n_owned + 255) / 256)), dim3(256), 0, s->stream,
                               src, s->d_local_to_old.p, s->d_snap[k].p, (int)s->n_owned);
        HIP_CHECK(hipEventRecord(s->ev_snap[k], s->stream));
        // ... D2H on the copy stream, overlapping whatever the compute stream does next
        HIP_CHECK(hipStreamWaitEvent(s->copy_stream, s->ev_snap[k], 0));
        if (!compact)
            HIP_CHECK(hipMemcpyAsync(s->h_snap[k], s->d_snap[k].p, (size_t)s->n * 3 * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream));
        s->snap_has_normals[k] = false;
        s->snap_compact[k] = compact;
        s->snap_has_render_set[k] = !s->render_tri.empty();
        if (compact && !single && !s->render_set.empty())
            HIP_CHECK(hipMemcpyAsync(s->h_cpos[k], s->d_cpos[k].p, s->render_set.size() * 3 * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream));
        if (!s->render_tri.empty() && single) {
            const int count = compact ? (int)s->render_set.size() : (int)s->n;
            hipLaunchKernelGGL(sbk::normals_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s->copy_stream, s->d_snap[k].p,
                               s->d_adj_off.p, s->d_adj_tri.p, s->d_tri.p, s->d_nrm[k].p, count,
                               compact ? s->d_render_set.p : (const int32_t *)nullptr, compact ? s->d_cpos[k].p : (float *)nullptr);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(s->h_nrm[k], s->d_nrm[k].p, (size_t)count * 3 * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream));
            if (compact)
                HIP_CHECK(hipMemcpyAsync(s->h_cpos[k], s->d_cpos[k].p, (size_t)count * 3 * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream));
            s->snap_has_normals[k] = true;
        }
        HIP_CHECK(hipEventRecord(s->ev_copied[k], s->copy_stream));
        ++s->snap_pending;
        return SB_OK;
    });
}

int sb_readback_end(sb_solver *s, const float **pos_xyz_out) {
    if (!s || !pos_xyz_out) return fail(SB_ERR_INVALID_ARG, "sb_readback_end: null argument");
    if (s->snap_pending == 0) return fail(SB_ERR_STATE, "sb_readback_end without a pending sb_readback_begin");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        const int k = s->snap_head;
        HIP_CHECK(hipEventSynchronize(s->ev_copied[k]));
        check_peer_error(s);       // (the snapshot was taken behind every tick enqueued before it)
        *pos_xyz_out = s->snap_compact[k] ? s->h_cpos[k] : s->h_snap[k];
        s->snap_last_ended = k;
        s->snap_head = (s->snap_head + 1) % sb_solver::kSnapSlots; --s->snap_pending;
        return SB_OK;
    });
}

int sb_set_render_triangles(sb_solver *s, const int32_t *tri, int32_t m) {
    if (!s || m < 0 || (m > 0 && !tri)) return fail(SB_ERR_INVALID_ARG, "sb_set_render_triangles: bad argument");
    if (s->n <= 0) return fail(SB_ERR_STATE, "sb_set_render_triangles before sb_set_particles");
    if (s->snap_pending) return fail(SB_ERR_STATE, "sb_set_render_triangles while a readback is pending");
    return guarded([&]() -> int {
        for (int64_t c = 0; c < 3 * (int64_t)m; ++c)
            if (tri[c] < 0 || tri[c] >= s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_render_triangles: particle index out of range");
        s->render_tri.assign(tri, tri + 3 * (size_t)m);
        s->render_dirty = true;
        if (m == 0) s->render_set_only = false;
        for (bool &b : s->snap_has_normals) b = false;
        for (bool &b : s->snap_has_render_set) b = false;
        return SB_OK;
    });
}

int sb_readback_get_normals(sb_solver *s, const float **out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_readback_get_normals: null argument");
    if (s->desc.world > 1)
        return fail(SB_ERR_UNSUPPORTED, "sb_readback_get_normals: a rank of a partitioned solver does not hold its neighbours' particles; vertex normals of a partitioned "
                    "body are computed on the gathered snapshot (sb_group_readback_get_normals)");
    if (s->snap_last_ended < 0 || !s->snap_has_normals[s->snap_last_ended])
        return fail(SB_ERR_STATE, "sb_readback_get_normals: no finished readback with render triangles set");
    *out = s->h_nrm[s->snap_last_ended];
    return SB_OK;
}

int sb_set_readback_render_set_only(sb_solver *s, int32_t on) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_set_readback_render_set_only: null handle");
    if (s->snap_pending) return fail(SB_ERR_STATE, "sb_set_readback_render_set_only while a readback is pending");
    if (on && s->render_tri.empty()) return fail(SB_ERR_STATE, "sb_set_readback_render_set_only: set the render triangles first");
    s->render_set_only = on != 0;
    return SB_OK;
}

int sb_readback_get_render_set(sb_solver *s, const int32_t **ids, int32_t *count) {
    if (!s || !ids || !count) return fail(SB_ERR_INVALID_ARG, "sb_readback_get_render_set: null argument");
    if (s->snap_last_ended < 0 || !s->snap_has_render_set[s->snap_last_ended])
        return fail(SB_ERR_STATE, "sb_readback_get_render_set: no finished readback with render triangles set");
    *ids = s->render_set.data();
    *count = (int32_t)s->render_set.size();
    return SB_OK;
}

}  // extern "C"
