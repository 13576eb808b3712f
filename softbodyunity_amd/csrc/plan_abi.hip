// plan_abi.hip — host-only planner inspection (include/softbody_plan.h): works without a GPU
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"

using namespace sbi;

extern "C" {

/* ---- plan inspection (host only) ---------------------------------------------------------------- */

int sb_plan_build(const float *rest, int32_t n, const int32_t *dist_ij, int32_t m_d, const int32_t *vol, int32_t m_v,
                  const int32_t *bend, int32_t m_b, const sb_plan_opts *opts, sb_plan **out) {
    if (!rest || !out || n <= 0 || m_d < 0 || m_v < 0 || m_b < 0) return fail(SB_ERR_INVALID_ARG, "sb_plan_build: bad argument");
    *out = nullptr;
    return guarded([&]() -> int {
        if (opts && (opts->partition < SB_PARTITION_AUTO || opts->partition > SB_PARTITION_RCB)) return fail(SB_ERR_INVALID_ARG, "sb_plan_build: bad partition");
        if (opts && (opts->plan_flags & ~kPlanFlagsAll)) return fail(SB_ERR_INVALID_ARG, "sb_plan_build: unknown bit in plan_flags");
        // (opts == NULL: every field 0 -- one rank, automatic tile size by the same rule as sb_finalize)
        if (opts && ((opts->domain != nullptr) != (opts->global_id != nullptr))) return fail(SB_ERR_INVALID_ARG, "sb_plan_build: domain and global_id go together");
        const sbp::Opts o = opts ? plan_opts(opts->rank, opts->world, opts->part_dims, opts->tile_particles, opts->partition, opts->plan_flags, m_v, m_b, opts->domain)
                                 : plan_opts(0, 1, nullptr, 0, SB_PARTITION_AUTO, 0u, m_v, m_b);
        sbp::Input in = make_input(rest, n, dist_ij, m_d, vol, m_v, bend, m_b);
        if (opts && opts->domain) in.global_id = opts->global_id;
        auto p = std::make_unique<sb_plan>();
        sbp::build_plan(in, o, p->plan);
        sbp::extract_local(p->plan, in, o.rank, p->local);
        *out = p.release();
        return SB_OK;
    });
}
int sb_plan_destroy(sb_plan *p) {
    if (!p) return fail(SB_ERR_INVALID_ARG, "sb_plan_destroy: null");
    delete p;
    return SB_OK;
}
int sb_get_plan(sb_solver *s, const sb_plan **out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_get_plan: null");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_plan before sb_finalize");
    *out = s->plan.get();
    return SB_OK;
}
#define PARITY_OK(fn) if (!p || parity < 0 || parity > 1) return fail(SB_ERR_INVALID_ARG, fn ": null plan or parity not 0/1")
int64_t sb_plan_order_count(const sb_plan *p) { return p ? (int64_t)p->plan.order_id[0].size() : -1; }
int sb_plan_get_order(const sb_plan *p, int32_t parity, uint8_t *type_out, int32_t *id_out) {
    PARITY_OK("sb_plan_get_order");
    if (!type_out || !id_out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_order: null");
    std::memcpy(type_out, p->plan.order_type[parity].data(), p->plan.order_type[parity].size());
    std::memcpy(id_out, p->plan.order_id[parity].data(), p->plan.order_id[parity].size() * sizeof(int32_t));
    return SB_OK;
}
int32_t sb_plan_phase_count(const sb_plan *p, int32_t parity) {
    if (!p || parity < 0 || parity > 1) return -1;
    return (int32_t)p->plan.phases[parity].size();
}
int sb_plan_get_phases(const sb_plan *p, int32_t parity, sb_phase_info *out) {
    PARITY_OK("sb_plan_get_phases");
    if (!out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_phases: null");
    for (size_t k = 0; k < p->plan.phases[parity].size(); ++k) {
        const sbp::Phase &F = p->plan.phases[parity][k];
        out[k].kind = F.kind; out[k].type = F.type; out[k].tiling = F.tiling; out[k].halo_slot = F.halo_slot;
        out[k].order_begin = F.order_begin; out[k].order_end = F.order_end;
        out[k].task_begin = F.task_begin; out[k].task_end = F.task_end;
    }
    return SB_OK;
}
int64_t sb_plan_task_count(const sb_plan *p, int32_t parity) {
    if (!p || parity < 0 || parity > 1) return -1;
    return (int64_t)p->plan.task_off[parity].size() - 1;
}
int sb_plan_get_tasks(const sb_plan *p, int32_t parity, int64_t *out) {
    PARITY_OK("sb_plan_get_tasks");
    if (!out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_tasks: null");
    std::memcpy(out, p->plan.task_off[parity].data(), p->plan.task_off[parity].size() * sizeof(int64_t));
    return SB_OK;
}
int64_t sb_plan_group_count(const sb_plan *p, int32_t parity) {
    if (!p || parity < 0 || parity > 1) return -1;
    return (int64_t)p->plan.group_off[parity].size() - 1;
}
int sb_plan_get_groups(const sb_plan *p, int32_t parity, int64_t *out) {
    PARITY_OK("sb_plan_get_groups");
    if (!out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_groups: null");
    std::memcpy(out, p->plan.group_off[parity].data(), p->plan.group_off[parity].size() * sizeof(int64_t));
    return SB_OK;
}
int sb_plan_get_owner(const sb_plan *p, int32_t *out) {
    if (!p || !out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_owner: null");
    std::memcpy(out, p->plan.owner_of_old.data(), p->plan.owner_of_old.size() * sizeof(int32_t));
    return SB_OK;
}
int64_t sb_plan_local_count(const sb_plan *p, int64_t *owned_out) {
    if (!p) return -1;
    if (owned_out) *owned_out = p->local.n_owned;
    return (int64_t)p->local.local_to_old.size();
}
int sb_plan_get_local_particles(const sb_plan *p, int32_t *out) {
    if (!p || !out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_local_particles: null");
    std::memcpy(out, p->local.local_to_old.data(), p->local.local_to_old.size() * sizeof(int32_t));
    return SB_OK;
}
int32_t sb_plan_halo_slot_count(const sb_plan *p) { return p ? (int32_t)p->local.halo.size() : -1; }
int sb_plan_halo_counts(const sb_plan *p, int32_t slot, int32_t *send_cnt, int32_t *recv_cnt) {
    if (!p || !send_cnt || !recv_cnt) return fail(SB_ERR_INVALID_ARG, "sb_plan_halo_counts: null");
    if (slot < 0 || slot >= (int32_t)p->local.halo.size()) return fail(SB_ERR_INVALID_ARG, "sb_plan_halo_counts: bad slot");
    const sbp::HaloSlot &H = p->local.halo[slot];
    for (int r = 0; r < p->local.world; ++r) {
        send_cnt[r] = (int32_t)H.send_idx[r].size();
        recv_cnt[r] = (int32_t)H.recv_idx[r].size();
    }
    return SB_OK;
}
int sb_plan_get_halo(const sb_plan *p, int32_t slot, int32_t peer, int32_t *send_ids, int32_t *recv_ids) {
    if (!p) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_halo: null");
    if (slot < 0 || slot >= (int32_t)p->local.halo.size() || peer < 0 || peer >= p->local.world)
        return fail(SB_ERR_INVALID_ARG, "sb_plan_get_halo: bad slot/peer");
    const sbp::HaloSlot &H = p->local.halo[slot];
    // published as caller-numbering (global) particle ids
    if (send_ids) for (size_t k = 0; k < H.send_idx[peer].size(); ++k) send_ids[k] = p->local.local_to_old[H.send_idx[peer][k]];
    if (recv_ids) for (size_t k = 0; k < H.recv_idx[peer].size(); ++k) recv_ids[k] = p->local.local_to_old[H.recv_idx[peer][k]];
    return SB_OK;
}
int sb_plan_get_pair_hashes(const sb_plan *p, uint64_t *out) {
    if (!p || !out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_pair_hashes: null");
    for (int r = 0; r < p->local.world; ++r) out[r] = p->local.pair_hash.empty() ? 0 : p->local.pair_hash[(size_t)r];
    return SB_OK;
}
int sb_plan_get_local_order_mask(const sb_plan *p, int32_t parity, uint8_t *out) {
    PARITY_OK("sb_plan_get_local_order_mask");
    if (!out) return fail(SB_ERR_INVALID_ARG, "sb_plan_get_local_order_mask: null");
    std::memcpy(out, p->local.order_mask[parity].data(), p->local.order_mask[parity].size());
    return SB_OK;
}

}  // extern "C"
