/*
 * softbody_plan.h — host-only planner entry points of libsoftbody_mi355x.so: the published constraint order (SPEC.md §3), tiles,
 * partition and halo lists of a mesh, and the frame / window helpers of sharded authoring. Pure host code: every function here works
 * on a machine without any GPU (the C# component's CPU branch, config 1 of BASELINE.json, calls sb_plan_build / sb_plan_get_order only).
 *
 * Reference interface replaced: NONE EXISTS (/root/reference/README.md:1 is the whole reference tree); [BUILDER-DEFINED], SURVEY.md §8b.
 */
#ifndef SOFTBODY_MI355X_PLAN_H
#define SOFTBODY_MI355X_PLAN_H

#include "softbody.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sb_plan sb_plan;     /* opaque, plugin-owned */

/* ---- plan inspection (pure host code; works without a GPU) ------------------------------------ */
/* The planner (SPEC.md §3) publishes one sequential constraint order per substep parity (0,1,0,1,...
 * restarting at 0 every tick). parity arguments below are 0 or 1. */
typedef struct {
    int32_t rank, world;
    int32_t part_dims[3];
    int32_t tile_particles;  /* as sb_desc.tile_particles (0 = automatic by the same rule); opts == NULL: all fields 0 */
    int32_t partition;       /* SB_PARTITION_* */
    uint32_t plan_flags;     /* SB_PLAN_* */
    const sb_domain *domain;    /* NULL = the input is the whole mesh; else the input is this rank's window of it ... */
    const int32_t *global_id;   /* ... and these are its particles' ids in the whole mesh, strictly ascending (n of them) */
} sb_plan_opts;
/* The frame of a whole mesh, measured exactly as a whole-mesh plan measures it (a lattice generator can also fill sb_domain in
 * closed form: lo / hi = the lattice's corners, spacing = the spring length). */
int sb_domain_from_mesh(const float *rest_xyz, int32_t n, const int32_t *dist_ij, int32_t m_d, const int32_t *vol_ijkl, int32_t m_v,
                        const int32_t *bend_ijkl, int32_t m_b, sb_domain *out);
/* The box (rest coordinates, lo inclusive, hi exclusive; +-1e300 where the window reaches the rim of the grid) whose particles rank
 * opts->rank must hand over. Only rank, world, part_dims and tile_particles of opts are read. */
int sb_domain_window(const sb_domain *domain, const sb_plan_opts *opts, double lo_out[3], double hi_out[3]);
typedef struct {
    int32_t kind;               /* 0 = global colour, 1 = a tiling's tiles, first phase of the substep, 2 = the other tiling's tiles,
                                   last phase, 3 = the sparse tiles of one T2 layer (after kind 1, before the global colours) */
    int32_t type;               /* kind 0: constraint type 0/1/2; else -1 */
    int32_t tiling;             /* kind 1/2: 0 or 1; kind 0: -1 */
    int32_t halo_slot;          /* -1 none; 1 = before the T1 tile kernel; 2+c = before global colour c;
                                   2+G+l = before the kernel of T2 layer l (G = number of global colours) */
    int64_t order_begin, order_end; /* slice of the parity's published order */
    int64_t task_begin, task_end;   /* slice of the task table: tasks of one phase touch disjoint particles */
} sb_phase_info;

int sb_plan_build(const float *rest_xyz, int32_t n,
                  const int32_t *dist_ij, int32_t m_d,
                  const int32_t *vol_ijkl, int32_t m_v,
                  const int32_t *bend_ijkl, int32_t m_b,
                  const sb_plan_opts *opts, sb_plan **out);
int sb_plan_destroy(sb_plan *p);
/* Borrowed view of a finalized solver's plan (valid until sb_destroy). */
int sb_get_plan(sb_solver *s, const sb_plan **out);

int64_t sb_plan_order_count(const sb_plan *p);
/* The published sequential order (SPEC.md §3): type 0/1/2 + index into that type's input arrays. */
int sb_plan_get_order(const sb_plan *p, int32_t parity, uint8_t *type_out, int32_t *id_out);
int32_t sb_plan_phase_count(const sb_plan *p, int32_t parity);
int sb_plan_get_phases(const sb_plan *p, int32_t parity, sb_phase_info *out);
int64_t sb_plan_task_count(const sb_plan *p, int32_t parity);
int sb_plan_get_tasks(const sb_plan *p, int32_t parity, int64_t *task_off_out /* task_count+1 */);
/* Finest independent sets (one round of one tile / one chunk of a global colour): constraints of one
 * group share no particle; the GPU runs a group's constraints concurrently. */
int64_t sb_plan_group_count(const sb_plan *p, int32_t parity);
int sb_plan_get_groups(const sb_plan *p, int32_t parity, int64_t *group_off_out /* group_count+1 */);
int sb_plan_get_owner(const sb_plan *p, int32_t *owner_rank_out /* n */);
/* Per-rank view: particles this rank keeps (owned first, then ghosts), in device order. */
int64_t sb_plan_local_count(const sb_plan *p, int64_t *owned_out);
int sb_plan_get_local_particles(const sb_plan *p, int32_t *global_id_out);
/* Halo schedule of this rank for halo slot `slot` (sb_phase_info.halo_slot): for every peer, which of its
 * own particles it sends and which ghosts it receives (caller particle ids, identical order on both
 * sides). Slot 1 carries positions and previous positions, slots >= 2 positions only. */
int32_t sb_plan_halo_slot_count(const sb_plan *p);
int sb_plan_halo_counts(const sb_plan *p, int32_t slot, int32_t *send_count_per_rank /* world */,
                        int32_t *recv_count_per_rank /* world */);
int sb_plan_get_halo(const sb_plan *p, int32_t slot, int32_t peer, int32_t *send_ids, int32_t *recv_ids);
/* Per peer rank: a hash of everything this rank and that peer must agree on -- the ghost lists between them (whole-mesh particle
 * ids, both directions, every slot) and the programs of the tiles both run (constraint sequences as whole-mesh particle ids).
 * Symmetric: rank a's entry for b equals rank b's entry for a exactly when the two planned consistently; sb_finalize (RCCL) and the
 * peer transport's link step compare them. out_per_rank: world entries, the own rank's is 0. */
int sb_plan_get_pair_hashes(const sb_plan *p, uint64_t *out_per_rank);
/* Which order entries this rank executes (1) or skips (0) — cut constraints run on every rank that owns
 * one of their particles. */
int sb_plan_get_local_order_mask(const sb_plan *p, int32_t parity, uint8_t *mask_out);

#ifdef __cplusplus
}
#endif
#endif /* SOFTBODY_MI355X_PLAN_H */
