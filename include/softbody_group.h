/*
 * softbody_group.h — ONE process driving SEVERAL MI355X behind one `Softbody : MonoBehaviour` component.
 *
 * A Unity player is one process. The per-rank entry points of softbody.h (sb_desc.rank / world, sb_comm_init) are the form a
 * one-process-per-GPU job uses (bench.py under torch.distributed.run); a component in a player uses this header instead: one handle
 * that owns one solver per device, authored ONCE with the whole mesh, stepped with ONE call, read back gathered in the caller's
 * numbering -- csharp/Softbody.cs `deviceCount`.
 *
 * Reference interface replaced: NONE EXISTS (/root/reference/README.md:1 is the whole reference tree). [BUILDER-DEFINED] from
 * BASELINE.json:5 ("keeping the reference's Softbody component / MonoBehaviour update API surface ... the mesh is spatially partitioned
 * across the 8 GPUs of one node with RCCL halo exchange") and SURVEY.md §1 L1' ("one process x 8 devices"), §8b (`device_count`).
 *
 * How a tick runs. Two host models (sb_group_create flags), same results:
 *   default        ONE HOST THREAD PER RANK, owned by the group: sb_group_step hands the tick to the ranks' threads, each enqueues its own
 *                  rank's launches and exchanges (exactly what a one-process-per-GPU job does, hipGraph replay and captured schedules included)
 *                  and the call returns when all have ENQUEUED -- the host cost of a tick is that of one rank, whatever the number of GPUs.
 *                  A tick of 256^3 on 8 ranks is ~30 launches + 10 exchanges per rank and ~0.7 ms of GPU time: one thread issuing all of it for
 *                  8 devices would be the bottleneck (measured on one device: profiles/r04_group_host_models.txt).
 *   SB_GROUP_WALK  no thread of the plugin's own: the CALLING thread walks the tick LAUNCH BY LAUNCH across the ranks (every rank's tile
 *                  kernel K_it, then every rank's next launch, ...), never rank by rank, so the launches of one step run side by side on the
 *                  devices; a ghost exchange is issued for all ranks together: pack kernels on every rank's stream, then the sends and receives
 *                  of ALL ranks inside ONE ncclGroupStart / ncclGroupEnd (a single thread that posted one rank's receive outside a group would
 *                  wait for a send it has not posted yet), then the unpack kernels (none where the T1 kernels read the receive buffer
 *                  themselves). The communicators are created together inside one ncclGroupStart / ncclGroupEnd. Eager schedules only, and
 *                  SB_SCHEDULE_AUTO is the serialised one here (its measurement of the two eager schedules runs on the ranks' own threads).
 * Transports (sb_desc.halo_transport): SB_TRANSPORT_RCCL as above; SB_TRANSPORT_PEER = the mailbox transport of softbody.h with the mailboxes
 * connected by plain pointer (sb_group_finalize enables peer access between the devices): push and unpack kernels only, nothing on the host.
 * The lazy tick boundary, the peek, kinematic targets and the render readback work as for a single solver; results are bit-identical to a
 * single-device solver of the same mesh (same published order: the plan does not depend on the partition).
 *
 * UNVERIFIED BETWEEN TWO DEVICES: the build and test boxes have one GPU. Everything here is verified with all ranks on ONE device
 * (tests/test_gpu_group.py: bitwise against the CPU oracle); several ranks on one device need a hardware queue each for the PEER transport
 * (environment GPU_MAX_HW_QUEUES >= ranks); RCCL refuses two ranks on one device, so its legs run as self-exchanges there (SB_DEBUG_LOOPBACK).
 */
#ifndef SOFTBODY_MI355X_GROUP_H
#define SOFTBODY_MI355X_GROUP_H

#include "softbody.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SB_GROUP_WALK 1u               /* sb_group_create flags: no plugin threads, the calling thread walks the tick across the ranks (see above) */
#define SB_GROUP_WHOLE_MESH 2u         /* never cut windows: every rank is handed (and plans) the whole mesh, whatever the partition (see sb_group_finalize) */

typedef struct sb_group sb_group;      /* opaque, plugin-owned */

/* ---- lifecycle ------------------------------------------------------------------------------------------------------------------------- */
/* desc: the settings of EVERY rank (gravity, damping, tile_particles, partition, part_dims, plan_flags, halo_transport, halo_schedule,
 * debug_flags); its device / rank / world fields are ignored. devices: n_devices HIP ordinals, rank r runs on devices[r] (NULL = 0, 1, ...,
 * n_devices - 1); an ordinal may repeat (several ranks on one device: tests). n_devices = 1 is a plain single-device solver behind the same
 * entry points, so a component needs only this header. The ranks' solvers exist from here on (sb_group_get_rank: sb_set_tuning).
 * Calling convention: calls on ONE group must not overlap in time (a component calls from Unity's main thread; the group's own threads
 * are an implementation detail behind each call); different groups are independent. A call that fails changes nothing it can avoid
 * changing, with one exception: after a failed sb_group_finalize (or a failed sb_group_step: a rank's device or transport error) the group
 * is good for sb_group_destroy only -- its ranks may no longer be in the same tick state. */
int sb_group_create(const sb_desc *desc, const int32_t *devices, int32_t n_devices, uint32_t flags, sb_group **out);
int sb_group_destroy(sb_group *g);

/* ---- authoring: the whole mesh, once (same arguments as the sb_set_* of softbody.h) ------------------------------------------------------- */
int sb_group_set_particles(sb_group *g, const float *pos_xyz, const float *vel_xyz /* may be NULL = 0 */, const float *inv_mass, int32_t n);
int sb_group_set_rest_positions(sb_group *g, const float *rest_xyz, int32_t n);
int sb_group_set_distance_constraints(sb_group *g, const int32_t *ij, const float *rest_len, int32_t m, float compliance);
int sb_group_set_volume_constraints(sb_group *g, const int32_t *ijkl, const float *rest_vol, int32_t m, float compliance);
int sb_group_set_bending_constraints(sb_group *g, const int32_t *ijkl, const float *rest_cs, int32_t m, float compliance);
int sb_group_set_ground_plane(sb_group *g, float nx, float ny, float nz, float d, int32_t enabled);
/* Plan, partition, upload on every device (the ranks plan side by side on host threads), connect the transport, verify that the ranks
 * planned consistently (plan hash + pair hashes, as sb_finalize does across processes).
 * Sharded authoring: every rank is handed only ITS WINDOW of the mesh (cut here from the one copy the group holds) instead of the whole
 * mesh where that is known to reproduce the whole-mesh plan -- a LATTICE-LIKE body: distance constraints only, filling its bounding box
 * (sb_domain.fill = 1) -- and the partition is the block grid: sb_desc.partition = SB_PARTITION_BLOCKS, or SB_PARTITION_AUTO on at least
 * 2 M particles (for which AUTO takes the block grid anyway; below that, eight whole-mesh plans cost less than they save). Every other mesh
 * (tets, hinges, a body that fills its box unevenly: their colouring and leftover layers are not local to a window) is planned whole on
 * every rank, under whatever partition was asked for (SB_PARTITION_AUTO may then choose RCB). SB_GROUP_WHOLE_MESH forces that for any mesh.
 * The rule is a filter, not the guarantee: the ranks' plans are compared before anything is uploaded, and where windows did NOT reproduce the
 * whole-mesh plan (e.g. particles on a line joined by nearest-neighbour springs pass the filter) every rank is handed the whole mesh and
 * plans again -- sb_group_finalize succeeds either way, only slower. */
int sb_group_finalize(sb_group *g);

/* ---- the hot path (FixedUpdate) ---------------------------------------------------------------------------------------------------------- */
int sb_group_step(sb_group *g, float dt, int32_t substeps);

/* ---- state, gathered in the caller's numbering --------------------------------------------------------------------------------------------- */
int sb_group_get_positions(sb_group *g, float *pos_xyz_out, int32_t n);      /* every entry written: each rank delivers what it owns */
int sb_group_get_velocities(sb_group *g, float *vel_xyz_out, int32_t n);
int sb_group_set_state(sb_group *g, const float *pos_xyz, const float *vel_xyz, int32_t n);
/* Kinematic targets (sb_set_kinematic_positions): ids in the caller's numbering, each at most once; every rank takes the ones it owns. */
int sb_group_set_kinematic_positions(sb_group *g, const int32_t *ids, const float *pos_xyz, int32_t count);

/* ---- render readback (same contract as sb_readback_* / sb_set_render_triangles of softbody.h) -------------------------------------------- */
/* Every rank snapshots the particles it owns (peeking while its tick's last kernel is held back) straight into ONE buffer on the render device
 * (rank 0's; peer stores), whole array or render set; the vertex normals are computed there on the gathered snapshot, then one copy to pinned
 * host memory. */
int sb_group_set_render_triangles(sb_group *g, const int32_t *tri_abc, int32_t m);
int sb_group_set_readback_render_set_only(sb_group *g, int32_t render_set_only);
int sb_group_readback_begin(sb_group *g);
int sb_group_readback_end(sb_group *g, const float **pos_xyz_out);
int sb_group_readback_get_normals(sb_group *g, const float **normal_xyz_out);
int sb_group_readback_get_render_set(sb_group *g, const int32_t **ids_out, int32_t *count_out);

/* ---- synchronisation, inspection ------------------------------------------------------------------------------------------------------------ */
int sb_group_synchronize(sb_group *g);
int32_t sb_group_rank_count(const sb_group *g);
/* Borrowed handle of rank r (valid until sb_group_destroy) for sb_get_stats, sb_get_owner, sb_debug_validate, sb_get_plan: inspection only
 * -- stepping or reading state through it desynchronises the group. */
int sb_group_get_rank(sb_group *g, int32_t rank, sb_solver **out);

#ifdef __cplusplus
}
#endif
#endif /* SOFTBODY_MI355X_GROUP_H */
