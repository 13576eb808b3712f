/*
 * softbody.h — C ABI of libsoftbody_mi355x.so, the MI355X-native plugin behind the Unity
 * `Softbody : MonoBehaviour` component (Start -> sb_create/sb_set_x/sb_finalize, FixedUpdate -> sb_step +
 * sb_get_positions, OnDestroy -> sb_destroy).
 *
 * Reference interface replaced: NONE EXISTS. The reference tree is one line
 * (/root/reference/README.md:1, "# SoftbodyUnity"); it holds no C#, no plugin and no FFI. The export
 * set below is therefore the [BUILDER-DEFINED] boundary of SURVEY.md §8b, fixed only in kind by
 * BASELINE.json:5 ("host code in C# ... calling into a thin C-ABI native plugin"). The C# side that
 * binds it is csharp/SoftbodyNative.cs ([DllImport("softbody_mi355x", CallingConvention = Cdecl)]);
 * softbodyunity_amd/native.py is the ctypes mirror used by the tests.
 *
 * Conventions:
 *  - Host arrays are AoS: positions/velocities float xyz with 12-byte stride (Unity Vector3[] pins
 *    directly), indices int32. The caller owns every array argument; the plugin copies before returning
 *    and never retains a pointer.
 *  - Every function returns 0 (SB_OK) or a negative sb_status; no exception or abort crosses the
 *    boundary; sb_last_error() gives a thread-local, plugin-owned message for the last failure.
 *  - One handle is single-threaded; distinct handles are independent. sb_step is synchronous w.r.t.
 *    sb_get_*. There is NO CPU fallback: without a usable gfx950 device sb_create fails with
 *    SB_ERR_NO_DEVICE.
 *  - Semantics of one tick: SPEC.md.
 *
 * This header is the whole PRODUCT surface of a solver handle (32 functions). Beside it:
 *   softbody_group.h  one process -- a Unity player -- driving several GPUs behind the same component (sb_group_*)
 *   softbody_plan.h   host-only planner inspection (published order, tiles, halo lists; frame / window of sharded authoring)
 *   softbody_debug.h  test hooks, the table validator, per-launch timing and the tuning switches of A/B measurements
 */
#ifndef SOFTBODY_MI355X_H
#define SOFTBODY_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SB_ABI_VERSION 8

typedef enum {
    SB_OK = 0,
    SB_ERR_INVALID_ARG = -1,   /* null pointer, negative count, index out of range, bad enum */
    SB_ERR_STATE = -2,         /* call out of order (e.g. sb_step before sb_finalize) */
    SB_ERR_NO_DEVICE = -3,     /* no HIP device / not gfx950 / device index out of range */
    SB_ERR_HIP = -4,           /* a HIP runtime call failed; message has the hipError string */
    SB_ERR_RCCL = -5,          /* an RCCL call failed */
    SB_ERR_NOMEM = -6,
    SB_ERR_UNSUPPORTED = -7
} sb_status;

typedef struct sb_solver sb_solver; /* opaque, plugin-owned */

/* Spatial partition of a world > 1 solver (which rank owns which particles). Ownership is always by whole T0 cells. */
#define SB_PARTITION_AUTO   0   /* block grid when it balances the ranks within 10 % (regular meshes), else RCB */
#define SB_PARTITION_BLOCKS 1   /* part_dims[0] x part_dims[1] x part_dims[2] blocks of cells (lattices: 2x2x2 for 8) */
#define SB_PARTITION_RCB    2   /* recursive coordinate bisection over the occupied cells, weighted by constraint cost */

/* Switches that CHANGE THE PLAN (the published order, the tiles, the ghost lists). Every rank of a partitioned solver must
 * pass the same value: sb_finalize compares a hash of plan, options and halo lists across the ranks and fails with
 * SB_ERR_STATE on a mismatch. 0 = the product's plan; the bits exist for A/B measurements. */
#define SB_PLAN_NO_T2             1u   /* no third tiling: constraints inside neither T0 nor T1 go to global colours */
#define SB_PLAN_NO_THIRD_LIST     2u   /* irregular meshes: the first T2 layer takes the leftovers only */
#define SB_PLAN_NO_CLUSTER_LAYERS 4u   /* T2 layers from grids only */
#define SB_PLAN_NO_MIXED_GROUPS   8u   /* one constraint type per group of a tile */
#define SB_PLAN_NO_BANK_ORDER    16u   /* keep the colouring order inside a group (no LDS-bank-aware lane order) */
#define SB_PLAN_NO_TILE_MERGE     32u   /* irregular meshes: do not merge small tiles of a balanced list around a leftover constraint */
#define SB_PLAN_BALANCED_LISTS(n) ((uint32_t)(n) << 8)   /* irregular meshes: 1..3 balanced extra lists (grids) beside T0 / T1; 0 = default (2) */

/* Ghost exchange of a world > 1 solver. */
#define SB_TRANSPORT_RCCL 0            /* pack -> grouped ncclSend/ncclRecv -> unpack (default) */
#define SB_TRANSPORT_PEER 1            /* peer-store mailboxes (opt-in; see sb_peer_connect) */
#define SB_SCHEDULE_AUTO               0   /* an EAGER schedule (nothing is asked of the bound RCCL / HIP runtime beyond plain send/recv, sb_runtime_info),
                                              chosen by MEASUREMENT on the devices at hand: SB_SCHEDULE_SERIAL_EAGER, and where the overlapped eager
                                              schedule applies (RCCL, lattice-type plan) the first six ticks alternate between the two -- the bits are
                                              the same either way -- under HIP events; then every rank keeps the one whose slowest rank was faster
                                              (one all-gather; sb_stats.halo_schedule / halo_auto_*). The captured schedules stay opt-in; bench.py
                                              --gpus N times every admitted one (config.schedule_ab) */
#define SB_SCHEDULE_SERIAL_EAGER       1
#define SB_SCHEDULE_SERIAL_GRAPH       2   /* the tick, exchange included, captured in a hipGraph */
#define SB_SCHEDULE_OVERLAP_EAGER      3   /* exchange on a second stream beside the interior tiles */
#define SB_SCHEDULE_OVERLAP_GRAPH      4   /* refused (SB_ERR_UNSUPPORTED) on a HIP runtime it is known to fault on */

/* Mirrors the [SerializeField] block of csharp/Softbody.cs. Zero-initialise (or sb_desc_default), then set fields. */
typedef struct {
    int32_t device;          /* HIP device ordinal for this rank (LOCAL_RANK in a multi-process job) */
    int32_t rank;            /* this process' part of the spatial partition, 0..world-1 */
    int32_t world;           /* number of partitions == number of GPUs (1,2,4,8); 0 is read as 1 */
    int32_t part_dims[3];    /* SB_PARTITION_BLOCKS: blocks per axis, product == world; {0,0,0} = auto (2x2x2 for 8, ...) */
    float   gravity[3];
    float   damping;
    int32_t tile_particles;  /* target particles per LDS tile; 0 = automatic (512; 256 when the mesh has volume or
                                bending constraints); -1 = no tiling:
                                every constraint goes through the global-colour kernels */
    int32_t use_graph;       /* 1 = replay the substep loop as a hipGraph (default 1 via sb_desc_default) */
    int32_t partition;       /* SB_PARTITION_* */
    uint32_t plan_flags;     /* SB_PLAN_* */
    int32_t halo_transport;  /* SB_TRANSPORT_* */
    int32_t halo_schedule;   /* SB_SCHEDULE_* */
    uint32_t debug_flags;    /* SB_DEBUG_* of softbody_debug.h (test-only behaviour); a product host leaves this 0 */
    int32_t reserved[3];     /* must be 0 */
} sb_desc;

void sb_desc_default(sb_desc *d);

/* ---- lifecycle (Start / OnDestroy) ---------------------------------------------------------- */
int sb_create(const sb_desc *desc, sb_solver **out);
int sb_destroy(sb_solver *s);

/* ---- authoring (Start), all before sb_finalize ----------------------------------------------- */
int sb_set_particles(sb_solver *s, const float *pos_xyz, const float *vel_xyz /* may be NULL = 0 */,
                     const float *inv_mass, int32_t n);
/* Optional: rest pose used only to group particles into tiles (default: the sb_set_particles pose). */
int sb_set_rest_positions(sb_solver *s, const float *rest_xyz, int32_t n);
int sb_set_distance_constraints(sb_solver *s, const int32_t *ij, const float *rest_len, int32_t m, float compliance);
int sb_set_volume_constraints(sb_solver *s, const int32_t *ijkl, const float *rest_vol, int32_t m, float compliance);
/* rest_cs: 2 floats per hinge = (cos, sin) of the rest dihedral angle (SPEC.md §6). */
int sb_set_bending_constraints(sb_solver *s, const int32_t *ijkl, const float *rest_cs, int32_t m, float compliance);
/* Optional frictionless ground plane n.x >= d applied at the end of every substep (SPEC.md §2 step 2b);
 * n should be unit length. May be called before or after sb_finalize; takes effect at the next sb_step. */
int sb_set_ground_plane(sb_solver *s, float nx, float ny, float nz, float d, int32_t enabled);
/* Sharded authoring (world > 1, block partition): a rank may hand over only ITS WINDOW of the mesh -- the particles whose rest
 * position lies in the box sb_domain_window returns for it (its block of cells plus two cells of margin), the constraints among
 * them, in the order of the whole mesh -- instead of the whole mesh on every rank. sb_domain is the frame all ranks agree on:
 * every window is cut from the one grid made from it, so the ranks' tiles, ghost lists and shared-tile programs fit together
 * exactly as when every rank plans the whole mesh -- where that plan is LATTICE-TYPE (two tilings, no leftover layer, no global colour:
 * lattices at the usual tile sizes): colours and leftover layers are not local to a window (sb_finalize verifies pair by pair what two ranks share, and that all ranks planned tick programs of the same
 * shape; a mismatch is SB_ERR_STATE on every rank). A whole-mesh host never needs this. */
typedef struct {
    int64_t n_global;                 /* particles of the whole mesh */
    double  lo[3], hi[3];             /* bounding box of the whole rest pose */
    double  spacing;                  /* mean rest length of the whole mesh's distance constraints */
    double  fill;                     /* fraction of the bounding box the mesh occupies, in (0, 1]; 1 for a lattice (0 is read as 1) */
    int32_t four_vertex_constraints;  /* the whole mesh has volume or bending constraints (automatic tile size: 256 instead of 512) */
    int32_t reserved;
} sb_domain;
/* Declare, before sb_finalize, that the arrays given to sb_set_particles / sb_set_*_constraints are this rank's window of a
 * larger mesh. global_id: the n particles' ids in the whole mesh, strictly ascending. Every later n (sb_get_positions,
 * sb_get_owner, ...) is the window's n, in the window's numbering. */
int sb_set_domain(sb_solver *s, const sb_domain *domain, const int32_t *global_id, int32_t n);
/* Plan (colour + tile + partition), upload, capture. After this the authoring calls are rejected. */
int sb_finalize(sb_solver *s);

/* ---- multi-GPU: one process per GPU; RCCL communicator over xGMI ------------------------------ */
/* Rank 0 calls sb_comm_unique_id, the host broadcasts the 128 bytes (any channel it likes), every
 * rank passes them to sb_comm_init before sb_finalize. world == 1 needs neither call. */
#define SB_UNIQUE_ID_BYTES 128
int sb_comm_unique_id(uint8_t out_id[SB_UNIQUE_ID_BYTES]);
int sb_comm_init(sb_solver *s, const uint8_t id[SB_UNIQUE_ID_BYTES]);

/* Opt-in peer-store halo transport (sb_desc.halo_transport = SB_TRANSPORT_PEER): instead of pack -> ncclSend /
 * ncclRecv -> unpack, a rank stores its neighbours' ghosts straight into their mailboxes (one device allocation per rank,
 * mapped by IPC handle, or by plain pointer inside one process) and raises a flag there; the neighbour's unpack kernel
 * waits for the flags, copies and acknowledges. With an RCCL communicator present sb_finalize exchanges the handles by
 * itself (one ncclAllGather at setup). Without one the host connects the mailboxes after sb_finalize: every rank
 * exports its handle, every rank connects each neighbour's (ranks it shares no halo with may be skipped). A rank that
 * lives in the SAME process is connected by passing its solver (handle may then be NULL): hipIpcOpenMemHandle refuses
 * handles of the opening process. Unverified between two devices (1-GPU box); RCCL stays the default.
 *  - Waits are bounded (about 2^24 polls, tens of seconds): a flag that never arrives sets an error word instead of hanging
 *    the GPU; sb_step (before it enqueues), sb_readback_end, sb_synchronize and sb_get_* report it as SB_ERR_RCCL.
 *  - One process driving SEVERAL ranks on ONE device must give every rank a hardware queue of its own (environment
 *    GPU_MAX_HW_QUEUES >= ranks, set before the first HIP call): a waiting kernel blocks the streams that share its queue.
 *  - sb_destroy of a peer-transport solver frees the mailbox its neighbours store into: every neighbour must be quiescent
 *    first (sb_synchronize on every rank, then a host barrier, then sb_destroy). */
#define SB_IPC_HANDLE_BYTES 64
int sb_peer_mailbox_handle(sb_solver *s, uint8_t out_handle[SB_IPC_HANDLE_BYTES]);
int sb_peer_connect(sb_solver *s, int32_t rank, const uint8_t handle[SB_IPC_HANDLE_BYTES], sb_solver *same_process_peer /* or NULL */);

/* ---- the hot path (FixedUpdate) -------------------------------------------------------------- */
/* One tick of `substeps` substeps (SPEC.md §2). Asynchronous: work is enqueued on the solver's stream. The last
 * kernel of a tick may be held back and fused with the next tick when dt/substeps/plane are unchanged; every call
 * that reads or writes state (sb_get_velocities, sb_set_state, sb_synchronize, ...) completes it first, so the laziness is
 * not observable. Position reads (sb_get_positions, sb_readback_begin; any rank) do not even need that: they
 * PEEK -- the held-back kernel's constraint rounds and collision run on the tiles in question into a side array, bit for bit
 * what the completed tick would hold, and the tick stays fusable with the next one (sb_stats.readback_peeks). Only where that pays:
 * tilings of at least 2 048 workgroups, whose launches are bandwidth-bound; smaller ones complete the tick as before. */
int sb_step(sb_solver *s, float dt, int32_t substeps);

/* ---- readback / state round trip ------------------------------------------------------------- */
/* Arrays are full size n in the caller's particle numbering. With world > 1 only entries of particles
 * this rank owns are written; the rest are left untouched (see sb_get_owner). */
int sb_get_positions(sb_solver *s, float *pos_xyz_out, int32_t n);
int sb_get_velocities(sb_solver *s, float *vel_xyz_out, int32_t n);
int sb_set_state(sb_solver *s, const float *pos_xyz, const float *vel_xyz, int32_t n); /* after finalize */
/* Kinematic particles (SPEC.md 2; attachments to animated objects): between two ticks, move particles whose inverse mass is 0 to new
 * positions -- count entries, ids in the caller's numbering (each at most once), pos_xyz 3 floats per entry. The next tick's constraints
 * pull their neighbours along. An id whose inverse mass is not 0 is refused (SB_ERR_INVALID_ARG, nothing is changed). Asynchronous like
 * sb_step (the targets are copied before the call returns). The targets are pending until the next tick starts: when that tick's first
 * kernel also finishes the previous tick (the lazy tick boundary of sb_step) they are applied inside it, so a host that moves its pins
 * every tick keeps the fusion (sb_stats.ticks_fused_kinematic); position reads in between already show them. A rank of a partitioned solver
 * (world > 1) is given the same list in its own numbering (the whole list on every rank is fine): every entry is validated, the rank applies
 * the particles it OWNS and skips the others -- their owners apply them, and the ghost copies arrive with the next exchange. */
int sb_set_kinematic_positions(sb_solver *s, const int32_t *ids, const float *pos_xyz, int32_t count);
/* Asynchronous render readback: sb_readback_begin snapshots the positions as of every sb_step issued so far
 * (a small kernel on the compute stream) and starts a D2H copy into plugin-owned pinned memory on a second
 * stream; the next sb_step overlaps with that copy. sb_readback_end waits for the OLDEST pending snapshot and
 * returns a pointer to n*3 floats in caller numbering (entries of particles another rank owns stay 0), valid
 * until the second sb_readback_begin after it (the plugin keeps three snapshot buffers for at most two pending
 * snapshots, so the buffer handed out last is never the next one filled). sb_set_render_triangles re-allocates the
 * render-set buffers and invalidates pointers returned earlier.
 * Cost between two ticks: the tick's held-back last kernel is not forced out; with render_set_only the peek
 * runs only the T0 tiles that hold a render particle (256^3 cube: 5 768 of 32 768 workgroups). */
int sb_readback_begin(sb_solver *s);
int sb_readback_end(sb_solver *s, const float **pos_xyz_out);
/* Render normals (SPEC.md 6a; replaces Unity's Mesh.RecalculateNormals on the main thread): give the render
 * triangles once (particle indices, caller numbering, 3*m ints; m = 0 switches it off), any time after
 * sb_set_particles. Every later sb_readback_begin then also computes area-weighted vertex normals of the snapshot on
 * the copy stream and brings them to pinned memory; after the matching sb_readback_end, sb_readback_get_normals
 * returns n*3 floats (zero for particles in no triangle), valid as long as that snapshot's positions.
 * A rank of a partitioned solver (world > 1) takes the triangles too, but serves POSITIONS only: its render set is the render particles it
 * owns. Vertex normals need the neighbours' particles at the tick's end, which a rank does not hold: sb_readback_get_normals then returns
 * SB_ERR_UNSUPPORTED, and the normals of a partitioned body are computed on the gathered snapshot (sb_group_readback_get_normals). */
int sb_set_render_triangles(sb_solver *s, const int32_t *tri_abc, int32_t m);
int sb_readback_get_normals(sb_solver *s, const float **normal_xyz_out);
/* Render-set readback: a volumetric body renders only its surface. With render_set_only != 0 (and render triangles
 * set) a readback brings just the particles the triangles use -- their ids ascending -- instead of all n: the
 * pointers of sb_readback_end and sb_readback_get_normals then address count*3 floats, entry k belonging to particle
 * ids[k] (sb_readback_get_render_set). 256^3 cube: 390 k surface particles = 9 MB per snapshot instead of 201 MB.
 * world > 1: the render particles THIS RANK owns, ids ascending in its numbering. Not while a readback is pending. */
int sb_set_readback_render_set_only(sb_solver *s, int32_t render_set_only);
int sb_readback_get_render_set(sb_solver *s, const int32_t **ids_out, int32_t *count_out);
int sb_get_owner(sb_solver *s, int32_t *owner_rank_out, int32_t n);


/* ---- synchronisation, statistics ----------------------------------------------------------------- */
int sb_synchronize(sb_solver *s);
typedef struct {
    int64_t n_particles_owned, n_particles_local;   /* local = owned + ghost */
    int64_t n_constraints_local[3];                 /* distance, volume, bending (incl. redundant cut copies) */
    int32_t n_tilings;                              /* 2 with tiling, 1 without (tile_particles = -1) */
    int32_t n_global_colours;
    int64_t n_tiles[2];                             /* workgroups this rank launches per tiling (under-full tiles share one) */
    int64_t tile_constraints[2];                    /* constraints stored in the tile streams per tiling, this rank */
    int64_t constraints_in_tiles, constraints_in_global;   /* whole mesh */
    int64_t halo_particles_t1;                      /* ghosts sent before every T1 kernel */
    int64_t halo_particles_global;                  /* ghosts sent per substep for the global colours */
    int64_t device_bytes;                           /* device memory held by the solver */
    int64_t n_t2_layers;                            /* third-tiling layers (constraints inside neither T0 nor T1 that got LDS tiles) */
    int64_t n_t2_tiles, t2_constraints;             /* workgroups per substep / constraints of all T2 layers, this rank */
    /* Compulsory HBM bytes of ONE launch (every array the kernel touches counted once: particle state read and written,
     * the tiles' constraint streams, descriptors and particle lists), from the tables actually uploaded -- the model the
     * PMC-measured traffic in profiles/ is checked against (bench.py). Indexed like the slots of sb_step_profiled with
     * G = 0: 0 / 1 = mid-tick kernel on T0 / T1, 2 = first kernel of a tick, 3 = last kernel of a tick (on T0; on T1 it
     * moves launch_bytes[1] - (launch_bytes[0] - launch_bytes[3])), 4 = all T2 layers of one substep. */
    int64_t launch_bytes[5];
    /* partition (world > 1): how the ownership was made and what it balanced */
    int32_t partition;                              /* SB_PARTITION_BLOCKS or SB_PARTITION_RCB (what AUTO resolved to) */
    int32_t halo_peers;                             /* ranks this rank exchanges ghosts with (any slot) */
    int64_t partition_cost;                         /* cost units of the particles this rank owns (12 per particle + the vertex
                                                       shares of their constraints: 12 / 24 / 48 per spring / tet / hinge) */
    int64_t partition_cost_max, partition_cost_total;   /* over all ranks: max / mean = partition_cost_max * world / total */
    int64_t halo_particles_recv;                    /* ghosts received per refresh, all slots */
    uint64_t plan_hash;                             /* hash of the published orders, ownership and plan options: equal on every rank */
    int32_t halo_schedule;                          /* SB_SCHEDULE_* in force (what AUTO resolved to) */
    int32_t halo_unpack_fused;                      /* 1 = the T1 kernels read their ghosts straight from the receive buffer (no unpack launch) */
    /* position reads served by a peek so far (see sb_readback_begin), and the T0 workgroups one render-set peek launches
     * (-1 = no render-set peek has been set up) */
    int64_t readback_peeks;
    int64_t readback_peek_tiles;
    int64_t ticks_fused;                            /* sb_step calls whose first kernel also finished the tick before (lazy tick boundary kept) */
    int64_t ticks_fused_kinematic;                  /* ... of which that kernel also applied kinematic targets (sb_set_kinematic_positions) */
    int64_t lane_packed_tiles[2];                   /* workgroups per tiling whose spring slots are lane-packed (16 B per lane of 128, 8 B per lane of 256, instead of 4 B per slot) */
    /* SB_SCHEDULE_AUTO's calibration: 0 = none (not applicable or switched off), 1 = still measuring, 2 = decided; the ticks timed and the
     * per-tick times of the slowest rank, [0] serialised, [1] overlapped, in ms (0 until decided) */
    int32_t halo_auto_state, halo_auto_ticks;
    double halo_auto_ms[2];
} sb_stats;
int sb_get_stats(sb_solver *s, sb_stats *out);

/* ---- what the plugin is running on -------------------------------------------------------------------- */
/* The plugin links the HIP runtime and loads RCCL on first use (sb_comm_unique_id / sb_comm_init / this call) with dlopen: the
 * librccl.so.1 ALREADY IN THE PROCESS when there is one (a host that imported PyTorch first brings PyTorch's own RCCL and HIP
 * runtime; two ROCm stacks in one process is what must never happen), else the system's (the plugin's RUNPATH, /opt/rocm/lib).
 * world == 1 hosts never load RCCL. The schedules a world > 1 solver admits depend on what was bound:
 *   capture_overlap_ok = 0 on HIP runtimes older than 7.2: hipStreamEndCapture recurses without bound when a captured stream that
 *   was itself forked (the exchange stream) is forked again by RCCL's internal stream (profiles/r03a_*). */
typedef struct {
    int32_t hip_runtime_version;       /* hipRuntimeGetVersion, e.g. 70226015 */
    int32_t hip_driver_version;
    int32_t rccl_version;              /* ncclGetVersion, e.g. 22707; 0 = RCCL could not be loaded */
    int32_t rccl_header_version;       /* NCCL_VERSION_CODE of the rccl.h the plugin was compiled against */
    int32_t rccl_was_resident;         /* 1 = the process had an RCCL loaded already (it was used), 0 = the plugin loaded the system's */
    int32_t capture_serial_ok;         /* SB_SCHEDULE_SERIAL_GRAPH admitted */
    int32_t capture_overlap_ok;        /* SB_SCHEDULE_OVERLAP_GRAPH admitted */
    int32_t reserved;
    char hip_library[256];             /* file the HIP runtime symbols resolve to */
    char rccl_library[256];            /* file the RCCL symbols resolve to ("" = not loaded) */
} sb_runtime_info_t;
int sb_runtime_info(sb_runtime_info_t *out);

const char *sb_last_error(void);
int sb_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SOFTBODY_MI355X_H */

