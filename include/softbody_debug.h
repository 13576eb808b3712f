/*
 * softbody_debug.h — what tests, profilers and A/B measurements use of libsoftbody_mi355x.so and a Unity maintainer does not:
 * test modes of sb_desc.debug_flags, launch-by-launch hooks (the host as the wire of a partitioned solver on a one-GPU box), the
 * table validator ("race detector"), HIP-event timing on the solver's own stream, and the tuning switches of the
 * A/B measurements in profiles/ (DESIGN.md §6). Everything here is exported by the product library, none of it is needed to run it.
 *
 * Reference interface replaced: NONE EXISTS (/root/reference/README.md:1 is the whole reference tree); [BUILDER-DEFINED].
 */
#ifndef SOFTBODY_MI355X_DEBUG_H
#define SOFTBODY_MI355X_DEBUG_H

#include "softbody.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sb_desc.debug_flags: test-only behaviour; a product host leaves the field 0. */
#define SB_DEBUG_NO_COMM  1u           /* world > 1 without any transport: the host carries the halo through sb_debug_* */
#define SB_DEBUG_LOOPBACK 2u           /* every peer is this rank itself (size-1 communicator): one-GPU pipeline tests */

/* ---- tuning (A/B measurements only; every setting gives the same bits) ---------------------------------------------------------------
 * Kernel selection and table layout switches that used to be environment variables of the plugin (round 3: 23 getenv calls). The
 * plugin now reads NO environment variable for them: a measurement harness fills sb_tuning (tools/ and tests/ translate the old SB_*
 * variable names in softbodyunity_amd/native.py tuning_from_env) and hands it over between sb_create and sb_finalize. A product host
 * never calls this. None of the switches changes the plan (the published order) or the results. */
#define SB_TUNE_NO_MASS_PALETTE   (1u << 0)   /* float inverse masses instead of 1-byte palette indices */
#define SB_TUNE_NO_UNIFORM_MASS   (1u << 1)   /* read the per-particle mass index even when every mass is equal */
#define SB_TUNE_NO_PALETTE        (1u << 2)   /* 8-byte {i|j<<16, rest} slots instead of dictionary-coded 4-byte slots */
#define SB_TUNE_NO_WAVE_ITEMS     (1u << 3)   /* decode the group words in the kernel instead of reading host-dealt wave items */
#define SB_TUNE_NO_LANE_PACK      (1u << 4)   /* 4 bytes per dictionary-coded slot everywhere (no 16-byte lane-packed words) */
#define SB_TUNE_NO_COST_ORDER     (1u << 5)   /* keep the plan's tile order inside small launches */
#define SB_TUNE_NO_FUSED_UNPACK   (1u << 6)   /* world > 1: keep the unpack kernel behind the exchange */
#define SB_TUNE_PEER_COARSE       (1u << 7)   /* peer transport: an ordinary cached mailbox (ONE-device timing experiments only) */
#define SB_TUNE_NO_LAZY_TICK      (1u << 8)   /* never defer the last kernel of a tick */
#define SB_TUNE_NO_PACK           (1u << 9)   /* one workgroup per plan tile (no tile packing) */
#define SB_TUNE_NO_PEEK           (1u << 10)  /* position reads complete the tick instead of peeking */
#define SB_TUNE_NO_KIN_FUSE       (1u << 11)  /* pending kinematic targets always complete the previous tick first */
#define SB_TUNE_NO_WIDE_SLOTS     (1u << 12)  /* 256-lane launches stage their slots through LDS (no 8-byte packed words) */
#define SB_TUNE_AUTO_PREFER_OVERLAP (1u << 13) /* SB_SCHEDULE_AUTO's calibration decides for the overlapped schedule whatever it measured (tests) */
#define SB_TUNE_NO_AUTO_CALIBRATION (1u << 14) /* SB_SCHEDULE_AUTO = SB_SCHEDULE_SERIAL_EAGER without measuring */
typedef struct {
    uint32_t flags;                    /* SB_TUNE_* */
    int32_t tile_lanes;                /* 0 = by launch size; 128 | 256 | 512 forces the workgroup width of small spring tiles */
    int32_t quad_lanes;                /* 0 = default (512); 256: tiles with tets / hinges as 4-wave workgroups */
    int32_t narrow_min_tiles;          /* 0 = default (10 240): launches of at least this many tiles run 128-lane workgroups */
    int32_t store_through_max_tiles;   /* -1 = default (6 144): launches of at most this many tiles store their state through the L2; 0 = never */
    int32_t store_through_large;       /* the same for larger launches: bit 0 previous positions, bit 1 positions (default 0) */
    int32_t peek_min_tiles;            /* -1 = default (2 048): position reads peek from this many T0 workgroups on */
    int32_t lds_pad_bytes;             /* unused LDS per workgroup (occupancy experiments) */
    int32_t win_dwords;                /* 0 = default: shrink the LDS constraint window */
    int32_t prev_offset_bytes;         /* placement experiment (profiles/r04_placement_modes.txt): the previous-position array starts this many bytes
                                          into its allocation, i.e. the RELATIVE offset of the two arrays every launch streams side by side moves
                                          (multiple of 16; default 0) */
    int32_t reserved[6];               /* must be 0 */
} sb_tuning;
void sb_tuning_default(sb_tuning *t);
int sb_set_tuning(sb_solver *s, const sb_tuning *t);      /* after sb_create, before sb_finalize */

/* ---- measurement --------------------------------------------------------------------------------------------------------------------- */
/* HIP-event timing on the solver's own stream (torch.cuda.Event cannot see it). */
int sb_profile_begin(sb_solver *s);
int sb_profile_end(sb_solver *s, float *elapsed_ms_out);
/* One tick launched eagerly with a HIP-event pair around every kernel launch on the solver's stream.
 * Slots: 0 / 1 = mid-tick tile kernels on tiling T0 / T1 (rounds + velocity/integrate + rounds),
 * 2+c = global colour c, 2+G = the first kernel of the tick, 3+G = the last, 4+G = the kernels of the T2 layers
 * (G = n_global_colours). n_slots must be 5 + n_global_colours (sb_get_stats). Same results as sb_step. */
int sb_step_profiled(sb_solver *s, float dt, int32_t substeps, float *slot_ms_out, int32_t *slot_launches_out,
                     int32_t n_slots);
/* Ghost-exchange timing of a world > 1 solver: while enabled, every exchange of sb_step (eager schedules) is bracketed by HIP events --
 * pack kernel, transport (grouped send/recv, or the peer transport's push + unpack), and the whole exchange as the compute stream sees it.
 * sb_debug_exchange_timing_read waits for the stream and returns the sums since the last read. Costs a few events per exchange: leave it
 * off in timed regions whose figure is quoted. */
typedef struct {
    int64_t exchanges;                 /* exchanges timed since the last read */
    double pack_ms, transport_ms, total_ms;      /* sums over those exchanges */
    double exposed_wait_ms;            /* what the compute stream waited: = total_ms for a serialised exchange; for an overlapped one the time its
                                          wait for the exchange stream lasted (what the interior tiles did not hide; an upper bound: the events
                                          around the wait also see the tail of the launch before it) */
} sb_exchange_timing;
int sb_debug_exchange_timing(sb_solver *s, int32_t enabled);
int sb_debug_exchange_timing_read(sb_solver *s, sb_exchange_timing *out);

/* ---- launch-by-launch hooks --------------------------------------------------------------------------------------------------------- */
/* Test hooks (used by tests/test_gpu_multirank.py to check the multi-rank device path on a box with one
 * GPU, where RCCL cannot form a communicator): run ONE launch of a tick — tile kernel K_it (gcolour = -1),
 * global colour `gcolour` of the substep that K_it started, or T2 layer l of that substep (gcolour = -2 - l) — without any ghost exchange, and move one
 * halo slot's send / receive buffer through host memory. Buffer layout = what goes over the wire: peers in
 * increasing rank order, each peer's particles back to back, 3 floats (position) per particle, slot 1: 6 floats
 * (position, previous position). */
int sb_debug_launch(sb_solver *s, float dt, int32_t substeps, int32_t it, int32_t gcolour);
int sb_debug_halo_pack(sb_solver *s, int32_t slot, float *host_out, int64_t capacity_floats, int64_t *count_floats_out);
int sb_debug_halo_unpack(sb_solver *s, int32_t slot, const float *host_in, int64_t count_floats);

/* ---- table validator ------------------------------------------------------------------------------------------------------------------ */
/* The "race detector" of this design: the only race the tile kernels can have is two constraints of one
 * group -- or two tiles of one launch -- touching the same particle. A GPU kernel re-reads everything the tile kernels read (the
 * uploaded descriptors, run tables / particle lists, group words, 4- and 8-byte spring slots, four-vertex slots, wave items: after
 * packing, lane dealing and cost ordering) with their own decoding rules and counts violations; the global colours likewise.
 * inject_fault != 0 runs the same check on a COPY of tiling T0's tables with one fault planted (1: a slot copied over its
 * neighbour = a particle twice in one group; 2: a descriptor copied over its neighbour = a particle staged by two tiles), so a host
 * can see the detector detect. The solver's own tables are never modified. Returns SB_OK when the check RAN; look at errors[]. */
typedef struct {
    int64_t tiles_checked, groups_checked, constraints_checked;
    /* 0 index out of range, 1 a particle twice in one group (or one global colour), 2 a particle staged by two tiles of one launch,
     * 3 a group's data leaves the tile's stream, 4 malformed run table / particle list, 5 wave items disagree with the group words */
    int64_t errors[6];
    int32_t first_stage;                            /* -1 none; 0 / 1 = tiling T0 / T1, 2 = a T2 layer, 3 = a global colour */
    int32_t first_tile, first_group, first_kind;    /* device tile index of that tiling (global colour: -2 - its number), group (colour: constraint), errors[] index */
} sb_validate_report;
int sb_debug_validate(sb_solver *s, int32_t inject_fault, sb_validate_report *out);

/* ---- last words ----------------------------------------------------------------------------------------------------------------------- */
/* A measurement harness that must leave ONE result line whatever happens (bench.py --gpus N: the one run a multi-GPU node ever makes)
 * registers the line as it stands; should the process then die of a fatal signal (SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL: a GPU
 * fault ends in abort()) or be told to stop (SIGTERM: a launcher tearing the job down after another rank died), the handler writes the
 * registered bytes to fd and _exit()s with `exit_code`. text == NULL or len == 0 removes the registration (handlers restored). The text is
 * copied. Async-signal-safe in the handler (write + _exit only). */
int sb_debug_last_words(int32_t fd, const char *text, int64_t len, int32_t exit_code);

#ifdef __cplusplus
}
#endif
#endif /* SOFTBODY_MI355X_DEBUG_H */
