// kernel_types.hpp — plain structs and constants the host units and the gfx950 kernels share (no device code)
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbk {

struct TickParams {           // SPEC.md §2 host-side scalars, uploaded once per (dt, S)
    float h, inv_h, hgx, hgy, hgz, kd, at_d, at_v, at_b;
    float pnx, pny, pnz, pd;  // ground plane n.x >= d (SPEC.md §2 step 2b)
    int32_t plane_on;
    float pad[2];
};

// One LDS tile (or pack of tiles) = one workgroup. 128 B, read with scalar loads.
// The tile's constraint stream lives at stream[s_begin ...], 16-byte aligned, in dwords:
//   [round words, padded to 4] [rest-length palette, padded to 4] [rounds' data]
// group word: bits 0-9 distance, 10-19 volume, 20-29 bending constraint count (each <= 256), bit 30 = the distance slots
// are dictionary-coded. A group's constraints share no particle (one barrier per group). Its data: the distance slots,
// count x {i | j<<16, rest length}, or -- when a tile's distance constraints use at most 256 distinct rest lengths
// (regular meshes) -- count x {i | j<<12 | palette index<<24} (one dword each) with the values in the tile's palette,
// padded to 4 dwords; then the volume slots, then the bending slots, each {i0|i1<<16, i2|i3<<16, rest.x, rest.y}.
struct TileDesc {
    int32_t n_local, run_count, n_rounds;
    int32_t gather_begin;      // KIND 3 (T2 tiles): the tile's particles are gather[gather_begin .. +n_local) instead of runs
    uint32_t s_begin;          // dword offset of the tile's stream
    uint32_t s_hdr;            // dwords of round words + palette (each padded to 4): round data starts at s_begin + s_hdr
    uint32_t s_len;            // total dwords (multiple of 4)
    uint32_t packed_lanes;     // 0, or 128 / 256: LANE-PACKED slots (see kLanePack* / kWidePack*) for workgroups of that many lanes -- the launch must use that width
    int32_t run_overflow;      // runs beyond kInlineRuns live at runs_overflow[run_overflow ...]
    int32_t n_pal;             // palette entries (0 = no dictionary coding), stored after the round words
    int32_t n_steps;           // wave items per wave (meshes with tets / hinges; 0 = none), see kItem* below
    uint32_t s_items;          // dword offset (from s_begin) of the items: kItemWaves runs of n_steps dwords, inside the header
    int2 runs[10];             // {first particle (device numbering), first tile-local index}; unused entries: {0, INT_MAX}
};
constexpr int kInlineRuns = 10;
// Lane-packed slots (round 3, second session). A register-resident spring tile run by 128-lane workgroups gives every lane two slots in
// each of its (at most three) rounds and keeps them in registers for both passes; the slots need 9 + 9 bits of tile-local indices and a
// palette index. Stored as ONE 16-byte word per lane -- six 21-bit fields {i:9 | j:9 | palette:3}, field 2 r + u = the lane's slot u of
// round r (constraint lane + 128 u of that round; beyond the round's count the field is 0) -- the tile's data is 2 KiB instead of 3 KiB
// (4 bytes per slot), it arrives in the lane's first window load and never touches LDS. build_device decides per tile (solver.hip).
// A tile whose slots are NOT dictionary-coded (per-spring rest lengths) packs the same way with its rest values behind the index words:
// [128 x 16 B index word][128 x 16 B: rest of fields 0..3][128 x 8 B: rest of fields 4, 5] = 40 bytes per lane instead of 48
// (n_pal == 0 marks this form; only in the kernels that read inverse masses as floats, WPAL = false -- the host packs accordingly).
constexpr int kLanePackLanes = 128, kLanePackRounds = 3, kLanePackFieldBits = 21, kLanePackMaxPalette = 8;
constexpr uint32_t kLanePackDwordsCompact = 4 * kLanePackLanes, kLanePackDwordsFull = 10 * kLanePackLanes;
// The same for 256-lane workgroups (round 4): a lane has ONE slot per round, so three 21-bit fields = ONE 8-byte word per lane, field r = the
// lane's slot of round r (constraint `lane` of that round) -- again 2 KiB per 512-particle tile instead of 3 KiB, loaded by the lane itself
// with the first batch, never staged in LDS. Dictionary-coded tiles only. For the launches between the 512-lane and the 128-lane regimes:
// mid-size meshes (128^3: 4 096 tiles) and the ranks of a partitioned solver (256^3 on 8 ranks: 4 096 tiles each).
constexpr int kWidePackLanes = 256;
constexpr uint32_t kWidePackDwords = 2 * kWidePackLanes;
constexpr int kWide8MaxTiles = 768;     // launches of at most this many spring-only small tiles run 512-lane workgroups (schedule.hip launch_tile)
// Register-resident programs (tile_kernel): a tile of at most this many distance rounds keeps its slots and rest lengths in registers.
// Build switches of A/B timing variants (make EXTRA=-D...): the HOST reads the same constants when it decides which tiles to lane-pack
// (tables.hip) -- a lane-packed tile is never staged in LDS, so only the register-resident path can decode it.
#ifndef SB_REG_ROUNDS
#define SB_REG_ROUNDS 4          // 256-lane workgroups (one constraint per lane and round)
#endif
#ifndef SB_REG_ROUNDS_NARROW
#define SB_REG_ROUNDS_NARROW 3   // 128-lane workgroups (two per lane): the register budget of 6 waves per SIMD allows 3
#endif
constexpr int kRegRoundsWide = SB_REG_ROUNDS, kRegRoundsNarrow = SB_REG_ROUNDS_NARROW;
#ifdef SB_REG_COMPACT_ONLY      // A/B timing builds only: the round-2 condition (dictionary-coded tiles only)
constexpr bool kRegFullSlots = false;
#else
constexpr bool kRegFullSlots = true;
#endif
constexpr bool kLanePackDecodable = kRegRoundsNarrow >= kLanePackRounds;       // else the host emits no lane-packed tile at all
// Wave items (meshes with tets / hinges, 4-wave tiles): the host deals every group's work to the four waves ahead of time --
// hinges first, then tets (16 four-lane constraints per wave), then springs (64 per wave), boustrophedon over the rows of
// a group -- and stores for every wave one dword per STEP (= one row of one group): what to project, how many, where the
// slots lie in the tile's data, and whether the group ends here (workgroup barrier). The kernel's loop over a list is then a
// v_readlane, three bit-field extracts and one uniform branch per step instead of decoding the group word, the window test
// and the slot arithmetic (90 scalar instructions and a dozen branches per group on the critical path of every group).
constexpr int kItemWaves = 4;        // dealt for 4-wave tiles (8 with SB_QUAD_LANES=512)
constexpr uint32_t kItemIdle = 0, kItemDistCompact = 1, kItemDistFull = 2, kItemVolume = 3, kItemBending = 4;
constexpr int kItemCountShift = 3, kItemBarrierBit = 10, kItemOffsetShift = 11;     // type:3 | count:7 | barrier:1 | dword offset:21
constexpr int kMaxRoundsLds = 128;   // round words cached in LDS; longer programs read them from memory
constexpr int kMaxPalette = 256;     // rest-length dictionary entries per tile

// Positions in HBM: packed xyz (12 B) + the static inverse mass in a side array (never rewritten).
typedef float f32x3_t __attribute__((ext_vector_type(3)));
struct PosView {
    float *xyz;        // 3 floats per local particle
    const float *w;    // inverse mass per local particle
};

constexpr int kMaxMassPalette = 64;   // distinct inverse masses that fit the one-byte-per-particle coding (one per lane)

struct TileArgs {
    PosView pos;              // packed xyz + inverse mass per local particle
    const uint8_t *w8;        // WPAL kernels: index of the particle's inverse mass in wpal (1 B instead of 4 B per read)
    const float *wpal;        // kMaxMassPalette entries
    float *prev;              // packed xyz
    float *vel;               // packed xyz (read by KIND 0, written by KIND 2)
    const TileDesc *tiles;
    const int2 *runs_overflow;
    const uint32_t *stream;
    const TickParams *tp;
    const int32_t *gather;    // KIND 3: particle lists of the sparse T2 tiles (device numbering)
    int32_t max_local;        // LDS carve: [max_local float4][rounds_dwords][pal_dwords][win_dwords][16 spare bytes]
    int32_t rounds_dwords;    // round words cached in LDS (multiple of 4, <= kMaxRoundsLds); longer programs read memory
    int32_t pal_dwords;       // largest rest-length dictionary of the tiling, padded to 4 (0 = none)
    int32_t win_dwords;       // constraint window held in LDS (multiple of 4, >= the largest round)
    int32_t tile_base;        // this launch covers tiles tile_base + blockIdx.x (boundary / interior split)
    int32_t w_uniform;        // WPAL kernels: every particle has the same inverse mass -> all lanes read index 0 (one cache line, no per-particle byte)
    int32_t item_waves;       // waves per tile the wave items of this tiling were dealt for (0 = none)
    int32_t store_through;    // bit 0: previous positions, bit 1: positions are stored through the L2 (small launches, see store3_through)
    // GHOSTS kernels (world > 1, T1 launches of a lattice-type plan): particle g >= n_owned is ghost g - n_owned and its position
    // and previous position are read straight from the receive buffer of the exchange that just ended (6 floats per ghost, in
    // ghost order) instead of from the arrays -- the unpack kernel between the exchange and this launch is gone
    const float *ghost_src;
    int32_t n_owned;
    // KIND 4 (peek): where the tick-end positions of the launch's tiles go (packed xyz, device numbering); the state arrays stay as
    // they are
    float *peek_out;
    // KIND 5 (kinematic targets inside the fused tick boundary): kin_map[g] = slot of pinned particle g (-1: free particle),
    // kin_target[3 slot ..] = its pending target or NaN; the lane that applies a target writes NaN back
    const int32_t *kin_map;
    float *kin_target;
};
// (A PACK variant -- T0 tiles writing the send buffer themselves, entries {tile-local index, send slot} per tile -- was built and
// measured in round 3: bit-exact, but 0.786 -> 0.861 ms per tick in the serialised W = 8 loopback schedule and no change in the
// overlapped one (profiles/r03g_loopback_w8_fused_pack_ab_not_kept.txt); removed.)
constexpr int kHaloNone = 0, kHaloGhosts = 1;

// Lanes per tile (template parameter THREADS of tile_kernel). A round holds up to kRoundSlots independent constraints,
// so a lane projects kRoundSlots / THREADS of them per round. Fewer waves per tile = more tiles resident per CU (the
// wave slots, not LDS, cap a 256-lane workgroup at 8 tiles per CU; 128 lanes reach the 12-14 that LDS allows): +4.4 %
// at 256^3. A launch whose tiles all fit on the chip at once is latency-bound instead and wants the wide workgroup
// (64^3: 256 lanes are 22 % faster), so the host picks per launch (solver.hip launch_tile).
constexpr int kNarrowTileThreads = 128;                     // small tiles, launches that oversubscribe the chip
constexpr int kQuadTileThreads = 512;                       // tiles with four-lane constraints (tets, hinges): 8 wave slots per group row
constexpr int kWideTileThreads = 256;                       // small tiles in latency-bound launches, and all large tiles
constexpr int kRoundSlots = 256;                            // plan.hpp kRoundThreads
constexpr int kSmallTile = 512, kLargeTile = 1024;          // the two particle capacities the kernels are built for
typedef float f32x3 __attribute__((ext_vector_type(3)));

typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vectors: one 16-byte load/store, no struct copies
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- peer-store halo transport (opt-in, SB_HALO_TRANSPORT=peer; solver.hip) ---------------------------------------
// Instead of pack -> ncclSend/ncclRecv -> unpack, the push kernel stores every peer's ghosts STRAIGHT into that peer's
// mailbox (one fine-grained device allocation per rank, mapped into the senders by IPC handle or, inside one process, by
// plain pointer) and raises a flag there; its last workgroup then waits until the flags of this rank's own senders have
// arrived, so the unpack kernel behind it in the stream can copy the ghosts into the arrays and acknowledge. A segment
// has two buffers used alternately (epoch & 1): exchange e writes the buffer exchange e-2 used, and a rank finishes
// exchange e only after its receivers acknowledged e-1, so no sender ever waits before writing. Epochs count the
// exchanges of a halo slot, live in device memory and are advanced by the kernels themselves: the launches sit in a
// captured hipGraph unchanged. Every wait is bounded: a flag that never arrives sets an error word instead of hanging.
constexpr int kMaxPeers = 8;
struct PeerSlot {
    int32_t n_send, n_recv;                 // peers this rank sends to / receives from on this halo slot
    int32_t send_off[kMaxPeers + 1];        // first ghost of every send peer in send_idx, order of the send peers
    int32_t send_cap[kMaxPeers];            // ghosts pushed to that peer (a loopback self-exchange may push fewer than it packs)
    float *remote_data[kMaxPeers];          // per send peer: where this rank's segment starts inside the peer's mailbox
    int32_t remote_stride[kMaxPeers];       // floats between the two buffers of that segment (exchanges alternate: epoch & 1)
    int32_t my_stride[kMaxPeers];           // per recv peer: the same for the segments in this rank's mailbox
    int32_t recv_off[kMaxPeers + 1];        // first ghost of every recv peer in recv_idx, order of the recv peers
    int32_t recv_cnt[kMaxPeers];            // ghosts per recv peer
    const float *my_data[kMaxPeers];        // per recv peer: its segment in this rank's mailbox (16-byte aligned)
    uint32_t *remote_data_flag[kMaxPeers];  // per send peer: the peer's "data from this rank arrived" word
    uint32_t *my_ack_flag[kMaxPeers];       // per send peer: local word the peer writes when it has consumed the segment
    uint32_t *my_data_flag[kMaxPeers];      // per recv peer: local word the peer writes when its data is in the mailbox
    uint32_t *remote_ack_flag[kMaxPeers];   // per recv peer: the peer's "consumed" word for this rank
    uint32_t *local;                        // this rank's own (ordinary, cached) words for the slot: [0] exchanges completed so far,
                                            // [1] push / [2] unpack workgroups finished
    uint32_t *error;                        // set to 1 when a wait gave up
};
constexpr int kPeerSpinLimit = 1 << 24;

// ---- table validator (debug entry sb_debug_validate; SURVEY.md 5 "race detection") ---------------------------------------------------
// The only race this design can have is two constraints of one group (or two tiles of one launch) touching the same particle. The
// planner's output is checked on the host (tests/test_plan.py); THIS kernel checks what the tile kernels actually read -- the uploaded
// descriptors, run tables / particle lists, group words, dictionary-coded or full slots, four-vertex slots and wave items, after
// packing, lane dealing and cost ordering -- with the tile kernels' own decoding rules. One workgroup per device tile.
struct ValidateCounters {
    unsigned long long tiles, groups, constraints;
    // 0 index out of range (particle or tile-local), 1 a particle twice in one group, 2 a particle staged by two tiles of one launch,
    // 3 the group walk leaves the tile's stream, 4 malformed run table / particle list, 5 wave items disagree with the group words
    unsigned int errors[6];
    int first[4];      // tile, group, kind of the first error seen (-1 = none), spare
};
constexpr int kValidateThreads = 256;

}  // namespace sbk
