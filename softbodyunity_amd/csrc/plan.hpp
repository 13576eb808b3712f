// plan.hpp — host-side planner: tiles, colours, partition, halo schedule (pure C++, no HIP).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the
// planner implements SPEC.md §3 and DESIGN.md §3 ("two tilings, static split"):
//
//   tiling T0 : cells of a uniform grid over the rest pose            (a partition of the particles)
//   tiling T1 : cells of the same grid shifted by about half a cell   (another partition)
//   S0 / S1   : a static split of the constraints: S0 is projected on the tiles of T0, S1 on the tiles of T1
//               (a constraint can only go where all its particles share a tile; constraints inside both
//               tilings are assigned so that colour classes stay whole and the two sides balance)
//   T2 layers : constraints inside neither tiling that share a cell of one of up to six further shifted grids: sparse
//               LDS tiles (explicit particle lists), one extra tile kernel per layer and substep
//   G         : what is left, greedy edge-coloured, one global kernel per colour
//
//   substep of parity p (0,1,0,1,... restarting at 0 every tick) projects, in this order,
//     S_p on T_p's tiles, then the T2 layers, then G, then S_(1-p) on T_(1-p)'s tiles   (tile by tile, round by round)
//
// S_q of one substep and S_q of the next are the same constraint list on the same tiles, so the GPU runs
// them in ONE kernel with the per-particle velocity update + integrate between them: every particle and
// every constraint word is read once per kernel. The flat sequential order equivalent to that execution is
// published per parity for the oracle.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>

namespace sbp {

// The frame a plan's grid is made from: bounding box of the rest pose, mean rest length, particle count. A whole-mesh plan
// measures it (compute_domain); the ranks of a SHARDED solver -- each plans only its window of the mesh -- are all given the
// same one, so that every window is cut from one grid (set = true).
struct Domain {
    bool set = false;
    int64_t n_global = 0;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double ell = 0;
    double fill = 1.0;          // fraction of the bounding box the mesh occupies (1 for anything regular: compute_domain)
};
struct Grid {
    int kk = 2;                 // cell edge in mean rest lengths (even)
    double cs = 0, org[3] = {0, 0, 0};
    int nc[3] = {1, 1, 1};
    int shift_units = 1;
    double shift_frac = 0.5, first_t2_frac = 0.75;
    float ext[3] = {0, 0, 0};
};
// cells of margin around a rank's block that its window must cover: the T1 tiles it runs reach less than one cell beyond the
// block, and the labels of the static split inside them depend on constraints less than one further cell away (plan.cpp)
constexpr int kWindowMarginCells = 2;

struct Opts {
    int rank = 0, world = 1;
    int dims[3] = {0, 0, 0};
    int tile_particles = 512;  // -1: no tiling
    int partition = 0;         // world > 1: 0 = automatic (block grid when that balances the ranks within 10 %, else RCB), 1 = block grid
                               // (dims), 2 = recursive coordinate bisection over whole T0 cells weighted by constraint cost
    bool third_tiling = true;       // constraints inside neither T0 nor T1 get LDS tiles of their own (T2) where they can (SB_NO_T2: A/B runs)
    bool third_list = true;         // irregular meshes: the first T2 layers take a balanced share of the constraints, not only the leftovers (SB_NO_THIRD_LIST: A/B runs)
    bool merge_tiles = true;        // irregular meshes: merge small tiles of a balanced list around a leftover constraint (SB_PLAN_NO_TILE_MERGE: A/B runs)
    int balanced_lists = 2;         // irregular meshes: how many T2 layers are balanced lists (grids of their own), 1 .. kMaxBalancedLists
    bool cluster_layers = true;     // once few constraints are left, T2 layers are made of connected components instead of grid cells (SB_NO_CLUSTER_LAYERS: A/B runs)
    bool mixed_groups = true;       // colour the constraint types of a tile together (SB_NO_MIXED_GROUPS: one type per group, A/B runs)
    bool bank_aware_lanes = true;   // order the constraints of a round for conflict-free LDS gathers (SB_NO_BANK_ORDER: A/B runs)
    Domain domain;                  // set: the input is this rank's window of a larger mesh (Input::global_id), block partition only
};

struct Run {            // a contiguous range of particles
    int32_t start;      // first particle (global-new numbering in Plan, local numbering in LocalPlan)
    int32_t len;
};

constexpr int kRoundThreads = 256;          // constraints of one type per group (round)
constexpr int kMaxTileLocal = 1024;         // particles staged per tile (4 per lane)
constexpr int kMaxTileRuns = 64;
constexpr int64_t kMergedTileCap = 512;     // particles (of the grid cells) a merged tile of a balanced list may hold: stays a small tile
constexpr int kMaxMergedCells = 8;           // ... and of how many original tiles (grid cells) it may be the union
constexpr int kMaxBalancedLists = 3;        // of the T2 layers, at most this many are balanced lists (plan.cpp static split)
constexpr int kMaxT2Layers = 6;             // shifted grids tried in turn for the constraints inside neither T0 nor T1
// partition cost units: a particle 12; a constraint 12 / 24 / 48 (distance / volume / bending), split evenly over its vertices
constexpr int64_t kCostParticle = 12;
constexpr int64_t kCostVertexShare[3] = {6, 6, 12};
constexpr int kLdsGroup = 8;                // lanes whose 16-byte LDS accesses are served together (measured: 8 beats 16, 32, 64)

struct Tile {
    int32_t owner;          // owning rank if all its particles have one owner, else -1
    int32_t run_begin, run_count;
    int32_t n_local;        // particles staged in LDS
    int32_t round_begin;    // into Tiling::rounds: the rounds of the tile's constraint list
    int32_t n_rounds;
    int64_t d_begin, q_begin;   // the tile's constraints in the tiling's arrays
    int64_t d_end, q_end;
    int64_t seq_begin, seq_end; // slice of the tiling's sequence
    int64_t order_begin[2], order_end[2];   // slices of the two published orders
    int64_t gather_begin = 0;               // T2 only: the tile's particles are Tiling::gather[gather_begin .. +n_local) (run_count == 0)
};

struct Tiling {
    std::vector<Tile> tiles;
    std::vector<Run> runs;
    std::vector<uint32_t> rounds;       // one word per group of a tile: distance | volume << 10 | bending << 20 constraint counts (each <= kRoundThreads)
    // tile constraints in execution order, tile-local particle indices (16 bit each)
    std::vector<uint32_t> t_dist;       // lo16 = i, hi16 = j
    std::vector<int32_t> t_dist_id;     // original constraint id (for rest values)
    std::vector<uint32_t> t_quad;       // 2 words per 4-vertex constraint
    std::vector<int32_t> t_quad_id;     // original id within its type
    std::vector<uint8_t> t_quad_type;   // 1 volume, 2 bending
    std::vector<int32_t> gather;        // T2 only: particle lists (global-new numbering), ascending inside a tile
    int32_t max_local = 0, max_runs = 0;
};

struct GColour {            // one global colour: constraints of one type that share no particle
    int type;
    std::vector<int32_t> ids;   // original ids, increasing
    bool cut = false;           // touches particles of more than one rank
};

struct Phase {              // one entry of a parity's phase list (inspection / oracle task parallelism)
    int kind;               // 0 global colour, 1 = S_p on T_p's tiles (first in the substep), 2 = S_(1-p) on T_(1-p)'s tiles (last),
                            // 3 = S2 on T2's tiles (after kind 1, before the global colours)
    int type;               // kind 0: constraint type, else -1
    int tiling;             // kind 1/2: which tiling's tiles; kind 0: -1
    int gcolour;            // kind 0: index into Plan::gcolours
    int64_t order_begin, order_end;
    int64_t task_begin, task_end;
    int halo_slot;          // -1 none; 1 = before the T1 kernel (x and xprev); 2+c = before global colour c (x);
                            // 2+|gcolours|+l = before the kernel of T2 layer l (x)
    int layer = -1;         // kind 3: which T2 layer
};

struct Plan {
    Opts opts;
    int32_t n = 0;
    int64_t m[3] = {0, 0, 0};
    int dims[3] = {1, 1, 1};
    Domain domain;              // the frame the grid was made from (measured, or imposed on a sharded plan)
    int partition = 1;          // what the ownership was made with: 1 = block grid (dims), 2 = RCB over T0 cells
    std::vector<int64_t> rank_cost;     // [world] cost units of the particles a rank owns (kCostParticle per particle + the vertex
                                        // shares of their constraints): what the partitioner balances
    bool tiling = true;
    // particle numbering
    std::vector<int32_t> new_of_old, old_of_new, owner_of_old;
    Tiling T[3];            // T[1] is empty when tiling is off; T[2] (sparse, single-owner tiles) only when constraints lie inside neither
    std::vector<std::pair<int32_t, int32_t>> t2_layers;     // tile ranges [begin, end) of T[2], one per layer, in execution order
    std::vector<GColour> gcolours;
    // published orders per substep parity (original constraint ids)
    std::vector<uint8_t> order_type[2];
    std::vector<int32_t> order_id[2];
    std::vector<Phase> phases[2];
    std::vector<int64_t> task_off[2];
    std::vector<int64_t> group_off[2];
    // stats
    int64_t cons_in_tiles = 0, cons_in_global = 0;
};

// What one rank uploads and executes.
struct LocalTiling {
    std::vector<int32_t> tile_ids;          // global tile ids, execution order
    std::vector<Run> runs;                  // local numbering, concatenated per tile
    std::vector<int32_t> run_begin;         // per local tile (+1)
    std::vector<int32_t> gather;            // T2: local particle indices, concatenated per tile
    std::vector<int32_t> gather_begin;      // T2: per local tile (+1)
};
struct LocalGColour {
    int type;
    std::vector<int32_t> idx;               // 2 or 4 local particle indices per constraint
    std::vector<int32_t> id;                // original id
};
struct HaloSlot {                           // per peer: local indices to send / to receive into
    std::vector<std::vector<int32_t>> send_idx, recv_idx;
};

struct LocalPlan {
    int rank = 0, world = 1;
    int64_t n_owned = 0;
    std::vector<int32_t> local_to_old;      // owned first (global-new order), then ghosts
    LocalTiling T[3];
    std::vector<LocalGColour> gcolours;
    std::vector<HaloSlot> halo;             // slot 0 unused, 1 = before T1 kernels, 2+c = before global colour c
    std::vector<uint8_t> order_mask[2];     // which order entries this rank executes
    // per peer: a hash of everything the two ranks must agree on -- the ghost lists between them (global particle ids, both
    // directions) and the programs of the tiles both execute (constraint sequence as global particle ids). Symmetric: rank a's
    // entry for b equals rank b's entry for a exactly when they planned consistently (checked at sb_finalize / peer_link).
    std::vector<uint64_t> pair_hash;
};

struct Input {
    const float *rest;
    int32_t n;
    const int32_t *dist_ij; int64_t m_d;
    const int32_t *vol; int64_t m_v;
    const int32_t *bend; int64_t m_b;
    const int32_t *global_id = nullptr;     // sharded plans: ids of the window's particles in the whole mesh, strictly ascending
};

// f(chunk, begin, end) for the chunks of [0, n) of `chunk_size` elements, on the planner's host threads (SB_PLAN_THREADS,
// default min(hardware threads, 16)). The chunking never depends on the thread count. Re-throws the first exception.
void parallel_for_chunks(int64_t n, int64_t chunk_size, const std::function<void(int64_t, int64_t, int64_t)> &f);

// bounding box + mean rest length of a whole mesh, exactly as build_plan measures them
void compute_domain(const Input &in, Domain &out);
Grid make_grid(const Domain &dom, int tile_target);
// the cells (and the box in rest coordinates, +-1e300 at the rim of the grid) opts.rank's window must cover: its block of the
// block partition + kWindowMarginCells. opts.tile_particles must be the resolved target (> 0, or -1 for no tiling).
void rank_window(const Domain &dom, const Opts &opts, int cell_lo[3], int cell_hi[3], double box_lo[3], double box_hi[3]);
// Throws std::runtime_error on invalid input.
void build_plan(const Input &in, const Opts &opts, Plan &out);
void extract_local(const Plan &plan, const Input &in, int rank, LocalPlan &out);

}  // namespace sbp
