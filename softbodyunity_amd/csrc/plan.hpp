// plan.hpp — host-side planner: tiles, colours, partition, halo schedule (pure C++, no HIP).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the
// planner implements SPEC.md §3 and the design in DESIGN.md §3: the constraint graph is cut into
//   phase P1  : vertex-disjoint tiles = cells of a uniform grid over the rest pose (every particle is in
//               exactly one P1 tile, so P1 also carries integrate/velocity),
//   phase P2  : tiles = cells of the same grid shifted by half a cell, holding the constraints P1 left,
//   phases G* : whatever is still left, greedy edge-coloured, one global kernel per colour,
// and the flat sequential order equivalent to that execution is published for the oracle.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sbp {

struct Opts {
    int rank = 0, world = 1;
    int dims[3] = {0, 0, 0};
    int tile_particles = 512;  // -1: no tiling
};

struct Run {            // a contiguous range of particles
    int32_t start;      // first particle (global-new numbering in Plan, local numbering in LocalPlan)
    int32_t len;
};

struct ColourEntry {    // one colour class of one constraint type inside a tile
    int32_t type;       // 0 distance, 1 volume, 2 bending
    int64_t begin;      // into the phase's per-type tile-constraint arrays
    int32_t count;
};

struct Cluster {
    int32_t owner;          // P1: owning rank; P2: -1 (executed by every rank owning one of its runs)
    int32_t run_begin, run_count;
    int32_t n_local;        // particles staged in LDS
    int32_t col_begin, col_count;
    int64_t order_begin, order_end;
};

struct Phase {
    int kind;               // 0 global colour, 1 tile
    int type;               // kind 0: constraint type
    int64_t order_begin, order_end;
    int64_t task_begin, task_end;
    int32_t cluster_begin = 0, cluster_end = 0;  // kind 1
    bool fused_integrate = false;                // P1
    bool needs_halo = false;
};

struct Plan {
    Opts opts;
    int32_t n = 0;
    int64_t m[3] = {0, 0, 0};
    int dims[3] = {1, 1, 1};
    // particle numbering
    std::vector<int32_t> new_of_old, old_of_new, owner_of_old;
    // published order (original constraint ids)
    std::vector<uint8_t> order_type;
    std::vector<int32_t> order_id;
    std::vector<Phase> phases;
    std::vector<int64_t> task_off;
    std::vector<int64_t> group_off;   // finest independent sets: one tile colour class / one global-colour chunk
    // tile data (all ranks)
    std::vector<Cluster> clusters;
    std::vector<Run> runs;
    std::vector<ColourEntry> colours;
    // tile constraints in execution order, cluster-local particle indices (16 bit each)
    std::vector<uint32_t> t_dist;       // lo16 = i, hi16 = j
    std::vector<int32_t> t_dist_id;     // original constraint id (for rest values)
    std::vector<uint32_t> t_quad;       // 2 words per 4-vertex constraint (volume then bending share the array)
    std::vector<int32_t> t_quad_id;     // original id within its type
    std::vector<uint8_t> t_quad_type;
    int32_t max_tile_local = 0, max_tile_runs = 0;
    // stats
    int64_t cons_in_tiles = 0, cons_in_global = 0;
    int n_tile_phases = 0, n_global_colours = 0;
};

// What one rank uploads and executes.
struct LocalPhase {
    int kind, type;
    bool fused_integrate, needs_halo;
    // kind 1
    std::vector<int32_t> cluster_ids;       // global cluster ids, execution order
    std::vector<Run> runs;                  // local numbering, concatenated per cluster
    std::vector<int32_t> run_begin;         // per local cluster (+1)
    // kind 0: constraints with local particle indices
    std::vector<int32_t> g_idx;             // 2 or 4 per constraint
    std::vector<int32_t> g_id;              // original id
    // halo before this phase: per peer, local indices to send / to receive into
    std::vector<std::vector<int32_t>> send_idx, recv_idx;
};

struct LocalPlan {
    int rank = 0, world = 1;
    int64_t n_owned = 0;
    std::vector<int32_t> local_to_old;      // owned first (global-new order), then ghosts
    std::vector<LocalPhase> phases;
    std::vector<uint8_t> order_mask;        // which order entries this rank executes
};

struct Input {
    const float *rest;
    int32_t n;
    const int32_t *dist_ij; int64_t m_d;
    const int32_t *vol; int64_t m_v;
    const int32_t *bend; int64_t m_b;
};

// Throws std::runtime_error on invalid input.
void build_plan(const Input &in, const Opts &opts, Plan &out);
void extract_local(const Plan &plan, const Input &in, int rank, LocalPlan &out);

constexpr int kMaxTileLocal = 4096;   // particles staged per tile (64 KiB of LDS as float4)
constexpr int kMaxTileRuns = 64;

}  // namespace sbp
