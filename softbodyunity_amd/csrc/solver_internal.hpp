// solver_internal.hpp — what the translation units of libsoftbody_mi355x.so share: error plumbing, the run-time RCCL binding, device
// buffers, the solver object. Nothing declared here is exported (exports.map keeps the dynamic symbol table to sb_*).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
// Units: binding.hip (errors, RCCL / HIP runtime binding), tables.hip (plan -> device tables), schedule.hip (launches, ghost exchange, the
// tick), readback.hip (state reads / writes, render readback, kinematic targets), validate.hip (table validator), abi.hip (lifecycle,
// authoring, finalize, stats), plan_abi.hip (host-only planner inspection), group.hip (one process driving several devices).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is bound at run time (RcclApi below)
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/softbody.h"
#include "../../include/softbody_plan.h"
#include "../../include/softbody_debug.h"
#include "../../include/softbody_group.h"
#include "kernel_types.hpp"
#include "plan.hpp"

namespace sbi {

int fail(int code, const std::string &msg);      // records the thread's last error (sb_last_error) and returns code
const char *last_error_text();

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            throw HipError(SB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)
// RCCL is NOT a link-time dependency. A world == 1 host never loads it (the library is 570 MB); a world > 1 host binds, on
// first use, the librccl.so.1 that is ALREADY in the process when there is one -- a host that imported PyTorch first brought
// PyTorch's own RCCL together with PyTorch's own HIP runtime, which this plugin's libamdhip64.so.7 dependency resolved to as
// well, and a second ROCm stack in one process is the one thing that must not happen -- and the system's otherwise (the
// plugin's RUNPATH: /opt/rocm/lib). What was bound is reported by sb_runtime_info and decides which schedules are admitted.
struct RcclApi {
    void *handle = nullptr;
    bool was_resident = false;
    int version = 0;
    std::string path, error;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    bool ok() const { return handle != nullptr; }
};
RcclApi &rccl(bool required = true);
#define NCCL_CHECK(expr)                                                                                  \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess)                                                                            \
            throw HipError(SB_ERR_RCCL, std::string(#expr) + ": " + rccl().GetErrorString(r_));          \
    } while (0)

int hip_runtime_version();
bool capture_overlap_ok();      // see binding.hip

template <class T>
struct DevBuf {
    T *p = nullptr;
    T *base = nullptr;           // what hipMalloc returned (p = base + lead: placement experiments, sb_tuning.prev_offset_bytes)
    size_t count = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;                 // owns device memory
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), base(o.base), count(o.count) { o.p = o.base = nullptr; o.count = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { free(); p = o.p; base = o.base; count = o.count; o.p = o.base = nullptr; o.count = 0; } return *this; }
    void alloc(size_t n, int64_t &acct, size_t lead_elems = 0) {
        free();
        count = n;
        if (n) { HIP_CHECK(hipMalloc((void **)&base, (n + lead_elems) * sizeof(T))); p = base + lead_elems; acct += (int64_t)((n + lead_elems) * sizeof(T)); }
    }
    void upload(const std::vector<T> &h, int64_t &acct) {
        alloc(h.size(), acct);
        if (!h.empty()) HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void free() { if (base) { (void)hipFree(base); } base = nullptr; p = nullptr; count = 0; }
    ~DevBuf() { free(); }
};

struct DevHalo {                 // one halo slot: who we talk to and which particles travel
    std::vector<int> peers;
    std::vector<int32_t> send_off, recv_off;  // per peer (+1), in particles
    DevBuf<int32_t> send_idx, recv_idx;
    bool active() const { return !peers.empty(); }
};

struct DevTiling {
    int32_t n_tiles = 0;
    size_t lds_bytes = 0;
    int64_t n_slots = 0;         // constraints stored in the tile streams
    int64_t staged_particles = 0;   // sum of n_local over the device tiles
    int64_t stream_bytes = 0;    // bytes of the tile streams (round words, palettes, slots)
    int32_t max_local = 0, win_dwords = 4, pal_dwords = 0, rounds_dwords = 0;
    int32_t n_boundary = 0;      // world > 1: T0 -- the FIRST n_boundary tiles hold every particle some peer needs; T1 -- the LAST
                                 // n_boundary tiles hold every ghost and every sent particle
    bool has_quads = false;
    int32_t item_waves = 0;      // waves per tile the wave items were dealt for (0 = the streams hold none)
    int32_t packed_lanes = 0;    // 128: tiles of this tiling may hold lane-packed slots (kernels.hip.hpp kLanePack*): EVERY launch of it runs 128-lane workgroups
    int64_t n_packed_tiles = 0;
    DevBuf<sbk::TileDesc> tiles;
    DevBuf<int2> runs_overflow;
    DevBuf<uint32_t> stream;     // per tile: [round words][rest-length dictionary][round data], see kernels.hip.hpp
    DevBuf<int32_t> gather;      // T2: particle lists of the tiles (local numbering)
};

struct DevGColour {
    int type = 0;
    int32_t count = 0;
    DevBuf<int2> ij;
    DevBuf<int4> quad;
    DevBuf<float> rest;
    DevBuf<float2> rest2;
};

}  // namespace sbi

namespace sbi {

// The tick as a program (schedule.hip tick_program): launches and exchanges in order, tile ranges by name.
enum class StepKind : uint8_t { Tile, T2Layer, GColour, Exchange, ForkExchange, JoinExchange };
enum class TileRange : uint8_t { All, T0Boundary, T0Interior, T1Interior, T1Boundary };
struct TickStep {
    StepKind kind;
    TileRange range;        // Tile: which of the rank's tiles
    bool kin;               // Tile: the fused first kernel carries kinematic targets (tile_kernel KIND 5)
    int it, substeps;       // Tile: kernel K_it of a tick of `substeps` substeps
    int index;              // T2Layer: layer; GColour: colour; *Exchange: halo slot
};
struct TickShape { int substeps; bool fuse, defer_last, kin; };

// sb_debug_exchange_timing: three events per exchange (start, after the pack / push kernel, end) on the stream it runs on
struct ExchangeTimer {
    bool enabled = false;
    std::vector<hipEvent_t> pending, join_pending, free_list;      // join_pending: pairs around the compute stream's wait for an overlapped exchange
    void mark(hipStream_t st, bool join = false);
    ~ExchangeTimer();
};

}  // namespace sbi

using sbi::DevBuf; using sbi::DevGColour; using sbi::DevHalo; using sbi::DevTiling;

struct sb_plan {
    sbp::Plan plan;
    sbp::LocalPlan local;
    // keep inputs needed by inspection calls
    bool owns_local = true;
};

struct sb_solver {
    sb_desc desc;
    bool finalized = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ncclComm_t comm = nullptr;
    bool loopback = false;           // SB_DEBUG_LOOPBACK: every peer is this rank itself (1-GPU pipeline test)
    int schedule = SB_SCHEDULE_SERIAL_EAGER;   // world > 1: what desc.halo_schedule resolved to (sb_finalize)
    uint64_t plan_hash = 0;          // hash of the published orders, ownership and plan options (equal on every rank)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_boundary = nullptr, ev_halo = nullptr;
    bool overlap_halo = false;       // T0 boundary tiles first, ghost exchange on comm_stream beside the interior
    // authoring copies
    int32_t n = 0;
    std::vector<float> pos, vel, invm, rest;
    bool sharded = false;            // sb_set_domain: the authoring arrays are this rank's window of a larger mesh
    sb_domain domain{};
    std::vector<int32_t> global_id;  // [n] ids of the window's particles in the whole mesh (sharded only)
    std::vector<int32_t> dist_ij, vol_ijkl, bend_ijkl;
    std::vector<float> dist_rest, vol_rest, bend_rest;
    float compliance[3] = {0, 0, 0};
    float plane[4] = {0, 1, 0, 0};
    int32_t plane_on = 0;
    // plan
    std::unique_ptr<sb_plan> plan;
    // device state
    int64_t dev_bytes = 0;
    int64_t n_owned = 0, n_local = 0;
    DevBuf<float> d_pos3;            // packed xyz per local particle
    DevBuf<float> d_wf;              // inverse mass per local particle (static)
    DevBuf<uint8_t> d_w8;            // palette index of the inverse mass (when <= 64 distinct values)
    DevBuf<float> d_wpal;
    bool w_palette = false;
    bool w_uniform = false;          // one distinct inverse mass: the tile kernels skip the per-particle index read
    sbk::PosView pos_view() const { return sbk::PosView{d_pos3.p, d_wf.p}; }
    DevBuf<float> d_prev, d_vel;
    DevBuf<sbk::TickParams> d_tp;
    DevBuf<float> d_sendbuf, d_recvbuf;   // 3 (slot 1: 6) floats per ghost, peers back to back
    DevTiling tiling[3];             // T0, T1, and the sparse T2 tiles (all layers; see t2_layer_range)
    std::vector<std::pair<int32_t, int32_t>> t2_layer_range;   // device-tile ranges of tiling[2], one per T2 layer
    std::vector<std::unique_ptr<DevGColour>> gcolours;
    std::vector<std::unique_ptr<DevHalo>> halos;   // indexed by halo slot
    sbk::TickParams tp_host{};
    bool tp_valid = false;
    struct CachedGraph { hipGraphExec_t exec; uint64_t last_use; };
    std::map<int, CachedGraph> graphs;         // key = substeps * 8 + (1: tick starts with the fused kernel) + (2: last kernel deferred) + (4: that kernel carries kinematic targets)
    uint64_t graph_clock = 0;                  // least recently used entry is evicted beyond kMaxGraphs (a host that varies substeps)
    static constexpr size_t kMaxGraphs = 8;
    // Lazy tick boundary: the last kernel of a tick (rounds + collide + velocity write) is deferred; if the next tick
    // has the same parameters it is FUSED with that tick's first kernel into one ordinary mid-tick kernel, otherwise
    // (or whenever state is read or written) it is flushed first. Results are identical either way.
    bool deferred = false;
    int deferred_substeps = 0;
    // Tuning (sb_set_tuning of softbody_debug.h: A/B measurements; the plugin reads no environment variable for any of this)
    uint32_t tune_flags = 0;         // SB_TUNE_* as given; the fields below are what they resolve to
    int win_dwords_cap = 0;          // sb_tuning.win_dwords (0 = the tiling's own window)
    int prev_offset_bytes = 0;       // sb_tuning.prev_offset_bytes: the previous-position array starts this far into its allocation (placement experiment)
    bool lazy_tick = true;           // !SB_TUNE_NO_LAZY_TICK
    int tile_lanes = 0;              // sb_tuning.tile_lanes = 128|256|512 forces the workgroup width of small tiles (0 = by launch size)
    int quad_lanes = 512;            // sb_tuning.quad_lanes = 256|512: workgroup width of tiles that hold tets / hinges (8 waves: every group of the
                                     // 100 k surrogate fits one row of wave slots; 1.99 against 2.11 ms per tick with 4 waves)
    int store_through_max_tiles = 6144;   // sb_tuning.store_through_max_tiles: launches of at most this many tiles store their state through the L2
                                          // (measured: 96^3 -18 %, 128^3 = 4096 tiles -3 %, 160^3 = 8000 tiles +2 %, 256^3 +4 %)
    int store_through_large = 0;          // sb_tuning.store_through_large = mask: the same for larger launches (experiments; bit 0 previous positions, bit 1 positions)
    int narrow_min_tiles = 10240;    // sb_tuning.narrow_min_tiles; measured crossover: 160^3 (8000 tiles) ties, 192^3 (13824) +4 % narrow
    size_t lds_pad = 0;              // sb_tuning.lds_pad_bytes of unused LDS per workgroup (occupancy experiments)
    bool pack_tiles = true;          // !SB_TUNE_NO_PACK: under-full tiles share a workgroup (build_device)
    bool fused_unpack = false;       // the T1 kernels read ghosts from the receive buffer: no unpack launch behind the slot-1 exchange
    bool graph_rccl = false;         // a multi-rank tick, exchange included, is captured in the hipGraph (SB_SCHEDULE_*_GRAPH)
    std::vector<float> h_stage;
    // SB_SCHEDULE_AUTO measured on the devices at hand (schedule.hip calibrate_*): ticks 0 / 1 warm the two eager schedules up (first use of
    // the communicator, of the second stream), ticks 2 .. 5 alternate serialised / overlapped under HIP events, the 7th sb_step decides.
    struct AutoSchedule {
        int state = 0;                       // 0 off, 1 calibrating, 2 decided
        int tick = 0;                        // sb_step calls so far
        hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // start / end of the four timed ticks
        int n[2] = {0, 0};                   // ticks timed per schedule
        double decided_ms[2] = {0, 0};       // per tick, slowest rank: [0] serialised, [1] overlapped
    } calib;
    bool group_walk = false;         // a rank of a group whose host thread walks the tick across the ranks (group.hip): exchanges are issued there
    bool capturing = false;          // sb_step is recording the tick into a hipGraph right now
    sbi::ExchangeTimer xtimer;       // sb_debug_exchange_timing
    // peer-store halo transport (SB_HALO_TRANSPORT=peer; kernels.hip.hpp): one mailbox per rank = [header words | ghost segments]
    struct PeerState {
        bool enabled = false, linked = false, fine_grained = false;
        uint32_t *mailbox = nullptr;            // header: words 0-1 = the rank's plan hash, 2 = sharded?, 4 .. 4+2W = its pair hashes; per slot: data flags[world], ack flags[world], epoch, 2 counters; then the offset table
        size_t bytes = 0, data_off_words = 0, off_table = 0;
        int n_slots = 0;
        std::vector<uint32_t *> remote;         // [world]: the ranks' mailboxes as this process sees them (own pointer for itself)
        std::vector<uint8_t> opened;            // remote[r] was mapped with hipIpcOpenMemHandle
        std::vector<std::vector<uint32_t>> my_off;   // [slot][rank]: first word (from the mailbox start) of rank's segment in MY mailbox
        std::vector<sbk::PeerSlot> slots;
        uint32_t *local = nullptr;              // 8 ordinary (cached) words per slot: epoch, workgroup counters, go words
        uint32_t *h_error = nullptr;            // pinned host word the kernels set when a wait gives up: the host reads it without a copy
        size_t slot_base(int slot, int world) const { return 4 + 2 * (size_t)world + (size_t)slot * (2 * (size_t)world + 3); }
    } peer;
    // asynchronous render readback (sb_readback_begin / sb_readback_end): two snapshot slots
    hipStream_t copy_stream = nullptr;
    DevBuf<int32_t> d_local_to_old;
    DevBuf<float> d_get_scratch;           // caller-numbered staging of the blocking sb_get_* calls (world == 1)
    // three slots, at most two pending: the slot sb_readback_end handed out last is never the next one to be filled, so
    // its pointer stays valid until the SECOND sb_readback_begin after it (softbody.h)
    static constexpr int kSnapSlots = 3;
    DevBuf<float> d_snap[kSnapSlots];
    float *h_snap[kSnapSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_snap[kSnapSlots] = {nullptr, nullptr, nullptr}, ev_copied[kSnapSlots] = {nullptr, nullptr, nullptr};
    int snap_head = 0, snap_pending = 0;   // ring: slots snap_head .. snap_head + snap_pending - 1 (mod kSnapSlots) are in flight
    // render normals of the snapshots (sb_set_render_triangles): incident-triangle lists per particle, caller numbering
    std::vector<int32_t> render_tri;
    bool render_dirty = false;             // triangles changed since the last upload
    DevBuf<int32_t> d_tri, d_adj_off, d_adj_tri, d_render_set, d_render_local;
    std::vector<int32_t> render_set;       // particles used by the render triangles, ascending
    bool render_set_only = false;          // readbacks bring the render set only (compact positions + normals)
    DevBuf<float> d_cpos[kSnapSlots];      // compact positions of the render set
    float *h_cpos[kSnapSlots] = {nullptr, nullptr, nullptr};
    bool snap_compact[kSnapSlots] = {false, false, false};
    DevBuf<float> d_nrm[kSnapSlots];
    float *h_nrm[kSnapSlots] = {nullptr, nullptr, nullptr};
    bool snap_has_normals[kSnapSlots] = {false, false, false};
    bool snap_has_render_set[kSnapSlots] = {false, false, false};
    std::vector<int32_t> render_local;     // device numbering of render_set's particles
    int snap_last_ended = -1;
    // kinematic targets (sb_set_kinematic_positions): a ring of pinned host tables the scatter kernel reads directly; a table is reused
    // only after the kernel that read it has finished (its event)
    static constexpr int kKinSlots = 4;
    int32_t *h_kin_idx[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};
    float *h_kin_pos[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *d_kin_idx[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};      // device-side aliases of the mapped tables
    float *d_kin_pos[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint32_t> kin_seen; uint32_t kin_stamp = 0;                    // duplicate-id check of sb_set_kinematic_positions
    std::vector<int32_t> local_of_old;     // the rank's numbering -> device numbering (-1: not held), built at first use (readback.hip)
    size_t kin_cap[kKinSlots] = {0, 0, 0, 0};
    hipEvent_t ev_kin[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};
    int kin_next = 0;
    // Targets are PENDING until the next tick starts: if that tick's first kernel also finishes the tick before (lazy tick boundary),
    // they travel into it (tile_kernel KIND 5 applies them between the old tick's velocity and the new tick's integrate) and the
    // fusion is kept; any other way across the boundary (state read or written, parameters changed, first tick) completes the old
    // tick and scatters them onto the positions (materialise_kinematic).
    int kin_pending = -1, kin_pending_count = 0;      // ring slot that holds them, or -1
    DevBuf<int32_t> d_kin_map;             // per local particle: slot of a pinned particle, -1 for a free one (built at the first use)
    DevBuf<float> d_kin_target;            // 3 floats per pinned particle: pending target or NaN
    int64_t n_kin_fused = 0;               // ticks whose fused first kernel carried targets
    bool kin_fuse = true;                  // !SB_TUNE_NO_KIN_FUSE (A/B: pending targets always complete the previous tick first)
    // Peek (world == 1): a position read while the tick's last kernel is deferred runs tile_kernel<4> -- the same rounds + collide on
    // the same inputs, written to d_peek instead of the state -- so the deferred kernel can still be fused with the next tick's first
    // one. A render-set-only readback peeks only at the T0 tiles that hold a render particle (peek_tiles: copies of their descriptors).
    DevBuf<float> d_peek;                  // packed xyz per local particle; only the peeked tiles' entries are ever written or read
    DevBuf<sbk::TileDesc> peek_tiles;
    int32_t n_peek_tiles = -1;             // -1: not built for the current render set
    bool peek_enabled = true;              // !SB_TUNE_NO_PEEK
    // A peek is one more launch; what it saves is the difference between a fused first kernel and a separate last + first kernel. That
    // pays where launches are bandwidth-bound (256^3 render-set readback: 3.49 -> 3.35 ms per tick) and costs where a launch is a fixed
    // latency whatever it covers (every tile resident at once -- 100 k tet mesh: +15 us per tick, 64^3: +2 us; profiles/
    // r03s2_soak_peek.jsonl): peek only from this many T0 workgroups on (sb_tuning.peek_min_tiles).
    int peek_min_tiles = 2048;
    int64_t n_peeks = 0;                   // launches so far (sb_stats.readback_peeks)
    int64_t n_fused = 0;                   // ticks that started with the fused kernel (sb_stats.ticks_fused)

    ~sb_solver() {
        // Teardown order: everything the device may still be running for this solver first (compute, exchange and copy
        // streams), then the graph executables (captured RCCL launches hold references into the communicator), then the
        // communicator, then the streams and events.
        if (stream) (void)hipStreamSynchronize(stream);
        if (comm_stream) (void)hipStreamSynchronize(comm_stream);
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
        for (auto &g : graphs) (void)hipGraphExecDestroy(g.second.exec);
        graphs.clear();
        if (comm) (void)sbi::rccl(false).CommDestroy(comm);
        for (size_t r = 0; r < peer.remote.size(); ++r) if (peer.opened[r] && peer.remote[r]) (void)hipIpcCloseMemHandle(peer.remote[r]);
        if (peer.mailbox) (void)hipFree(peer.mailbox);
        if (peer.local) (void)hipFree(peer.local);
        if (peer.h_error) (void)hipHostFree(peer.h_error);
        for (hipEvent_t e : calib.ev) if (e) (void)hipEventDestroy(e);
        if (ev_boundary) (void)hipEventDestroy(ev_boundary);
        if (ev_halo) (void)hipEventDestroy(ev_halo);
        if (comm_stream) (void)hipStreamDestroy(comm_stream);
        gcolours.clear(); halos.clear();
        for (int k = 0; k < kSnapSlots; ++k) {
            if (h_snap[k]) (void)hipHostFree(h_snap[k]);
            if (h_nrm[k]) (void)hipHostFree(h_nrm[k]);
            if (h_cpos[k]) (void)hipHostFree(h_cpos[k]);
            if (ev_snap[k]) (void)hipEventDestroy(ev_snap[k]);
            if (ev_copied[k]) (void)hipEventDestroy(ev_copied[k]);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (int k = 0; k < kKinSlots; ++k) {
            if (h_kin_idx[k]) (void)hipHostFree(h_kin_idx[k]);
            if (h_kin_pos[k]) (void)hipHostFree(h_kin_pos[k]);
            if (ev_kin[k]) (void)hipEventDestroy(ev_kin[k]);
        }
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace sbi {

// ---- tables.hip: plan -> device tables ---------------------------------------------------------------------------------------------
sbp::Input make_input(const float *rest, int32_t n, const int32_t *d, int64_t md, const int32_t *v, int64_t mv, const int32_t *b, int64_t mb);
sbp::Domain to_domain(const sb_domain &d);
// The planner options behind the ABI's fields: ONE rule for sb_finalize and sb_plan_build
sbp::Opts plan_opts(int rank, int world, const int32_t dims[3], int32_t tile_particles, int32_t partition, uint32_t plan_flags,
                    int64_t m_v, int64_t m_b, const sb_domain *domain = nullptr);
constexpr uint32_t kPlanFlagsAll = SB_PLAN_NO_T2 | SB_PLAN_NO_THIRD_LIST | SB_PLAN_NO_CLUSTER_LAYERS | SB_PLAN_NO_MIXED_GROUPS | SB_PLAN_NO_BANK_ORDER | SB_PLAN_NO_TILE_MERGE | SB_PLAN_BALANCED_LISTS(3);
uint64_t hash_plan(const sbp::Plan &P);
void build_device(sb_solver *s);

// ---- schedule.hip: launches, ghost exchange, the tick -----------------------------------------------------------------------------------
sbk::TickParams tick_params(const sb_solver *s, float dt, int substeps);
void upload_tick_params(sb_solver *s, float dt, int substeps);
void peer_link(sb_solver *s);
void halo_exchange(sb_solver *s, int slot, hipStream_t st = nullptr);
bool halo_slot_active(const sb_solver *s, int slot);
void halo_exchange_pre(sb_solver *s, int slot, hipStream_t st);       // the three parts of an exchange, for a host thread that drives several ranks
void halo_exchange_calls(sb_solver *s, int slot, hipStream_t st);
void halo_exchange_post(sb_solver *s, int slot, hipStream_t st);
std::vector<TickStep> tick_program(const sb_solver *s, int substeps, bool fused_first, bool defer_last, bool kin);
struct LaunchTimer;
void run_step(sb_solver *s, const TickStep &st, LaunchTimer *lt);
TickShape begin_tick(sb_solver *s, float dt, int substeps);
void end_tick(sb_solver *s, const TickShape &t);
void flush_deferred(sb_solver *s);
void scatter_kinematic(sb_solver *s, float *dst);
bool can_peek(const sb_solver *s);
void build_peek_subset(sb_solver *s, const std::vector<int32_t> &wanted);
void peek_positions(sb_solver *s, bool subset);
void check_peer_error(sb_solver *s);

// ---- readback.hip -----------------------------------------------------------------------------------------------------------------------
const std::vector<int32_t> &local_of_old(sb_solver *s);
int get_state_owned(sb_solver *s, float *out, bool velocity, const int32_t *id_map);
int set_state_from(sb_solver *s, const float *pos, const float *vel, const int32_t *id_map);
int set_kinematic(sb_solver *s, const int32_t *ids, const float *pos, int32_t count);
const float *render_source(sb_solver *s, bool compact, const std::vector<int32_t> &wanted_local);
void launch_snapshot_all(sb_solver *s, const float *src_xyz, const int32_t *d_target_of_local, float *dst_xyz);
void launch_snapshot_subset(sb_solver *s, const float *src_xyz, const int32_t *d_ids, const int32_t *d_local, int count, float *dst_xyz);
void launch_normals(hipStream_t st, const float *snap_xyz, const int32_t *adj_off, const int32_t *adj_tri, const int32_t *tri, float *nrm_xyz, int count,
                    const int32_t *subset, float *subset_pos_xyz);

// ---- abi.hip: the phases of sb_finalize (a group runs them itself) ----------------------------------------------------------------------
int finalize_local(sb_solver *s);                       // plan + tables, this rank alone (= finalize_plan, then finalize_device)
int finalize_plan(sb_solver *s);                        //   host work only: schedule, plan, plan hash
int finalize_device(sb_solver *s);                      //   device tables, streams
void reset_authoring(sb_solver *s);                     // forget a window (sb_set_domain) and the plan made from it
std::vector<uint64_t> agreement_record(const sb_solver *s, bool failed);
uint32_t plan_shape(const sb_solver *s);                // the tick program's shape: tiling on / off, leftover (T2) layers, global colours (= halo slots)
int check_agreement(const std::vector<uint64_t> &all, int W, int me);
int finalize_agree(sb_solver *s, int local_rc);         // RCCL all-gather of the agreement records (entered by a failed rank too)
int finalize_link(sb_solver *s);                        // peer mailboxes over the communicator, bookkeeping
template <class F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const HipError &e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc &) {
        return fail(SB_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(SB_ERR_INVALID_ARG, e.what());
    }
}

int set_device(const sb_solver *s);

}  // namespace sbi
