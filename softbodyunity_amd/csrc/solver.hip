// solver.hip — solver runtime + C ABI of libsoftbody_mi355x.so (include/softbody.h).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the
// exported functions are the [BUILDER-DEFINED] boundary of SURVEY.md §8b. This file owns device
// memory, the substep loop (hipGraph replay), the RCCL ghost exchange and the plan inspection API.
// There is deliberately no CPU execution path: without a gfx950 device sb_create fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is bound at run time (RcclApi below)
#include <dlfcn.h>

#include <algorithm>
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
#include "kernels.hip.hpp"
#include "plan.hpp"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

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
RcclApi &rccl(bool required = true) {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *nm : names) if (!api.handle) { api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD); api.was_resident = api.handle != nullptr; }
        for (const char *nm : names) if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) { const char *e = dlerror(); api.error = std::string("RCCL could not be loaded: ") + (e ? e : "librccl.so.1 not found"); return; }
        bool all = true;
        auto sym = [&](auto &fn, const char *name) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name)); all &= fn != nullptr; };
        sym(api.GetVersion, "ncclGetVersion"); sym(api.GetUniqueId, "ncclGetUniqueId"); sym(api.CommInitRank, "ncclCommInitRank");
        sym(api.CommDestroy, "ncclCommDestroy"); sym(api.GetErrorString, "ncclGetErrorString"); sym(api.GroupStart, "ncclGroupStart");
        sym(api.GroupEnd, "ncclGroupEnd"); sym(api.Send, "ncclSend"); sym(api.Recv, "ncclRecv"); sym(api.AllGather, "ncclAllGather");
        if (!all) { api.error = "RCCL library lacks a symbol the plugin needs"; api.handle = nullptr; return; }
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(api.Send), &di) && di.dli_fname) api.path = di.dli_fname;
        if (api.GetVersion(&api.version) != ncclSuccess) api.version = 0;
        // the plugin uses only calls whose signatures have not changed since NCCL 2.7 (send/recv); refuse anything older
        if (api.version < 20700) { api.error = "RCCL " + std::to_string(api.version) + " is older than 2.7 (no ncclSend/ncclRecv)"; api.handle = nullptr; }
    });
    if (required && !api.ok()) throw HipError(SB_ERR_UNSUPPORTED, api.error);
    return api;
}
#define NCCL_CHECK(expr)                                                                                  \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess)                                                                            \
            throw HipError(SB_ERR_RCCL, std::string(#expr) + ": " + rccl().GetErrorString(r_));          \
    } while (0)

// HIP runtimes older than 7.2 recurse without bound in hipStreamEndCapture when a captured stream that was itself forked
// from the origin (the exchange stream of the overlapped schedule) is forked again (RCCL's internal stream joins the capture
// from the stream it is called on): the list of parallel capture streams becomes cyclic. Found with a native backtrace on
// PyTorch's bundled HIP 7.0.51831 (profiles/r03a_overlap_capture_backtrace.txt); the same process on HIP 7.2.26015 is fine.
int hip_runtime_version() {
    static const int v = [] { int x = 0; return hipRuntimeGetVersion(&x) == hipSuccess ? x : 0; }();
    return v;
}
bool capture_overlap_ok() { return hip_runtime_version() >= 70200000; }

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    void alloc(size_t n, int64_t &acct) {
        free();
        count = n;
        if (n) { HIP_CHECK(hipMalloc((void **)&p, n * sizeof(T))); acct += (int64_t)(n * sizeof(T)); }
    }
    void upload(const std::vector<T> &h, int64_t &acct) {
        alloc(h.size(), acct);
        if (!h.empty()) HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void free() { if (p) { (void)hipFree(p); p = nullptr; } count = 0; }
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

}  // namespace

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
    bool lazy_tick = true;           // SB_NO_LAZY_TICK unset (read once in sb_create)
    int tile_lanes = 0;              // SB_TILE_LANES=128|256|512 forces the workgroup width of small tiles (0 = by launch size)
    int quad_lanes = 512;            // SB_QUAD_LANES=256|512: workgroup width of tiles that hold tets / hinges (8 waves: every group of the
                                     // 100 k surrogate fits one row of wave slots; 1.99 against 2.11 ms per tick with 4 waves)
    int store_through_max_tiles = 6144;   // SB_STORE_THROUGH_MAX_TILES: launches of at most this many tiles store their state through the L2
                                          // (measured: 96^3 -18 %, 128^3 = 4096 tiles -3 %, 160^3 = 8000 tiles +2 %, 256^3 +4 %)
    int store_through_large = 0;          // SB_STORE_THROUGH_LARGE=mask: the same for larger launches (experiments; bit 0 previous positions, bit 1 positions)
    int narrow_min_tiles = 10240;    // SB_NARROW_MIN_TILES; measured crossover: 160^3 (8000 tiles) ties, 192^3 (13824) +4 % narrow
    size_t lds_pad = 0;              // SB_LDS_PAD bytes of unused LDS per workgroup (occupancy experiments)
    bool pack_tiles = true;          // SB_NO_PACK unset: under-full tiles share a workgroup (build_device)
    bool fused_unpack = false;       // the T1 kernels read ghosts from the receive buffer: no unpack launch behind the slot-1 exchange
    bool graph_rccl = false;         // a multi-rank tick, exchange included, is captured in the hipGraph (SB_SCHEDULE_*_GRAPH)
    std::vector<float> h_stage;
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
    int snap_last_ended = -1;
    // kinematic targets (sb_set_kinematic_positions): a ring of pinned host tables the scatter kernel reads directly; a table is reused
    // only after the kernel that read it has finished (its event)
    static constexpr int kKinSlots = 4;
    int32_t *h_kin_idx[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};
    float *h_kin_pos[kKinSlots] = {nullptr, nullptr, nullptr, nullptr};
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
    bool kin_fuse = true;                  // SB_NO_KIN_FUSE unset (A/B: pending targets always complete the previous tick first)
    // Peek (world == 1): a position read while the tick's last kernel is deferred runs tile_kernel<4> -- the same rounds + collide on
    // the same inputs, written to d_peek instead of the state -- so the deferred kernel can still be fused with the next tick's first
    // one. A render-set-only readback peeks only at the T0 tiles that hold a render particle (peek_tiles: copies of their descriptors).
    DevBuf<float> d_peek;                  // packed xyz per local particle; only the peeked tiles' entries are ever written or read
    DevBuf<sbk::TileDesc> peek_tiles;
    int32_t n_peek_tiles = -1;             // -1: not built for the current render set
    bool peek_enabled = true;              // SB_NO_PEEK unset (read once in sb_create)
    // A peek is one more launch; what it saves is the difference between a fused first kernel and a separate last + first kernel. That
    // pays where launches are bandwidth-bound (256^3 render-set readback: 3.49 -> 3.35 ms per tick) and costs where a launch is a fixed
    // latency whatever it covers (every tile resident at once -- 100 k tet mesh: +15 us per tick, 64^3: +2 us; profiles/
    // r03s2_soak_peek.jsonl): peek only from this many T0 workgroups on (SB_PEEK_MIN_TILES).
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
        if (comm) (void)rccl(false).CommDestroy(comm);
        for (size_t r = 0; r < peer.remote.size(); ++r) if (peer.opened[r] && peer.remote[r]) (void)hipIpcCloseMemHandle(peer.remote[r]);
        if (peer.mailbox) (void)hipFree(peer.mailbox);
        if (peer.local) (void)hipFree(peer.local);
        if (peer.h_error) (void)hipHostFree(peer.h_error);
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

namespace {

sbp::Input make_input(const float *rest, int32_t n, const int32_t *d, int64_t md, const int32_t *v, int64_t mv,
                      const int32_t *b, int64_t mb) {
    sbp::Input in;
    in.rest = rest; in.n = n; in.dist_ij = d; in.m_d = md; in.vol = v; in.m_v = mv; in.bend = b; in.m_b = mb;
    return in;
}

// The planner options behind the ABI's fields: ONE rule for sb_finalize and sb_plan_build, so the CPU schedule a host builds
// with sb_plan_build is the one the GPU solver of the same mesh runs.
sbp::Domain to_domain(const sb_domain &d) {
    sbp::Domain D;
    D.set = true; D.n_global = d.n_global; D.ell = d.spacing; D.fill = d.fill > 0 && d.fill <= 1 ? d.fill : 1.0;
    for (int a = 0; a < 3; ++a) { D.lo[a] = d.lo[a]; D.hi[a] = d.hi[a]; }
    return D;
}
sbp::Opts plan_opts(int rank, int world, const int32_t dims[3], int32_t tile_particles, int32_t partition, uint32_t plan_flags,
                    int64_t m_v, int64_t m_b, const sb_domain *domain = nullptr) {
    sbp::Opts o;
    if (domain) {       // sharded: the automatic tile size follows the WHOLE mesh, which only the domain knows
        o.domain = to_domain(*domain);
        if (domain->four_vertex_constraints) m_v += 1;
    }
    o.rank = rank; o.world = world <= 0 ? 1 : world;
    for (int a = 0; a < 3; ++a) o.dims[a] = dims ? dims[a] : 0;
    // automatic tile size: 512 particles for spring meshes (bandwidth-bound: the fewest rim tiles that still fill the chip),
    // 256 when tets or hinges are present (latency-bound: shorter programs per tile, more tiles in flight)
    o.tile_particles = tile_particles != 0 ? tile_particles : (m_v + m_b > 0 ? 256 : 512);
    o.partition = partition;
    o.third_tiling = !(plan_flags & SB_PLAN_NO_T2);
    o.third_list = !(plan_flags & SB_PLAN_NO_THIRD_LIST);
    o.cluster_layers = !(plan_flags & SB_PLAN_NO_CLUSTER_LAYERS);
    o.mixed_groups = !(plan_flags & SB_PLAN_NO_MIXED_GROUPS);
    o.bank_aware_lanes = !(plan_flags & SB_PLAN_NO_BANK_ORDER);
    o.merge_tiles = !(plan_flags & SB_PLAN_NO_TILE_MERGE);
    if ((plan_flags >> 8) & 3u) o.balanced_lists = (int)((plan_flags >> 8) & 3u);
    return o;
}
constexpr uint32_t kPlanFlagsAll = SB_PLAN_NO_T2 | SB_PLAN_NO_THIRD_LIST | SB_PLAN_NO_CLUSTER_LAYERS | SB_PLAN_NO_MIXED_GROUPS | SB_PLAN_NO_BANK_ORDER | SB_PLAN_NO_TILE_MERGE | SB_PLAN_BALANCED_LISTS(3);

// 64-bit FNV-1a over everything the ranks of a partitioned solver must agree on: the published orders, who owns which
// particle, the phase list with its halo slots, and the options that shaped them. (The halo lists are functions of these.)
uint64_t hash_plan(const sbp::Plan &P) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        // 8 bytes at a time (the arrays are tens of MB at 256^3), tail bytewise
        size_t k = 0;
        for (; k + 8 <= bytes; k += 8) { uint64_t w; std::memcpy(&w, b + k, 8); h = (h ^ w) * 1099511628211ull; }
        for (; k < bytes; ++k) h = (h ^ b[k]) * 1099511628211ull;
    };
    const int32_t head[8] = {P.n, P.opts.world, P.opts.tile_particles, P.partition,
                             (int32_t)((P.opts.third_tiling ? 0 : 1) | (P.opts.third_list ? 0 : 2) | (P.opts.cluster_layers ? 0 : 4) |
                                       (P.opts.mixed_groups ? 0 : 8) | (P.opts.bank_aware_lanes ? 0 : 16) | (P.opts.merge_tiles ? 0 : 32) | (P.opts.balanced_lists << 8)),
                             P.dims[0], P.dims[1], P.dims[2]};
    mix(head, sizeof(head));
    mix(P.m, sizeof(P.m));
    for (int p = 0; p < 2; ++p) {
        mix(P.order_type[p].data(), P.order_type[p].size());
        mix(P.order_id[p].data(), P.order_id[p].size() * sizeof(int32_t));
        for (const sbp::Phase &ph : P.phases[p]) {
            const int64_t rec[6] = {ph.kind, ph.tiling, ph.halo_slot, ph.layer, ph.order_begin, ph.order_end};
            mix(rec, sizeof(rec));
        }
    }
    mix(P.owner_of_old.data(), P.owner_of_old.size() * sizeof(int32_t));
    return h;
}

// SPEC.md §2 host-side scalars (same operation order as oracle.c orc_scalars_for).
sbk::TickParams tick_params(const sb_solver *s, float dt, int substeps) {
    sbk::TickParams t{};
    volatile float S = (float)substeps;
    volatile float h = dt / S;
    t.h = h;
    volatile float inv_h = 1.0f / h;
    t.inv_h = inv_h;
    volatile float hx = h * s->desc.gravity[0], hy = h * s->desc.gravity[1], hz = h * s->desc.gravity[2];
    t.hgx = hx; t.hgy = hy; t.hgz = hz;
    volatile float td = s->desc.damping * h;
    volatile float kd = 1.0f - td;
    t.kd = kd < 0.0f ? 0.0f : (float)kd;
    volatile float h2 = h * h;
    volatile float ad = s->compliance[0] / h2;
    volatile float av = s->compliance[1] / h2;
    volatile float av36 = 36.0f * av;
    volatile float ab = s->compliance[2] / h2;
    t.at_d = ad; t.at_v = av36; t.at_b = ab;
    t.pnx = s->plane[0]; t.pny = s->plane[1]; t.pnz = s->plane[2]; t.pd = s->plane[3]; t.plane_on = s->plane_on;
    return t;
}

void build_device(sb_solver *s) {
    const sbp::Plan &P = s->plan->plan;
    const sbp::LocalPlan &L = s->plan->local;
    s->n_owned = L.n_owned;
    s->n_local = (int64_t)L.local_to_old.size();
    // particle state
    std::vector<float> hp((size_t)s->n_local * 3), hw((size_t)s->n_local);
    std::vector<float> hv((size_t)s->n_local * 3, 0.0f);
    sbp::parallel_for_chunks(s->n_local, 1 << 18, [&](int64_t, int64_t lb, int64_t le) {
        for (int64_t l = lb; l < le; ++l) {
            int32_t o = L.local_to_old[l];
            for (int c = 0; c < 3; ++c) { hp[3 * (size_t)l + c] = s->pos[3 * (size_t)o + c]; hv[3 * (size_t)l + c] = s->vel[3 * (size_t)o + c]; }
            hw[l] = s->invm[o];
        }
    });
    s->d_pos3.upload(hp, s->dev_bytes);
    s->d_wf.upload(hw, s->dev_bytes);
    {   // one byte per particle instead of four when the mesh uses few distinct masses (the usual case)
        std::vector<uint32_t> vals(hw.size());
        for (size_t l = 0; l < hw.size(); ++l) std::memcpy(&vals[l], &hw[l], 4);
        std::vector<uint32_t> uniq;           // sorted distinct bit patterns, given up beyond the palette size
        bool few = true;
        {
            uint32_t last = 0; bool have_last = false;
            for (uint32_t v : vals) {
                if (have_last && v == last) continue;
                last = v; have_last = true;
                auto it = std::lower_bound(uniq.begin(), uniq.end(), v);
                if (it != uniq.end() && *it == v) continue;
                if ((int)uniq.size() == sbk::kMaxMassPalette) { few = false; break; }
                uniq.insert(it, v);
            }
        }
        std::vector<float> pal(sbk::kMaxMassPalette, 0.0f);
        if (few && !std::getenv("SB_NO_MASS_PALETTE")) {
            std::vector<uint8_t> w8(hw.size());
            sbp::parallel_for_chunks((int64_t)hw.size(), 1 << 20, [&](int64_t, int64_t lb, int64_t le) {
                for (int64_t l = lb; l < le; ++l) w8[(size_t)l] = (uint8_t)(std::lower_bound(uniq.begin(), uniq.end(), vals[(size_t)l]) - uniq.begin());
            });
            for (size_t k = 0; k < uniq.size(); ++k) std::memcpy(&pal[k], &uniq[k], 4);
            s->d_w8.upload(w8, s->dev_bytes);
            s->w_palette = true;
            s->w_uniform = uniq.size() == 1 && !std::getenv("SB_NO_UNIFORM_MASS");
        }
        s->d_wpal.upload(pal, s->dev_bytes);
    }
    s->d_vel.upload(hv, s->dev_bytes);
    s->d_prev.alloc((size_t)s->n_local * 3, s->dev_bytes);
    HIP_CHECK(hipMemset(s->d_prev.p, 0, (size_t)s->n_local * 3 * sizeof(float)));
    s->d_tp.alloc(1, s->dev_bytes);
    // tilings: re-base this rank's tiles onto compact device arrays
    for (int tl = 0; tl < 3; ++tl) {
        const sbp::Tiling &G = P.T[tl];
        sbp::LocalTiling LT = L.T[tl];     // copy: T0 is re-ordered boundary tiles first
        DevTiling &D = s->tiling[tl];
        // world > 1: T0 launches run the tiles that hold sent particles FIRST (n_boundary of them), T1 launches run the
        // tiles that hold a ghost or a sent particle LAST: the ghost exchange between a T0 and the following T1 kernel can
        // then travel beside the T0 interior tiles and the T1 interior tiles, which touch none of the particles the
        // pack kernel reads or the unpack kernel writes (enqueue_substeps, overlapped schedule).
        std::vector<uint8_t> tile_is_b;            // per plan tile of LT after the re-ordering (tilings 0 and 1)
        if ((tl == 0 || tl == 1) && L.world > 1 && L.halo.size() > 1) {
            std::vector<uint8_t> sent((size_t)s->n_local, 0);
            for (const auto &lst : L.halo[1].send_idx) for (int32_t li : lst) sent[li] = 1;
            std::vector<int32_t> order(LT.tile_ids.size());
            std::vector<uint8_t> is_b(LT.tile_ids.size(), 0);
            for (size_t ci = 0; ci < LT.tile_ids.size(); ++ci) {
                order[ci] = (int32_t)ci;
                for (int32_t r = LT.run_begin[ci]; r < LT.run_begin[ci + 1] && !is_b[ci]; ++r) {
                    if (tl == 1 && (int64_t)LT.runs[r].start + LT.runs[r].len > s->n_owned) { is_b[ci] = 1; break; }   // a ghost run
                    for (int32_t q = 0; q < LT.runs[r].len; ++q) if (sent[LT.runs[r].start + q]) { is_b[ci] = 1; break; }
                }
            }
            if (tl == 0) std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return is_b[a] > is_b[b]; });
            else std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return is_b[a] < is_b[b]; });
            sbp::LocalTiling R;
            R.run_begin.push_back(0);
            for (int32_t ci : order) {
                R.tile_ids.push_back(LT.tile_ids[ci]);
                for (int32_t r = LT.run_begin[ci]; r < LT.run_begin[ci + 1]; ++r) R.runs.push_back(LT.runs[r]);
                R.run_begin.push_back((int32_t)R.runs.size());
                tile_is_b.push_back(is_b[ci]);
            }
            LT = R;
        }
        // Packing: a tile is only a set of particles whose own constraints are projected in LDS, so several under-full
        // plan tiles (the rim of the shifted grid, surface cells of an irregular mesh) can share one workgroup: their
        // particles are staged side by side and round r of the pack is the union of the members' next rounds of one
        // type. Members share no particle and keep their own round order, so the result is bit-identical to running
        // them one after the other (the published order); only the number of workgroups changes.
        const size_t n_plan_tiles = LT.tile_ids.size();
        const int capacity = sbk::kSmallTile;         // packs stay small tiles; plan tiles above that size are left alone
        // only tiles with short programs share a workgroup (the rim of a lattice: 3-4 rounds): zipping long programs of
        // an irregular mesh (40+ rounds per tile) lengthens them, and such launches do not fill the chip anyway
        constexpr int kPackMaxRounds = 8;     // (16: -0.2 %, 32: +0.7 %, 64: +11 % on the 100 k surrogate, profiles/r02zq_pack_rounds.json)
        std::vector<std::vector<int32_t>> packs;      // members (indices into LT.tile_ids), in execution order
        {
            std::vector<int32_t> pack_of(n_plan_tiles, -1), cand;
            auto layer_of = [&](int32_t plan_tile) {
                int ly = 0;
                while (ly + 1 < (int)P.t2_layers.size() && plan_tile >= P.t2_layers[ly].second) ++ly;
                return ly;
            };
            auto cls = [&](int32_t ci) { return tl == 2 ? layer_of(LT.tile_ids[ci]) : (tile_is_b.empty() ? 0 : (int)tile_is_b[(size_t)ci]); };
            auto size_of = [&](int32_t ci) { return G.tiles[LT.tile_ids[ci]].n_local; };
            auto runs_of = [&](int32_t ci) { return LT.run_begin[ci + 1] - LT.run_begin[ci]; };
            if (s->pack_tiles)
                for (size_t ci = 0; ci < n_plan_tiles; ++ci)
                    if (size_of((int32_t)ci) < capacity && runs_of((int32_t)ci) <= sbk::kInlineRuns &&
                        G.tiles[LT.tile_ids[ci]].n_rounds <= kPackMaxRounds) cand.push_back((int32_t)ci);
            std::sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) {
                if (cls(a) != cls(b)) return cls(a) < cls(b);
                if (size_of(a) != size_of(b)) return size_of(a) > size_of(b);
                return a < b;
            });
            struct Bin { int32_t fill, runs, members; };
            std::vector<Bin> bins;
            std::vector<std::vector<int32_t>> open((size_t)capacity + 1);   // open[r]: bins of the current class with r free slots
            int cur_cls = -1;
            for (int32_t ci : cand) {     // best fit, largest first
                if (cls(ci) != cur_cls) { for (auto &o : open) o.clear(); cur_cls = cls(ci); }
                int32_t chosen = -1;
                for (int r = size_of(ci); r <= capacity && chosen < 0; ++r)
                    for (size_t k = open[r].size(); k-- > 0;) {
                        const Bin &B = bins[open[r][k]];
                        if (B.runs + runs_of(ci) <= sbk::kInlineRuns && B.members < 16) {
                            chosen = open[r][k];
                            open[r].erase(open[r].begin() + (std::ptrdiff_t)k);
                            break;
                        }
                    }
                if (chosen < 0) { chosen = (int32_t)bins.size(); bins.push_back({0, 0, 0}); }
                Bin &B = bins[chosen];
                B.fill += size_of(ci); B.runs += runs_of(ci); ++B.members;
                open[capacity - B.fill].push_back(chosen);
                pack_of[ci] = chosen;
            }
            std::vector<int32_t> slot_of_bin(bins.size(), -1);
            int32_t n_boundary_packs = 0;
            for (size_t ci = 0; ci < n_plan_tiles; ++ci) {
                if (pack_of[ci] < 0) { packs.push_back({(int32_t)ci}); }
                else if (slot_of_bin[pack_of[ci]] < 0) { slot_of_bin[pack_of[ci]] = (int32_t)packs.size(); packs.push_back({(int32_t)ci}); }
                else { packs[slot_of_bin[pack_of[ci]]].push_back((int32_t)ci); continue; }
                if (!tile_is_b.empty() && tile_is_b[ci]) ++n_boundary_packs;     // (a pack never mixes the two classes)
            }
            D.n_boundary = n_boundary_packs;
        }
        if (tl == 2) {
            s->t2_layer_range.assign(P.t2_layers.size(), {0, 0});
            for (size_t pk = 0; pk < packs.size(); ++pk) {
                int ly = 0;
                while (ly + 1 < (int)P.t2_layers.size() && LT.tile_ids[packs[pk][0]] >= P.t2_layers[ly].second) ++ly;
                auto &rg = s->t2_layer_range[ly];
                if (rg.second == rg.first) rg.first = (int32_t)pk;
                rg.second = (int32_t)pk + 1;
            }
        }
        // Packs are independent: chunks of packs build their pieces of the tables side by side on host threads, the pieces
        // are then laid end to end in pack order (offsets re-based), exactly as a pack-by-pack loop would fill them.
        struct Piece {
            std::vector<sbk::TileDesc> tiles;
            std::vector<int2> overflow;
            std::vector<uint32_t> stream;
            std::vector<int32_t> dev_gather;
            int32_t max_local = 0, max_pal = 0, max_rounds = 0;
            uint32_t max_data = 4;
            bool has_quads = false;
        };
        auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        struct Part { int32_t member; int32_t cnt[3]; int64_t first_d, first_q; };   // a member's group inside a pack group
        struct PackRound { int32_t cnt[3]; std::vector<Part> parts; };                 // constraints per type (distance, volume, bending)
        const bool no_palette = std::getenv("SB_NO_PALETTE") != nullptr;
        // meshes with tets / hinges: per-wave step lists beside the group words (springs-only meshes never run the kernels that read them)
        const bool emit_items = (!s->vol_rest.empty() || !s->bend_rest.empty()) && !std::getenv("SB_NO_WAVE_ITEMS");
        const int item_waves = s->quad_lanes / 64;
        D.item_waves = emit_items ? item_waves : 0;
        // Lane-packed slots (kernels.hip.hpp kLanePack*): only where every launch of the tiling is known to run 128-lane workgroups --
        // a single-rank solver (no boundary / interior ranges) whose launches oversubscribe the chip (launch_tile: narrow) -- and the
        // mesh has springs only. Which TILES then qualify is decided tile by tile below.
        D.packed_lanes = 0;
        bool all_small = true;       // (a tiling with a tile above 512 particles launches the 1 024-particle kernels)
        for (size_t ci = 0; ci < n_plan_tiles; ++ci) all_small = all_small && G.tiles[LT.tile_ids[ci]].n_local <= sbk::kSmallTile;
        if (tl < 2 && L.world == 1 && all_small && s->vol_rest.empty() && s->bend_rest.empty() && !no_palette && !std::getenv("SB_NO_LANE_PACK") &&
            (s->tile_lanes == 0 || s->tile_lanes == sbk::kLanePackLanes) && (int64_t)packs.size() >= (int64_t)s->narrow_min_tiles)
            D.packed_lanes = sbk::kLanePackLanes;
        std::atomic<int64_t> n_packed_tiles{0};
        constexpr int64_t kPacksPerChunk = 128;
        const int64_t n_chunks = ((int64_t)packs.size() + kPacksPerChunk - 1) / kPacksPerChunk;
        std::vector<Piece> pieces((size_t)n_chunks);
        sbp::parallel_for_chunks((int64_t)packs.size(), kPacksPerChunk, [&](int64_t chunk, int64_t pk_begin, int64_t pk_end) {
        Piece &Q = pieces[(size_t)chunk];
        std::vector<sbk::TileDesc> &tiles = Q.tiles;
        std::vector<int2> &overflow = Q.overflow;
        std::vector<uint32_t> &stream = Q.stream;
        std::vector<int32_t> &dev_gather = Q.dev_gather;
        int32_t &max_local = Q.max_local, &max_pal = Q.max_pal, &max_rounds = Q.max_rounds;
        uint32_t &max_data = Q.max_data;
        std::vector<PackRound> prog;
        for (int64_t pk = pk_begin; pk < pk_end; ++pk) {
            const std::vector<int32_t> &members = packs[(size_t)pk];
            sbk::TileDesc td{};
            td.run_overflow = (int32_t)overflow.size();
            std::vector<int32_t> base(members.size());
            int32_t lstart = 0, n_runs = 0;
            int64_t n_dist = 0, n_cons = 0;
            for (size_t m = 0; m < members.size(); ++m) {
                const int32_t ci = members[m];
                const sbp::Tile &T = G.tiles[LT.tile_ids[ci]];
                base[m] = lstart;
                if (tl == 2) {
                    if (m == 0) td.gather_begin = (int32_t)dev_gather.size();
                    for (int32_t q = LT.gather_begin[ci]; q < LT.gather_begin[ci + 1]; ++q) dev_gather.push_back(LT.gather[q]);
                    lstart += LT.gather_begin[ci + 1] - LT.gather_begin[ci];
                }
                for (int32_t r = LT.run_begin[ci]; r < LT.run_begin[ci + 1]; ++r, ++n_runs) {
                    const sbp::Run &rn = LT.runs[r];
                    if (n_runs < sbk::kInlineRuns) td.runs[n_runs] = make_int2(rn.start, lstart);
                    else overflow.push_back(make_int2(rn.start, lstart));
                    lstart += rn.len;
                }
                if (lstart - base[m] != T.n_local) throw std::runtime_error("internal: tile run lengths do not add up");
                n_dist += T.d_end - T.d_begin;
                n_cons += (T.d_end - T.d_begin) + (T.q_end - T.q_begin);
            }
            td.n_local = lstart;
            td.run_count = n_runs;
            for (int32_t r = n_runs; r < sbk::kInlineRuns; ++r) td.runs[r] = make_int2(0, INT32_MAX);   // never selected
            if (lstart > sbk::kLargeTile) throw std::runtime_error("internal: packed tile too large");
            max_local = std::max(max_local, lstart);
            // the pack's program: zip the members' round lists (same type, at most 256 constraints per round)
            prog.clear();
            {
                std::vector<int32_t> next(members.size(), 0);
                std::vector<int64_t> dk(members.size()), qk(members.size());
                for (size_t m = 0; m < members.size(); ++m) {
                    const sbp::Tile &T = G.tiles[LT.tile_ids[members[m]]];
                    dk[m] = T.d_begin; qk[m] = T.q_begin;
                }
                for (;;) {       // group r of the pack = the members' next groups, as many as fit (<= 256 constraints per type)
                    PackRound R{{0, 0, 0}, {}};
                    for (size_t m = 0; m < members.size(); ++m) {
                        const sbp::Tile &T = G.tiles[LT.tile_ids[members[m]]];
                        if (next[m] >= T.n_rounds) continue;
                        const uint32_t w = G.rounds[T.round_begin + next[m]];
                        const int32_t c[3] = {(int32_t)(w & 1023u), (int32_t)((w >> 10) & 1023u), (int32_t)((w >> 20) & 1023u)};
                        if (R.cnt[0] + c[0] > sbp::kRoundThreads || R.cnt[1] + c[1] > sbp::kRoundThreads || R.cnt[2] + c[2] > sbp::kRoundThreads) continue;
                        R.parts.push_back({(int32_t)m, {c[0], c[1], c[2]}, dk[m], qk[m]});
                        dk[m] += c[0]; qk[m] += c[1] + c[2];
                        for (int t = 0; t < 3; ++t) R.cnt[t] += c[t];
                        ++next[m];
                    }
                    if (R.parts.empty()) break;
                    prog.push_back(std::move(R));
                }
                for (size_t m = 0; m < members.size(); ++m) {
                    const sbp::Tile &T = G.tiles[LT.tile_ids[members[m]]];
                    if (dk[m] != T.d_end || qk[m] != T.q_end) throw std::runtime_error("internal: tile stream does not match its rounds");
                }
            }
            td.n_rounds = (int32_t)prog.size();
            if (stream.size() > 0xfffffff0ull - 4ull * (size_t)n_cons - 48ull * prog.size() - 1024ull)      // (group word + up to 40 wave items per group)
                throw std::runtime_error("tile constraint stream exceeds 2^32 dwords");
            td.s_begin = (uint32_t)stream.size();
            const size_t s0 = stream.size();
            // dictionary-code the rest lengths of this tile's distance constraints when few values repeat
            std::vector<uint32_t> pal;
            bool compact = !no_palette && n_dist > 0;
            if (compact) {
                std::vector<uint32_t> vals;
                vals.reserve((size_t)n_dist);
                for (int32_t ci : members) {
                    const sbp::Tile &T = G.tiles[LT.tile_ids[ci]];
                    for (int64_t k = T.d_begin; k < T.d_end; ++k) vals.push_back(fbits(s->dist_rest[G.t_dist_id[k]]));
                }
                std::sort(vals.begin(), vals.end());
                vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
                if ((int)vals.size() <= sbk::kMaxPalette && td.n_local <= 4096) pal = vals; else compact = false;
            }
            for (const PackRound &R : prog)      // group word: counts per type, bit 30 = dictionary-coded distance slots
                stream.push_back((uint32_t)R.cnt[0] | ((uint32_t)R.cnt[1] << 10) | ((uint32_t)R.cnt[2] << 20) | (compact ? 1u << 30 : 0u));
            while ((stream.size() - s0) & 3) stream.push_back(0);
            if (stream.size() == s0) stream.insert(stream.end(), 4, 0u);   // empty program: keep 16 readable bytes
            td.n_pal = (int32_t)pal.size();
            for (uint32_t v : pal) stream.push_back(v);
            while ((stream.size() - s0) & 3) stream.push_back(0);
            max_pal = std::max(max_pal, (int32_t)pal.size());
            max_rounds = std::max(max_rounds, td.n_rounds);
            if (emit_items && !prog.empty()) {
                // wave items (kernels.hip.hpp kItem*): the work of every group dealt to the four waves of a tile, one dword per
                // wave and step. Slots of a group: its hinges (16 per wave slot), its tets (16), its springs (64); rows of
                // four slots, dealt boustrophedon (the wave that took a hinge slot in one row takes the cheapest of the next).
                std::vector<uint32_t> it[8];
                uint32_t off = 0;                       // dwords from the start of the tile's data
                bool fits = true;
                for (const PackRound &R : prog) {
                    const uint32_t nd = (uint32_t)R.cnt[0], nv = (uint32_t)R.cnt[1], nb = (uint32_t)R.cnt[2];
                    const uint32_t dsize = compact ? ((nd + 3u) & ~3u) : ((2u * nd + 3u) & ~3u), qoff = off + dsize;
                    // (a hinge takes a row of 16 lanes in the wave-items path, kernels.hip.hpp project_bending_row: 4 per wave slot)
                    const int n_wb = (int)((nb + 3) >> 2), n_wv = (int)((nv + 15) >> 4), n_wd = (int)((nd + 63) >> 6);
                    const int n_slots = n_wb + n_wv + n_wd, rows = std::max(1, (n_slots + item_waves - 1) / item_waves);
                    for (int row = 0; row < rows; ++row)
                        for (int wave = 0; wave < item_waves; ++wave) {
                            const int sw = row * item_waves + ((row & 1) ? item_waves - 1 - wave : wave);
                            uint32_t type = sbk::kItemIdle, cnt = 0, o = 0;
                            if (sw < n_wb) { type = sbk::kItemBending; cnt = std::min(4u, nb - 4u * (uint32_t)sw); o = qoff + 4u * (nv + 4u * (uint32_t)sw); }
                            else if (sw < n_wb + n_wv) { const uint32_t c0 = 16u * (uint32_t)(sw - n_wb); type = sbk::kItemVolume; cnt = std::min(16u, nv - c0); o = qoff + 4u * c0; }
                            else if (sw < n_slots) {
                                const uint32_t c0 = 64u * (uint32_t)(sw - n_wb - n_wv);
                                type = compact ? sbk::kItemDistCompact : sbk::kItemDistFull; cnt = std::min(64u, nd - c0); o = off + (compact ? c0 : 2u * c0);
                            }
                            if (o >= (1u << (32 - sbk::kItemOffsetShift))) fits = false;
                            it[wave].push_back(type | (cnt << sbk::kItemCountShift) | (row + 1 == rows ? 1u << sbk::kItemBarrierBit : 0u) |
                                               (o << sbk::kItemOffsetShift));
                        }
                    off += dsize + 4u * (nv + nb);
                }
                if (fits) {
                    td.n_steps = (int32_t)it[0].size();
                    td.s_items = (uint32_t)(stream.size() - s0);
                    for (int wave = 0; wave < item_waves; ++wave) stream.insert(stream.end(), it[wave].begin(), it[wave].end());
                    while ((stream.size() - s0) & 3) stream.push_back(0);
                }
            }
            td.s_hdr = (uint32_t)(stream.size() - s0);
            // (dictionary-coded tiles with a palette of at most 8; tiles with per-spring rest lengths where the kernels read float inverse
            // masses -- the WPAL = false instantiations carry the loads for that form)
            bool lane_pack = D.packed_lanes == sbk::kLanePackLanes && (compact ? (int)pal.size() <= sbk::kLanePackMaxPalette : (!s->w_palette && n_dist > 0)) &&
                             td.n_local <= sbk::kSmallTile && !prog.empty() && (int)prog.size() <= sbk::kLanePackRounds;
            for (const PackRound &R : prog) lane_pack = lane_pack && R.cnt[1] == 0 && R.cnt[2] == 0 && R.cnt[0] <= 2 * sbk::kLanePackLanes;
            if (lane_pack) {
                // one 16-byte word per lane: six 21-bit fields {i:9 | j:9 | palette:3}, field 2 r + u = slot lane + 128 u of round r
                std::vector<uint32_t> words(compact ? sbk::kLanePackDwordsCompact : sbk::kLanePackDwordsFull, 0u);
                for (size_t r = 0; r < prog.size(); ++r) {
                    int32_t c = 0;
                    for (const Part &pt : prog[r].parts) {
                        const uint32_t b = (uint32_t)base[pt.member], b2 = b | (b << 16);
                        for (int64_t k = pt.first_d; k < pt.first_d + pt.cnt[0]; ++k, ++c) {
                            const uint32_t idx = G.t_dist[k] + b2, rb = fbits(s->dist_rest[G.t_dist_id[k]]);
                            const uint32_t pi = compact ? (uint32_t)(std::lower_bound(pal.begin(), pal.end(), rb) - pal.begin()) : 0u;
                            const uint32_t i = idx & 0xffffu, j = idx >> 16;
                            if (i > 511u || j > 511u || pi > 7u) throw std::runtime_error("internal: lane-packed slot out of range");
                            const uint64_t f = (uint64_t)(i | (j << 9) | (pi << 18));
                            const int lane = c % sbk::kLanePackLanes, u = c / sbk::kLanePackLanes;
                            const int bit = sbk::kLanePackFieldBits * (2 * (int)r + u), w0 = bit >> 5, sh = bit & 31;
                            uint32_t *wd = &words[4 * (size_t)lane];
                            wd[w0] |= (uint32_t)(f << sh);
                            if (sh + sbk::kLanePackFieldBits > 32) wd[w0 + 1] |= (uint32_t)(f >> (32 - sh));
                            if (!compact) {      // the slot's rest length: fields 0..3 in the second 16-byte sweep, 4 and 5 in the 8-byte one
                                const int fld = 2 * (int)r + u;
                                if (fld < 4) words[4 * (size_t)sbk::kLanePackLanes + 4 * (size_t)lane + (size_t)fld] = rb;
                                else words[8 * (size_t)sbk::kLanePackLanes + 2 * (size_t)lane + (size_t)(fld - 4)] = rb;
                            }
                        }
                    }
                }
                stream.insert(stream.end(), words.begin(), words.end());
                td.packed_lanes = (uint32_t)sbk::kLanePackLanes;
                n_packed_tiles.fetch_add(1, std::memory_order_relaxed);
            }
            else for (const PackRound &R : prog) {
                // a group's data: its distance slots (padded to 4 dwords), then its volume slots, then its bending slots
                for (const Part &pt : R.parts) {
                    const uint32_t b = (uint32_t)base[pt.member], b2 = b | (b << 16);   // added to both 16-bit local indices
                    for (int64_t k = pt.first_d; k < pt.first_d + pt.cnt[0]; ++k) {
                        const uint32_t idx = G.t_dist[k] + b2, rb = fbits(s->dist_rest[G.t_dist_id[k]]);
                        if (compact) {
                            const uint32_t pi = (uint32_t)(std::lower_bound(pal.begin(), pal.end(), rb) - pal.begin());
                            stream.push_back((idx & 0xffffu) | ((idx >> 16) << 12) | (pi << 24));
                        } else {
                            stream.push_back(idx);
                            stream.push_back(rb);
                        }
                    }
                }
                while ((stream.size() - s0) & 3) stream.push_back(0);
                for (int t = 1; t < 3; ++t)
                    for (const Part &pt : R.parts) {
                        const uint32_t b = (uint32_t)base[pt.member], b2 = b | (b << 16);
                        const int64_t kb = pt.first_q + (t == 2 ? pt.cnt[1] : 0);      // a member's group lists its tets, then its hinges
                        for (int64_t k = kb; k < kb + pt.cnt[t]; ++k) {
                            if (G.t_quad_type[k] != t) throw std::runtime_error("internal: group layout");
                            Q.has_quads = true;
                            stream.push_back(G.t_quad[2 * k] + b2); stream.push_back(G.t_quad[2 * k + 1] + b2);
                            const int32_t id = G.t_quad_id[k];
                            if (t == 1) { volatile float r6 = 6.0f * s->vol_rest[id]; stream.push_back(fbits(r6)); stream.push_back(0); }
                            else { stream.push_back(fbits(s->bend_rest[2 * (size_t)id])); stream.push_back(fbits(s->bend_rest[2 * (size_t)id + 1])); }
                        }
                    }
            }
            td.s_len = (uint32_t)(stream.size() - s0);
            if (!td.packed_lanes) max_data = std::max(max_data, td.s_len - td.s_hdr);     // (lane-packed tiles never use the LDS window)
            tiles.push_back(td);
        }
        });
        // place the pieces: stream / overflow / gather offsets of a descriptor are relative to its piece until now
        std::vector<sbk::TileDesc> tiles;
        std::vector<int2> overflow;
        std::vector<uint32_t> stream;
        std::vector<int32_t> dev_gather;
        int32_t max_local = 0, max_pal = 0, max_rounds = 0;
        uint32_t max_data = 4;
        {
            size_t nt = 0, no = 0, ns = 0, ng = 0;
            for (const Piece &Q : pieces) { nt += Q.tiles.size(); no += Q.overflow.size(); ns += Q.stream.size(); ng += Q.dev_gather.size(); }
            if (ns > 0xfffffff0ull) throw std::runtime_error("tile constraint stream exceeds 2^32 dwords");
            tiles.reserve(nt); overflow.reserve(no); stream.reserve(ns); dev_gather.reserve(ng);
            for (Piece &Q : pieces) {
                for (sbk::TileDesc td : Q.tiles) {
                    td.s_begin += (uint32_t)stream.size();
                    td.run_overflow += (int32_t)overflow.size();
                    td.gather_begin += (int32_t)dev_gather.size();
                    tiles.push_back(td);
                }
                overflow.insert(overflow.end(), Q.overflow.begin(), Q.overflow.end());
                stream.insert(stream.end(), Q.stream.begin(), Q.stream.end());
                dev_gather.insert(dev_gather.end(), Q.dev_gather.begin(), Q.dev_gather.end());
                max_local = std::max(max_local, Q.max_local); max_pal = std::max(max_pal, Q.max_pal);
                max_rounds = std::max(max_rounds, Q.max_rounds); max_data = std::max(max_data, Q.max_data);
                D.has_quads |= Q.has_quads;
                Piece().tiles.swap(Q.tiles); std::vector<uint32_t>().swap(Q.stream);
            }
        }
        // Cost order inside a launch. Tiles of one launch share no particle, so their order is free; workgroups are dispatched
        // in index order, and a launch of a few hundred tiles puts the first 256 on a compute unit each and the rest beside
        // them. On an irregular mesh the launch lasts as long as its longest tile (40+ groups against a mean of 29): run the
        // long tiles first, so that none of them starts late or beside another long one. The position of the w-th heaviest
        // tile is the one workgroup w reads (the XCD remap of tile_kernel). Large launches (a lattice: equal tiles, placed
        // for L2 locality) and launches of equal tiles are left alone.
        if (!std::getenv("SB_NO_COST_ORDER")) {
            constexpr int32_t kCostOrderMaxTiles = 2048;
            const int32_t n = (int32_t)tiles.size();
            std::vector<std::pair<int32_t, int32_t>> ranges;
            if (tl == 2) ranges = s->t2_layer_range;
            else if (s->overlap_halo && D.n_boundary > 0 && D.n_boundary < n) {
                // (only the overlapped schedule launches the boundary and the interior tiles separately; a launch of the whole
                // tiling remaps with its own workgroup count, so the placement must be made for that launch)
                const int32_t cut = tl == 0 ? D.n_boundary : n - D.n_boundary;
                ranges = {{0, cut}, {cut, n}};
            } else ranges = {{0, n}};
            for (const auto &rg : ranges) {
                const int32_t nr = rg.second - rg.first;
                if (nr < 2 || nr > kCostOrderMaxTiles) continue;
                std::vector<int32_t> cost((size_t)nr), idx((size_t)nr);
                for (int32_t k = 0; k < nr; ++k) {
                    const sbk::TileDesc &td = tiles[(size_t)(rg.first + k)];
                    int32_t c = 0;
                    for (int32_t r = 0; r < td.n_rounds; ++r) {
                        const uint32_t w = stream[(size_t)td.s_begin + (size_t)r];
                        const int32_t nd = (int32_t)(w & 1023u), nv = (int32_t)((w >> 10) & 1023u), nb = (int32_t)((w >> 20) & 1023u);
                        if (D.has_quads) {       // rows of wave slots (as dealt for the wave items), a step with a hinge counts double
                            const int nw = std::max(1, (int)D.item_waves ? (int)D.item_waves : 4);
                            c += std::max(1, (((nd + 63) >> 6) + ((nv + 15) >> 4) + ((nb + 3) >> 2) + nw - 1) / nw) + (nb > 0 ? 1 : 0);
                        }
                        else c += std::max(1, (nd + sbk::kRoundSlots - 1) / sbk::kRoundSlots);
                    }
                    cost[(size_t)k] = c; idx[(size_t)k] = k;
                }
                const auto mm = std::minmax_element(cost.begin(), cost.end());
                if ((int64_t)*mm.second * 4 <= (int64_t)*mm.first * 5) continue;       // equal within 25 %
                std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b2) { return cost[(size_t)a] > cost[(size_t)b2]; });
                std::vector<sbk::TileDesc> placed((size_t)nr);
                const int32_t xq = nr >> 3, xr = nr & 7;
                for (int32_t w = 0; w < nr; ++w) {
                    const int32_t xcd = w & 7;
                    const int32_t pos = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (w >> 3);
                    placed[(size_t)pos] = tiles[(size_t)(rg.first + idx[(size_t)w])];
                }
                std::copy(placed.begin(), placed.end(), tiles.begin() + rg.first);
            }
        }
        D.n_tiles = (int32_t)tiles.size();
        D.n_packed_tiles = n_packed_tiles.load();
        D.max_local = std::max(max_local, 1);
        D.win_dwords = (int32_t)std::min<uint32_t>(max_data, 8192u);     // <= 32 KiB of LDS; >= one round (4 KiB)
        if (const char *e = std::getenv("SB_WIN_DWORDS")) D.win_dwords = std::max(1024, std::min(D.win_dwords, std::atoi(e)) & ~3);   // tuning experiments
        D.pal_dwords = (max_pal + 3) & ~3;
        D.rounds_dwords = std::min(sbk::kMaxRoundsLds, (max_rounds + 3) & ~3);
        D.lds_bytes = (size_t)D.max_local * sizeof(float4) + (size_t)D.rounds_dwords * 4 + (size_t)D.pal_dwords * 4 + (size_t)D.win_dwords * 4 + 16;
        D.n_slots = 0;
        for (size_t ci = 0; ci < LT.tile_ids.size(); ++ci) {
            const sbp::Tile &T = G.tiles[LT.tile_ids[ci]];
            D.n_slots += (T.d_end - T.d_begin) + (T.q_end - T.q_begin);
        }
        D.staged_particles = 0;
        for (const sbk::TileDesc &td : tiles) D.staged_particles += td.n_local;
        D.stream_bytes = (int64_t)stream.size() * 4;
        D.tiles.upload(tiles, s->dev_bytes); D.runs_overflow.upload(overflow, s->dev_bytes);
        D.stream.upload(stream, s->dev_bytes);
        D.gather.upload(dev_gather, s->dev_bytes);
    }
    for (const sbp::LocalGColour &LG : L.gcolours) {
        auto D = std::make_unique<DevGColour>();
        D->type = LG.type; D->count = (int32_t)LG.id.size();
        if (LG.type == 0) {
            std::vector<int2> ij(LG.id.size()); std::vector<float> rest(LG.id.size());
            for (size_t k = 0; k < LG.id.size(); ++k) { ij[k] = make_int2(LG.idx[2 * k], LG.idx[2 * k + 1]); rest[k] = s->dist_rest[LG.id[k]]; }
            D->ij.upload(ij, s->dev_bytes); D->rest.upload(rest, s->dev_bytes);
        } else {
            std::vector<int4> q(LG.id.size()); std::vector<float2> rest(LG.id.size());
            for (size_t k = 0; k < LG.id.size(); ++k) {
                q[k] = make_int4(LG.idx[4 * k], LG.idx[4 * k + 1], LG.idx[4 * k + 2], LG.idx[4 * k + 3]);
                const int32_t id = LG.id[k];
                if (LG.type == 1) { volatile float r6 = 6.0f * s->vol_rest[id]; rest[k] = make_float2(r6, 0.0f); }
                else rest[k] = make_float2(s->bend_rest[2 * (size_t)id], s->bend_rest[2 * (size_t)id + 1]);
            }
            D->quad.upload(q, s->dev_bytes); D->rest2.upload(rest, s->dev_bytes);
        }
        s->gcolours.push_back(std::move(D));
    }
    size_t max_send = 0, max_recv = 0;
    for (size_t slot = 0; slot < L.halo.size(); ++slot) {
        const sbp::HaloSlot &H = L.halo[slot];
        auto D = std::make_unique<DevHalo>();
        std::vector<int32_t> sidx, ridx;
        D->send_off.push_back(0); D->recv_off.push_back(0);
        for (int peer = 0; peer < L.world; ++peer) {
            if (H.send_idx[peer].empty() && H.recv_idx[peer].empty()) continue;
            D->peers.push_back(peer);
            sidx.insert(sidx.end(), H.send_idx[peer].begin(), H.send_idx[peer].end());
            ridx.insert(ridx.end(), H.recv_idx[peer].begin(), H.recv_idx[peer].end());
            D->send_off.push_back((int32_t)sidx.size()); D->recv_off.push_back((int32_t)ridx.size());
        }
        D->send_idx.upload(sidx, s->dev_bytes); D->recv_idx.upload(ridx, s->dev_bytes);
        const size_t fl = slot == 1 ? 6 : 3;   // floats per ghost (slot 1 also carries previous positions)
        max_send = std::max(max_send, sidx.size() * fl); max_recv = std::max(max_recv, ridx.size() * fl);
        s->halos.push_back(std::move(D));
    }
    s->d_sendbuf.alloc(max_send, s->dev_bytes);
    s->d_recvbuf.alloc(max_recv, s->dev_bytes);
    if (max_recv) HIP_CHECK(hipMemset(s->d_recvbuf.p, 0, max_recv * sizeof(float)));
    // Fused unpack. Ownership is contiguous in the planner's numbering and a rank's ghosts are numbered in that order, so when the
    // exchange before the T1 kernels is the plan's ONLY exchange (lattice-type plans: no T2 layers, no global colours) its
    // receive buffer -- peers in rank order, each peer's ghosts in its own order -- IS the ghost range [n_owned, n_local) in
    // order. The T1 kernels then read ghost k at 6 k floats into the buffer (tile_kernel GHOSTS) and the unpack launch is dropped.
    s->fused_unpack = false;
    if (L.world > 1 && !s->peer.enabled && P.tiling && P.gcolours.empty() && P.t2_layers.empty() && !s->tiling[1].has_quads &&
        s->halos.size() > 1 && !std::getenv("SB_NO_FUSED_UNPACK")) {
        std::vector<int32_t> ridx;
        for (int peer = 0; peer < L.world; ++peer) ridx.insert(ridx.end(), L.halo[1].recv_idx[(size_t)peer].begin(), L.halo[1].recv_idx[(size_t)peer].end());
        bool identity = (int64_t)ridx.size() == s->n_local - s->n_owned && !ridx.empty();
        for (size_t k = 0; identity && k < ridx.size(); ++k) identity = ridx[k] == (int32_t)(s->n_owned + (int64_t)k);
        s->fused_unpack = identity;
    }
    if (s->peer.enabled && L.world > 1) {
        // the mailbox: header words, then one segment per (halo slot, sending rank) in slot order, ranks increasing
        auto &PS = s->peer;
        const int W = L.world;
        if (W > sbk::kMaxPeers + 1) throw std::runtime_error("peer transport: at most 9 ranks");
        PS.n_slots = (int)s->halos.size();
        PS.off_table = PS.slot_base(PS.n_slots, W);
        const size_t hdr_words = PS.off_table + (size_t)PS.n_slots * W;
        PS.data_off_words = (hdr_words + 63) & ~(size_t)63;
        std::vector<uint32_t> header(PS.data_off_words, 0u);
        header[0] = (uint32_t)s->plan_hash; header[1] = (uint32_t)(s->plan_hash >> 32);      // compared by the neighbours (peer_link)
        header[2] = s->sharded ? 1u : 0u;
        for (int r = 0; r < W; ++r) { const uint64_t ph = L.pair_hash[(size_t)r]; header[4 + 2 * (size_t)r] = (uint32_t)ph; header[5 + 2 * (size_t)r] = (uint32_t)(ph >> 32); }
        PS.my_off.assign((size_t)PS.n_slots, std::vector<uint32_t>((size_t)W, 0u));
        size_t words = PS.data_off_words;
        for (int slot = 0; slot < PS.n_slots; ++slot) {
            const DevHalo &D = *s->halos[(size_t)slot];
            const size_t fl = slot == 1 ? 6 : 3;
            for (size_t k = 0; k < D.peers.size(); ++k) {          // one 16-byte aligned segment per sending neighbour
                PS.my_off[(size_t)slot][(size_t)D.peers[k]] = (uint32_t)words;
                header[PS.off_table + (size_t)slot * W + (size_t)D.peers[k]] = (uint32_t)words;
                words += 2 * ((fl * (size_t)(D.recv_off[k + 1] - D.recv_off[k]) + 3) & ~(size_t)3);      // two buffers, used alternately
            }
            words = (words + 63) & ~(size_t)63;
        }
        PS.bytes = words * 4;
        void *mb = nullptr;
        // (SB_PEER_COARSE: an ordinary cached allocation -- timing experiments on ONE device only; between devices the flags and
        // segments must be uncached for the stores of one agent to reach the loads of another without cache maintenance)
        if (!std::getenv("SB_PEER_COARSE") && hipExtMallocWithFlags(&mb, PS.bytes, hipDeviceMallocFinegrained) == hipSuccess) PS.fine_grained = true;
        else { (void)hipGetLastError(); HIP_CHECK(hipMalloc(&mb, PS.bytes)); }
        PS.mailbox = (uint32_t *)mb;
        s->dev_bytes += (int64_t)PS.bytes;
        HIP_CHECK(hipMemset(PS.mailbox, 0, PS.bytes));
        HIP_CHECK(hipMemcpy(PS.mailbox, header.data(), header.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc((void **)&PS.local, (size_t)PS.n_slots * 8 * sizeof(uint32_t)));
        HIP_CHECK(hipMemset(PS.local, 0, (size_t)PS.n_slots * 8 * sizeof(uint32_t)));
        HIP_CHECK(hipHostMalloc((void **)&PS.h_error, sizeof(uint32_t), hipHostMallocMapped));
        *PS.h_error = 0;
        PS.remote.assign((size_t)W, nullptr);
        PS.opened.assign((size_t)W, 0);
        PS.remote[(size_t)L.rank] = PS.mailbox;
    }
}

// Peer transport: once every sending / receiving neighbour's mailbox is mapped, fill the per-slot tables the kernels take.
void peer_link(sb_solver *s) {
    auto &PS = s->peer;
    const int W = s->desc.world, me = s->desc.rank;
    PS.slots.assign((size_t)PS.n_slots, sbk::PeerSlot{});
    for (int slot = 0; slot < PS.n_slots; ++slot) {
        const DevHalo &D = *s->halos[(size_t)slot];
        sbk::PeerSlot &P = PS.slots[(size_t)slot];
        const size_t base = PS.slot_base(slot, W);
        const size_t fl = slot == 1 ? 6 : 3;
        P.local = PS.local + (size_t)slot * 8;
        P.error = PS.h_error;         // (pinned host memory is device-accessible at the same address)
        for (size_t k = 0; k < D.peers.size(); ++k) {
            const int r = s->loopback ? me : D.peers[k];
            int cs = D.send_off[k + 1] - D.send_off[k], cr = D.recv_off[k + 1] - D.recv_off[k];
            if (s->loopback && (cs == 0 || cr == 0)) cs = cr = 0;       // a self-exchange needs both directions
            uint32_t *rm = PS.remote[(size_t)r];
            if ((cs || cr) && !rm) throw HipError(SB_ERR_STATE, "peer transport: the mailbox of rank " + std::to_string(r) + " is not connected (sb_peer_connect)");
            if ((cs || cr) && !s->loopback) {       // ranks plan independently: the neighbour must have arrived at a matching plan
                std::vector<uint32_t> hw(4 + 2 * (size_t)W, 0u);
                HIP_CHECK(hipMemcpy(hw.data(), rm, hw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                const uint64_t their_plan = (uint64_t)hw[0] | ((uint64_t)hw[1] << 32);
                const uint64_t their_pair = (uint64_t)hw[4 + 2 * (size_t)me] | ((uint64_t)hw[5 + 2 * (size_t)me] << 32);
                const uint64_t my_pair = s->plan->local.pair_hash[(size_t)r];
                char msg[320];
                if (!s->sharded && !hw[2] && their_plan != s->plan_hash) {
                    std::snprintf(msg, sizeof msg, "peer transport: rank %d planned a different schedule than rank %d (plan hash %016llx vs %016llx): every rank "
                                  "must pass the same mesh, tile_particles, partition and plan_flags", r, me, (unsigned long long)their_plan, (unsigned long long)s->plan_hash);
                    throw HipError(SB_ERR_STATE, msg);
                }
                if (their_pair != my_pair) {
                    std::snprintf(msg, sizeof msg, "peer transport: ranks %d and %d disagree on what they share (ghost lists / programs of the tiles both run: pair hash "
                                  "%016llx vs %016llx)", me, r, (unsigned long long)my_pair, (unsigned long long)their_pair);
                    throw HipError(SB_ERR_STATE, msg);
                }
            }
            if (cs) {
                if (P.n_send >= sbk::kMaxPeers) throw std::runtime_error("peer transport: too many neighbours");
                // where my segment starts inside the peer's mailbox: the peer's own offset table says
                uint32_t off = 0;
                if (s->loopback) off = PS.my_off[(size_t)slot][(size_t)D.peers[k]];
                else HIP_CHECK(hipMemcpy(&off, rm + PS.off_table + (size_t)slot * W + (size_t)me, 4, hipMemcpyDeviceToHost));
                if (off == 0) throw std::runtime_error("peer transport: a neighbour's mailbox has no segment for this rank");
                P.send_off[P.n_send] = D.send_off[k];
                P.send_cap[P.n_send] = s->loopback ? std::min(cs, cr) : cs;
                P.remote_data[P.n_send] = reinterpret_cast<float *>(rm + off);
                P.remote_stride[P.n_send] = (int32_t)((fl * (size_t)(s->loopback ? cr : cs) + 3) & ~(size_t)3);     // = the receiver's segment size
                P.remote_data_flag[P.n_send] = rm + base + (size_t)(s->loopback ? D.peers[k] : me);
                P.my_ack_flag[P.n_send] = PS.mailbox + base + (size_t)W + (size_t)D.peers[k];
                ++P.n_send;
                P.send_off[P.n_send] = D.send_off[k + 1];
            }
            if (cr) {
                P.recv_off[P.n_recv] = D.recv_off[k];
                P.recv_off[P.n_recv + 1] = D.recv_off[k + 1];
                P.recv_cnt[P.n_recv] = cr;
                P.my_data[P.n_recv] = reinterpret_cast<const float *>(PS.mailbox + PS.my_off[(size_t)slot][(size_t)D.peers[k]]);
                P.my_stride[P.n_recv] = (int32_t)((fl * (size_t)cr + 3) & ~(size_t)3);
                P.my_data_flag[P.n_recv] = PS.mailbox + base + (size_t)D.peers[k];
                P.remote_ack_flag[P.n_recv] = rm + base + (size_t)W + (size_t)(s->loopback ? D.peers[k] : me);
                ++P.n_recv;
            }
        }
        for (int q = P.n_send + 1; q <= sbk::kMaxPeers; ++q) P.send_off[q] = INT32_MAX;
        for (int q = P.n_recv + 1; q <= sbk::kMaxPeers; ++q) P.recv_off[q] = INT32_MAX;
    }
    PS.linked = true;
}

// Ghost refresh for one halo slot. Buffers hold every peer's particles back to back (3 floats each, slot 1: 6 with the
// previous position), so each peer gets exactly one message per direction.
void halo_exchange(sb_solver *s, int slot, hipStream_t st = nullptr) {
    if (!st) st = s->stream;
    if (slot < 0 || slot >= (int)s->halos.size()) return;
    DevHalo &D = *s->halos[slot];
    if (!D.active()) return;
    const bool with_prev = slot == 1;
    const int ns = D.send_off.back(), nr = D.recv_off.back();
    if (s->peer.enabled) {
        if (!s->peer.linked) peer_link(s);
        const sbk::PeerSlot &P = s->peer.slots[(size_t)slot];
        const int push_chunks = ns, unpack_chunks = nr;       // one lane per ghost (send_idx / recv_idx are indexed by the absolute position)
        // two launches per exchange, both always (the push also carries the waits, the unpack advances the slot's epoch);
        // at most kPeerGrid workgroups each: they end with an atomic on one word
        constexpr int kPeerGrid = 128;
        if (with_prev) {
            hipLaunchKernelGGL(sbk::peer_push_kernel<true>, dim3(std::min(kPeerGrid, std::max(1, (push_chunks + 255) / 256))), dim3(256), 0, st, s->pos_view(), s->d_prev.p, D.send_idx.p, push_chunks, P);
            hipLaunchKernelGGL(sbk::peer_unpack_kernel<true>, dim3(std::min(kPeerGrid, std::max(1, (unpack_chunks + 255) / 256))), dim3(256), 0, st, s->pos_view(), s->d_prev.p, D.recv_idx.p, unpack_chunks, P);
        } else {
            hipLaunchKernelGGL(sbk::peer_push_kernel<false>, dim3(std::min(kPeerGrid, std::max(1, (push_chunks + 255) / 256))), dim3(256), 0, st, s->pos_view(), s->d_prev.p, D.send_idx.p, push_chunks, P);
            hipLaunchKernelGGL(sbk::peer_unpack_kernel<false>, dim3(std::min(kPeerGrid, std::max(1, (unpack_chunks + 255) / 256))), dim3(256), 0, st, s->pos_view(), s->d_prev.p, D.recv_idx.p, unpack_chunks, P);
        }
        return;
    }
    if (!s->comm) throw HipError(SB_ERR_STATE, "world > 1 needs sb_comm_init before sb_finalize");
    if (ns) {
        if (with_prev)
            hipLaunchKernelGGL(sbk::halo_pack_kernel<true>, dim3((ns + 255) / 256), dim3(256), 0, st, s->pos_view(),
                               s->d_prev.p, D.send_idx.p, s->d_sendbuf.p, ns);
        else
            hipLaunchKernelGGL(sbk::halo_pack_kernel<false>, dim3((ns + 255) / 256), dim3(256), 0, st, s->pos_view(),
                               s->d_prev.p, D.send_idx.p, s->d_sendbuf.p, ns);
    }
    NCCL_CHECK(rccl().GroupStart());
    try {
        for (size_t k = 0; k < D.peers.size(); ++k) {
            int cs = D.send_off[k + 1] - D.send_off[k], cr = D.recv_off[k + 1] - D.recv_off[k];
            if (s->loopback) cs = cr = std::min(cs, cr);   // a self-exchange must post equal sizes (real peers always do)
            const size_t fl = with_prev ? 6 : 3;   // floats per ghost; one message per peer and direction
            if (cs) NCCL_CHECK(rccl().Send(s->d_sendbuf.p + fl * D.send_off[k], fl * (size_t)cs, ncclFloat, s->loopback ? 0 : D.peers[k], s->comm, st));
            if (cr) NCCL_CHECK(rccl().Recv(s->d_recvbuf.p + fl * D.recv_off[k], fl * (size_t)cr, ncclFloat, s->loopback ? 0 : D.peers[k], s->comm, st));
        }
    } catch (...) {
        (void)rccl().GroupEnd();      // never leave the group open behind an error
        throw;
    }
    NCCL_CHECK(rccl().GroupEnd());
    if (nr && !(with_prev && s->fused_unpack)) {
        if (with_prev)
            hipLaunchKernelGGL(sbk::halo_unpack_kernel<true>, dim3((nr + 255) / 256), dim3(256), 0, st, s->pos_view(),
                               s->d_prev.p, D.recv_idx.p, s->d_recvbuf.p, nr);
        else
            hipLaunchKernelGGL(sbk::halo_unpack_kernel<false>, dim3((nr + 255) / 256), dim3(256), 0, st, s->pos_view(),
                               s->d_prev.p, D.recv_idx.p, s->d_recvbuf.p, nr);
    }
}

// `table` (KIND 4 only): a descriptor table of its own -- copies of some of D's descriptors -- instead of D's tiles [tile_begin, tile_end)
template <int KIND>
void launch_tile(sb_solver *s, DevTiling &D, int tile_begin = 0, int tile_end = -1, int halo = sbk::kHaloNone, const sbk::TileDesc *table = nullptr) {
    const bool ghosts = halo == sbk::kHaloGhosts;
    if (tile_end < 0) tile_end = D.n_tiles;
    if (tile_end <= tile_begin) return;
    sbk::TileArgs A{};
    A.pos = s->pos_view(); A.w8 = s->d_w8.p; A.wpal = s->d_wpal.p; A.prev = s->d_prev.p; A.vel = s->d_vel.p;
    A.tiles = D.tiles.p; A.runs_overflow = D.runs_overflow.p; A.stream = D.stream.p;
    A.tp = s->d_tp.p;
    A.gather = D.gather.p;
    A.w_uniform = s->w_uniform ? 1 : 0;
    A.item_waves = D.item_waves;
    A.store_through = tile_end - tile_begin <= s->store_through_max_tiles ? 3 : s->store_through_large;
    A.ghost_src = ghosts ? s->d_recvbuf.p : nullptr; A.n_owned = (int32_t)s->n_owned;
    A.peek_out = KIND == 4 ? s->d_peek.p : nullptr;
    A.kin_map = KIND == 5 ? s->d_kin_map.p : nullptr; A.kin_target = KIND == 5 ? s->d_kin_target.p : nullptr;
    A.max_local = D.max_local; A.win_dwords = D.win_dwords; A.tile_base = tile_begin; A.pal_dwords = D.pal_dwords; A.rounds_dwords = D.rounds_dwords;
    const sbk::TileDesc *tiles_at_base = table ? table : D.tiles.p + tile_begin;      // the two preloaded kernel arguments (tile_kernel)
    const int n_wg = tile_end - tile_begin;
    const bool small = D.max_local <= sbk::kSmallTile;   // every tile <= 512 particles
    // narrow (2-wave) workgroups once the launch oversubscribes the chip; wide ones while every tile is resident at once
    // (a tiling with lane-packed slots runs 128-lane workgroups in EVERY launch, also the peek's subset of its tiles)
    const bool narrow = small && (D.packed_lanes ? true : (s->tile_lanes ? s->tile_lanes == sbk::kNarrowTileThreads : tile_end - tile_begin >= s->narrow_min_tiles));
    // tiles with tets / hinges: optionally 8 waves, so that a group's wave slots (16 four-lane constraints or 64 springs each) fit one row
    const bool quad8 = D.has_quads && s->quad_lanes == sbk::kQuadTileThreads;
    // spring-only small tiles on 8 waves (one particle per lane in the load / MARK / store phases; the rounds use half the lanes) while
    // every workgroup of the launch is resident even at that width (4 per compute unit): 64^3 0.1244 -> 0.1213, 48^3 0.1027 -> 0.1005 ms
    // per tick; 96^3 (1 728 tiles) 0.206 -> 0.230, so only launches of at most kWide8MaxTiles (profiles/r03o_lanes512_small_cubes.txt)
    constexpr int kWide8MaxTiles = 768;
    const bool wide8 = small && !D.has_quads && !D.packed_lanes && (s->tile_lanes ? s->tile_lanes == 512 : tile_end - tile_begin <= kWide8MaxTiles);
    const dim3 grid(tile_end - tile_begin), block(quad8 || wide8 ? sbk::kQuadTileThreads : (narrow ? sbk::kNarrowTileThreads : sbk::kWideTileThreads));
#define SB_LAUNCH_TILE(Q, W, G)                                                                                               \
    do {                                                                                                                      \
        if (Q && quad8) {                                                                                                     \
            if (small) hipLaunchKernelGGL((sbk::tile_kernel<KIND, true, sbk::kQuadTileThreads, sbk::kSmallTile / sbk::kQuadTileThreads, W, sbk::kHaloNone>), \
                                          grid, block, D.lds_bytes + s->lds_pad, s->stream, tiles_at_base, n_wg, A);                                           \
            else hipLaunchKernelGGL((sbk::tile_kernel<KIND, true, sbk::kQuadTileThreads, sbk::kLargeTile / sbk::kQuadTileThreads, W, sbk::kHaloNone>), \
                                    grid, block, D.lds_bytes + s->lds_pad, s->stream, tiles_at_base, n_wg, A);                                                 \
        } else if (wide8) hipLaunchKernelGGL((sbk::tile_kernel<KIND, false, sbk::kQuadTileThreads, sbk::kSmallTile / sbk::kQuadTileThreads, W, G>), \
                                       grid, block, D.lds_bytes + s->lds_pad, s->stream, tiles_at_base, n_wg, A);                                              \
        else if (narrow) hipLaunchKernelGGL((sbk::tile_kernel<KIND, Q, sbk::kNarrowTileThreads, sbk::kSmallTile / sbk::kNarrowTileThreads, W, G>), \
                                       grid, block, D.lds_bytes + s->lds_pad, s->stream, tiles_at_base, n_wg, A);                                              \
        else if (small) hipLaunchKernelGGL((sbk::tile_kernel<KIND, Q, sbk::kWideTileThreads, sbk::kSmallTile / sbk::kWideTileThreads, W, G>), \
                                           grid, block, D.lds_bytes + s->lds_pad, s->stream, tiles_at_base, n_wg, A);                                          \
        else hipLaunchKernelGGL((sbk::tile_kernel<KIND, Q, sbk::kWideTileThreads, sbk::kLargeTile / sbk::kWideTileThreads, W, G>), \
                                grid, block, D.lds_bytes + s->lds_pad, s->stream, tiles_at_base, n_wg, A);                                                     \
    } while (0)
    // the ghost-reading variant exists for the kernels that can meet ghosts behind a fused exchange: mid-tick and last kernels of
    // spring-only tilings (launch_tick_kernel decides; s->fused_unpack is never set for a tiling with tets / hinges)
    constexpr bool kCanGhost = KIND == 1 || KIND == 2;
    if (ghosts && !(kCanGhost && !D.has_quads)) throw std::runtime_error("internal: ghost-reading tile kernel requested for a launch that has none");
    if (kCanGhost && ghosts) {
        if (s->w_palette) SB_LAUNCH_TILE(false, true, (kCanGhost ? sbk::kHaloGhosts : sbk::kHaloNone)); else SB_LAUNCH_TILE(false, false, (kCanGhost ? sbk::kHaloGhosts : sbk::kHaloNone));
    } else
    if (s->w_palette) { if (D.has_quads) SB_LAUNCH_TILE(true, true, sbk::kHaloNone); else SB_LAUNCH_TILE(false, true, sbk::kHaloNone); }
    else { if (D.has_quads) SB_LAUNCH_TILE(true, false, sbk::kHaloNone); else SB_LAUNCH_TILE(false, false, sbk::kHaloNone); }
#undef SB_LAUNCH_TILE
}

struct LaunchTimer {            // optional HIP-event pair around every launch of one tick (sb_step_profiled)
    std::vector<hipEvent_t> ev;
    std::vector<int> slot;      // see sb_step_profiled in softbody.h
    hipStream_t stream;
    void begin(int which) {
        hipEvent_t a, b;
        HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        ev.push_back(a); ev.push_back(b); slot.push_back(which);
        HIP_CHECK(hipEventRecord(a, stream));
    }
    void end() { HIP_CHECK(hipEventRecord(ev.back(), stream)); }
    ~LaunchTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
};

// Launch the tile kernel K_it of a tick of `substeps` substeps (no halo).
void launch_tick_kernel(sb_solver *s, int it, int substeps, LaunchTimer *lt, int tile_begin = 0, int tile_end = -1, bool kin = false) {
    const int tl = s->plan->plan.tiling ? (it & 1) : 0;
    DevTiling &D = s->tiling[tl];
    if (lt && D.n_tiles) lt->begin(it == 0 ? 2 + (int)s->gcolours.size() : (it == substeps ? 3 + (int)s->gcolours.size() : tl));
    const int halo_in = s->fused_unpack && tl == 1 ? sbk::kHaloGhosts : sbk::kHaloNone;      // T1 tiles read their ghosts straight from the receive buffer
    if (it == 0) launch_tile<0>(s, D, tile_begin, tile_end);
    else if (it < substeps && kin) launch_tile<5>(s, D, tile_begin, tile_end);        // (world == 1: the fused first kernel of a tick, with kinematic targets)
    else if (it < substeps) launch_tile<1>(s, D, tile_begin, tile_end, halo_in);
    else launch_tile<2>(s, D, tile_begin, tile_end, halo_in);
    if (lt && D.n_tiles) lt->end();
}

// The kernel of one T2 layer (constraints inside neither T0 nor T1, in LDS tiles of their own), after its ghost refresh.
void launch_t2_layer(sb_solver *s, int layer, LaunchTimer *lt, bool with_halo = true) {
    const auto rg = s->t2_layer_range[layer];
    if (with_halo) halo_exchange(s, 2 + (int)s->gcolours.size() + layer);
    if (rg.second <= rg.first) return;
    if (lt) lt->begin(4 + (int)s->gcolours.size());
    launch_tile<3>(s, s->tiling[2], rg.first, rg.second);
    if (lt) lt->end();
}

void launch_gcolour(sb_solver *s, int gc, LaunchTimer *lt) {
    DevGColour &G = *s->gcolours[gc];
    if (G.count == 0) return;
    if (lt) lt->begin(2 + gc);
    dim3 grid((G.count + 255) / 256);
    if (G.type == 0)
        hipLaunchKernelGGL(sbk::global_distance_kernel, grid, dim3(256), 0, s->stream, s->pos_view(), G.ij.p, G.rest.p, G.count,
                           s->d_tp.p);
    else
        hipLaunchKernelGGL(sbk::global_quad_kernel, grid, dim3(256), 0, s->stream, s->pos_view(), G.quad.p, G.rest2.p, G.count,
                           G.type, s->d_tp.p);
    if (lt) lt->end();
}

// One tick (SPEC.md §2/§3): kernel K_s runs on the tiles of tiling T_(s&1): the tile's rounds (end of substep
// s-1), collide + velocity update + integrate, the same rounds again (start of substep s).
// fused_first: the tick starts with an ordinary mid-tick kernel on T0 that also finishes the PREVIOUS tick (its
// deferred last kernel); defer_last: leave K_substeps to the next tick / to flush_deferred().
void enqueue_substeps(sb_solver *s, int substeps, LaunchTimer *lt = nullptr, bool fused_first = false, bool defer_last = false, bool kin = false) {
    const bool two = s->plan->plan.tiling;
    if (s->overlap_halo) {
        // Overlapped schedule (opt-in, SB_HALO_OVERLAP): a T0 kernel runs its boundary tiles first; the ghost exchange
        // for the following T1 kernel then travels on comm_stream beside the T0 interior tiles AND the T1 interior
        // tiles; the T1 tiles that hold a ghost or a sent particle run last, after the exchange. The interior tiles of
        // either tiling touch none of the particles the pack kernel reads or the unpack kernel writes (build_device).
        DevTiling &T0 = s->tiling[0], &T1 = s->tiling[1];
        const int t1_interior = T1.n_tiles - T1.n_boundary;
        for (int it = 0; it <= substeps; ++it) {
            if (it == substeps && defer_last) break;       // (lazy tick boundary, as in the serialised schedule below)
            // the first kernel of a tick that also finishes the previous one is an ordinary mid-tick kernel on T0: an even, interior step index
            const int kit = it == 0 && fused_first ? 2 : it, ks = it == 0 && fused_first ? 4 : substeps;
            if (it & 1) {
                launch_tick_kernel(s, kit, ks, lt, 0, t1_interior);
                HIP_CHECK(hipStreamWaitEvent(s->stream, s->ev_halo, 0));
                launch_tick_kernel(s, kit, ks, lt, t1_interior, T1.n_tiles);
            } else {
                launch_tick_kernel(s, kit, ks, lt, 0, T0.n_boundary);
                if (it < substeps) {
                    HIP_CHECK(hipEventRecord(s->ev_boundary, s->stream));
                    HIP_CHECK(hipStreamWaitEvent(s->comm_stream, s->ev_boundary, 0));
                    halo_exchange(s, 1, s->comm_stream);
                    HIP_CHECK(hipEventRecord(s->ev_halo, s->comm_stream));
                }
                launch_tick_kernel(s, kit, ks, lt, T0.n_boundary, T0.n_tiles);
            }
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    for (int it = 0; it <= substeps; ++it) {
        if (it == substeps && defer_last) break;
        if (two && (it & 1)) halo_exchange(s, 1);
        if (it == 0 && fused_first) launch_tick_kernel(s, 2, 4, lt, 0, -1, kin);   // an even, interior step index: KIND 1 (5 with kinematic targets) on T0
        else launch_tick_kernel(s, it, substeps, lt);
        if (it == substeps) break;
        for (size_t ly = 0; ly < s->t2_layer_range.size(); ++ly) launch_t2_layer(s, (int)ly, lt);
        for (size_t gc = 0; gc < s->gcolours.size(); ++gc) {
            halo_exchange(s, 2 + (int)gc);
            launch_gcolour(s, (int)gc, lt);
        }
    }
    HIP_CHECK(hipGetLastError());
}

// Launch the deferred last kernel of the previous tick (uses the tick parameters still on the device).
// Pending kinematic targets onto `dst` (the positions, or the peek's side array): scatter kernel over the pinned host table.
void scatter_kinematic(sb_solver *s, float *dst) {
    const int q = s->kin_pending, count = s->kin_pending_count;
    hipLaunchKernelGGL(sbk::kinematic_scatter_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s->stream, dst, s->h_kin_idx[q], s->h_kin_pos[q], count);
    HIP_CHECK(hipGetLastError());
}
// The table of the pending targets has been handed to its last reader: it may be reused once that kernel is done.
void retire_kinematic(sb_solver *s) {
    HIP_CHECK(hipEventRecord(s->ev_kin[s->kin_pending], s->stream));
    s->kin_pending = -1; s->kin_pending_count = 0;
}

void flush_deferred(sb_solver *s) {
    if (s->deferred) {
        const int S = s->deferred_substeps;
        s->deferred = false;
        if (s->plan->plan.tiling && (S & 1)) halo_exchange(s, 1);
        launch_tick_kernel(s, S, S, nullptr);
        HIP_CHECK(hipGetLastError());
    }
    if (s->kin_pending >= 0) {       // the tick they follow is complete: the targets take effect
        scatter_kinematic(s, s->d_pos3.p);
        retire_kinematic(s);
    }
}

// Everything a fused first kernel needs to apply the pending targets itself: the particle -> slot map (built once), the slot
// array (NaN = no target), and this tick's targets written into their slots on the solver's stream.
void stage_kinematic_for_fusion(sb_solver *s) {
    if (!s->d_kin_map.p) {
        const sbp::LocalPlan &L = s->plan->local;
        std::vector<int32_t> map((size_t)s->n_local, -1);
        int32_t n_pinned = 0;
        for (int64_t l = 0; l < s->n_local; ++l) if (s->invm[(size_t)L.local_to_old[(size_t)l]] == 0.0f) map[(size_t)l] = n_pinned++;
        s->d_kin_map.upload(map, s->dev_bytes);
        std::vector<float> nan((size_t)std::max(n_pinned, 1) * 3, std::numeric_limits<float>::quiet_NaN());
        s->d_kin_target.upload(nan, s->dev_bytes);
    }
    const int q = s->kin_pending, count = s->kin_pending_count;
    hipLaunchKernelGGL(sbk::kinematic_fill_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s->stream, s->d_kin_map.p, s->d_kin_target.p,
                       s->h_kin_idx[q], s->h_kin_pos[q], count);
    HIP_CHECK(hipGetLastError());
    retire_kinematic(s);
}

// ---- peek: tick-end positions without completing the tick ------------------------------------------------------------------------
// While the last kernel K_S of a tick is deferred, the positions the tick ends with are K_S's rounds + collide applied to the state
// in memory. tile_kernel<4> computes exactly that -- same tiles, same programs, same inputs, hence the same bits -- into d_peek and
// leaves the state alone, so the next sb_step still fuses K_S with its first kernel (one launch instead of two) and a render
// readback after every tick no longer costs a whole extra pass over the mesh. world == 1 only (a rank of a partitioned solver would
// need its ghosts refreshed first).
bool can_peek(const sb_solver *s) {
    if (!s->peek_enabled || !s->deferred || s->desc.world != 1) return false;
    const int tl = s->plan->plan.tiling ? (s->deferred_substeps & 1) : 0;
    return tl == 0 && s->tiling[0].n_tiles > 0 && s->tiling[0].n_tiles >= s->peek_min_tiles;
}

// The T0 device tiles (packs) that hold at least one particle of `wanted` (device numbering): copies of their descriptors.
void build_peek_subset(sb_solver *s, const std::vector<int32_t> &wanted) {
    DevTiling &D = s->tiling[0];
    std::vector<sbk::TileDesc> tiles((size_t)D.n_tiles);
    std::vector<int2> ovf(D.runs_overflow.count);
    if (!tiles.empty()) HIP_CHECK(hipMemcpy(tiles.data(), D.tiles.p, tiles.size() * sizeof(sbk::TileDesc), hipMemcpyDeviceToHost));
    if (!ovf.empty()) HIP_CHECK(hipMemcpy(ovf.data(), D.runs_overflow.p, ovf.size() * sizeof(int2), hipMemcpyDeviceToHost));
    std::vector<uint8_t> is_wanted((size_t)s->n_local, 0);
    for (int32_t g : wanted) is_wanted[(size_t)g] = 1;
    std::vector<sbk::TileDesc> keep;
    for (const sbk::TileDesc &td : tiles) {
        bool hit = false;
        for (int r = 0; r < td.run_count && !hit; ++r) {
            auto run = [&](int q) { return q < sbk::kInlineRuns ? td.runs[q] : ovf[(size_t)(td.run_overflow + q - sbk::kInlineRuns)]; };
            const int2 a = run(r);
            const int end = r + 1 < td.run_count ? run(r + 1).y : td.n_local;
            for (int l = a.y; l < end && !hit; ++l) hit = is_wanted[(size_t)(a.x + (l - a.y))] != 0;
        }
        if (hit) keep.push_back(td);
    }
    s->peek_tiles.upload(keep, s->dev_bytes);
    s->n_peek_tiles = (int32_t)keep.size();
}

// Enqueue the peek on the solver's stream; afterwards d_peek holds the tick-end positions of every particle of the peeked tiles.
// subset = only the tiles of build_peek_subset (render-set readback), else every T0 tile.
void peek_positions(sb_solver *s, bool subset) {
    DevTiling &D = s->tiling[0];
    if (!s->d_peek.p) s->d_peek.alloc((size_t)s->n_local * 3, s->dev_bytes);
    if (subset) {
        if (s->n_peek_tiles > 0) launch_tile<4>(s, D, 0, s->n_peek_tiles, sbk::kHaloNone, s->peek_tiles.p);
    } else launch_tile<4>(s, D);
    HIP_CHECK(hipGetLastError());
    ++s->n_peeks;
}

void upload_tick_params(sb_solver *s, float dt, int substeps) {
    sbk::TickParams tp = tick_params(s, dt, substeps);
    if (!s->tp_valid || std::memcmp(&tp, &s->tp_host, sizeof(tp)) != 0) {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        HIP_CHECK(hipMemcpyAsync(s->d_tp.p, &tp, sizeof(tp), hipMemcpyHostToDevice, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        s->tp_host = tp; s->tp_valid = true;
    }
}

// Peer transport: a wait that gave up (a neighbour never delivered / never acknowledged) must not pass silently.
void check_peer_error(sb_solver *s) {
    if (!s->peer.enabled || !s->peer.h_error) return;
    const uint32_t flag = *reinterpret_cast<volatile uint32_t *>(s->peer.h_error);     // a host load: cheap enough for every sb_step
    if (flag) throw HipError(SB_ERR_RCCL, "peer transport: a halo wait gave up (a neighbour never delivered or never acknowledged)");
}

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

int set_device(const sb_solver *s) {
    hipError_t e = hipSetDevice(s->desc.device);
    if (e != hipSuccess) return fail(SB_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return SB_OK;
}

}  // namespace

extern "C" {

const char *sb_last_error(void) { return g_err.c_str(); }
int sb_abi_version(void) { return SB_ABI_VERSION; }

int sb_runtime_info(sb_runtime_info_t *out) {
    if (!out) return fail(SB_ERR_INVALID_ARG, "sb_runtime_info: null argument");
    std::memset(out, 0, sizeof(*out));
    return guarded([&]() -> int {
        out->hip_runtime_version = hip_runtime_version();
        int drv = 0;
        if (hipDriverGetVersion(&drv) == hipSuccess) out->hip_driver_version = drv;
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(&hipRuntimeGetVersion), &di) && di.dli_fname) std::snprintf(out->hip_library, sizeof out->hip_library, "%s", di.dli_fname);
        out->rccl_header_version = NCCL_VERSION_CODE;
        RcclApi &R = rccl(false);
        if (R.ok()) {
            out->rccl_version = R.version;
            out->rccl_was_resident = R.was_resident ? 1 : 0;
            std::snprintf(out->rccl_library, sizeof out->rccl_library, "%s", R.path.c_str());
            out->capture_serial_ok = R.version >= 22606 ? 1 : 0;
            out->capture_overlap_ok = (R.version >= 22606 && capture_overlap_ok()) ? 1 : 0;
        }
        return SB_OK;
    });
}

void sb_desc_default(sb_desc *d) {
    if (!d) return;
    std::memset(d, 0, sizeof(*d));
    d->world = 1;
    d->gravity[1] = -9.81f;
    d->tile_particles = 0;         // automatic: 512, or 256 when the mesh has 4-vertex constraints (sb_finalize)
    d->use_graph = 1;
}

int sb_create(const sb_desc *desc, sb_solver **out) {
    if (!desc || !out) return fail(SB_ERR_INVALID_ARG, "sb_create: null argument");
    *out = nullptr;
    return guarded([&]() -> int {
        sb_desc d = *desc;
        if (d.world <= 0) d.world = 1;
        if (d.rank < 0 || d.rank >= d.world) return fail(SB_ERR_INVALID_ARG, "sb_create: rank out of range");
        if (!(d.damping >= 0.0f)) return fail(SB_ERR_INVALID_ARG, "sb_create: damping must be >= 0");
        if (d.tile_particles > sbp::kMaxTileLocal) return fail(SB_ERR_INVALID_ARG, "sb_create: tile_particles too large");
        if (d.partition < SB_PARTITION_AUTO || d.partition > SB_PARTITION_RCB) return fail(SB_ERR_INVALID_ARG, "sb_create: partition must be SB_PARTITION_AUTO, _BLOCKS or _RCB");
        if (d.plan_flags & ~kPlanFlagsAll) return fail(SB_ERR_INVALID_ARG, "sb_create: unknown bit in plan_flags");
        if (d.halo_transport != SB_TRANSPORT_RCCL && d.halo_transport != SB_TRANSPORT_PEER) return fail(SB_ERR_INVALID_ARG, "sb_create: halo_transport must be SB_TRANSPORT_RCCL or _PEER");
        if (d.halo_schedule < SB_SCHEDULE_AUTO || d.halo_schedule > SB_SCHEDULE_OVERLAP_GRAPH) return fail(SB_ERR_INVALID_ARG, "sb_create: halo_schedule must be one of SB_SCHEDULE_*");
        if (d.debug_flags & ~(SB_DEBUG_NO_COMM | SB_DEBUG_LOOPBACK)) return fail(SB_ERR_INVALID_ARG, "sb_create: unknown bit in debug_flags");
        if (d.reserved[0] || d.reserved[1] || d.reserved[2]) return fail(SB_ERR_INVALID_ARG, "sb_create: reserved fields must be 0 (zero-initialise sb_desc or call sb_desc_default)");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            return fail(SB_ERR_NO_DEVICE, "sb_create: no HIP device (this plugin has no CPU path)");
        if (d.device < 0 || d.device >= ndev) return fail(SB_ERR_NO_DEVICE, "sb_create: device ordinal out of range");
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, d.device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SB_ERR_NO_DEVICE, std::string("sb_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        HIP_CHECK(hipSetDevice(d.device));
        auto s = std::make_unique<sb_solver>();
        s->desc = d;
        s->lazy_tick = !std::getenv("SB_NO_LAZY_TICK");
        s->peer.enabled = d.world > 1 && d.halo_transport == SB_TRANSPORT_PEER;
        s->loopback = d.world > 1 && (d.debug_flags & SB_DEBUG_LOOPBACK) != 0;
        s->pack_tiles = !std::getenv("SB_NO_PACK");
        s->peek_enabled = !std::getenv("SB_NO_PEEK");
        s->kin_fuse = !std::getenv("SB_NO_KIN_FUSE");
        if (const char *e = std::getenv("SB_PEEK_MIN_TILES")) s->peek_min_tiles = std::max(0, std::atoi(e));
        if (const char *e = std::getenv("SB_LDS_PAD")) s->lds_pad = (size_t)std::max(0, std::atoi(e));
        if (const char *e = std::getenv("SB_TILE_LANES")) s->tile_lanes = std::atoi(e) == 128 ? 128 : (std::atoi(e) == 256 ? 256 : (std::atoi(e) == 512 ? 512 : 0));
        if (const char *e = std::getenv("SB_QUAD_LANES")) s->quad_lanes = std::atoi(e) == 256 ? 256 : 512;
        if (const char *e = std::getenv("SB_STORE_THROUGH_MAX_TILES")) s->store_through_max_tiles = std::max(0, std::atoi(e));
        if (const char *e = std::getenv("SB_STORE_THROUGH_LARGE")) s->store_through_large = std::atoi(e) & 3;
        if (const char *e = std::getenv("SB_NARROW_MIN_TILES")) s->narrow_min_tiles = std::max(1, std::atoi(e));
        HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreate(&s->ev0));
        HIP_CHECK(hipEventCreate(&s->ev1));
        *out = s.release();
        return SB_OK;
    });
}

int sb_destroy(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_destroy: null handle");
    (void)hipSetDevice(s->desc.device);
    delete s;          // (the destructor drains the solver's streams first)
    return SB_OK;
}

int sb_set_particles(sb_solver *s, const float *pos, const float *vel, const float *inv_mass, int32_t n) {
    if (!s || !pos || !inv_mass || n <= 0) return fail(SB_ERR_INVALID_ARG, "sb_set_particles: bad argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_particles after sb_finalize");
    return guarded([&]() -> int {
        for (int32_t p = 0; p < n; ++p) if (!(inv_mass[p] >= 0.0f)) return fail(SB_ERR_INVALID_ARG, "sb_set_particles: inverse mass must be >= 0");
        s->n = n;
        s->pos.assign(pos, pos + 3 * (size_t)n);
        if (vel) s->vel.assign(vel, vel + 3 * (size_t)n); else s->vel.assign(3 * (size_t)n, 0.0f);
        s->invm.assign(inv_mass, inv_mass + n);
        return SB_OK;
    });
}

int sb_set_rest_positions(sb_solver *s, const float *rest, int32_t n) {
    if (!s || !rest) return fail(SB_ERR_INVALID_ARG, "sb_set_rest_positions: bad argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_rest_positions after sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_rest_positions: n differs from sb_set_particles");
    return guarded([&]() -> int { s->rest.assign(rest, rest + 3 * (size_t)n); return SB_OK; });
}

static int set_cons(sb_solver *s, const char *who, const int32_t *idx, const float *rest, int32_t m, float compliance,
                    int type, int nv, int nrest) {
    if (!s || m < 0 || (m > 0 && (!idx || !rest))) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": bad argument");
    if (s->finalized) return fail(SB_ERR_STATE, std::string(who) + " after sb_finalize");
    if (s->n <= 0) return fail(SB_ERR_STATE, std::string(who) + " before sb_set_particles");
    if (!(compliance >= 0.0f)) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": compliance must be >= 0");
    return guarded([&]() -> int {
        std::atomic<bool> bad{false};
        sbp::parallel_for_chunks((int64_t)m * nv, 1 << 22, [&](int64_t, int64_t kb, int64_t ke) {
            for (int64_t k = kb; k < ke; ++k) if (idx[k] < 0 || idx[k] >= s->n) { bad = true; return; }
        });
        if (bad) return fail(SB_ERR_INVALID_ARG, std::string(who) + ": particle index out of range");
        std::vector<int32_t> &I = type == 0 ? s->dist_ij : (type == 1 ? s->vol_ijkl : s->bend_ijkl);
        std::vector<float> &R = type == 0 ? s->dist_rest : (type == 1 ? s->vol_rest : s->bend_rest);
        I.assign(idx, idx + (size_t)m * nv);
        R.assign(rest, rest + (size_t)m * nrest);
        s->compliance[type] = compliance;
        return SB_OK;
    });
}

int sb_set_distance_constraints(sb_solver *s, const int32_t *ij, const float *rest_len, int32_t m, float compliance) {
    return set_cons(s, "sb_set_distance_constraints", ij, rest_len, m, compliance, 0, 2, 1);
}
int sb_set_volume_constraints(sb_solver *s, const int32_t *ijkl, const float *rest_vol, int32_t m, float compliance) {
    return set_cons(s, "sb_set_volume_constraints", ijkl, rest_vol, m, compliance, 1, 4, 1);
}
int sb_set_bending_constraints(sb_solver *s, const int32_t *ijkl, const float *rest_cs, int32_t m, float compliance) {
    return set_cons(s, "sb_set_bending_constraints", ijkl, rest_cs, m, compliance, 2, 4, 2);
}

int sb_set_ground_plane(sb_solver *s, float nx, float ny, float nz, float d, int32_t enabled) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_set_ground_plane: null handle");
    if (!(nx == nx) || !(ny == ny) || !(nz == nz) || !(d == d)) return fail(SB_ERR_INVALID_ARG, "sb_set_ground_plane: NaN");
    s->plane[0] = nx; s->plane[1] = ny; s->plane[2] = nz; s->plane[3] = d;
    s->plane_on = enabled ? 1 : 0;
    return SB_OK;   // picked up by the next sb_step (tick parameters are re-uploaded when they change)
}

int sb_comm_unique_id(uint8_t out_id[SB_UNIQUE_ID_BYTES]) {
    if (!out_id) return fail(SB_ERR_INVALID_ARG, "sb_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) <= SB_UNIQUE_ID_BYTES, "unique id size");
    return guarded([&]() -> int {
        ncclUniqueId id;
        NCCL_CHECK(rccl().GetUniqueId(&id));
        std::memset(out_id, 0, SB_UNIQUE_ID_BYTES);
        std::memcpy(out_id, &id, sizeof(id));
        return SB_OK;
    });
}

int sb_comm_init(sb_solver *s, const uint8_t id_bytes[SB_UNIQUE_ID_BYTES]) {
    if (!s || !id_bytes) return fail(SB_ERR_INVALID_ARG, "sb_comm_init: null");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_comm_init after sb_finalize");
    if (s->comm) return fail(SB_ERR_STATE, "sb_comm_init called twice");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        ncclUniqueId id;
        std::memcpy(&id, id_bytes, sizeof(id));
        if (s->loopback) {
            // pipeline test on one GPU (SB_DEBUG_LOOPBACK): a communicator of size 1, every peer replaced by this rank itself
            NCCL_CHECK(rccl().CommInitRank(&s->comm, 1, id, 0));
        } else {
            NCCL_CHECK(rccl().CommInitRank(&s->comm, s->desc.world, id, s->desc.rank));
        }
        return SB_OK;
    });
}

int sb_peer_mailbox_handle(sb_solver *s, uint8_t out_handle[SB_IPC_HANDLE_BYTES]) {
    if (!s || !out_handle) return fail(SB_ERR_INVALID_ARG, "sb_peer_mailbox_handle: null argument");
    if (!s->finalized || !s->peer.mailbox) return fail(SB_ERR_STATE, "sb_peer_mailbox_handle: needs a finalized world > 1 solver with SB_HALO_TRANSPORT=peer");
    static_assert(sizeof(hipIpcMemHandle_t) <= SB_IPC_HANDLE_BYTES, "ipc handle size");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        hipIpcMemHandle_t h;
        HIP_CHECK(hipIpcGetMemHandle(&h, s->peer.mailbox));
        std::memset(out_handle, 0, SB_IPC_HANDLE_BYTES);
        std::memcpy(out_handle, &h, sizeof(h));
        return SB_OK;
    });
}

int sb_peer_connect(sb_solver *s, int32_t rank, const uint8_t handle[SB_IPC_HANDLE_BYTES], sb_solver *same_process_peer) {
    if (!s || (!handle && !same_process_peer)) return fail(SB_ERR_INVALID_ARG, "sb_peer_connect: null argument");
    if (!s->finalized || !s->peer.mailbox) return fail(SB_ERR_STATE, "sb_peer_connect: needs a finalized world > 1 solver with SB_HALO_TRANSPORT=peer");
    if (rank < 0 || rank >= s->desc.world || rank == s->desc.rank) return fail(SB_ERR_INVALID_ARG, "sb_peer_connect: bad rank");
    if (s->peer.remote[(size_t)rank]) return fail(SB_ERR_STATE, "sb_peer_connect: rank already connected");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        if (same_process_peer) {             // the peer's handle lives in this process: its pointer is directly usable
            if (!same_process_peer->peer.mailbox || same_process_peer->desc.rank != rank) return fail(SB_ERR_INVALID_ARG, "sb_peer_connect: the given solver is not that rank");
            s->peer.remote[(size_t)rank] = same_process_peer->peer.mailbox;
        } else {
            hipIpcMemHandle_t h;
            std::memcpy(&h, handle, sizeof(h));
            void *p = nullptr;
            HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            s->peer.remote[(size_t)rank] = (uint32_t *)p; s->peer.opened[(size_t)rank] = 1;
        }
        s->peer.linked = false;
        return SB_OK;
    });
}

int sb_set_domain(sb_solver *s, const sb_domain *domain, const int32_t *global_id, int32_t n) {
    if (!s || !domain || !global_id) return fail(SB_ERR_INVALID_ARG, "sb_set_domain: null argument");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_set_domain after sb_finalize");
    if (n != s->n || n <= 0) return fail(SB_ERR_INVALID_ARG, "sb_set_domain: n differs from sb_set_particles");
    if (domain->n_global < n || !(domain->spacing > 0) || domain->reserved != 0) return fail(SB_ERR_INVALID_ARG, "sb_set_domain: bad domain");
    return guarded([&]() -> int {
        s->domain = *domain;
        s->global_id.assign(global_id, global_id + n);
        s->sharded = true;
        return SB_OK;
    });
}

int sb_domain_from_mesh(const float *rest, int32_t n, const int32_t *dist_ij, int32_t m_d, const int32_t *vol, int32_t m_v,
                        const int32_t *bend, int32_t m_b, sb_domain *out) {
    if (!rest || !out || n <= 0 || m_d < 0 || m_v < 0 || m_b < 0) return fail(SB_ERR_INVALID_ARG, "sb_domain_from_mesh: bad argument");
    return guarded([&]() -> int {
        sbp::Domain D;
        sbp::compute_domain(make_input(rest, n, dist_ij, m_d, vol, m_v, bend, m_b), D);
        std::memset(out, 0, sizeof(*out));
        out->n_global = D.n_global; out->spacing = D.ell; out->fill = D.fill;
        for (int a = 0; a < 3; ++a) { out->lo[a] = D.lo[a]; out->hi[a] = D.hi[a]; }
        out->four_vertex_constraints = (m_v + m_b > 0) ? 1 : 0;
        return SB_OK;
    });
}

int sb_domain_window(const sb_domain *domain, const sb_plan_opts *opts, double lo_out[3], double hi_out[3]) {
    if (!domain || !opts || !lo_out || !hi_out) return fail(SB_ERR_INVALID_ARG, "sb_domain_window: null argument");
    if (opts->world < 1 || opts->rank < 0 || opts->rank >= opts->world) return fail(SB_ERR_INVALID_ARG, "sb_domain_window: bad rank / world");
    return guarded([&]() -> int {
        const sbp::Opts o = plan_opts(opts->rank, opts->world, opts->part_dims, opts->tile_particles, SB_PARTITION_BLOCKS, 0u, 0, 0, domain);
        int clo[3], chi[3];
        sbp::rank_window(o.domain, o, clo, chi, lo_out, hi_out);
        return SB_OK;
    });
}

int sb_finalize(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_finalize: null handle");
    if (s->finalized) return fail(SB_ERR_STATE, "sb_finalize called twice");
    if (s->n <= 0) return fail(SB_ERR_STATE, "sb_finalize before sb_set_particles");
    // SB_DEBUG_NO_COMM: the hosted-halo test hooks (sb_debug_*) drive world > 1 without any transport
    const bool no_comm = (s->desc.debug_flags & SB_DEBUG_NO_COMM) != 0;
    if (s->desc.world > 1 && !s->comm && !s->peer.enabled && !no_comm)
        return fail(SB_ERR_STATE, "sb_finalize: world > 1 needs sb_comm_init first (or halo_transport = SB_TRANSPORT_PEER with sb_peer_connect)");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        // ---- which schedule (world > 1): decided before any work, from what the process is actually bound to ----
        int sched = s->desc.world > 1 ? s->desc.halo_schedule : SB_SCHEDULE_SERIAL_EAGER;
        const bool sched_auto = sched == SB_SCHEDULE_AUTO;       // resolved below, once the size of the exchange is known
        if (sched_auto) sched = SB_SCHEDULE_SERIAL_EAGER;
        if (s->desc.world > 1 && !no_comm) {
            const bool graph = sched == SB_SCHEDULE_SERIAL_GRAPH || sched == SB_SCHEDULE_OVERLAP_GRAPH;
            if (graph && !s->desc.use_graph) return fail(SB_ERR_INVALID_ARG, "sb_finalize: a captured halo schedule needs use_graph = 1");
            if (graph && !s->peer.enabled && rccl().version < 22606)
                return fail(SB_ERR_UNSUPPORTED, "sb_finalize: capturing ncclSend/ncclRecv in a hipGraph is verified on RCCL >= 2.26.6 only; this process is bound to RCCL " +
                            std::to_string(rccl().version) + " (" + rccl().path + "): use SB_SCHEDULE_SERIAL_EAGER");
            if (sched == SB_SCHEDULE_OVERLAP_GRAPH && !capture_overlap_ok())
                return fail(SB_ERR_UNSUPPORTED, "sb_finalize: SB_SCHEDULE_OVERLAP_GRAPH faults on HIP runtimes older than 7.2 (hipStreamEndCapture recurses over the forked "
                            "exchange stream); this process is bound to HIP runtime " + std::to_string(hip_runtime_version()) +
                            " (a host that loaded PyTorch first runs on PyTorch's bundled runtime): use SB_SCHEDULE_SERIAL_GRAPH or an eager schedule");
        }
        const std::vector<float> &rest = s->rest.empty() ? s->pos : s->rest;
        sbp::Input in = make_input(rest.data(), s->n, s->dist_ij.data(), (int64_t)s->dist_rest.size(), s->vol_ijkl.data(),
                                   (int64_t)s->vol_rest.size(), s->bend_ijkl.data(), (int64_t)s->bend_rest.size() / 2);
        if (s->sharded) {
            if (s->desc.world < 2) return fail(SB_ERR_INVALID_ARG, "sb_finalize: sharded authoring (sb_set_domain) is for world > 1");
            if (s->desc.partition == SB_PARTITION_RCB) return fail(SB_ERR_UNSUPPORTED, "sb_finalize: the RCB partition needs the whole mesh on every rank (no sb_set_domain)");
            in.global_id = s->global_id.data();
        }
        sbp::Opts o = plan_opts(s->desc.rank, s->desc.world, s->desc.part_dims, s->desc.tile_particles, s->desc.partition, s->desc.plan_flags,
                                in.m_v, in.m_b, s->sharded ? &s->domain : nullptr);
        s->plan = std::make_unique<sb_plan>();
        const bool timing = std::getenv("SB_PLAN_TIMING") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        sbp::build_plan(in, o, s->plan->plan);
        sbp::extract_local(s->plan->plan, in, o.rank, s->plan->local);
        s->plan_hash = hash_plan(s->plan->plan);
        auto t1 = std::chrono::steady_clock::now();
        {   // the overlapped schedules apply to plans whose only exchange is the one before the T1 kernels (lattices)
            const sbp::Plan &P = s->plan->plan;
            const sbp::LocalPlan &L = s->plan->local;
            bool t1_halo = false;
            if (L.halo.size() > 1) for (int r = 0; r < L.world; ++r) t1_halo |= !L.halo[1].send_idx[(size_t)r].empty() || !L.halo[1].recv_idx[(size_t)r].empty();
            // SB_SCHEDULE_AUTO (RCCL between devices): xGMI is point-to-point, one link per pair of GPUs, and a rank's ghosts for one
            // peer travel as one message -- an exchange cannot end before its largest message has crossed its link (<= 77 GB/s per
            // direction). From kAutoOverlapBytes per peer on (13 us at the link's peak, several times that in practice) the exchange
            // lasts longer than the two cross-stream events and the two extra launches of the overlapped schedule cost (measured on
            // one device, where the link time is zero: + 5 .. 15 us per exchange, profiles/r03q_loopback_w2_w4_schedules.txt), so it is
            // run beside the interior tiles; below that the serialised schedule. A model-based choice: no schedule has run between two
            // devices yet. Eager in both cases (captured schedules stay opt-in).
            constexpr int64_t kAutoOverlapBytes = 1 << 20;
            if (sched_auto && s->comm && !s->peer.enabled && L.halo.size() > 1) {
                int64_t largest = 0;
                for (int r = 0; r < L.world; ++r) largest = std::max<int64_t>(largest, (int64_t)L.halo[1].send_idx[(size_t)r].size() * 24);
                if (largest >= kAutoOverlapBytes) sched = SB_SCHEDULE_OVERLAP_EAGER;
            }
            const bool want_overlap = sched == SB_SCHEDULE_OVERLAP_EAGER || sched == SB_SCHEDULE_OVERLAP_GRAPH;
            s->overlap_halo = want_overlap && s->desc.world > 1 && s->comm && P.tiling && P.gcolours.empty() && P.t2_layers.empty() && t1_halo;
            if (want_overlap && !s->overlap_halo)      // T2 layers / global colours (irregular mesh) or no T1 halo: the serialised form
                sched = sched == SB_SCHEDULE_OVERLAP_GRAPH ? SB_SCHEDULE_SERIAL_GRAPH : SB_SCHEDULE_SERIAL_EAGER;
        }
        s->schedule = sched;
        s->graph_rccl = sched == SB_SCHEDULE_SERIAL_GRAPH || sched == SB_SCHEDULE_OVERLAP_GRAPH;
        build_device(s);
        if (timing)
            std::fprintf(stderr, "[finalize] plan %.1f ms, build_device + upload %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        // opt in to the LDS size the largest tile needs
        for (int tl = 0; tl < 3; ++tl)
            if (s->tiling[tl].lds_bytes > 64 * 1024) throw std::runtime_error("internal: tile LDS budget exceeded");
        // Ranks plan independently: before the first exchange make sure they all arrived at the same plan (same published
        // order, ownership, halo slots, plan options). One 8-byte all-gather over the communicator; the peer transport without
        // a communicator compares the hashes when the mailboxes are linked (peer_link).
        if (s->desc.world > 1 && s->comm && !s->loopback) {
            // per rank: [sharded?, plan hash, pair hash with rank 0 .. W-1]
            const int W = s->desc.world;
            const size_t rec = (size_t)W + 2;
            std::vector<uint64_t> mine(rec, 0);
            mine[0] = s->sharded ? 1 : 0; mine[1] = s->plan_hash;
            for (int r = 0; r < W; ++r) mine[2 + (size_t)r] = s->plan->local.pair_hash[(size_t)r];
            DevBuf<uint64_t> d_all; int64_t acct = 0;
            d_all.alloc((size_t)W * rec, acct);
            HIP_CHECK(hipMemcpy(d_all.p + (size_t)s->desc.rank * rec, mine.data(), rec * sizeof(uint64_t), hipMemcpyHostToDevice));
            NCCL_CHECK(rccl().AllGather(d_all.p + (size_t)s->desc.rank * rec, d_all.p, rec * sizeof(uint64_t), ncclUint8, s->comm, s->stream));
            HIP_CHECK(hipStreamSynchronize(s->stream));
            std::vector<uint64_t> all((size_t)W * rec);
            HIP_CHECK(hipMemcpy(all.data(), d_all.p, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            // Every rank holds the whole table and checks EVERY pair, so that all ranks fail together (a rank that went on alone would
            // wait for its neighbours' first exchange forever). Whole-mesh ranks must hold the identical plan; any two ranks must agree
            // on what they share (ghost lists both ways, programs of the tiles both run) -- the only check a sharded rank, which sees
            // just its window, can make.
            for (int a = 0; a < W; ++a)
                for (int b = a + 1; b < W; ++b) {
                    const uint64_t *ra = all.data() + (size_t)a * rec, *rb = all.data() + (size_t)b * rec;
                    char msg[320];
                    if (!ra[0] && !rb[0] && ra[1] != rb[1]) {
                        std::snprintf(msg, sizeof msg, "sb_finalize: ranks %d and %d planned different schedules (plan hash %016llx vs %016llx): "
                                      "every rank must pass the same mesh, tile_particles, partition and plan_flags", a, b,
                                      (unsigned long long)ra[1], (unsigned long long)rb[1]);
                        return fail(SB_ERR_STATE, msg);
                    }
                    if (ra[2 + (size_t)b] != rb[2 + (size_t)a]) {
                        std::snprintf(msg, sizeof msg, "sb_finalize: ranks %d and %d disagree on what they share (ghost lists / programs of the tiles both run: pair hash "
                                      "%016llx vs %016llx): same mesh, domain, tile_particles, partition and plan_flags on every rank? window complete (sb_domain_window)?",
                                      a, b, (unsigned long long)ra[2 + (size_t)b], (unsigned long long)rb[2 + (size_t)a]);
                        return fail(SB_ERR_STATE, msg);
                    }
                }
        }
        if (s->overlap_halo) {
            // opt-in: on the one measurement available (RCCL loopback on one GPU, 8-rank share of 256^3) splitting the T0
            // launch and running the exchange beside the interior tiles pays only inside a captured graph
            // (an ordinary stream: a highest-priority one -- meant to keep the pack kernel and RCCL's few workgroups from queueing behind
            // the interior launch -- made the eager overlapped tick FOUR TIMES slower on this runtime, 0.93 -> 3.35 ms in the W = 8
            // loopback, profiles/r03r_loopback_w8_priority_stream_not_kept.txt)
            HIP_CHECK(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&s->ev_boundary, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming));
        }
        if (s->peer.enabled && s->desc.world > 1) {
            if (s->loopback) {
                peer_link(s);                      // every neighbour is this rank itself
            } else if (s->comm) {
                // exchange the mailbox handles over the communicator the host already set up (setup time only)
                const int W = s->desc.world;
                hipIpcMemHandle_t mine;
                HIP_CHECK(hipIpcGetMemHandle(&mine, s->peer.mailbox));
                DevBuf<uint8_t> d_all; int64_t acct = 0;
                d_all.alloc((size_t)W * sizeof(mine), acct);
                HIP_CHECK(hipMemcpy(d_all.p + (size_t)s->desc.rank * sizeof(mine), &mine, sizeof(mine), hipMemcpyHostToDevice));
                NCCL_CHECK(rccl().AllGather(d_all.p + (size_t)s->desc.rank * sizeof(mine), d_all.p, sizeof(mine), ncclUint8, s->comm, s->stream));
                HIP_CHECK(hipStreamSynchronize(s->stream));
                std::vector<hipIpcMemHandle_t> all((size_t)W);
                HIP_CHECK(hipMemcpy(all.data(), d_all.p, (size_t)W * sizeof(mine), hipMemcpyDeviceToHost));
                for (int r = 0; r < W; ++r) {
                    if (r == s->desc.rank) continue;
                    bool needed = false;
                    for (const auto &H : s->halos) for (int pr : H->peers) needed |= pr == r;
                    if (!needed) continue;
                    void *p = nullptr;
                    HIP_CHECK(hipIpcOpenMemHandle(&p, all[(size_t)r], hipIpcMemLazyEnablePeerAccess));
                    s->peer.remote[(size_t)r] = (uint32_t *)p; s->peer.opened[(size_t)r] = 1;
                }
                peer_link(s);
            }
            // otherwise (test mode without a communicator) the host connects the mailboxes: sb_peer_mailbox_handle / sb_peer_connect
        }
        if (std::getenv("SB_PRINT_ALLOC"))       // diagnosis: where the arrays landed (run-to-run timing modes)
            std::fprintf(stderr, "[alloc] pos3 %p prev %p vel %p wf %p w8 %p T0.stream %p T1.stream %p T0.tiles %p T1.tiles %p\n", (void *)s->d_pos3.p, (void *)s->d_prev.p,
                         (void *)s->d_vel.p, (void *)s->d_wf.p, (void *)s->d_w8.p, (void *)s->tiling[0].stream.p, (void *)s->tiling[1].stream.p,
                         (void *)s->tiling[0].tiles.p, (void *)s->tiling[1].tiles.p);
        HIP_CHECK(hipDeviceSynchronize());
        // authoring copies are no longer needed (keep rest values out of memory for 50M-constraint meshes)
        std::vector<float>().swap(s->pos); std::vector<float>().swap(s->vel); std::vector<float>().swap(s->rest);
        std::vector<int32_t>().swap(s->dist_ij); std::vector<int32_t>().swap(s->vol_ijkl); std::vector<int32_t>().swap(s->bend_ijkl);
        std::vector<float>().swap(s->dist_rest); std::vector<float>().swap(s->vol_rest); std::vector<float>().swap(s->bend_rest);
        s->finalized = true;
        return SB_OK;
    });
}

int sb_step(sb_solver *s, float dt, int32_t substeps) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_step: null handle");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_step before sb_finalize");
    if (!(dt > 0.0f) || substeps <= 0) return fail(SB_ERR_INVALID_ARG, "sb_step: dt and substeps must be positive");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        check_peer_error(s);       // a halo wait of an earlier tick gave up: do not pile further ticks on stale ghosts
        if (s->peer.enabled && s->desc.world > 1 && !s->peer.linked) peer_link(s);     // (reads the neighbours' offset tables: not inside a capture)
        // lazy tick boundary: fuse with the previous tick's deferred last kernel when nothing changed
        const sbk::TickParams tp_new = tick_params(s, dt, substeps);
        const bool can_defer = s->lazy_tick && (!s->plan->plan.tiling || (substeps & 1) == 0);
        const bool fuse = s->deferred && can_defer && s->deferred_substeps == substeps && s->tp_valid &&
                          std::memcmp(&tp_new, &s->tp_host, sizeof(tp_new)) == 0 && (s->kin_fuse || s->kin_pending < 0);
        if (!fuse) flush_deferred(s); else ++s->n_fused;
        // pending kinematic targets cross a fused tick boundary INSIDE the fused kernel (tile_kernel KIND 5)
        const bool kin = fuse && s->kin_pending >= 0;
        if (kin) { stage_kinematic_for_fusion(s); ++s->n_kin_fused; }
        upload_tick_params(s, dt, substeps);
        // world > 1: the exchange inside a captured graph is opt-in (SB_SCHEDULE_*_GRAPH), see DESIGN.md §7
        const bool graph_ok = s->desc.use_graph && (s->desc.world == 1 || s->graph_rccl);     // the overlapped schedule forks onto comm_stream inside the capture
        if (!graph_ok) {
            enqueue_substeps(s, substeps, nullptr, fuse, can_defer, kin);
        } else {
            const int key = substeps * 8 + (fuse ? 1 : 0) + (can_defer ? 2 : 0) + (kin ? 4 : 0);
            auto it = s->graphs.find(key);
            if (it == s->graphs.end()) {
                hipGraph_t g = nullptr;
                HIP_CHECK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
                try {
                    enqueue_substeps(s, substeps, nullptr, fuse, can_defer, kin);
                } catch (...) {
                    (void)hipStreamEndCapture(s->stream, &g);
                    if (g) (void)hipGraphDestroy(g);
                    throw;
                }
                HIP_CHECK(hipStreamEndCapture(s->stream, &g));
                hipGraphExec_t ge = nullptr;
                hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e != hipSuccess) throw HipError(SB_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
                if (s->graphs.size() >= sb_solver::kMaxGraphs) {     // evict the least recently used executable
                    auto old = s->graphs.begin();
                    for (auto q = s->graphs.begin(); q != s->graphs.end(); ++q) if (q->second.last_use < old->second.last_use) old = q;
                    HIP_CHECK(hipStreamSynchronize(s->stream));          // it may still be running
                    (void)hipGraphExecDestroy(old->second.exec);
                    s->graphs.erase(old);
                }
                it = s->graphs.emplace(key, sb_solver::CachedGraph{ge, 0}).first;
            }
            it->second.last_use = ++s->graph_clock;
            HIP_CHECK(hipGraphLaunch(it->second.exec, s->stream));
        }
        s->deferred = can_defer;
        s->deferred_substeps = substeps;
        return SB_OK;
    });
}

int sb_step_profiled(sb_solver *s, float dt, int32_t substeps, float *slot_ms, int32_t *slot_launches, int32_t n_slots) {
    if (!s || !slot_ms || !slot_launches) return fail(SB_ERR_INVALID_ARG, "sb_step_profiled: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_step_profiled before sb_finalize");
    if (!(dt > 0.0f) || substeps <= 0) return fail(SB_ERR_INVALID_ARG, "sb_step_profiled: dt and substeps must be positive");
    if (n_slots != (int32_t)s->gcolours.size() + 5) return fail(SB_ERR_INVALID_ARG, "sb_step_profiled: n_slots must be 5 + n_global_colours");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        if (s->peer.enabled && s->desc.world > 1 && !s->peer.linked) peer_link(s);
        flush_deferred(s);
        upload_tick_params(s, dt, substeps);
        LaunchTimer lt; lt.stream = s->stream;
        enqueue_substeps(s, substeps, &lt);
        HIP_CHECK(hipStreamSynchronize(s->stream));
        for (int k = 0; k < n_slots; ++k) { slot_ms[k] = 0.0f; slot_launches[k] = 0; }
        for (size_t k = 0; k < lt.slot.size(); ++k) {
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, lt.ev[2 * k], lt.ev[2 * k + 1]));
            slot_ms[lt.slot[k]] += ms; ++slot_launches[lt.slot[k]];
        }
        return SB_OK;
    });
}

/* ---- test hooks: drive one tick launch by launch with the halo carried by the host -------------------- */

int sb_debug_launch(sb_solver *s, float dt, int32_t substeps, int32_t it, int32_t gcolour) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_debug_launch: null handle");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_debug_launch before sb_finalize");
    if (!(dt > 0.0f) || substeps <= 0 || it < 0 || it > substeps || gcolour >= (int32_t)s->gcolours.size())
        return fail(SB_ERR_INVALID_ARG, "sb_debug_launch: bad argument");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        flush_deferred(s);
        upload_tick_params(s, dt, substeps);
        if (gcolour <= -2) {
            if (-2 - gcolour >= (int32_t)s->t2_layer_range.size()) return fail(SB_ERR_INVALID_ARG, "sb_debug_launch: no such T2 layer");
            launch_t2_layer(s, -2 - gcolour, nullptr, false);
        } else if (gcolour < 0) launch_tick_kernel(s, it, substeps, nullptr);
        else launch_gcolour(s, gcolour, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s->stream));
        return SB_OK;
    });
}

int sb_debug_halo_pack(sb_solver *s, int32_t slot, float *host_out, int64_t capacity_floats, int64_t *count_floats) {
    if (!s || !count_floats) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_pack: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_debug_halo_pack before sb_finalize");
    if (slot < 0 || slot >= (int32_t)s->halos.size()) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_pack: bad slot");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        flush_deferred(s);
        DevHalo &D = *s->halos[slot];
        const int ns = D.send_off.back();
        const int64_t need = (int64_t)ns * (slot == 1 ? 6 : 3);
        *count_floats = need;
        if (need == 0) return SB_OK;
        if (!host_out || capacity_floats < need) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_pack: buffer too small");
        if (slot == 1)
            hipLaunchKernelGGL(sbk::halo_pack_kernel<true>, dim3((ns + 255) / 256), dim3(256), 0, s->stream, s->pos_view(), s->d_prev.p,
                               D.send_idx.p, s->d_sendbuf.p, ns);
        else
            hipLaunchKernelGGL(sbk::halo_pack_kernel<false>, dim3((ns + 255) / 256), dim3(256), 0, s->stream, s->pos_view(), s->d_prev.p,
                               D.send_idx.p, s->d_sendbuf.p, ns);
        HIP_CHECK(hipMemcpyAsync(host_out, s->d_sendbuf.p, (size_t)need * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        return SB_OK;
    });
}

int sb_debug_halo_unpack(sb_solver *s, int32_t slot, const float *host_in, int64_t count_floats) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_unpack: null handle");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_debug_halo_unpack before sb_finalize");
    if (slot < 0 || slot >= (int32_t)s->halos.size()) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_unpack: bad slot");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        DevHalo &D = *s->halos[slot];
        const int nr = D.recv_off.back();
        const int64_t need = (int64_t)nr * (slot == 1 ? 6 : 3);
        if (count_floats != need) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_unpack: wrong element count");
        if (need == 0) return SB_OK;
        if (!host_in) return fail(SB_ERR_INVALID_ARG, "sb_debug_halo_unpack: null buffer");
        HIP_CHECK(hipMemcpyAsync(s->d_recvbuf.p, host_in, (size_t)need * sizeof(float), hipMemcpyHostToDevice, s->stream));
        if (slot == 1 && s->fused_unpack) {
            // (the T1 kernels read the ghosts from the receive buffer: nothing to scatter)
        } else if (slot == 1)
            hipLaunchKernelGGL(sbk::halo_unpack_kernel<true>, dim3((nr + 255) / 256), dim3(256), 0, s->stream, s->pos_view(), s->d_prev.p,
                               D.recv_idx.p, s->d_recvbuf.p, nr);
        else
            hipLaunchKernelGGL(sbk::halo_unpack_kernel<false>, dim3((nr + 255) / 256), dim3(256), 0, s->stream, s->pos_view(), s->d_prev.p,
                               D.recv_idx.p, s->d_recvbuf.p, nr);
        HIP_CHECK(hipStreamSynchronize(s->stream));
        return SB_OK;
    });
}

int sb_debug_validate(sb_solver *s, int32_t inject_fault, sb_validate_report *out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_debug_validate: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_debug_validate before sb_finalize");
    if (inject_fault < 0 || inject_fault > 2) return fail(SB_ERR_INVALID_ARG, "sb_debug_validate: inject_fault must be 0, 1 or 2");
    std::memset(out, 0, sizeof(*out));
    out->first_stage = out->first_tile = out->first_group = out->first_kind = -1;
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        int64_t acct = 0;
        DevBuf<int32_t> owner; owner.alloc((size_t)std::max<int64_t>(s->n_local, 1), acct);
        DevBuf<sbk::ValidateCounters> d_cnt; d_cnt.alloc(1, acct);
        sbk::ValidateCounters zero{}; zero.first[0] = zero.first[1] = zero.first[2] = zero.first[3] = -1;
        auto stage = [&](int which, auto &&launch) {
            HIP_CHECK(hipMemcpyAsync(d_cnt.p, &zero, sizeof(zero), hipMemcpyHostToDevice, s->stream));
            HIP_CHECK(hipMemsetAsync(owner.p, 0xff, owner.count * sizeof(int32_t), s->stream));
            launch();
            HIP_CHECK(hipGetLastError());
            sbk::ValidateCounters c{};
            HIP_CHECK(hipMemcpyAsync(&c, d_cnt.p, sizeof(c), hipMemcpyDeviceToHost, s->stream));
            HIP_CHECK(hipStreamSynchronize(s->stream));
            out->tiles_checked += (int64_t)c.tiles; out->groups_checked += (int64_t)c.groups; out->constraints_checked += (int64_t)c.constraints;
            bool any = false;
            for (int k = 0; k < 6; ++k) { out->errors[k] += c.errors[k]; any |= c.errors[k] != 0; }
            if (any && out->first_stage < 0) { out->first_stage = which; out->first_tile = c.first[0]; out->first_group = c.first[1]; out->first_kind = c.first[2]; }
        };
        auto tiles_launch = [&](DevTiling &D, int tl, int begin, int end, const sbk::TileDesc *tiles, const uint32_t *stream) {
            if (end <= begin) return;
            // (tiles of a launch number from `begin`: the kernel indexes the table it is given with tile_begin + blockIdx.x)
            hipLaunchKernelGGL(sbk::validate_tiles_kernel, dim3((unsigned)(end - begin)), dim3(sbk::kValidateThreads), 0, s->stream, tiles, begin,
                               D.runs_overflow.p, stream, D.gather.p, tl == 2 ? 1 : 0, (int)s->n_local, (int)D.item_waves, owner.p, d_cnt.p);
        };
        if (inject_fault) {
            DevTiling &D = s->tiling[0];
            if (D.n_tiles < 2) return fail(SB_ERR_STATE, "sb_debug_validate: fault injection needs a tiling of at least two tiles");
            std::vector<sbk::TileDesc> tiles((size_t)D.n_tiles);
            HIP_CHECK(hipMemcpy(tiles.data(), D.tiles.p, tiles.size() * sizeof(sbk::TileDesc), hipMemcpyDeviceToHost));
            DevBuf<sbk::TileDesc> t_copy; DevBuf<uint32_t> s_copy;
            s_copy.alloc(D.stream.count, acct);
            HIP_CHECK(hipMemcpy(s_copy.p, D.stream.p, D.stream.count * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            if (inject_fault == 2) tiles[1] = tiles[0];       // two workgroups of one launch stage the same particles
            else {
                bool planted = false;
                for (size_t t = 0; t < tiles.size() && !planted; ++t) {
                    const sbk::TileDesc &td = tiles[t];
                    std::vector<uint32_t> words((size_t)std::max(td.n_rounds, 1));
                    HIP_CHECK(hipMemcpy(words.data(), D.stream.p + td.s_begin, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                    uint32_t off = td.s_hdr;
                    if (td.packed_lanes && td.n_rounds > 0 && (words[0] & 1023u) >= 2) {      // lane-packed: lane 0's word over lane 1's
                        uint32_t *base = s_copy.p + td.s_begin + off;
                        HIP_CHECK(hipMemcpy(base + 4, base, 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice));
                        planted = true;
                    }
                    for (int r = 0; r < td.n_rounds && !planted && !td.packed_lanes; ++r) {
                        const uint32_t w = words[(size_t)r], cnt = w & 1023u, nq = ((w >> 10) & 1023u) + ((w >> 20) & 1023u);
                        const bool compact = (w >> 30) & 1u;
                        const uint32_t dsize = compact ? ((cnt + 3u) & ~3u) : ((2u * cnt + 3u) & ~3u);
                        uint32_t *base = s_copy.p + td.s_begin + off;
                        if (cnt >= 2) { HIP_CHECK(hipMemcpy(base + (compact ? 1 : 2), base, (compact ? 1 : 2) * sizeof(uint32_t), hipMemcpyDeviceToDevice)); planted = true; }
                        else if (nq >= 2) { HIP_CHECK(hipMemcpy(base + dsize + 4, base + dsize, 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice)); planted = true; }
                        off += dsize + 4u * nq;
                    }
                }
                if (!planted) return fail(SB_ERR_STATE, "sb_debug_validate: no group with two constraints of one type to plant the fault in");
            }
            t_copy.upload(tiles, acct);
            stage(0, [&] { tiles_launch(D, 0, 0, D.n_tiles, t_copy.p, s_copy.p); });
            return SB_OK;
        }
        for (int tl = 0; tl < 2; ++tl) {
            DevTiling &D = s->tiling[tl];
            if (D.n_tiles) stage(tl, [&] { tiles_launch(D, tl, 0, D.n_tiles, D.tiles.p, D.stream.p); });
        }
        for (const auto &rg : s->t2_layer_range) {       // the tiles of ONE layer share no particle; different layers do
            DevTiling &D = s->tiling[2];
            stage(2, [&] { tiles_launch(D, 2, rg.first, rg.second, D.tiles.p, D.stream.p); });
        }
        for (size_t gc = 0; gc < s->gcolours.size(); ++gc) {
            DevGColour &G = *s->gcolours[gc];
            if (!G.count) continue;
            stage(3, [&] {
                const int per = G.type == 0 ? 2 : 4;
                const int32_t *idx = G.type == 0 ? reinterpret_cast<const int32_t *>(G.ij.p) : reinterpret_cast<const int32_t *>(G.quad.p);
                hipLaunchKernelGGL(sbk::validate_gcolour_kernel, dim3((unsigned)((G.count + 255) / 256)), dim3(256), 0, s->stream, idx, per, (int)G.count,
                                   (int)s->n_local, (int)gc, owner.p, d_cnt.p);
            });
        }
        return SB_OK;
    });
}

int sb_synchronize(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_synchronize: null handle");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        flush_deferred(s);
        HIP_CHECK(hipStreamSynchronize(s->stream));
        check_peer_error(s);
        return SB_OK;
    });
}

static int get_state(sb_solver *s, float *out, int32_t n, bool velocity) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_get_*: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_* before sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_get_*: n differs from sb_set_particles");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        const sbp::LocalPlan &L = s->plan->local;
        // positions while the tick's last kernel is deferred: peek instead of completing the tick (the next sb_step keeps its fusion)
        const bool peek = !velocity && can_peek(s);
        if (peek) { peek_positions(s, false); if (s->kin_pending >= 0) scatter_kinematic(s, s->d_peek.p); }     // (pending targets show in what is read; they stay pending)
        else flush_deferred(s);
        if (s->desc.world == 1) {
            // single rank: every entry is ours, so the permutation to caller numbering runs on the GPU and one copy
            // lands in the caller's array (the host-side scatter below costs 25 ms for 16.7 M particles)
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
        }
        HIP_CHECK(hipStreamSynchronize(s->stream));
        check_peer_error(s);
        // world > 1: only the entries this rank owns may be written (the caller merges the ranks' arrays)
        s->h_stage.resize((size_t)s->n_owned * 3);
        const float *src = velocity ? s->d_vel.p : s->d_pos3.p;
        if (s->n_owned) HIP_CHECK(hipMemcpy(s->h_stage.data(), src, (size_t)s->n_owned * 3 * sizeof(float), hipMemcpyDeviceToHost));
        sbp::parallel_for_chunks(s->n_owned, 1 << 18, [&](int64_t, int64_t lb, int64_t le) {     // owned particles have distinct caller ids
            for (int64_t l = lb; l < le; ++l) {
                int32_t o = L.local_to_old[l];
                for (int c = 0; c < 3; ++c) out[3 * (size_t)o + c] = s->h_stage[3 * (size_t)l + c];
            }
        });
        return SB_OK;
    });
}

int sb_get_positions(sb_solver *s, float *out, int32_t n) { return get_state(s, out, n, false); }
int sb_get_velocities(sb_solver *s, float *out, int32_t n) { return get_state(s, out, n, true); }

int sb_set_state(sb_solver *s, const float *pos, const float *vel, int32_t n) {
    if (!s || !pos || !vel) return fail(SB_ERR_INVALID_ARG, "sb_set_state: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_set_state before sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_state: n differs from sb_set_particles");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        const sbp::LocalPlan &L = s->plan->local;
        flush_deferred(s);
        HIP_CHECK(hipStreamSynchronize(s->stream));
        std::vector<float> hp((size_t)s->n_local * 3), hv((size_t)s->n_local * 3);
        for (int64_t l = 0; l < s->n_local; ++l) {
            int32_t o = L.local_to_old[l];
            for (int c = 0; c < 3; ++c) { hp[3 * (size_t)l + c] = pos[3 * (size_t)o + c]; hv[3 * (size_t)l + c] = vel[3 * (size_t)o + c]; }
        }
        HIP_CHECK(hipMemcpy(s->d_pos3.p, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(s->d_vel.p, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice));
        return SB_OK;
    });
}

int sb_set_kinematic_positions(sb_solver *s, const int32_t *ids, const float *pos, int32_t count) {
    if (!s || count < 0 || (count > 0 && (!ids || !pos))) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: bad argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_set_kinematic_positions before sb_finalize");
    if (s->desc.world != 1) return fail(SB_ERR_UNSUPPORTED, "sb_set_kinematic_positions: single-rank solvers only (world == 1)");
    if (count == 0) return SB_OK;
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        for (int32_t k = 0; k < count; ++k) {
            if (ids[k] < 0 || ids[k] >= s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle index out of range");
            if (s->invm[(size_t)ids[k]] != 0.0f)
                return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: particle " + std::to_string(ids[k]) + " has a non-zero inverse mass (only pinned particles are kinematic)");
            for (int c = 0; c < 3; ++c) if (!(pos[3 * (size_t)k + c] == pos[3 * (size_t)k + c])) return fail(SB_ERR_INVALID_ARG, "sb_set_kinematic_positions: NaN");
        }
        if (s->kin_pending >= 0) flush_deferred(s);       // two moves without a tick between them: the earlier one takes effect first
        const int q = s->kin_next;
        s->kin_next = (q + 1) % sb_solver::kKinSlots;
        if (!s->ev_kin[q]) HIP_CHECK(hipEventCreateWithFlags(&s->ev_kin[q], hipEventDisableTiming));
        else HIP_CHECK(hipEventSynchronize(s->ev_kin[q]));       // the kernel that read this table (four calls ago) is done
        if (s->kin_cap[q] < (size_t)count) {
            if (s->h_kin_idx[q]) { (void)hipHostFree(s->h_kin_idx[q]); s->h_kin_idx[q] = nullptr; }
            if (s->h_kin_pos[q]) { (void)hipHostFree(s->h_kin_pos[q]); s->h_kin_pos[q] = nullptr; }
            const size_t cap = std::max<size_t>(256, (size_t)count * 2);
            HIP_CHECK(hipHostMalloc((void **)&s->h_kin_idx[q], cap * sizeof(int32_t), hipHostMallocDefault));
            HIP_CHECK(hipHostMalloc((void **)&s->h_kin_pos[q], cap * 3 * sizeof(float), hipHostMallocDefault));
            s->kin_cap[q] = cap;
        }
        const std::vector<int32_t> &new_of_old = s->plan->plan.new_of_old;      // world == 1: device numbering = the planner's numbering
        for (int32_t k = 0; k < count; ++k) s->h_kin_idx[q][k] = new_of_old[(size_t)ids[k]];
        std::memcpy(s->h_kin_pos[q], pos, (size_t)count * 3 * sizeof(float));
        // PENDING until the next tick starts (the previous tick's held-back last kernel still reads the old positions of these
        // particles): a fused first kernel takes them along, every other way across the tick boundary scatters them (flush_deferred)
        s->kin_pending = q; s->kin_pending_count = count;
        return SB_OK;
    });
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
        const bool peek = can_peek(s);       // the tick's last kernel is deferred: snapshot a peek and leave it deferred
        if (!peek) flush_deferred(s);
        if (!s->render_tri.empty() && s->render_dirty) {     // (re)build the incident-triangle lists: triangle ids ascending per particle
            HIP_CHECK(hipStreamSynchronize(s->copy_stream));
            const int64_t m = (int64_t)s->render_tri.size() / 3;
            std::vector<int32_t> off((size_t)s->n + 1, 0), adj((size_t)3 * m);
            for (int64_t c = 0; c < 3 * m; ++c) ++off[(size_t)s->render_tri[c] + 1];
            s->render_set.clear();
            for (int32_t v = 0; v < s->n; ++v) { if (off[(size_t)v + 1]) s->render_set.push_back(v); off[(size_t)v + 1] += off[v]; }
            std::vector<int32_t> cur(off.begin(), off.end() - 1);
            for (int64_t t = 0; t < m; ++t)
                for (int j = 0; j < 3; ++j) adj[(size_t)cur[s->render_tri[3 * t + j]]++] = (int32_t)t;
            std::vector<int32_t> local_of(s->render_set.size());      // world == 1: local numbering = the planner's new numbering
            for (size_t q = 0; q < local_of.size(); ++q) local_of[q] = s->plan->plan.new_of_old[s->render_set[q]];
            s->d_tri.upload(s->render_tri, s->dev_bytes);
            s->d_adj_off.upload(off, s->dev_bytes);
            s->d_adj_tri.upload(adj, s->dev_bytes);
            s->d_render_set.upload(s->render_set, s->dev_bytes);
            s->d_render_local.upload(local_of, s->dev_bytes);
            for (int q = 0; q < sb_solver::kSnapSlots; ++q) {
                if (!s->h_nrm[q]) {
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
        if (peek) {
            if (compact && s->n_peek_tiles < 0) {
                std::vector<int32_t> wanted(s->render_set.size());
                for (size_t q = 0; q < wanted.size(); ++q) wanted[q] = s->plan->plan.new_of_old[s->render_set[q]];
                build_peek_subset(s, wanted);
            }
            peek_positions(s, compact);
            if (s->kin_pending >= 0) scatter_kinematic(s, s->d_peek.p);
            src.xyz = s->d_peek.p;
        }
        if (compact) {      // only the render set leaves the device: snapshot just those particles
            const int cnt = (int)s->render_set.size();
            if (cnt)
                hipLaunchKernelGGL(sbk::snapshot_subset_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, src,
                                   s->d_render_set.p, s->d_render_local.p, s->d_snap[k].p, cnt);
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
        if (!s->render_tri.empty()) {
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
    if (s->desc.world > 1) return fail(SB_ERR_STATE, "sb_set_render_triangles: render normals need a single-rank solver (world == 1)");
    if (s->snap_pending) return fail(SB_ERR_STATE, "sb_set_render_triangles while a readback is pending");
    return guarded([&]() -> int {
        for (int64_t c = 0; c < 3 * (int64_t)m; ++c)
            if (tri[c] < 0 || tri[c] >= s->n) return fail(SB_ERR_INVALID_ARG, "sb_set_render_triangles: particle index out of range");
        s->render_tri.assign(tri, tri + 3 * (size_t)m);
        s->render_dirty = true;
        if (m == 0) s->render_set_only = false;
        for (bool &b : s->snap_has_normals) b = false;
        return SB_OK;
    });
}

int sb_readback_get_normals(sb_solver *s, const float **out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_readback_get_normals: null argument");
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
    if (s->snap_last_ended < 0 || !s->snap_has_normals[s->snap_last_ended])
        return fail(SB_ERR_STATE, "sb_readback_get_render_set: no finished readback with render triangles set");
    *ids = s->render_set.data();
    *count = (int32_t)s->render_set.size();
    return SB_OK;
}

int sb_get_owner(sb_solver *s, int32_t *owner, int32_t n) {
    if (!s || !owner) return fail(SB_ERR_INVALID_ARG, "sb_get_owner: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_owner before sb_finalize");
    if (n != s->n) return fail(SB_ERR_INVALID_ARG, "sb_get_owner: n mismatch");
    std::memcpy(owner, s->plan->plan.owner_of_old.data(), (size_t)n * sizeof(int32_t));
    return SB_OK;
}

int sb_profile_begin(sb_solver *s) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_profile_begin: null handle");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        HIP_CHECK(hipEventRecord(s->ev0, s->stream));
        return SB_OK;
    });
}
int sb_profile_end(sb_solver *s, float *ms) {
    if (!s || !ms) return fail(SB_ERR_INVALID_ARG, "sb_profile_end: null argument");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        HIP_CHECK(hipEventRecord(s->ev1, s->stream));
        HIP_CHECK(hipEventSynchronize(s->ev1));
        HIP_CHECK(hipEventElapsedTime(ms, s->ev0, s->ev1));
        return SB_OK;
    });
}

int sb_get_stats(sb_solver *s, sb_stats *out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_get_stats: null argument");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_get_stats before sb_finalize");
    std::memset(out, 0, sizeof(*out));
    const sbp::Plan &P = s->plan->plan;
    const sbp::LocalPlan &L = s->plan->local;
    out->n_particles_owned = s->n_owned;
    out->n_particles_local = s->n_local;
    for (size_t k = 0; k < L.order_mask[0].size(); ++k) if (L.order_mask[0][k]) ++out->n_constraints_local[P.order_type[0][k]];
    out->n_tilings = P.tiling ? 2 : 1;
    out->n_global_colours = (int32_t)P.gcolours.size();
    for (int tl = 0; tl < 2; ++tl) { out->n_tiles[tl] = s->tiling[tl].n_tiles; out->tile_constraints[tl] = s->tiling[tl].n_slots; }
    out->n_t2_layers = (int64_t)s->t2_layer_range.size();
    out->n_t2_tiles = s->tiling[2].n_tiles;
    out->t2_constraints = s->tiling[2].n_slots;
    out->constraints_in_tiles = P.cons_in_tiles;
    out->constraints_in_global = P.cons_in_global;
    for (size_t slot = 0; slot < s->halos.size(); ++slot) {
        const int64_t cnt = s->halos[slot]->send_off.back();
        if (slot == 1) out->halo_particles_t1 = cnt; else out->halo_particles_global += cnt;   // global colours and T2 layers
    }
    out->device_bytes = s->dev_bytes;
    {   // compulsory bytes per launch (see softbody.h): particle state + the tables a launch reads
        const int64_t mb = s->w_uniform ? 0 : (s->w_palette ? 1 : 4);     // inverse mass: nothing (uniform), palette index, or float
        auto tables = [&](const DevTiling &D) { return D.stream_bytes + (int64_t)D.n_tiles * (int64_t)sizeof(sbk::TileDesc) + (int64_t)D.runs_overflow.count * 8; };
        for (int tl = 0; tl < 2; ++tl) {
            const DevTiling &D = s->tiling[tl];
            if (!D.n_tiles) continue;
            out->launch_bytes[tl] = D.staged_particles * (12 + mb + 12 + 12 + 12) + tables(D);   // x, w, xprev in; x, xprev out
        }
        const DevTiling &D0 = s->tiling[0];
        if (D0.n_tiles) {
            out->launch_bytes[2] = D0.staged_particles * (12 + mb + 12 + 12 + 12) + tables(D0);  // x, w, v in; x, xprev out
            out->launch_bytes[3] = D0.staged_particles * (12 + mb + 12 + 12 + 12) + tables(D0);  // x, w, xprev in; x, v out
        }
        const DevTiling &D2 = s->tiling[2];
        if (D2.n_tiles) out->launch_bytes[4] = D2.staged_particles * (4 + 12 + mb + 12) + tables(D2);
    }
    out->partition = P.partition;
    out->halo_schedule = s->schedule;
    out->halo_unpack_fused = s->fused_unpack ? 1 : 0;
    out->readback_peeks = s->n_peeks;
    out->readback_peek_tiles = s->n_peek_tiles;
    out->ticks_fused = s->n_fused;
    out->ticks_fused_kinematic = s->n_kin_fused;
    for (int tl = 0; tl < 2; ++tl) out->lane_packed_tiles[tl] = s->tiling[tl].n_packed_tiles;
    out->plan_hash = s->plan_hash;
    if (!P.rank_cost.empty()) {
        out->partition_cost = P.rank_cost[(size_t)s->desc.rank];
        for (int64_t c : P.rank_cost) { out->partition_cost_total += c; out->partition_cost_max = std::max(out->partition_cost_max, c); }
    }
    {
        std::vector<uint8_t> is_peer((size_t)s->desc.world, 0);
        for (const auto &H : s->halos) {
            for (int pr : H->peers) is_peer[(size_t)pr] = 1;
            out->halo_particles_recv += H->recv_off.back();
        }
        for (uint8_t b : is_peer) out->halo_peers += b;
    }
    return SB_OK;
}

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
