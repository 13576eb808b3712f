// schedule.hip — kernel launches, the ghost exchange and the tick: sb_step and the launch-by-launch test hooks
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"
#include "tile_kernel.hip.hpp"
#include "aux_kernels.hip.hpp"

namespace sbi {

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
                if (hw[3] != plan_shape(s)) {       // (first: the layout of the neighbour's mailbox follows its number of halo slots)
                    std::snprintf(msg, sizeof msg, "peer transport: ranks %d and %d planned tick programs of different shape (leftover layers / global colours: %x vs %x): "
                                  "a window that does not reproduce the whole-mesh plan, or different meshes on the ranks", me, r, plan_shape(s), hw[3]);
                    throw HipError(SB_ERR_STATE, msg);
                }
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
// previous position), so each peer gets exactly one message per direction. In three parts, so that a host thread that drives
// SEVERAL ranks (group.hip, walk mode) can issue the middle part of all of them inside ONE ncclGroupStart / ncclGroupEnd:
//   halo_exchange_pre    RCCL: the pack kernel; peer transport: push + unpack kernels (the whole exchange)
//   halo_exchange_calls  RCCL: this rank's ncclSend / ncclRecv, one pair per neighbour -- the CALLER opens and closes the group
//   halo_exchange_post   RCCL: the unpack kernel (none behind a fused exchange)
bool halo_slot_active(const sb_solver *s, int slot) { return slot >= 0 && slot < (int)s->halos.size() && s->halos[(size_t)slot]->active(); }

void halo_exchange_pre(sb_solver *s, int slot, hipStream_t st) {
    if (!halo_slot_active(s, slot)) return;
    DevHalo &D = *s->halos[slot];
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
}

void halo_exchange_calls(sb_solver *s, int slot, hipStream_t st) {
    if (!halo_slot_active(s, slot) || s->peer.enabled) return;
    DevHalo &D = *s->halos[slot];
    const bool with_prev = slot == 1;
    for (size_t k = 0; k < D.peers.size(); ++k) {
        int cs = D.send_off[k + 1] - D.send_off[k], cr = D.recv_off[k + 1] - D.recv_off[k];
        if (s->loopback) cs = cr = std::min(cs, cr);   // a self-exchange must post equal sizes (real peers always do)
        const size_t fl = with_prev ? 6 : 3;   // floats per ghost; one message per peer and direction
        if (cs) NCCL_CHECK(rccl().Send(s->d_sendbuf.p + fl * D.send_off[k], fl * (size_t)cs, ncclFloat, s->loopback ? 0 : D.peers[k], s->comm, st));
        if (cr) NCCL_CHECK(rccl().Recv(s->d_recvbuf.p + fl * D.recv_off[k], fl * (size_t)cr, ncclFloat, s->loopback ? 0 : D.peers[k], s->comm, st));
    }
}

void halo_exchange_post(sb_solver *s, int slot, hipStream_t st) {
    if (!halo_slot_active(s, slot) || s->peer.enabled) return;
    DevHalo &D = *s->halos[slot];
    const bool with_prev = slot == 1;
    const int nr = D.recv_off.back();
    if (nr && !(with_prev && s->fused_unpack)) {
        if (with_prev)
            hipLaunchKernelGGL(sbk::halo_unpack_kernel<true>, dim3((nr + 255) / 256), dim3(256), 0, st, s->pos_view(),
                               s->d_prev.p, D.recv_idx.p, s->d_recvbuf.p, nr);
        else
            hipLaunchKernelGGL(sbk::halo_unpack_kernel<false>, dim3((nr + 255) / 256), dim3(256), 0, st, s->pos_view(),
                               s->d_prev.p, D.recv_idx.p, s->d_recvbuf.p, nr);
    }
}

// Exchange timing (sb_debug_exchange_timing): three events per exchange on the stream it runs on -- start, after the pack (or push)
// kernel, end -- resolved when the host reads the sums.
void ExchangeTimer::mark(hipStream_t st, bool join) {
    hipEvent_t e;
    if (free_list.empty()) HIP_CHECK(hipEventCreate(&e)); else { e = free_list.back(); free_list.pop_back(); }
    HIP_CHECK(hipEventRecord(e, st));
    (join ? join_pending : pending).push_back(e);
}
ExchangeTimer::~ExchangeTimer() {
    for (auto e : pending) (void)hipEventDestroy(e);
    for (auto e : join_pending) (void)hipEventDestroy(e);
    for (auto e : free_list) (void)hipEventDestroy(e);
}

void halo_exchange(sb_solver *s, int slot, hipStream_t st) {
    if (!st) st = s->stream;
    if (!halo_slot_active(s, slot)) return;
    // (events inside a capture would become graph nodes without a host-visible time: timing applies to the eager schedules)
    const bool timed = s->xtimer.enabled && !s->capturing;
    // (peer transport: push + unpack kernels ARE the transport, there is no pack step of its own)
    if (timed) s->xtimer.mark(st);
    if (!s->peer.enabled) halo_exchange_pre(s, slot, st);
    if (timed) s->xtimer.mark(st);
    if (s->peer.enabled) halo_exchange_pre(s, slot, st);
    if (!s->peer.enabled) {
        NCCL_CHECK(rccl().GroupStart());
        try {
            halo_exchange_calls(s, slot, st);
        } catch (...) {
            (void)rccl().GroupEnd();      // never leave the group open behind an error
            throw;
        }
        NCCL_CHECK(rccl().GroupEnd());
    }
    halo_exchange_post(s, slot, st);
    if (timed) s->xtimer.mark(st);
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
    // (a tiling with 8-byte wide-packed slots likewise runs 256-lane workgroups in every launch)
    const bool narrow = small && (D.packed_lanes ? D.packed_lanes == sbk::kLanePackLanes : (s->tile_lanes ? s->tile_lanes == sbk::kNarrowTileThreads : tile_end - tile_begin >= s->narrow_min_tiles));
    // tiles with tets / hinges: optionally 8 waves, so that a group's wave slots (16 four-lane constraints or 64 springs each) fit one row
    const bool quad8 = D.has_quads && s->quad_lanes == sbk::kQuadTileThreads;
    // spring-only small tiles on 8 waves (one particle per lane in the load / MARK / store phases; the rounds use half the lanes) while
    // every workgroup of the launch is resident even at that width (4 per compute unit): 64^3 0.1244 -> 0.1213, 48^3 0.1027 -> 0.1005 ms
    // per tick; 96^3 (1 728 tiles) 0.206 -> 0.230, so only launches of at most kWide8MaxTiles (profiles/r03o_lanes512_small_cubes.txt)
    constexpr int kWide8MaxTiles = sbk::kWide8MaxTiles;
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
    else if (it < substeps && kin) launch_tile<5>(s, D, tile_begin, tile_end);        // (the fused first kernel of a tick, with kinematic targets: a T0 launch, owned particles only)
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
//
// The tick is first written down as a PROGRAM -- the launches and exchanges in order, tile ranges by name -- and then run. Every rank of a
// partitioned solver has the same program (the phases are a property of the plan, the schedule is the same on every rank; only what a
// range NAME means differs from rank to rank), so a host thread that drives several ranks can walk it step by step across them and
// issue the RCCL calls of one exchange for all of them together (group.hip).
std::vector<TickStep> tick_program(const sb_solver *s, int substeps, bool fused_first, bool defer_last, bool kin) {
    std::vector<TickStep> P;
    const bool two = s->plan->plan.tiling;
    auto tile = [&](int it, TileRange rg, bool k = false) {
        // the first kernel of a tick that also finishes the previous one is an ordinary mid-tick kernel on T0: an even, interior step index
        const bool ff = it == 0 && fused_first;
        P.push_back(TickStep{StepKind::Tile, rg, k && ff, ff ? 2 : it, ff ? 4 : substeps, 0});
    };
    if (s->overlap_halo) {
        // Overlapped schedule (opt-in): a T0 kernel runs its boundary tiles first; the ghost exchange for the following T1 kernel then
        // travels on comm_stream beside the T0 interior tiles AND the T1 interior tiles; the T1 tiles that hold a ghost or a sent
        // particle run last, after the exchange. The interior tiles of either tiling touch none of the particles the pack kernel
        // reads or the unpack kernel writes (build_device).
        for (int it = 0; it <= substeps; ++it) {
            if (it == substeps && defer_last) break;       // (lazy tick boundary, as in the serialised schedule below)
            if (it & 1) {
                tile(it, TileRange::T1Interior);
                P.push_back(TickStep{StepKind::JoinExchange, TileRange::All, false, it, substeps, 1});
                tile(it, TileRange::T1Boundary);
            } else {
                tile(it, TileRange::T0Boundary);
                if (it < substeps) P.push_back(TickStep{StepKind::ForkExchange, TileRange::All, false, it, substeps, 1});
                tile(it, TileRange::T0Interior);
            }
        }
        return P;
    }
    for (int it = 0; it <= substeps; ++it) {
        if (it == substeps && defer_last) break;
        if (two && (it & 1)) P.push_back(TickStep{StepKind::Exchange, TileRange::All, false, it, substeps, 1});
        tile(it, TileRange::All, kin);
        if (it == substeps) break;
        for (size_t ly = 0; ly < s->t2_layer_range.size(); ++ly) {
            P.push_back(TickStep{StepKind::Exchange, TileRange::All, false, it, substeps, 2 + (int)s->gcolours.size() + (int)ly});
            P.push_back(TickStep{StepKind::T2Layer, TileRange::All, false, it, substeps, (int)ly});
        }
        for (size_t gc = 0; gc < s->gcolours.size(); ++gc) {
            P.push_back(TickStep{StepKind::Exchange, TileRange::All, false, it, substeps, 2 + (int)gc});
            P.push_back(TickStep{StepKind::GColour, TileRange::All, false, it, substeps, (int)gc});
        }
    }
    return P;
}

// One step of the program on one rank. Exchange steps: the whole exchange (a group that walks several ranks takes them apart itself).
void run_step(sb_solver *s, const TickStep &st, LaunchTimer *lt) {
    switch (st.kind) {
    case StepKind::Tile: {
        DevTiling &T0 = s->tiling[0], &T1 = s->tiling[1];
        int b = 0, e = -1;
        if (st.range == TileRange::T0Boundary) { b = 0; e = T0.n_boundary; }
        else if (st.range == TileRange::T0Interior) { b = T0.n_boundary; e = T0.n_tiles; }
        else if (st.range == TileRange::T1Interior) { b = 0; e = T1.n_tiles - T1.n_boundary; }
        else if (st.range == TileRange::T1Boundary) { b = T1.n_tiles - T1.n_boundary; e = T1.n_tiles; }
        launch_tick_kernel(s, st.it, st.substeps, lt, b, e, st.kin);
        break;
    }
    case StepKind::T2Layer: launch_t2_layer(s, st.index, lt, false); break;
    case StepKind::GColour: launch_gcolour(s, st.index, lt); break;
    case StepKind::Exchange: halo_exchange(s, st.index); break;
    case StepKind::ForkExchange:
        HIP_CHECK(hipEventRecord(s->ev_boundary, s->stream));
        HIP_CHECK(hipStreamWaitEvent(s->comm_stream, s->ev_boundary, 0));
        halo_exchange(s, st.index, s->comm_stream);
        HIP_CHECK(hipEventRecord(s->ev_halo, s->comm_stream));
        break;
    case StepKind::JoinExchange: {
        // (timed: what the compute stream actually WAITS for an overlapped exchange -- the part the interior tiles did not hide)
        const bool timed = s->xtimer.enabled && !s->capturing;
        if (timed) s->xtimer.mark(s->stream, true);
        HIP_CHECK(hipStreamWaitEvent(s->stream, s->ev_halo, 0));
        if (timed) s->xtimer.mark(s->stream, true);
        break;
    }
    }
}

void enqueue_substeps(sb_solver *s, int substeps, LaunchTimer *lt = nullptr, bool fused_first = false, bool defer_last = false, bool kin = false) {
    for (const TickStep &st : tick_program(s, substeps, fused_first, defer_last, kin)) run_step(s, st, lt);
    HIP_CHECK(hipGetLastError());
}

// Launch the deferred last kernel of the previous tick (uses the tick parameters still on the device).
// Pending kinematic targets onto `dst` (the positions, or the peek's side array): scatter kernel over the pinned host table.
void scatter_kinematic(sb_solver *s, float *dst) {
    const int q = s->kin_pending, count = s->kin_pending_count;
    hipLaunchKernelGGL(sbk::kinematic_scatter_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s->stream, dst, s->d_kin_idx[q], s->d_kin_pos[q], count);
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
                       s->d_kin_idx[q], s->d_kin_pos[q], count);
    HIP_CHECK(hipGetLastError());
    retire_kinematic(s);
}

// ---- peek: tick-end positions without completing the tick ------------------------------------------------------------------------
// While the last kernel K_S of a tick is deferred, the positions the tick ends with are K_S's rounds + collide applied to the state
// in memory. tile_kernel<4> computes exactly that -- same tiles, same programs, same inputs, hence the same bits -- into d_peek and
// leaves the state alone, so the next sb_step still fuses K_S with its first kernel (one launch instead of two) and a render
// readback after every tick no longer costs a whole extra pass over the mesh. On every rank of a partitioned solver too (round 4): the
// held-back kernel runs on T0 tiles, which hold owned particles only and have no exchange in front of them.
bool can_peek(const sb_solver *s) {
    if (!s->peek_enabled || !s->deferred) return false;
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
        if (s->n_peek_tiles <= 0) return;       // (no tile holds a wanted particle: nothing is launched, nothing is counted)
        launch_tile<4>(s, D, 0, s->n_peek_tiles, sbk::kHaloNone, s->peek_tiles.p);
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

}  // namespace sbi

using namespace sbi;

extern "C" {

}  // extern "C"  (the pieces of a tick are shared with group.hip)

namespace sbi {

// What the next tick will look like, decided from the solver's state alone -- the ranks of a partitioned solver are in the same state,
// so they all decide the same: `fuse` = its first kernel also finishes the previous tick (lazy tick boundary), `defer_last` = its own
// last kernel is held back, `kin` = pending kinematic targets travel inside the fused kernel.
TickShape begin_tick(sb_solver *s, float dt, int substeps) {
    check_peer_error(s);       // a halo wait of an earlier tick gave up: do not pile further ticks on stale ghosts
    if (s->peer.enabled && s->desc.world > 1 && !s->peer.linked) peer_link(s);     // (reads the neighbours' offset tables: not inside a capture)
    const sbk::TickParams tp_new = tick_params(s, dt, substeps);
    TickShape t{};
    t.substeps = substeps;
    t.defer_last = s->lazy_tick && (!s->plan->plan.tiling || (substeps & 1) == 0);
    // (pending kinematic targets ride in the fused kernel of the serialised schedules; the overlapped ones launch that kernel in two
    // pieces and complete the previous tick first instead)
    const bool kin_ok = s->kin_pending < 0 || (s->kin_fuse && !s->overlap_halo);
    t.fuse = s->deferred && t.defer_last && s->deferred_substeps == substeps && s->tp_valid &&
             std::memcmp(&tp_new, &s->tp_host, sizeof(tp_new)) == 0 && kin_ok;
    if (!t.fuse) flush_deferred(s); else ++s->n_fused;
    t.kin = t.fuse && s->kin_pending >= 0;
    if (t.kin) { stage_kinematic_for_fusion(s); ++s->n_kin_fused; }
    upload_tick_params(s, dt, substeps);
    return t;
}
void end_tick(sb_solver *s, const TickShape &t) {
    s->deferred = t.defer_last;
    s->deferred_substeps = t.substeps;
}

// ---- SB_SCHEDULE_AUTO, measured ---------------------------------------------------------------------------------------------------------
// Both eager schedules project the same constraints in the same order on the same operands -- only WHEN the exchange travels differs -- so a
// solver may change between them from tick to tick without a bit moving. The calibration uses that: ticks 0 and 1 run one schedule each
// untimed (first use of the communicator's connections, of the second stream), ticks 2 .. 5 alternate serialised / overlapped between two
// HIP events on the compute stream, and the next sb_step gathers every rank's two sums (one 16-byte all-gather, the only blocking step) and
// keeps the overlapped schedule when its slowest rank beat the serialised schedule's slowest rank by more than 3 %. Every rank decides
// from the same table, hence alike.
constexpr int kCalibWarmTicks = 2, kCalibTimedTicks = 4;
void calibrate_before_tick(sb_solver *s) {
    auto &C = s->calib;
    const int k = C.tick;
    if (k < kCalibWarmTicks + kCalibTimedTicks) {
        s->overlap_halo = (k & 1) != 0;
        if (k >= kCalibWarmTicks) {
            const int q = 2 * (k - kCalibWarmTicks);
            if (!C.ev[q]) { HIP_CHECK(hipEventCreate(&C.ev[q])); HIP_CHECK(hipEventCreate(&C.ev[q + 1])); }
            HIP_CHECK(hipEventRecord(C.ev[q], s->stream));
        }
        return;
    }
    // decide
    HIP_CHECK(hipStreamSynchronize(s->stream));
    float mine[2] = {0.0f, 0.0f};
    for (int t = 0; t < kCalibTimedTicks; ++t) {
        float ms = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms, C.ev[2 * t], C.ev[2 * t + 1]));
        mine[t & 1] += ms; ++C.n[t & 1];
    }
    const int W = s->loopback ? 1 : s->desc.world, me = s->loopback ? 0 : s->desc.rank;
    std::vector<float> all((size_t)2 * W, 0.0f);
    {
        DevBuf<float> d_all; int64_t acct = 0;
        d_all.alloc((size_t)2 * W, acct);
        HIP_CHECK(hipMemcpy(d_all.p + 2 * (size_t)me, mine, sizeof(mine), hipMemcpyHostToDevice));
        NCCL_CHECK(rccl().AllGather(d_all.p + 2 * (size_t)me, d_all.p, sizeof(mine), ncclUint8, s->comm, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        HIP_CHECK(hipMemcpy(all.data(), d_all.p, all.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    float worst[2] = {0.0f, 0.0f};
    for (int r = 0; r < W; ++r) for (int q = 0; q < 2; ++q) worst[q] = std::max(worst[q], all[2 * (size_t)r + (size_t)q]);
    for (int q = 0; q < 2; ++q) C.decided_ms[q] = C.n[q] ? (double)worst[q] / C.n[q] : 0.0;
    bool overlap = worst[1] < 0.97f * worst[0];
    if (s->tune_flags & SB_TUNE_AUTO_PREFER_OVERLAP) overlap = true;
    s->overlap_halo = overlap;
    s->schedule = overlap ? SB_SCHEDULE_OVERLAP_EAGER : SB_SCHEDULE_SERIAL_EAGER;
    C.state = 2;
}
void calibrate_after_tick(sb_solver *s) {
    auto &C = s->calib;
    const int k = C.tick++;
    if (k >= kCalibWarmTicks && k < kCalibWarmTicks + kCalibTimedTicks) HIP_CHECK(hipEventRecord(C.ev[2 * (k - kCalibWarmTicks) + 1], s->stream));
}

}  // namespace sbi

extern "C" {

int sb_step(sb_solver *s, float dt, int32_t substeps) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_step: null handle");
    if (!s->finalized) return fail(SB_ERR_STATE, "sb_step before sb_finalize");
    if (!(dt > 0.0f) || substeps <= 0) return fail(SB_ERR_INVALID_ARG, "sb_step: dt and substeps must be positive");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        const bool calibrating = s->calib.state == 1;      // SB_SCHEDULE_AUTO: which eager schedule this tick runs (or: decide now)
        if (calibrating) calibrate_before_tick(s);
        const TickShape t = begin_tick(s, dt, substeps);
        // world > 1: the exchange inside a captured graph is opt-in (SB_SCHEDULE_*_GRAPH), see DESIGN.md §7
        const bool graph_ok = s->desc.use_graph && (s->desc.world == 1 || s->graph_rccl);     // the overlapped schedule forks onto comm_stream inside the capture
        if (!graph_ok) {
            enqueue_substeps(s, substeps, nullptr, t.fuse, t.defer_last, t.kin);
        } else {
            const int key = substeps * 8 + (t.fuse ? 1 : 0) + (t.defer_last ? 2 : 0) + (t.kin ? 4 : 0);
            auto it = s->graphs.find(key);
            if (it == s->graphs.end()) {
                hipGraph_t g = nullptr;
                HIP_CHECK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
                s->capturing = true;
                try {
                    enqueue_substeps(s, substeps, nullptr, t.fuse, t.defer_last, t.kin);
                } catch (...) {
                    s->capturing = false;
                    (void)hipStreamEndCapture(s->stream, &g);
                    if (g) (void)hipGraphDestroy(g);
                    throw;
                }
                s->capturing = false;
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
        end_tick(s, t);
        if (calibrating && s->calib.state == 1) calibrate_after_tick(s);
        return SB_OK;
    });
}

// Exchange timing (softbody_debug.h): sums over the exchanges of the eager ticks enqueued since the last read.
int sb_debug_exchange_timing(sb_solver *s, int32_t enabled) {
    if (!s) return fail(SB_ERR_INVALID_ARG, "sb_debug_exchange_timing: null handle");
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        if (!enabled && s->xtimer.enabled) {       // drop what was not read
            HIP_CHECK(hipStreamSynchronize(s->stream));
            if (s->comm_stream) HIP_CHECK(hipStreamSynchronize(s->comm_stream));
            s->xtimer.free_list.insert(s->xtimer.free_list.end(), s->xtimer.pending.begin(), s->xtimer.pending.end());
            s->xtimer.free_list.insert(s->xtimer.free_list.end(), s->xtimer.join_pending.begin(), s->xtimer.join_pending.end());
            s->xtimer.pending.clear(); s->xtimer.join_pending.clear();
        }
        s->xtimer.enabled = enabled != 0;
        return SB_OK;
    });
}
int sb_debug_exchange_timing_read(sb_solver *s, sb_exchange_timing *out) {
    if (!s || !out) return fail(SB_ERR_INVALID_ARG, "sb_debug_exchange_timing_read: null argument");
    std::memset(out, 0, sizeof(*out));
    return guarded([&]() -> int {
        int rc = set_device(s); if (rc) return rc;
        HIP_CHECK(hipStreamSynchronize(s->stream));
        if (s->comm_stream) HIP_CHECK(hipStreamSynchronize(s->comm_stream));
        auto &X = s->xtimer;
        for (size_t k = 0; k + 3 <= X.pending.size(); k += 3) {
            float a = 0.0f, b = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&a, X.pending[k], X.pending[k + 1]));
            HIP_CHECK(hipEventElapsedTime(&b, X.pending[k + 1], X.pending[k + 2]));
            out->pack_ms += a; out->transport_ms += b; out->total_ms += (double)a + (double)b;
            ++out->exchanges;
        }
        if (X.join_pending.empty()) out->exposed_wait_ms = out->total_ms;       // a serialised exchange is exposed in full
        for (size_t k = 0; k + 2 <= X.join_pending.size(); k += 2) {
            float w = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&w, X.join_pending[k], X.join_pending[k + 1]));
            out->exposed_wait_ms += w;
        }
        X.free_list.insert(X.free_list.end(), X.pending.begin(), X.pending.end());
        X.free_list.insert(X.free_list.end(), X.join_pending.begin(), X.join_pending.end());
        X.pending.clear(); X.join_pending.clear();
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

}  // extern "C"
