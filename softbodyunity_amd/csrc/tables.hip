// tables.hip — from the planner's output to the tables the kernels read: particle state, tile descriptors and constraint streams, global colours, halo lists, the peer mailbox
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"

namespace sbi {

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
                    int64_t m_v, int64_t m_b, const sb_domain *domain) {
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
        if (few && !(s->tune_flags & SB_TUNE_NO_MASS_PALETTE)) {
            std::vector<uint8_t> w8(hw.size());
            sbp::parallel_for_chunks((int64_t)hw.size(), 1 << 20, [&](int64_t, int64_t lb, int64_t le) {
                for (int64_t l = lb; l < le; ++l) w8[(size_t)l] = (uint8_t)(std::lower_bound(uniq.begin(), uniq.end(), vals[(size_t)l]) - uniq.begin());
            });
            for (size_t k = 0; k < uniq.size(); ++k) std::memcpy(&pal[k], &uniq[k], 4);
            s->d_w8.upload(w8, s->dev_bytes);
            s->w_palette = true;
            s->w_uniform = uniq.size() == 1 && !(s->tune_flags & SB_TUNE_NO_UNIFORM_MASS);
        }
        s->d_wpal.upload(pal, s->dev_bytes);
    }
    s->d_vel.upload(hv, s->dev_bytes);
    s->d_prev.alloc((size_t)s->n_local * 3, s->dev_bytes, (size_t)s->prev_offset_bytes / sizeof(float));
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
        const bool no_palette = (s->tune_flags & SB_TUNE_NO_PALETTE) != 0;
        // meshes with tets / hinges: per-wave step lists beside the group words (springs-only meshes never run the kernels that read them)
        const bool emit_items = (!s->vol_rest.empty() || !s->bend_rest.empty()) && !(s->tune_flags & SB_TUNE_NO_WAVE_ITEMS);
        const int item_waves = s->quad_lanes / 64;
        D.item_waves = emit_items ? item_waves : 0;
        // Lane-packed slots (kernels.hip.hpp kLanePack*): only where every launch of the tiling is known to run 128-lane workgroups --
        // a single-rank solver (no boundary / interior ranges) whose launches oversubscribe the chip (launch_tile: narrow) -- and the
        // mesh has springs only. Which TILES then qualify is decided tile by tile below.
        D.packed_lanes = 0;
        bool all_small = true;       // (a tiling with a tile above 512 particles launches the 1 024-particle kernels)
        for (size_t ci = 0; ci < n_plan_tiles; ++ci) all_small = all_small && G.tiles[LT.tile_ids[ci]].n_local <= sbk::kSmallTile;
        // (round 4: also for the ranks of a partitioned solver -- a packed tiling forces its width on every launch, the boundary / interior
        // pieces of the overlapped schedule included -- and, in the 8-byte form for 256-lane workgroups (kWidePack*), for tilings between the
        // 512-lane and the 128-lane regimes: 128^3 on one GPU, a rank's 4 096 tiles of 256^3 on 8)
        const bool packable = tl < 2 && all_small && s->vol_rest.empty() && s->bend_rest.empty() && !no_palette;
        if (packable && !(s->tune_flags & SB_TUNE_NO_LANE_PACK) && (s->tile_lanes == 0 || s->tile_lanes == sbk::kLanePackLanes) &&
            (int64_t)packs.size() >= (int64_t)s->narrow_min_tiles)
            D.packed_lanes = sbk::kLanePackLanes;
        else if (packable && !(s->tune_flags & SB_TUNE_NO_WIDE_SLOTS) && (s->tile_lanes == 0 || s->tile_lanes == sbk::kWidePackLanes) &&
                 (int64_t)packs.size() > (int64_t)sbk::kWide8MaxTiles && (int64_t)packs.size() < (int64_t)s->narrow_min_tiles && sbk::kRegRoundsWide >= sbk::kLanePackRounds)
            D.packed_lanes = sbk::kWidePackLanes;
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
            bool lane_pack = D.packed_lanes == sbk::kLanePackLanes && sbk::kLanePackDecodable &&
                             (compact ? (int)pal.size() <= sbk::kLanePackMaxPalette : (sbk::kRegFullSlots && !s->w_palette && n_dist > 0)) &&
                             td.n_local <= sbk::kSmallTile && !prog.empty() && (int)prog.size() <= sbk::kLanePackRounds;
            for (const PackRound &R : prog) lane_pack = lane_pack && R.cnt[1] == 0 && R.cnt[2] == 0 && R.cnt[0] <= 2 * sbk::kLanePackLanes;
            bool wide_pack = D.packed_lanes == sbk::kWidePackLanes && compact && (int)pal.size() <= sbk::kLanePackMaxPalette &&
                             td.n_local <= sbk::kSmallTile && !prog.empty() && (int)prog.size() <= sbk::kLanePackRounds;
            {   // (the packed form is 2 KiB whatever the tile holds: only where that is LESS than 4 bytes per slot -- full-size tiles, not rim packs)
                uint32_t unpacked = 0;
                for (const PackRound &R : prog) { wide_pack = wide_pack && R.cnt[1] == 0 && R.cnt[2] == 0 && R.cnt[0] <= sbk::kWidePackLanes; unpacked += ((uint32_t)R.cnt[0] + 3u) & ~3u; }
                wide_pack = wide_pack && unpacked > sbk::kWidePackDwords;
            }
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
            else if (wide_pack) {
                // one 8-byte word per lane: three 21-bit fields {i:9 | j:9 | palette:3}, field r = slot `lane` of round r
                std::vector<uint32_t> words(sbk::kWidePackDwords, 0u);
                for (size_t r = 0; r < prog.size(); ++r) {
                    int32_t c = 0;
                    for (const Part &pt : prog[r].parts) {
                        const uint32_t b = (uint32_t)base[pt.member], b2 = b | (b << 16);
                        for (int64_t k = pt.first_d; k < pt.first_d + pt.cnt[0]; ++k, ++c) {
                            const uint32_t idx = G.t_dist[k] + b2, rb = fbits(s->dist_rest[G.t_dist_id[k]]);
                            const uint32_t pi = (uint32_t)(std::lower_bound(pal.begin(), pal.end(), rb) - pal.begin());
                            const uint32_t i = idx & 0xffffu, j = idx >> 16;
                            if (i > 511u || j > 511u || pi > 7u || c >= sbk::kWidePackLanes) throw std::runtime_error("internal: wide-packed slot out of range");
                            const uint64_t f = (uint64_t)(i | (j << 9) | (pi << 18)) << (sbk::kLanePackFieldBits * (int)r);
                            words[2 * (size_t)c] |= (uint32_t)f;
                            words[2 * (size_t)c + 1] |= (uint32_t)(f >> 32);
                        }
                    }
                }
                stream.insert(stream.end(), words.begin(), words.end());
                td.packed_lanes = (uint32_t)sbk::kWidePackLanes;
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
        if (!(s->tune_flags & SB_TUNE_NO_COST_ORDER)) {
            constexpr int32_t kCostOrderMaxTiles = 2048;
            const int32_t n = (int32_t)tiles.size();
            std::vector<std::pair<int32_t, int32_t>> ranges;
            if (tl == 2) ranges = s->t2_layer_range;
            else if ((s->overlap_halo || s->calib.state == 1) && D.n_boundary > 0 && D.n_boundary < n) {
                // (SB_SCHEDULE_AUTO's calibration runs ticks of BOTH eager schedules on these tables: the boundary / interior pieces must stay
                // pieces; a whole-tiling launch of the serialised ticks then merely finds its tiles placed for two launches)
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
        if (s->win_dwords_cap > 0) D.win_dwords = std::max(1024, std::min(D.win_dwords, s->win_dwords_cap) & ~3);   // tuning experiments (sb_tuning.win_dwords)
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
        s->halos.size() > 1 && !(s->tune_flags & SB_TUNE_NO_FUSED_UNPACK)) {
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
        header[3] = plan_shape(s);                    // compared by the neighbours too: the mailbox LAYOUT follows the number of halo slots
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
        // (SB_TUNE_PEER_COARSE: an ordinary cached allocation -- timing experiments on ONE device only; between devices the flags and
        // segments must be uncached for the stores of one agent to reach the loads of another without cache maintenance)
        if (!(s->tune_flags & SB_TUNE_PEER_COARSE) && hipExtMallocWithFlags(&mb, PS.bytes, hipDeviceMallocFinegrained) == hipSuccess) PS.fine_grained = true;
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

}  // namespace sbi
