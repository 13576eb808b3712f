// validate_kernels.hip.hpp — the table validator (sb_debug_validate; SURVEY.md §5 race detection)
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
#pragma once
#include "kernel_types.hpp"

namespace sbk {

// ---- table validator (debug entry sb_debug_validate; SURVEY.md 5 "race detection") ---------------------------------------------------
// The only race this design can have is two constraints of one group (or two tiles of one launch) touching the same particle. The
// planner's output is checked on the host (tests/test_plan.py); THIS kernel checks what the tile kernels actually read -- the uploaded
// descriptors, run tables / particle lists, group words, dictionary-coded or full slots, four-vertex slots and wave items, after
// packing, lane dealing and cost ordering -- with the tile kernels' own decoding rules. One workgroup per device tile.
__global__ __launch_bounds__(kValidateThreads) void validate_tiles_kernel(const TileDesc *tiles, int tile_begin, const int2 *runs_overflow, const uint32_t *stream,
                                                                         const int32_t *gather, int use_gather, int n_particles, int item_waves,
                                                                         int32_t *owner, ValidateCounters *out) {
    __shared__ uint32_t bitmap[kLargeTile / 32];
    __shared__ unsigned int s_err[6];
    __shared__ int s_first_group, s_first_kind;
    const int tid = threadIdx.x;
    const int tile = tile_begin + (int)blockIdx.x;
    const TileDesc td = tiles[tile];
    const uint32_t *ts = stream + td.s_begin;
    if (tid < 6) s_err[tid] = 0;
    if (tid == 0) { s_first_group = -1; s_first_kind = -1; }
    __syncthreads();
    auto flag = [&](int kind, int group) {
        atomicAdd(&s_err[kind], 1u);
        if (atomicCAS(&s_first_kind, -1, kind) == -1) s_first_group = group;
    };
    const bool size_ok = td.n_local >= 0 && td.n_local <= kLargeTile && td.n_rounds >= 0 && td.s_hdr <= td.s_len;
    if (!size_ok) { if (tid == 0) flag(4, -1); }
    // ---- particles: the tile kernels' own lookup (runs sorted by first local index, or the explicit list of a sparse tile) ----
    if (size_ok) {
        if (!use_gather && tid == 0) {
            bool ok = td.run_count >= (td.n_local > 0 ? 1 : 0);
            int prev_y = -1;
            for (int r = 0; r < td.run_count && ok; ++r) {
                const int2 rn = r < kInlineRuns ? td.runs[r] : runs_overflow[td.run_overflow + r - kInlineRuns];
                ok = rn.y > prev_y && rn.y < td.n_local && rn.x >= 0 && (r > 0 || rn.y == 0);
                prev_y = rn.y;
            }
            if (!ok) flag(4, -1);
        }
        for (int l = tid; l < td.n_local; l += kValidateThreads) {
            int g;
            if (use_gather) g = gather[td.gather_begin + l];
            else {
                g = -1;
                for (int r = 0; r < td.run_count; ++r) {
                    const int2 rn = r < kInlineRuns ? td.runs[r] : runs_overflow[td.run_overflow + r - kInlineRuns];
                    if (rn.y <= l) g = rn.x + (l - rn.y);
                }
            }
            if (g < 0 || g >= n_particles) flag(0, -1);
            else if (atomicCAS(&owner[g], -1, tile) != -1) flag(2, -1);
        }
    }
    // ---- groups: walk the data exactly as the kernels do; inside a group every particle at most once ----
    unsigned int tot[3] = {0, 0, 0};
    uint32_t off = td.s_hdr;
    bool walk_ok = size_ok;
    if (size_ok && td.packed_lanes) {
        // lane-packed slots: 128 lanes (kLanePack*) -- one 16-byte word per lane, field 2 r + u = slot lane + 128 u of round r; 256 lanes
        // (kWidePack*) -- one 8-byte word per lane, field r = slot lane of round r
        const uint32_t P = td.packed_lanes;
        const bool wide = P == (uint32_t)kWidePackLanes;
        const uint32_t lane_dwords = wide ? 2u : 4u;
        // (n_pal == 0: the full form of the 128-lane packing, rest lengths behind the index words)
        if ((P != (uint32_t)kLanePackLanes && !wide) || td.n_rounds > kLanePackRounds || td.n_pal < 0 || td.n_pal > kLanePackMaxPalette || td.n_local > kSmallTile ||
            (wide && td.n_pal == 0) ||
            td.s_hdr + (wide ? kWidePackDwords : (td.n_pal > 0 ? kLanePackDwordsCompact : kLanePackDwordsFull)) > td.s_len) { if (tid == 0) flag(3, -1); walk_ok = false; }
        for (int r = 0; walk_ok && r < td.n_rounds; ++r) {
            const uint32_t w = ts[r];
            const uint32_t cnt = w & 1023u;
            if (cnt > (wide ? P : 2u * P) || ((w >> 10) & 0xfffffu) != 0u) { if (tid == 0) flag(3, r); walk_ok = false; break; }
            for (int q = tid; q < kLargeTile / 32; q += kValidateThreads) bitmap[q] = 0;
            __syncthreads();
            for (uint32_t c = tid; c < cnt; c += kValidateThreads) {
                const uint32_t lane = c % P, u = c / P, bit = (uint32_t)kLanePackFieldBits * (wide ? (uint32_t)r : 2u * (uint32_t)r + u), w0 = bit >> 5, sh = bit & 31u;
                const uint32_t *wd = ts + td.s_hdr + lane_dwords * lane;
                uint64_t two = (uint64_t)wd[w0] | ((uint64_t)(w0 + 1 < lane_dwords ? wd[w0 + 1] : 0u) << 32);
                const uint32_t f = (uint32_t)(two >> sh) & ((1u << kLanePackFieldBits) - 1u);
                const uint32_t p0 = f & 511u, p1 = (f >> 9) & 511u;
                if ((f >> 18) >= (uint32_t)max(td.n_pal, 1)) flag(0, r);
                for (int e = 0; e < 2; ++e) {
                    const uint32_t p = e ? p1 : p0;
                    if (p >= (uint32_t)td.n_local) flag(0, r);
                    else if (atomicOr(&bitmap[p >> 5], 1u << (p & 31)) & (1u << (p & 31))) flag(1, r);
                }
            }
            __syncthreads();
            tot[0] += cnt;
        }
    } else
    for (int r = 0; walk_ok && r < td.n_rounds; ++r) {
        const uint32_t w = ts[r];
        const uint32_t cnt = w & 1023u, n_vol = (w >> 10) & 1023u, n_bend = (w >> 20) & 1023u;
        const bool compact = (w >> 30) & 1u;
        const uint32_t dsize = compact ? ((cnt + 3u) & ~3u) : ((2u * cnt + 3u) & ~3u);
        const uint32_t size = dsize + 4u * (n_vol + n_bend);
        if (off + size > td.s_len || cnt > (uint32_t)kRoundSlots || n_vol > (uint32_t)kRoundSlots || n_bend > (uint32_t)kRoundSlots ||
            (compact && td.n_pal <= 0)) { if (tid == 0) flag(3, r); walk_ok = false; break; }
        for (int q = tid; q < kLargeTile / 32; q += kValidateThreads) bitmap[q] = 0;
        __syncthreads();
        auto mark = [&](uint32_t p) {
            if (p >= (uint32_t)td.n_local) { flag(0, r); return; }
            if (atomicOr(&bitmap[p >> 5], 1u << (p & 31)) & (1u << (p & 31))) flag(1, r);
        };
        for (uint32_t c = tid; c < cnt; c += kValidateThreads) {
            uint32_t i, k;
            if (compact) { const uint32_t e = ts[off + c]; i = e & 0xfffu; k = (e >> 12) & 0xfffu; if ((e >> 24) >= (uint32_t)td.n_pal) flag(0, r); }
            else { const uint32_t e = ts[off + 2 * c]; i = e & 0xffffu; k = e >> 16; }
            mark(i); mark(k);
        }
        for (uint32_t c = tid; c < n_vol + n_bend; c += kValidateThreads) {
            const uint32_t e0 = ts[off + dsize + 4 * c], e1 = ts[off + dsize + 4 * c + 1];
            mark(e0 & 0xffffu); mark(e0 >> 16); mark(e1 & 0xffffu); mark(e1 >> 16);
        }
        __syncthreads();
        tot[0] += cnt; tot[1] += n_vol; tot[2] += n_bend;
        off += size;
    }
    // ---- wave items: the same work dealt to the waves ahead of time; must add up to the group words and end every group on every wave ----
    if (walk_ok && td.n_steps > 0 && item_waves > 0 && tid == 0) {
        unsigned int it_tot[3] = {0, 0, 0};
        bool ok = td.s_items + (uint32_t)(item_waves * td.n_steps) <= td.s_hdr;
        for (int wv = 0; ok && wv < item_waves; ++wv) {
            int barriers = 0;
            for (int st = 0; st < td.n_steps; ++st) {
                const uint32_t it = ts[td.s_items + (uint32_t)(wv * td.n_steps + st)];
                const uint32_t type = it & 7u, c = (it >> kItemCountShift) & 127u, o = it >> kItemOffsetShift;
                if (it & (1u << kItemBarrierBit)) ++barriers;
                if (type == kItemIdle) continue;
                const uint32_t bytes4 = type == kItemDistCompact ? c : (type == kItemDistFull ? 2u * c : 4u * c);
                if (type > kItemBending || td.s_hdr + o + bytes4 > td.s_len) { ok = false; break; }
                it_tot[type == kItemVolume ? 1 : (type == kItemBending ? 2 : 0)] += c;
            }
            ok = ok && barriers == td.n_rounds;
        }
        if (!ok || it_tot[0] != tot[0] || it_tot[1] != tot[1] || it_tot[2] != tot[2]) flag(5, -1);
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&out->tiles, 1ull);
        atomicAdd(&out->groups, (unsigned long long)(walk_ok ? td.n_rounds : 0));
        atomicAdd(&out->constraints, (unsigned long long)tot[0] + tot[1] + tot[2]);
        bool any = false;
        for (int k = 0; k < 6; ++k) if (s_err[k]) { atomicAdd(&out->errors[k], s_err[k]); any = true; }
        if (any && atomicCAS(&out->first[0], -1, tile) == -1) { out->first[1] = s_first_group; out->first[2] = s_first_kind; }
    }
}
// One global colour: no particle twice (owner[] cleared to -1 before every colour).
__global__ __launch_bounds__(256) void validate_gcolour_kernel(const int32_t *idx, int per_constraint, int count, int n_particles, int colour, int32_t *owner,
                                                               ValidateCounters *out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    for (int c = 0; c < per_constraint; ++c) {
        const int g = idx[(size_t)per_constraint * k + c];
        int kind = -1;
        if (g < 0 || g >= n_particles) kind = 0;
        else if (atomicCAS(&owner[g], -1, k) != -1) kind = 1;
        if (kind >= 0) {
            atomicAdd(&out->errors[kind], 1u);
            if (atomicCAS(&out->first[0], -1, -2 - colour) == -1) { out->first[1] = k; out->first[2] = kind; }
        }
    }
    if (threadIdx.x == 0) atomicAdd(&out->constraints, (unsigned long long)min(256, count - (int)blockIdx.x * 256));
}

}  // namespace sbk
