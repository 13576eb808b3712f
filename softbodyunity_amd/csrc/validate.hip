// validate.hip — sb_debug_validate: the uploaded tables re-read by a GPU kernel with the tile kernels' own decoding rules
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"
#include "validate_kernels.hip.hpp"

using namespace sbi;

extern "C" {

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
                        const size_t lane_dwords = td.packed_lanes == (uint32_t)sbk::kWidePackLanes ? 2 : 4;
                        HIP_CHECK(hipMemcpy(base + lane_dwords, base, lane_dwords * sizeof(uint32_t), hipMemcpyDeviceToDevice));
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

}  // extern "C"
