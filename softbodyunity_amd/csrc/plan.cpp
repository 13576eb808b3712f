// plan.cpp — see plan.hpp. Pure host C++; deterministic (no hashing by address, no threads).
#include "plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace sbp {
namespace {

struct Mask128 {
    uint64_t lo = 0, hi = 0;
};
inline Mask128 operator|(const Mask128 &a, const Mask128 &b) { return {a.lo | b.lo, a.hi | b.hi}; }
inline int first_free(const Mask128 &m) {
    if (~m.lo) return __builtin_ctzll(~m.lo);
    if (~m.hi) return 64 + __builtin_ctzll(~m.hi);
    return -1;
}
inline void set_bit(Mask128 &m, int c) {
    if (c < 64) m.lo |= 1ull << c; else m.hi |= 1ull << (c - 64);
}

const int kVerts[3] = {2, 4, 4};

struct Cons {  // view over the three input arrays
    const Input *in;
    const int32_t *idx(int type, int64_t id) const {
        return type == 0 ? in->dist_ij + 2 * id : (type == 1 ? in->vol + 4 * id : in->bend + 4 * id);
    }
    int64_t count(int type) const { return type == 0 ? in->m_d : (type == 1 ? in->m_v : in->m_b); }
};

void resolve_dims(int world, const float ext[3], const int want[3], int dims[3]) {
    if (want[0] > 0 && want[1] > 0 && want[2] > 0) {
        if ((int64_t)want[0] * want[1] * want[2] != world) throw std::runtime_error("part_dims product != world");
        dims[0] = want[0]; dims[1] = want[1]; dims[2] = want[2];
        return;
    }
    dims[0] = dims[1] = dims[2] = 1;
    int w = world;
    for (int f = 2; w > 1;) {
        if (w % f) { ++f; continue; }
        w /= f;
        // give the factor to the axis with the largest extent per block (ties -> lowest axis)
        int best = 0; double bv = -1;
        for (int a = 0; a < 3; ++a) { double v = (double)ext[a] / dims[a]; if (v > bv * (1 + 1e-9)) { bv = v; best = a; } }
        dims[best] *= f;
    }
}

// Greedy colouring of a list of constraints over a small local index space. Returns colour per item.
// used: scratch masks indexed by local particle id, must be zero on entry for the touched ids; zeroed on exit.
template <class GetVerts>
int greedy_colour(int count, int nverts, GetVerts get, std::vector<Mask128> &used, std::vector<int> &colour_out) {
    colour_out.resize(count);
    int ncol = 0;
    for (int k = 0; k < count; ++k) {
        const int32_t *v = get(k);
        Mask128 m;
        for (int a = 0; a < nverts; ++a) m = m | used[v[a]];
        int c = first_free(m);
        if (c < 0) { ncol = -1; break; }
        for (int a = 0; a < nverts; ++a) set_bit(used[v[a]], c);
        colour_out[k] = c;
        ncol = std::max(ncol, c + 1);
    }
    for (int k = 0; k < count; ++k) {
        const int32_t *v = get(k);
        for (int a = 0; a < nverts; ++a) used[v[a]] = Mask128();
    }
    return ncol;
}

}  // namespace

void build_plan(const Input &in, const Opts &opts, Plan &P) {
    P = Plan();
    P.opts = opts;
    const int32_t n = in.n;
    if (n <= 0) throw std::runtime_error("no particles");
    if (opts.world < 1 || opts.rank < 0 || opts.rank >= opts.world) throw std::runtime_error("bad rank/world");
    Cons C{&in};
    P.n = n;
    P.m[0] = in.m_d; P.m[1] = in.m_v; P.m[2] = in.m_b;
    for (int t = 0; t < 3; ++t) {
        if (C.count(t) < 0) throw std::runtime_error("negative constraint count");
        if (C.count(t) > 0 && !C.idx(t, 0)) throw std::runtime_error("null constraint array");
        for (int64_t k = 0; k < C.count(t); ++k) {
            const int32_t *v = C.idx(t, k);
            for (int a = 0; a < kVerts[t]; ++a) {
                if (v[a] < 0 || v[a] >= n) throw std::runtime_error("constraint index out of range");
                for (int b = 0; b < a; ++b)
                    if (v[a] == v[b]) throw std::runtime_error("constraint repeats a particle");
            }
        }
    }
    // ---- geometry: spacing estimate, bounding box --------------------------------------------
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int32_t p = 0; p < n; ++p)
        for (int a = 0; a < 3; ++a) {
            double v = in.rest[3 * (int64_t)p + a];
            if (!(v == v) || std::fabs(v) > 1e30) throw std::runtime_error("non-finite rest position");
            lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v);
        }
    double ell = 0;
    {
        double acc = 0; int64_t cnt = 0;
        auto edge = [&](int32_t i, int32_t j) {
            double s = 0;
            for (int a = 0; a < 3; ++a) { double d = (double)in.rest[3 * (int64_t)i + a] - in.rest[3 * (int64_t)j + a]; s += d * d; }
            acc += std::sqrt(s); ++cnt;
        };
        for (int64_t k = 0; k < in.m_d; ++k) edge(in.dist_ij[2 * k], in.dist_ij[2 * k + 1]);
        if (cnt == 0) for (int64_t k = 0; k < in.m_v; ++k) edge(in.vol[4 * k], in.vol[4 * k + 1]);
        if (cnt == 0) for (int64_t k = 0; k < in.m_b; ++k) edge(in.bend[4 * k], in.bend[4 * k + 1]);
        if (cnt > 0) ell = acc / cnt;
        if (!(ell > 0)) {
            double vol = 1; for (int a = 0; a < 3; ++a) vol *= std::max(hi[a] - lo[a], 1e-6);
            ell = std::cbrt(vol / n);
        }
    }
    float ext[3];
    for (int a = 0; a < 3; ++a) ext[a] = (float)(hi[a] - lo[a] + ell);
    resolve_dims(opts.world, ext, opts.dims, P.dims);

    const bool tiling = opts.tile_particles > 0;
    const int target = tiling ? opts.tile_particles : 512;
    if (target > kMaxTileLocal) throw std::runtime_error("tile_particles too large");
    int kk;
    {
        double density = n / ((double)ext[0] * ext[1] * ext[2]);
        double per = density * ell * ell * ell;  // particles per ell^3
        kk = (int)std::lround(std::cbrt(target / std::max(per, 1e-9)));
        kk = std::max(kk, 2);
        if (kk & 1) ++kk;
    }
    const double cs = kk * ell;
    double org[3];
    int nc[3];
    for (int a = 0; a < 3; ++a) {
        org[a] = lo[a] - 0.5 * ell;
        nc[a] = (int)std::floor((hi[a] - org[a]) / cs) + 1;
    }
    if ((int64_t)nc[0] * nc[1] * nc[2] > (int64_t)1 << 40) throw std::runtime_error("grid too large");
    // per particle: cell coords, shifted-cell linear id, owner
    std::vector<int64_t> cell(n), scell(n);
    P.owner_of_old.resize(n);
    for (int32_t p = 0; p < n; ++p) {
        int c[3], s[3], blk[3];
        for (int a = 0; a < 3; ++a) {
            double r = (in.rest[3 * (int64_t)p + a] - org[a]) / cs;
            c[a] = std::min(std::max((int)std::floor(r), 0), nc[a] - 1);
            s[a] = std::min(std::max((int)std::floor(r + 0.5), 0), nc[a]);
            blk[a] = (int)((int64_t)c[a] * P.dims[a] / nc[a]);
        }
        cell[p] = ((int64_t)c[2] * nc[1] + c[1]) * nc[0] + c[0];
        scell[p] = ((int64_t)s[2] * (nc[1] + 1) + s[1]) * (nc[0] + 1) + s[0];
        P.owner_of_old[p] = (blk[2] * P.dims[1] + blk[1]) * P.dims[0] + blk[0];
    }
    // ---- P1 clusters: group particles by (owner, cell); split oversized groups ----------------
    // order particles by (owner, cell, old id)
    std::vector<int32_t> byc(n);
    std::iota(byc.begin(), byc.end(), 0);
    if (tiling) {
        std::sort(byc.begin(), byc.end(), [&](int32_t a, int32_t b) {
            if (P.owner_of_old[a] != P.owner_of_old[b]) return P.owner_of_old[a] < P.owner_of_old[b];
            if (cell[a] != cell[b]) return cell[a] < cell[b];
            return a < b;
        });
    } else {
        std::stable_sort(byc.begin(), byc.end(), [&](int32_t a, int32_t b) { return P.owner_of_old[a] < P.owner_of_old[b]; });
    }
    const int cap = tiling ? std::min(kMaxTileLocal, std::max(2 * target, 64)) : 512;
    std::vector<int32_t> p1_of_old(n);
    std::vector<int32_t> p1_begin;  // into byc (after splitting, byc is re-ordered inside a group)
    {
        // recursive median split along the longest axis until <= cap
        std::vector<std::pair<int32_t, int32_t>> stack;
        auto emit_group = [&](int32_t b, int32_t e) {
            stack.clear();
            stack.push_back({b, e});
            std::vector<std::pair<int32_t, int32_t>> done;
            while (!stack.empty()) {
                auto [gb, ge] = stack.back();
                stack.pop_back();
                if (ge - gb <= cap) { done.push_back({gb, ge}); continue; }
                if (!tiling) {  // plain chunks
                    for (int32_t s = gb; s < ge; s += cap) done.push_back({s, std::min(s + cap, ge)});
                    continue;
                }
                double l2[3] = {1e300, 1e300, 1e300}, h2[3] = {-1e300, -1e300, -1e300};
                for (int32_t q = gb; q < ge; ++q)
                    for (int a = 0; a < 3; ++a) {
                        double v = in.rest[3 * (int64_t)byc[q] + a];
                        l2[a] = std::min(l2[a], v); h2[a] = std::max(h2[a], v);
                    }
                int ax = 0;
                for (int a = 1; a < 3; ++a) if (h2[a] - l2[a] > h2[ax] - l2[ax]) ax = a;
                int32_t mid = gb + (ge - gb) / 2;
                std::sort(byc.begin() + gb, byc.begin() + ge, [&](int32_t a, int32_t b2) {
                    float va = in.rest[3 * (int64_t)a + ax], vb = in.rest[3 * (int64_t)b2 + ax];
                    if (va != vb) return va < vb;
                    return a < b2;
                });
                stack.push_back({mid, ge});
                stack.push_back({gb, mid});
            }
            std::sort(done.begin(), done.end());
            for (auto &d : done) p1_begin.push_back(d.first);
        };
        int32_t b = 0;
        while (b < n) {
            int32_t e = b + 1;
            if (tiling)
                while (e < n && P.owner_of_old[byc[e]] == P.owner_of_old[byc[b]] && cell[byc[e]] == cell[byc[b]]) ++e;
            else
                while (e < n && P.owner_of_old[byc[e]] == P.owner_of_old[byc[b]]) ++e;
            emit_group(b, e);
            b = e;
        }
        p1_begin.push_back(n);
    }
    const int32_t n_p1 = (int32_t)p1_begin.size() - 1;
    for (int32_t c = 0; c < n_p1; ++c)
        for (int32_t q = p1_begin[c]; q < p1_begin[c + 1]; ++q) p1_of_old[byc[q]] = c;

    // ---- classify constraints, shell flags ----------------------------------------------------
    // cls: 1 = P1, 2 = P2, 3 = global
    std::vector<uint8_t> cls[3];
    std::vector<uint8_t> shell(n, 0);
    for (int t = 0; t < 3; ++t) {
        cls[t].assign(C.count(t), 3);
        if (!tiling) continue;
        for (int64_t k = 0; k < C.count(t); ++k) {
            const int32_t *v = C.idx(t, k);
            bool same = true;
            for (int a = 1; a < kVerts[t]; ++a) same &= p1_of_old[v[a]] == p1_of_old[v[0]];
            if (same) cls[t][k] = 1;
            else for (int a = 0; a < kVerts[t]; ++a) shell[v[a]] = 1;
        }
    }
    // ---- final numbering: inside a P1 cluster order by (shell, shifted cell, old id) ----------
    P.old_of_new.resize(n);
    P.new_of_old.resize(n);
    for (int32_t c = 0; c < n_p1; ++c) {
        if (tiling)
            std::sort(byc.begin() + p1_begin[c], byc.begin() + p1_begin[c + 1], [&](int32_t a, int32_t b) {
                if (shell[a] != shell[b]) return shell[a] < shell[b];
                if (scell[a] != scell[b]) return scell[a] < scell[b];
                return a < b;
            });
        else
            std::sort(byc.begin() + p1_begin[c], byc.begin() + p1_begin[c + 1]);
    }
    for (int32_t q = 0; q < n; ++q) { P.old_of_new[q] = byc[q]; P.new_of_old[byc[q]] = q; }
    // shell segments: maximal runs of shell particles with equal (P1 cluster, shifted cell)
    std::vector<int32_t> seg_of_new(n, -1), seg_start, seg_len;
    if (tiling) {
        for (int32_t q = 0; q < n; ++q) {
            int32_t o = byc[q];
            if (!shell[o]) continue;
            bool fresh = q == 0 || !shell[byc[q - 1]] || p1_of_old[byc[q - 1]] != p1_of_old[o] || scell[byc[q - 1]] != scell[o];
            if (fresh) { seg_start.push_back(q); seg_len.push_back(0); }
            seg_of_new[q] = (int32_t)seg_start.size() - 1;
            ++seg_len.back();
        }
    }

    // ---- emit helpers -------------------------------------------------------------------------
    std::vector<Mask128> used((size_t)std::max<int>(kMaxTileLocal, 1));
    std::vector<int> col_tmp;
    auto push_order = [&](int type, int32_t id) { P.order_type.push_back((uint8_t)type); P.order_id.push_back(id); };
    P.order_type.reserve(P.m[0] + P.m[1] + P.m[2]);
    P.order_id.reserve(P.m[0] + P.m[1] + P.m[2]);
    P.task_off.push_back(0);
    P.group_off.push_back(0);

    // Emit one tile: items[t] = constraint ids of type t (increasing), local index via loc(new idx).
    // Returns false (and emits nothing) if some type needs more than 128 colours.
    std::vector<int32_t> lv;  // local vertex scratch
    auto emit_tile = [&](Cluster &cl, const std::vector<int32_t> items[3], auto loc) -> bool {
        struct Pending { int type; std::vector<std::vector<int32_t>> by_col; };
        std::vector<Pending> pend;
        for (int t = 0; t < 3; ++t) {
            const auto &it = items[t];
            if (it.empty()) continue;
            const int nv = kVerts[t];
            lv.resize(it.size() * nv);
            for (size_t k = 0; k < it.size(); ++k) {
                const int32_t *v = C.idx(t, it[k]);
                for (int a = 0; a < nv; ++a) lv[k * nv + a] = loc(P.new_of_old[v[a]]);
            }
            int ncol = greedy_colour((int)it.size(), nv, [&](int k) { return lv.data() + (size_t)k * nv; }, used, col_tmp);
            if (ncol < 0) return false;
            Pending pd; pd.type = t; pd.by_col.resize(ncol);
            for (size_t k = 0; k < it.size(); ++k) pd.by_col[col_tmp[k]].push_back((int32_t)k);
            // commit below (needs lv, which is per type) -> do it now into temporaries
            pend.push_back(std::move(pd));
            Pending &pp = pend.back();
            // stash local indices inside by_col as packed entries: replace item index by position; keep lv copy
            // (emit immediately: order inside a tile is type-major, so committing per type is fine)
            for (auto &colv : pp.by_col) {
                ColourEntry ce; ce.type = t; ce.count = (int32_t)colv.size();
                ce.begin = t == 0 ? (int64_t)P.t_dist.size() : (int64_t)P.t_quad_id.size();
                for (int32_t k : colv) {
                    const int32_t *l = lv.data() + (size_t)k * nv;
                    if (t == 0) {
                        P.t_dist.push_back((uint32_t)l[0] | ((uint32_t)l[1] << 16));
                        P.t_dist_id.push_back(it[k]);
                    } else {
                        P.t_quad.push_back((uint32_t)l[0] | ((uint32_t)l[1] << 16));
                        P.t_quad.push_back((uint32_t)l[2] | ((uint32_t)l[3] << 16));
                        P.t_quad_id.push_back(it[k]);
                        P.t_quad_type.push_back((uint8_t)t);
                    }
                    push_order(t, it[k]);
                }
                P.colours.push_back(ce);
                P.group_off.push_back((int64_t)P.order_id.size());
            }
        }
        return true;
    };

    // ---- phase P1 -----------------------------------------------------------------------------
    {
        Phase ph; ph.kind = 1; ph.type = -1; ph.fused_integrate = true; ph.needs_halo = false;
        ph.order_begin = 0; ph.task_begin = 0; ph.cluster_begin = 0;
        // bucket P1 constraints by cluster (ids increasing inside a bucket)
        std::vector<int64_t> off[3];
        std::vector<int32_t> lst[3];
        for (int t = 0; t < 3; ++t) {
            off[t].assign((size_t)n_p1 + 1, 0);
            for (int64_t k = 0; k < C.count(t); ++k) if (cls[t][k] == 1) ++off[t][p1_of_old[C.idx(t, k)[0]] + 1];
            for (int32_t c = 0; c < n_p1; ++c) off[t][c + 1] += off[t][c];
            lst[t].resize(off[t][n_p1]);
            std::vector<int64_t> cur(off[t].begin(), off[t].end() - 1);
            for (int64_t k = 0; k < C.count(t); ++k) if (cls[t][k] == 1) lst[t][cur[p1_of_old[C.idx(t, k)[0]]]++] = (int32_t)k;
        }
        std::vector<int32_t> items[3];
        for (int32_t c = 0; c < n_p1; ++c) {
            Cluster cl;
            cl.owner = P.owner_of_old[byc[p1_begin[c]]];
            cl.run_begin = (int32_t)P.runs.size(); cl.run_count = 1;
            P.runs.push_back({p1_begin[c], p1_begin[c + 1] - p1_begin[c]});
            cl.n_local = p1_begin[c + 1] - p1_begin[c];
            cl.col_begin = (int32_t)P.colours.size();
            cl.order_begin = (int64_t)P.order_id.size();
            for (int t = 0; t < 3; ++t) items[t].assign(lst[t].begin() + off[t][c], lst[t].begin() + off[t][c + 1]);
            const int32_t base = p1_begin[c];
            size_t sv_d = P.t_dist.size(), sv_q = P.t_quad_id.size(), sv_c = P.colours.size(), sv_o = P.order_id.size(), sv_g = P.group_off.size();
            if (!emit_tile(cl, items, [&](int32_t nw) { return nw - base; })) {
                // more than 128 colours inside one tile: push its constraints to the global phases
                P.t_dist.resize(sv_d); P.t_dist_id.resize(sv_d); P.t_quad.resize(2 * sv_q); P.t_quad_id.resize(sv_q);
                P.t_quad_type.resize(sv_q); P.colours.resize(sv_c); P.order_id.resize(sv_o); P.order_type.resize(sv_o); P.group_off.resize(sv_g);
                for (int t = 0; t < 3; ++t) for (int32_t k : items[t]) cls[t][k] = 3;
            }
            cl.col_count = (int32_t)P.colours.size() - cl.col_begin;
            cl.order_end = (int64_t)P.order_id.size();
            P.clusters.push_back(cl);
            P.task_off.push_back(cl.order_end);
            P.max_tile_local = std::max(P.max_tile_local, cl.n_local);
            P.max_tile_runs = std::max(P.max_tile_runs, 1);
        }
        ph.cluster_end = (int32_t)P.clusters.size();
        ph.order_end = (int64_t)P.order_id.size();
        ph.task_end = (int64_t)P.task_off.size() - 1;
        P.phases.push_back(ph);
        P.n_tile_phases = 1;
    }
    // ---- phase P2: shifted cells ---------------------------------------------------------------
    if (tiling) {
        // candidates: non-P1 constraints whose particles share a shifted cell
        struct Cand { int64_t sc; int32_t type; int32_t id; };
        std::vector<Cand> cand;
        for (int t = 0; t < 3; ++t)
            for (int64_t k = 0; k < C.count(t); ++k) {
                if (cls[t][k] != 3) continue;
                const int32_t *v = C.idx(t, k);
                bool same = true;
                for (int a = 1; a < kVerts[t]; ++a) same &= scell[v[a]] == scell[v[0]];
                if (same) cand.push_back({scell[v[0]], t, (int32_t)k});
            }
        std::sort(cand.begin(), cand.end(), [](const Cand &a, const Cand &b) {
            if (a.sc != b.sc) return a.sc < b.sc;
            if (a.type != b.type) return a.type < b.type;
            return a.id < b.id;
        });
        Phase ph; ph.kind = 1; ph.type = -1; ph.fused_integrate = false; ph.needs_halo = false;
        ph.order_begin = (int64_t)P.order_id.size(); ph.task_begin = (int64_t)P.task_off.size() - 1;
        ph.cluster_begin = (int32_t)P.clusters.size();
        std::vector<int32_t> items[3], segs, seg_lstart;
        size_t b = 0;
        while (b < cand.size()) {
            size_t e = b;
            while (e < cand.size() && cand[e].sc == cand[b].sc) ++e;
            for (int t = 0; t < 3; ++t) items[t].clear();
            segs.clear();
            for (size_t q = b; q < e; ++q) {
                items[cand[q].type].push_back(cand[q].id);
                const int32_t *v = C.idx(cand[q].type, cand[q].id);
                for (int a = 0; a < kVerts[cand[q].type]; ++a) segs.push_back(seg_of_new[P.new_of_old[v[a]]]);
            }
            std::sort(segs.begin(), segs.end());
            segs.erase(std::unique(segs.begin(), segs.end()), segs.end());
            int32_t nl = 0;
            seg_lstart.resize(segs.size());
            for (size_t s = 0; s < segs.size(); ++s) { seg_lstart[s] = nl; nl += seg_len[segs[s]]; }
            bool ok = nl <= kMaxTileLocal && (int)segs.size() <= kMaxTileRuns;
            if (ok) {
                Cluster cl; cl.owner = -1;
                cl.run_begin = (int32_t)P.runs.size(); cl.run_count = (int32_t)segs.size();
                bool multi = false; int own0 = -1;
                for (size_t s = 0; s < segs.size(); ++s) {
                    P.runs.push_back({seg_start[segs[s]], seg_len[segs[s]]});
                    int ow = P.owner_of_old[byc[seg_start[segs[s]]]];
                    if (own0 < 0) own0 = ow; else if (ow != own0) multi = true;
                }
                cl.owner = multi ? -1 : own0;
                cl.n_local = nl;
                cl.col_begin = (int32_t)P.colours.size();
                cl.order_begin = (int64_t)P.order_id.size();
                size_t sv_d = P.t_dist.size(), sv_q = P.t_quad_id.size(), sv_c = P.colours.size(), sv_o = P.order_id.size(), sv_g = P.group_off.size();
                auto loc = [&](int32_t nw) {
                    int32_t sg = seg_of_new[nw];
                    size_t s = std::lower_bound(segs.begin(), segs.end(), sg) - segs.begin();
                    return seg_lstart[s] + (nw - seg_start[sg]);
                };
                if (emit_tile(cl, items, loc)) {
                    cl.col_count = (int32_t)P.colours.size() - cl.col_begin;
                    cl.order_end = (int64_t)P.order_id.size();
                    P.clusters.push_back(cl);
                    P.task_off.push_back(cl.order_end);
                    P.max_tile_local = std::max(P.max_tile_local, cl.n_local);
                    P.max_tile_runs = std::max(P.max_tile_runs, cl.run_count);
                    for (size_t q = b; q < e; ++q) cls[cand[q].type][cand[q].id] = 2;
                    if (multi) ph.needs_halo = true;
                } else {
                    P.t_dist.resize(sv_d); P.t_dist_id.resize(sv_d); P.t_quad.resize(2 * sv_q); P.t_quad_id.resize(sv_q);
                    P.t_quad_type.resize(sv_q); P.colours.resize(sv_c); P.order_id.resize(sv_o); P.order_type.resize(sv_o); P.group_off.resize(sv_g);
                    P.runs.resize(cl.run_begin);
                }
            }
            b = e;
        }
        ph.cluster_end = (int32_t)P.clusters.size();
        ph.order_end = (int64_t)P.order_id.size();
        ph.task_end = (int64_t)P.task_off.size() - 1;
        if (ph.cluster_end > ph.cluster_begin) { P.phases.push_back(ph); P.n_tile_phases = 2; }
    }
    P.cons_in_tiles = (int64_t)P.order_id.size();
    // ---- global colour phases for the rest ----------------------------------------------------
    {
        std::vector<Mask128> gused;
        std::vector<int32_t> left;
        std::vector<int> colr;
        for (int t = 0; t < 3; ++t) {
            left.clear();
            for (int64_t k = 0; k < C.count(t); ++k) if (cls[t][k] == 3) left.push_back((int32_t)k);
            if (left.empty()) continue;
            if (gused.empty()) gused.assign(n, Mask128());
            const int nv = kVerts[t];
            int ncol = greedy_colour((int)left.size(), nv, [&](int k) { return C.idx(t, left[k]); }, gused, colr);
            if (ncol < 0) throw std::runtime_error("constraint graph needs more than 128 colours");
            std::vector<std::vector<int32_t>> by(ncol);
            for (size_t k = 0; k < left.size(); ++k) by[colr[k]].push_back(left[k]);
            for (int c = 0; c < ncol; ++c) {
                Phase ph; ph.kind = 0; ph.type = t; ph.fused_integrate = false; ph.needs_halo = false;
                ph.order_begin = (int64_t)P.order_id.size(); ph.task_begin = (int64_t)P.task_off.size() - 1;
                int cnt = 0;
                for (int32_t id : by[c]) {
                    push_order(t, id);
                    if (++cnt == 256) { P.task_off.push_back((int64_t)P.order_id.size()); P.group_off.push_back((int64_t)P.order_id.size()); cnt = 0; }
                    if (opts.world > 1 && !ph.needs_halo) {
                        const int32_t *v = C.idx(t, id);
                        for (int a = 1; a < nv; ++a) if (P.owner_of_old[v[a]] != P.owner_of_old[v[0]]) ph.needs_halo = true;
                    }
                }
                if (cnt) { P.task_off.push_back((int64_t)P.order_id.size()); P.group_off.push_back((int64_t)P.order_id.size()); }
                ph.order_end = (int64_t)P.order_id.size(); ph.task_end = (int64_t)P.task_off.size() - 1;
                P.phases.push_back(ph);
                ++P.n_global_colours;
            }
        }
    }
    P.cons_in_global = (int64_t)P.order_id.size() - P.cons_in_tiles;
    if ((int64_t)P.order_id.size() != P.m[0] + P.m[1] + P.m[2]) throw std::runtime_error("planner lost constraints");
}

void extract_local(const Plan &P, const Input &in, int rank, LocalPlan &L) {
    L = LocalPlan();
    L.rank = rank; L.world = P.opts.world;
    const int32_t n = P.n;
    const int world = P.opts.world;
    Cons C{&in};
    auto owner_new = [&](int32_t nw) { return P.owner_of_old[P.old_of_new[nw]]; };
    // owned range in new numbering (sorted by owner first)
    int32_t ob = 0, oe = 0;
    {
        ob = n; oe = n;
        for (int32_t q = 0; q < n; ++q) if (owner_new(q) == rank) { ob = q; break; }
        for (int32_t q = ob; q < n; ++q) if (owner_new(q) != rank) { oe = q; break; }
        if (ob == n) ob = oe = 0;
    }
    L.n_owned = oe - ob;
    // which items does `r` execute, and which particles do they touch? Walk all phases once, for all ranks,
    // collecting (phase, consumer rank, particle new idx) triples where particle owner != consumer.
    struct Need { int32_t phase, consumer, nw; };
    std::vector<Need> needs;
    L.order_mask.assign(P.order_id.size(), 0);
    L.phases.resize(P.phases.size());
    std::vector<int> owners;
    for (size_t ph = 0; ph < P.phases.size(); ++ph) {
        const Phase &F = P.phases[ph];
        LocalPhase &LP = L.phases[ph];
        LP.kind = F.kind; LP.type = F.type; LP.fused_integrate = F.fused_integrate; LP.needs_halo = F.needs_halo;
        LP.send_idx.assign(world, {}); LP.recv_idx.assign(world, {});
        if (F.kind == 1) {
            for (int32_t c = F.cluster_begin; c < F.cluster_end; ++c) {
                const Cluster &cl = P.clusters[c];
                owners.clear();
                for (int r = 0; r < cl.run_count; ++r) owners.push_back(owner_new(P.runs[cl.run_begin + r].start));
                std::sort(owners.begin(), owners.end());
                owners.erase(std::unique(owners.begin(), owners.end()), owners.end());
                bool mine = std::binary_search(owners.begin(), owners.end(), rank);
                if (mine) {
                    LP.cluster_ids.push_back(c);
                    for (int64_t k = cl.order_begin; k < cl.order_end; ++k) L.order_mask[k] = 1;
                }
                if (owners.size() > 1)
                    for (int r = 0; r < cl.run_count; ++r) {
                        const Run &rn = P.runs[cl.run_begin + r];
                        int ow = owner_new(rn.start);
                        for (int cons : owners) {
                            if (cons == ow) continue;
                            if (cons != rank && ow != rank) continue;  // only triples that involve me
                            for (int32_t q = 0; q < rn.len; ++q) needs.push_back({(int32_t)ph, cons, rn.start + q});
                        }
                    }
            }
        } else {
            const int nv = kVerts[F.type];
            for (int64_t k = F.order_begin; k < F.order_end; ++k) {
                const int32_t *v = C.idx(F.type, P.order_id[k]);
                bool mine = false, multi = false;
                for (int a = 0; a < nv; ++a) {
                    int ow = P.owner_of_old[v[a]];
                    mine |= ow == rank;
                    multi |= ow != P.owner_of_old[v[0]];
                }
                if (mine) L.order_mask[k] = 1;
                if (multi)
                    for (int a = 0; a < nv; ++a)
                        for (int b = 0; b < nv; ++b) {
                            int cons = P.owner_of_old[v[b]], ow = P.owner_of_old[v[a]];
                            if (cons == ow) continue;
                            if (cons != rank && ow != rank) continue;
                            needs.push_back({(int32_t)ph, cons, P.new_of_old[v[a]]});
                        }
            }
        }
    }
    std::sort(needs.begin(), needs.end(), [](const Need &a, const Need &b) {
        if (a.phase != b.phase) return a.phase < b.phase;
        if (a.consumer != b.consumer) return a.consumer < b.consumer;
        return a.nw < b.nw;
    });
    needs.erase(std::unique(needs.begin(), needs.end(), [](const Need &a, const Need &b) {
        return a.phase == b.phase && a.consumer == b.consumer && a.nw == b.nw; }), needs.end());
    // ghost set of this rank
    std::vector<int32_t> ghosts;
    for (const Need &nd : needs) if (nd.consumer == rank) ghosts.push_back(nd.nw);
    std::sort(ghosts.begin(), ghosts.end());
    ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
    std::vector<int32_t> local_of_new(n, -1);
    L.local_to_old.resize((size_t)L.n_owned + ghosts.size());
    for (int32_t q = ob; q < oe; ++q) { local_of_new[q] = q - ob; L.local_to_old[q - ob] = P.old_of_new[q]; }
    for (size_t g = 0; g < ghosts.size(); ++g) {
        local_of_new[ghosts[g]] = (int32_t)(L.n_owned + g);
        L.local_to_old[L.n_owned + g] = P.old_of_new[ghosts[g]];
    }
    // halo lists (both sides sorted by new idx -> same order)
    for (const Need &nd : needs) {
        LocalPhase &LP = L.phases[nd.phase];
        int ow = owner_new(nd.nw);
        if (nd.consumer == rank) LP.recv_idx[ow].push_back(local_of_new[nd.nw]);
        else if (ow == rank) LP.send_idx[nd.consumer].push_back(local_of_new[nd.nw]);
    }
    // local phases
    for (size_t ph = 0; ph < P.phases.size(); ++ph) {
        const Phase &F = P.phases[ph];
        LocalPhase &LP = L.phases[ph];
        if (F.kind == 1) {
            LP.run_begin.push_back(0);
            for (int32_t c : LP.cluster_ids) {
                const Cluster &cl = P.clusters[c];
                for (int r = 0; r < cl.run_count; ++r) {
                    const Run &rn = P.runs[cl.run_begin + r];
                    int32_t ls = local_of_new[rn.start];
                    if (ls < 0) throw std::runtime_error("internal: tile run not resident");
                    // runs stay contiguous locally: whole segments are resident and ghosts are in new-index order
                    if (local_of_new[rn.start + rn.len - 1] != ls + rn.len - 1) throw std::runtime_error("internal: run split");
                    LP.runs.push_back({ls, rn.len});
                }
                LP.run_begin.push_back((int32_t)LP.runs.size());
            }
        } else {
            const int nv = kVerts[F.type];
            for (int64_t k = F.order_begin; k < F.order_end; ++k) {
                if (!L.order_mask[k]) continue;
                const int32_t *v = C.idx(F.type, P.order_id[k]);
                for (int a = 0; a < nv; ++a) {
                    int32_t li = local_of_new[P.new_of_old[v[a]]];
                    if (li < 0) throw std::runtime_error("internal: constraint particle not resident");
                    LP.g_idx.push_back(li);
                }
                LP.g_id.push_back(P.order_id[k]);
            }
        }
    }
}

}  // namespace sbp
