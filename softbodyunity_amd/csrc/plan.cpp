// plan.cpp — see plan.hpp. Pure host C++; deterministic: no hashing by address, and the host threads it uses
// (std::thread, SB_PLAN_THREADS, default min(hardware threads, 16)) only ever split work into pieces whose results do
// not depend on how many threads ran them -- every rank of a partitioned solver must arrive at the same plan.
#include "plan.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <exception>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

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

int plan_threads() {
    static const int n = [] {
        if (const char *e = std::getenv("SB_PLAN_THREADS")) return std::max(1, std::atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(hw ? hw : 1u, 16u);
    }();
    return n;
}

// Runs f(chunk, begin, end) for the chunks [k*chunk_size, min(n, (k+1)*chunk_size)) of [0, n) on the planner's
// threads. The chunking depends on n and chunk_size only, never on the thread count, so per-chunk results (partial
// sums, output pieces) combine to the same answer on every machine. The first exception is re-thrown after the join.
template <class F>
void parallel_chunks(int64_t n, int64_t chunk_size, F f) {
    if (n <= 0) return;
    const int64_t n_chunks = (n + chunk_size - 1) / chunk_size;
    const int nt = (int)std::min<int64_t>(plan_threads(), n_chunks);
    if (nt <= 1) {
        for (int64_t c = 0; c < n_chunks; ++c) f(c, c * chunk_size, std::min(n, (c + 1) * chunk_size));
        return;
    }
    std::atomic<int64_t> next{0};
    std::exception_ptr err;
    std::mutex err_mu;
    auto work = [&] {
        try {
            for (int64_t c; (c = next.fetch_add(1)) < n_chunks;) f(c, c * chunk_size, std::min(n, (c + 1) * chunk_size));
        } catch (...) {
            std::lock_guard<std::mutex> g(err_mu);
            if (!err) err = std::current_exception();
            next.store(n_chunks);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// Stable counting sort of the ids 0..n-1 by key[id] in [0, n_keys): equals std::sort by (key, id).
void counting_sort_ids(const std::vector<int64_t> &key, int64_t n_keys, std::vector<int32_t> &out) {
    const int32_t n = (int32_t)key.size();
    std::vector<int32_t> cnt((size_t)n_keys + 1, 0);
    for (int32_t p = 0; p < n; ++p) ++cnt[(size_t)key[p] + 1];
    for (int64_t k = 0; k < n_keys; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
    out.resize(n);
    for (int32_t p = 0; p < n; ++p) out[(size_t)cnt[(size_t)key[p]]++] = p;
}

struct PlanTimer {      // SB_PLAN_TIMING=1: phase times of build_plan on stderr
    bool on = std::getenv("SB_PLAN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[plan] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

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

// Greedy colouring of `count` constraints over an index space. used[] must be zero for the touched
// indices on entry and is zeroed again on exit. Returns the colour count. Up to 128 colours run on one
// 128-bit mask per index; a graph that needs more (a hub particle with hundreds of springs) is recoloured by
// the general routine below with masks as wide as it takes -- same rule (lowest colour free at every endpoint,
// constraints in the given order), so the result for <= 128 colours is the same either way.
template <class GetVerts>
int greedy_colour_wide(int64_t count, GetVerts get, size_t index_space, std::vector<int> &colour_out) {
    colour_out.assign(count, 0);
    std::vector<std::vector<uint64_t>> used(index_space);     // only touched indices ever grow
    std::vector<uint64_t> m;
    int ncol = 0;
    for (int64_t k = 0; k < count; ++k) {
        int nverts;
        const int32_t *v = get(k, nverts);
        m.clear();
        for (int a = 0; a < nverts; ++a) {
            const std::vector<uint64_t> &u = used[v[a]];
            if (u.size() > m.size()) m.resize(u.size(), 0);
            for (size_t w = 0; w < u.size(); ++w) m[w] |= u[w];
        }
        int c = -1;
        for (size_t w = 0; w < m.size() && c < 0; ++w) if (~m[w]) c = (int)(64 * w) + __builtin_ctzll(~m[w]);
        if (c < 0) c = (int)(64 * m.size());
        for (int a = 0; a < nverts; ++a) {
            std::vector<uint64_t> &u = used[v[a]];
            if (u.size() <= (size_t)c / 64) u.resize((size_t)c / 64 + 1, 0);
            u[(size_t)c / 64] |= 1ull << (c % 64);
        }
        colour_out[k] = c;
        ncol = std::max(ncol, c + 1);
    }
    return ncol;
}

// get(k, nverts) returns the index list of constraint k and its length (2 or 4): constraints of different types may be
// coloured together.
template <class GetVerts>
int greedy_colour(int64_t count, GetVerts get, std::vector<Mask128> &used, std::vector<int> &colour_out) {
    colour_out.resize(count);
    int ncol = 0;
    for (int64_t k = 0; k < count; ++k) {
        int nverts;
        const int32_t *v = get(k, nverts);
        Mask128 m;
        for (int a = 0; a < nverts; ++a) m = m | used[v[a]];
        int c = first_free(m);
        if (c < 0) { ncol = -1; break; }
        for (int a = 0; a < nverts; ++a) set_bit(used[v[a]], c);
        colour_out[k] = c;
        ncol = std::max(ncol, c + 1);
    }
    for (int64_t k = 0; k < count; ++k) {
        int nverts;
        const int32_t *v = get(k, nverts);
        for (int a = 0; a < nverts; ++a) used[v[a]] = Mask128();
    }
    if (ncol < 0) ncol = greedy_colour_wide(count, get, used.size(), colour_out);
    return ncol;
}

// Groups sorted particle ids [b,e) of `byc` into pieces of at most `cap` by recursive median split along
// the longest axis (plain chunks when !spatial). Appends piece start offsets (ascending) to `begins`.
void split_group(std::vector<int32_t> &byc, int32_t b, int32_t e, int cap, bool spatial, const float *rest,
                 std::vector<int32_t> &begins) {
    std::vector<std::pair<int32_t, int32_t>> stack{{b, e}}, done;
    while (!stack.empty()) {
        auto [gb, ge] = stack.back();
        stack.pop_back();
        if (ge - gb <= cap) { done.push_back({gb, ge}); continue; }
        if (!spatial) {
            for (int32_t s = gb; s < ge; s += cap) done.push_back({s, std::min(s + cap, ge)});
            continue;
        }
        double l2[3] = {1e300, 1e300, 1e300}, h2[3] = {-1e300, -1e300, -1e300};
        for (int32_t q = gb; q < ge; ++q)
            for (int a = 0; a < 3; ++a) {
                double v = rest[3 * (int64_t)byc[q] + a];
                l2[a] = std::min(l2[a], v); h2[a] = std::max(h2[a], v);
            }
        int ax = 0;
        for (int a = 1; a < 3; ++a) if (h2[a] - l2[a] > h2[ax] - l2[ax]) ax = a;
        int32_t mid = gb + (ge - gb) / 2;
        std::sort(byc.begin() + gb, byc.begin() + ge, [&](int32_t a, int32_t b2) {
            float va = rest[3 * (int64_t)a + ax], vb = rest[3 * (int64_t)b2 + ax];
            if (va != vb) return va < vb;
            return a < b2;
        });
        stack.push_back({mid, ge});
        stack.push_back({gb, mid});
    }
    std::sort(done.begin(), done.end());
    for (auto &d : done) begins.push_back(d.first);
}

}  // namespace

void parallel_for_chunks(int64_t n, int64_t chunk_size, const std::function<void(int64_t, int64_t, int64_t)> &f) {
    parallel_chunks(n, chunk_size, f);
}

void compute_domain(const Input &in, Domain &dom) {
    const int32_t n = in.n;
    if (n <= 0) throw std::runtime_error("no particles");
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    {
        constexpr int64_t kChunk = 1 << 20;
        const int64_t nch = ((int64_t)n + kChunk - 1) / kChunk;
        std::vector<double> plo((size_t)nch * 3, 1e300), phi((size_t)nch * 3, -1e300);
        parallel_chunks(n, kChunk, [&](int64_t c, int64_t pb, int64_t pe) {
            double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
            for (int64_t p = pb; p < pe; ++p)
                for (int a = 0; a < 3; ++a) {
                    double v = in.rest[3 * p + a];
                    if (!(v == v) || std::fabs(v) > 1e30) throw std::runtime_error("non-finite rest position");
                    l[a] = std::min(l[a], v); h[a] = std::max(h[a], v);
                }
            for (int a = 0; a < 3; ++a) { plo[(size_t)c * 3 + a] = l[a]; phi[(size_t)c * 3 + a] = h[a]; }
        });
        for (int64_t c = 0; c < nch; ++c)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], plo[(size_t)c * 3 + a]); hi[a] = std::max(hi[a], phi[(size_t)c * 3 + a]); }
    }
    double ell = 0;
    {
        // mean spring length: partial sums over fixed chunks of the constraint list, added in chunk order (the same
        // value whatever the thread count)
        double acc = 0; int64_t cnt = 0;
        auto mean_edges = [&](const int32_t *idx, int nv, int64_t m_all) {
            if (m_all <= 0 || !idx) return;
            // a sample of about a million evenly spaced constraints is plenty for a length scale (every one below 2^21)
            const int64_t stride = std::max<int64_t>(1, m_all >> 20), m = (m_all + stride - 1) / stride;
            constexpr int64_t kChunk = 1 << 16;
            const int64_t nch = (m + kChunk - 1) / kChunk;
            std::vector<double> part((size_t)nch, 0.0);
            parallel_chunks(m, kChunk, [&](int64_t c, int64_t kb, int64_t ke) {
                double a2 = 0;
                for (int64_t ks = kb; ks < ke; ++ks) {
                    const int64_t k = ks * stride;
                    const int32_t i = idx[nv * k], j = idx[nv * k + 1];
                    if (i < 0 || i >= n || j < 0 || j >= n) throw std::runtime_error("constraint index out of range");
                    double s = 0;
                    for (int a = 0; a < 3; ++a) { double d = (double)in.rest[3 * (int64_t)i + a] - in.rest[3 * (int64_t)j + a]; s += d * d; }
                    a2 += std::sqrt(s);
                }
                part[(size_t)c] = a2;
            });
            for (double v : part) acc += v;
            cnt += m;
        };
        mean_edges(in.dist_ij, 2, in.m_d);
        if (cnt == 0) mean_edges(in.vol, 4, in.m_v);
        if (cnt == 0) mean_edges(in.bend, 4, in.m_b);
        if (cnt > 0) ell = acc / cnt;
        if (!(ell > 0)) {
            double vol = 1; for (int a = 0; a < 3; ++a) vol *= std::max(hi[a] - lo[a], 1e-6);
            ell = std::cbrt(vol / n);
        }
    }
    dom.set = false;            // (measured, not imposed)
    dom.n_global = n;
    for (int a = 0; a < 3; ++a) { dom.lo[a] = lo[a]; dom.hi[a] = hi[a]; }
    dom.ell = ell;
    // Which fraction of the bounding box the mesh fills: a coarse occupancy count (cells of 4 mean rest lengths). A lattice fills
    // its box (1.0, and anything above 0.8 is read as 1.0 so that no regular mesh's grid moves); a bunny fills 40 % of its box,
    // and a grid sized for the box's average density would give it cells of 2.5 x the particles asked for, every one of them
    // median-split along one axis -- whose split planes then coincide from grid to grid (plan.cpp static split).
    dom.fill = 1.0;
    {
        double cell = 4.0 * ell;
        int64_t nd[3];
        for (;;) {
            for (int a = 0; a < 3; ++a) nd[a] = std::max<int64_t>(1, (int64_t)std::ceil((hi[a] - lo[a] + ell) / cell));
            if (nd[0] * nd[1] * nd[2] <= ((int64_t)1 << 24)) break;
            cell *= 2.0;
        }
        std::vector<uint8_t> occ((size_t)(nd[0] * nd[1] * nd[2]), 0);
        for (int64_t p = 0; p < n; ++p) {
            int64_t c[3];
            for (int a = 0; a < 3; ++a) c[a] = std::min<int64_t>(nd[a] - 1, std::max<int64_t>(0, (int64_t)std::floor((in.rest[3 * p + a] - (lo[a] - 0.5 * ell)) / cell)));
            occ[(size_t)((c[2] * nd[1] + c[1]) * nd[0] + c[0])] = 1;
        }
        int64_t used = 0;
        for (uint8_t o : occ) used += o;
        const double f = (double)used / (double)occ.size();
        if (f <= 0.8) dom.fill = f;
    }
}

Grid make_grid(const Domain &dom, int target) {
    Grid G;
    const double ell = dom.ell;
    for (int a = 0; a < 3; ++a) G.ext[a] = (float)(dom.hi[a] - dom.lo[a] + ell);
    {
        double density = (double)dom.n_global / ((double)G.ext[0] * G.ext[1] * G.ext[2] * dom.fill);
        double per = density * ell * ell * ell;  // particles per ell^3
        G.kk = (int)std::lround(std::cbrt(target / std::max(per, 1e-9)));
        G.kk = std::max(G.kk, 2);
        if (G.kk & 1) ++G.kk;
    }
    G.cs = G.kk * ell;
    for (int a = 0; a < 3; ++a) {
        G.org[a] = dom.lo[a] - 0.5 * ell;
        G.nc[a] = (int)std::floor((dom.hi[a] - G.org[a]) / G.cs) + 1;
    }
    if ((int64_t)G.nc[0] * G.nc[1] * G.nc[2] > (int64_t)1 << 40) throw std::runtime_error("grid too large");
    // T1 = T0 shifted by an ODD number of mean spring lengths close to half a cell (kk is even): on a lattice the
    // springs that cross a T0 boundary and those that cross a T1 boundary then belong to different parity classes,
    // so each class is a complete matching inside one of the two tilings (see the static split in build_plan)
    G.shift_units = ((G.kk / 2) & 1) ? G.kk / 2 : std::max(G.kk / 2 - 1, 1);
    G.shift_frac = (double)G.shift_units / G.kk;
    G.first_t2_frac = G.shift_frac + 0.5 * (1.0 - G.shift_frac);     // grid of the first T2 layer: the middle of the widest gap between the T0 and T1 planes
    return G;
}

void rank_window(const Domain &dom, const Opts &opts, int cell_lo[3], int cell_hi[3], double box_lo[3], double box_hi[3]) {
    const int target = opts.tile_particles > 0 ? opts.tile_particles : 512;
    const Grid G = make_grid(dom, target);
    int dims[3];
    resolve_dims(opts.world, G.ext, opts.dims, dims);
    int b[3] = {opts.rank % dims[0], (opts.rank / dims[0]) % dims[1], opts.rank / (dims[0] * dims[1])};
    for (int a = 0; a < 3; ++a) {
        // cells c with floor(c * dims / nc) == b[a]
        const int64_t nc = G.nc[a], d = dims[a];
        const int c_lo = (int)(((int64_t)b[a] * nc + d - 1) / d), c_hi = (int)((((int64_t)b[a] + 1) * nc + d - 1) / d);
        cell_lo[a] = std::max(0, c_lo - kWindowMarginCells);
        cell_hi[a] = std::min((int)nc, c_hi + kWindowMarginCells);
        // the box in rest coordinates; the outermost cells reach to infinity (positions are clamped into the grid)
        box_lo[a] = cell_lo[a] == 0 ? -1e300 : G.org[a] + cell_lo[a] * G.cs;
        box_hi[a] = cell_hi[a] == (int)nc ? 1e300 : G.org[a] + cell_hi[a] * G.cs;
    }
}

void build_plan(const Input &in, const Opts &opts, Plan &P) {
    P = Plan();
    P.opts = opts;
    const int32_t n = in.n;
    if (n <= 0) throw std::runtime_error("no particles");
    if (opts.world < 1 || opts.rank < 0 || opts.rank >= opts.world) throw std::runtime_error("bad rank/world");
    PlanTimer timer;
    const bool bank_aware = opts.bank_aware_lanes;
    const bool mixed_groups = opts.mixed_groups;
    Cons C{&in};
    P.n = n;
    P.m[0] = in.m_d; P.m[1] = in.m_v; P.m[2] = in.m_b;
    for (int t = 0; t < 3; ++t) {
        if (C.count(t) < 0) throw std::runtime_error("negative constraint count");
        if (C.count(t) > 0 && !C.idx(t, 0)) throw std::runtime_error("null constraint array");
        parallel_chunks(C.count(t), 1 << 20, [&](int64_t, int64_t kb, int64_t ke) {
            for (int64_t k = kb; k < ke; ++k) {
                const int32_t *v = C.idx(t, k);
                for (int a = 0; a < kVerts[t]; ++a) {
                    if (v[a] < 0 || v[a] >= n) throw std::runtime_error("constraint index out of range");
                    for (int b = 0; b < a; ++b)
                        if (v[a] == v[b]) throw std::runtime_error("constraint repeats a particle");
                }
            }
        });
    }
    timer.lap("validate");
    // ---- geometry: the frame (bounding box, length scale, particle count) and the grid made from it -----------
    // A whole-mesh plan measures the frame on its input; a sharded plan (the input is one rank's window of a larger mesh)
    // takes the frame every rank agrees on from the caller, so that all windows are cut from ONE grid.
    Domain dom = opts.domain;
    const bool sharded = dom.set;
    if (sharded) {
        if (!in.global_id) throw std::runtime_error("a sharded plan needs the global ids of its particles");
        if (dom.n_global < n || !(dom.ell > 0)) throw std::runtime_error("bad domain (n_global < n, or spacing <= 0)");
        for (int a = 0; a < 3; ++a) if (!(dom.hi[a] >= dom.lo[a])) throw std::runtime_error("bad domain (hi < lo)");
        parallel_chunks(n, 1 << 20, [&](int64_t, int64_t pb, int64_t pe) {
            for (int64_t q = pb; q < pe; ++q) {
                if (in.global_id[q] < 0 || in.global_id[q] >= dom.n_global) throw std::runtime_error("global particle id out of range");
                if (q > 0 && in.global_id[q - 1] >= in.global_id[q]) throw std::runtime_error("global particle ids must be strictly ascending");
                for (int a = 0; a < 3; ++a) { double v = in.rest[3 * q + a]; if (!(v == v) || std::fabs(v) > 1e30) throw std::runtime_error("non-finite rest position"); }
            }
        });
    } else {
        compute_domain(in, dom);
    }
    P.domain = dom;
    const bool tiling = opts.tile_particles > 0;
    P.tiling = tiling;
    const int target = tiling ? opts.tile_particles : 512;
    if (target > kMaxTileLocal) throw std::runtime_error("tile_particles too large");
    const Grid G = make_grid(dom, target);
    resolve_dims(opts.world, G.ext, opts.dims, P.dims);
    const int kk = G.kk; (void)kk;
    const double cs = G.cs;
    const double *org = G.org;
    const int *nc = G.nc;
    const double shift_frac = G.shift_frac, first_t2_frac = G.first_t2_frac;
    int win_lo[3] = {0, 0, 0}, win_hi[3] = {nc[0], nc[1], nc[2]};     // sharded: the cells this rank's window must cover
    if (sharded) {
        if (opts.partition == 2) throw std::runtime_error("a sharded plan cannot use the RCB partition (it needs the whole mesh): pass the whole mesh");
        double blo[3], bhi[3];
        rank_window(dom, opts, win_lo, win_hi, blo, bhi);
    }
    std::vector<int64_t> cell(n), scell(n);
    P.owner_of_old.resize(n);
    parallel_chunks(n, 1 << 18, [&](int64_t, int64_t pb, int64_t pe) {
        for (int64_t p = pb; p < pe; ++p) {
            int c[3], s[3], blk[3];
            for (int a = 0; a < 3; ++a) {
                double r = (in.rest[3 * p + a] - org[a]) / cs;
                c[a] = std::min(std::max((int)std::floor(r), 0), nc[a] - 1);
                s[a] = std::min(std::max((int)std::floor(r - shift_frac) + 1, 0), nc[a]);
                blk[a] = (int)((int64_t)c[a] * P.dims[a] / nc[a]);
            }
            if (sharded)
                for (int a = 0; a < 3; ++a)
                    if (c[a] < win_lo[a] || c[a] >= win_hi[a]) throw std::runtime_error("a particle lies outside this rank's window (sb_domain_window)");
            cell[p] = ((int64_t)c[2] * nc[1] + c[1]) * nc[0] + c[0];
            scell[p] = ((int64_t)s[2] * (nc[1] + 1) + s[1]) * (nc[0] + 1) + s[0];
            P.owner_of_old[p] = (blk[2] * P.dims[1] + blk[1]) * P.dims[0] + blk[0];
        }
    });
    const int64_t n_cells = (int64_t)nc[0] * nc[1] * nc[2], n_scells = (int64_t)(nc[0] + 1) * (nc[1] + 1) * (nc[2] + 1);
    const int cap = tiling ? std::min(kMaxTileLocal, std::max(2 * target, 64)) : 512;
    // ---- ownership (world > 1) -------------------------------------------------------------------
    // A rank owns whole T0 cells, so T0 tiles are single-owner whatever the partition. Regular meshes: the block grid
    // above (dims). A mesh that fills its bounding box unevenly (a tet mesh of a bunny: the corner blocks are nearly
    // empty) is cut by recursive coordinate bisection over the occupied cells instead, each cell weighted by the cost
    // of its particles (kCostParticle each + their vertex shares of the constraints).
    P.partition = 1;
    P.rank_cost.assign((size_t)opts.world, 0);
    if (opts.world > 1) {
        if (opts.partition < 0 || opts.partition > 2) throw std::runtime_error("partition must be 0 (automatic), 1 (blocks) or 2 (RCB)");
        // cost of every rank under the ownership `own`: partial sums over fixed chunks (integers: any order gives the same)
        auto rank_costs = [&](const std::vector<int32_t> &own, std::vector<int64_t> &out) {
            out.assign((size_t)opts.world, 0);
            std::mutex mu;
            auto add = [&](const std::vector<int64_t> &part) { std::lock_guard<std::mutex> g(mu); for (int r = 0; r < opts.world; ++r) out[(size_t)r] += part[(size_t)r]; };
            parallel_chunks(n, 1 << 20, [&](int64_t, int64_t pb, int64_t pe) {
                std::vector<int64_t> part((size_t)opts.world, 0);
                for (int64_t q = pb; q < pe; ++q) part[(size_t)own[(size_t)q]] += kCostParticle;
                add(part);
            });
            for (int t = 0; t < 3; ++t)
                parallel_chunks(C.count(t), 1 << 20, [&](int64_t, int64_t kb, int64_t ke) {
                    std::vector<int64_t> part((size_t)opts.world, 0);
                    for (int64_t k = kb; k < ke; ++k) {
                        const int32_t *v = C.idx(t, k);
                        for (int a = 0; a < kVerts[t]; ++a) part[(size_t)own[(size_t)v[a]]] += kCostVertexShare[t];
                    }
                    add(part);
                });
        };
        rank_costs(P.owner_of_old, P.rank_cost);     // (sharded: the costs of the window's particles only)
        bool rcb = opts.partition == 2;
        if (opts.partition == 0 && !sharded && !(opts.dims[0] > 0 && opts.dims[1] > 0 && opts.dims[2] > 0)) {
            int64_t total = 0, worst = 0;
            for (int64_t c : P.rank_cost) { total += c; worst = std::max(worst, c); }
            rcb = worst * opts.world * 10 > total * 11;      // the block grid leaves a rank more than 10 % above the mean
        }
        if (rcb) {
            // occupied cells and their weights
            const bool dense = n_cells <= 8 * (int64_t)n + 4096;
            std::vector<int64_t> occ;                   // occupied cell ids, ascending
            std::vector<int32_t> dense_index;           // dense: cell id -> index into occ (-1: empty)
            if (dense) {
                dense_index.assign((size_t)n_cells, -1);
                for (int32_t q = 0; q < n; ++q) dense_index[(size_t)cell[(size_t)q]] = 0;
                for (int64_t c = 0; c < n_cells; ++c) if (dense_index[(size_t)c] == 0) { dense_index[(size_t)c] = (int32_t)occ.size(); occ.push_back(c); }
            } else {
                occ = cell;
                std::sort(occ.begin(), occ.end());
                occ.erase(std::unique(occ.begin(), occ.end()), occ.end());
            }
            auto index_of = [&](int64_t c) -> size_t {
                return dense ? (size_t)dense_index[(size_t)c] : (size_t)(std::lower_bound(occ.begin(), occ.end(), c) - occ.begin());
            };
            std::vector<int64_t> wcell(occ.size(), 0);
            for (int32_t q = 0; q < n; ++q) wcell[index_of(cell[(size_t)q])] += kCostParticle;
            for (int t = 0; t < 3; ++t)
                for (int64_t k = 0; k < C.count(t); ++k) {
                    const int32_t *v = C.idx(t, k);
                    for (int a = 0; a < kVerts[t]; ++a) wcell[index_of(cell[(size_t)v[a]])] += kCostVertexShare[t];
                }
            struct RcbCell { int32_t c[3]; int32_t idx; };
            std::vector<RcbCell> cells(occ.size());
            for (size_t k = 0; k < occ.size(); ++k) {
                int64_t id = occ[k];
                cells[k].c[0] = (int32_t)(id % nc[0]); id /= nc[0];
                cells[k].c[1] = (int32_t)(id % nc[1]); cells[k].c[2] = (int32_t)(id / nc[1]);
                cells[k].idx = (int32_t)k;
            }
            std::vector<int32_t> owner_of_cell(occ.size(), 0);
            struct Job { size_t b, e; int r0, r1; };
            std::vector<Job> jobs{{0, cells.size(), 0, opts.world}};
            while (!jobs.empty()) {
                const Job j = jobs.back();
                jobs.pop_back();
                if (j.r1 - j.r0 <= 1 || j.e <= j.b) {
                    for (size_t k = j.b; k < j.e; ++k) owner_of_cell[(size_t)cells[k].idx] = j.r0;
                    continue;
                }
                // cut across the longest axis of the subset (cells are cubes: count them), ties -> lowest axis; the cut
                // is a plane with one staircase step: cells in lexicographic order of (axis, next axis, third axis)
                int32_t lo3[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi3[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
                for (size_t k = j.b; k < j.e; ++k)
                    for (int a = 0; a < 3; ++a) { lo3[a] = std::min(lo3[a], cells[k].c[a]); hi3[a] = std::max(hi3[a], cells[k].c[a]); }
                int ax = 0;
                for (int a = 1; a < 3; ++a) if (hi3[a] - lo3[a] > hi3[ax] - lo3[ax]) ax = a;
                const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
                std::sort(cells.begin() + (std::ptrdiff_t)j.b, cells.begin() + (std::ptrdiff_t)j.e, [&](const RcbCell &x, const RcbCell &y) {
                    if (x.c[ax] != y.c[ax]) return x.c[ax] < y.c[ax];
                    if (x.c[a1] != y.c[a1]) return x.c[a1] < y.c[a1];
                    return x.c[a2] < y.c[a2];
                });
                const int nl = (j.r1 - j.r0) / 2, nr = (j.r1 - j.r0) - nl;
                int64_t total = 0;
                for (size_t k = j.b; k < j.e; ++k) total += wcell[(size_t)cells[k].idx];
                // the prefix closest to nl / (nl + nr) of the weight (first such position)
                const size_t size = j.e - j.b;
                size_t best = 0; int64_t best_err = INT64_MAX, prefix = 0;
                for (size_t k = 0; k <= size; ++k) {
                    const int64_t err = std::llabs(prefix * (int64_t)(nl + nr) - total * (int64_t)nl);
                    if (err < best_err) { best_err = err; best = k; }
                    if (k < size) prefix += wcell[(size_t)cells[j.b + k].idx];
                }
                // no side goes without a cell while there are enough cells
                const size_t k_lo = std::min<size_t>((size_t)nl, size), k_hi = std::max(k_lo, size - std::min<size_t>((size_t)nr, size - k_lo));
                best = std::min(std::max(best, k_lo), k_hi);
                jobs.push_back({j.b + best, j.e, j.r0 + nl, j.r1});
                jobs.push_back({j.b, j.b + best, j.r0, j.r0 + nl});
            }
            parallel_chunks(n, 1 << 18, [&](int64_t, int64_t pb, int64_t pe) {
                for (int64_t q = pb; q < pe; ++q) P.owner_of_old[(size_t)q] = owner_of_cell[index_of(cell[(size_t)q])];
            });
            rank_costs(P.owner_of_old, P.rank_cost);
            P.partition = 2;
        }
    }

    timer.lap("geometry + cells");
    // ---- tiling T1 (shifted cells), computed first so that T0 can order its particles by T1 tile ----
    std::vector<int32_t> t1_of_old(n, 0);
    int32_t n_t1 = 0;
    if (tiling) {
        std::vector<int32_t> bys;
        if (n_scells <= 8 * (int64_t)n + 4096) {
            counting_sort_ids(scell, n_scells, bys);       // = sort by (shifted cell, id)
        } else {
            bys.resize(n);
            std::iota(bys.begin(), bys.end(), 0);
            std::sort(bys.begin(), bys.end(), [&](int32_t a, int32_t b) {
                if (scell[a] != scell[b]) return scell[a] < scell[b];
                return a < b;
            });
        }
        std::vector<int32_t> begins;
        for (int32_t b = 0; b < n;) {
            int32_t e = b + 1;
            while (e < n && scell[bys[e]] == scell[bys[b]]) ++e;
            split_group(bys, b, e, cap, true, in.rest, begins);
            b = e;
        }
        begins.push_back(n);
        n_t1 = (int32_t)begins.size() - 1;
        for (int32_t c = 0; c < n_t1; ++c)
            for (int32_t q = begins[c]; q < begins[c + 1]; ++q) t1_of_old[bys[q]] = c;
    }
    timer.lap("tiling T1");
    // ---- tiling T0 (aligned cells, grouped by owner) ---------------------------------------------
    std::vector<int32_t> byc;
    if (tiling && (int64_t)opts.world * n_cells <= 8 * (int64_t)n + 4096) {
        std::vector<int64_t> key(n);                       // = sort by (owner, cell, id)
        parallel_chunks(n, 1 << 20, [&](int64_t, int64_t pb, int64_t pe) {
            for (int64_t p = pb; p < pe; ++p) key[p] = (int64_t)P.owner_of_old[p] * n_cells + cell[p];
        });
        counting_sort_ids(key, (int64_t)opts.world * n_cells, byc);
    } else {
        byc.resize(n);
        std::iota(byc.begin(), byc.end(), 0);
        if (tiling)
            std::sort(byc.begin(), byc.end(), [&](int32_t a, int32_t b) {
                if (P.owner_of_old[a] != P.owner_of_old[b]) return P.owner_of_old[a] < P.owner_of_old[b];
                if (cell[a] != cell[b]) return cell[a] < cell[b];
                return a < b;
            });
        else
            std::stable_sort(byc.begin(), byc.end(), [&](int32_t a, int32_t b) { return P.owner_of_old[a] < P.owner_of_old[b]; });
    }
    std::vector<int32_t> t0_begin;
    for (int32_t b = 0; b < n;) {
        int32_t e = b + 1;
        if (tiling)
            while (e < n && P.owner_of_old[byc[e]] == P.owner_of_old[byc[b]] && cell[byc[e]] == cell[byc[b]]) ++e;
        else
            while (e < n && P.owner_of_old[byc[e]] == P.owner_of_old[byc[b]]) ++e;
        split_group(byc, b, e, cap, tiling, in.rest, t0_begin);
        b = e;
    }
    t0_begin.push_back(n);
    const int32_t n_t0 = (int32_t)t0_begin.size() - 1;
    std::vector<int32_t> t0_of_old(n);
    parallel_chunks(n_t0, 256, [&](int64_t, int64_t cb, int64_t ce) {
        for (int64_t c = cb; c < ce; ++c) {
            // final order inside a T0 tile: by T1 tile, then original id -> T0∩T1 pieces are contiguous
            if (tiling)
                std::sort(byc.begin() + t0_begin[c], byc.begin() + t0_begin[c + 1], [&](int32_t a, int32_t b) {
                    if (t1_of_old[a] != t1_of_old[b]) return t1_of_old[a] < t1_of_old[b];
                    return a < b;
                });
            else
                std::sort(byc.begin() + t0_begin[c], byc.begin() + t0_begin[c + 1]);
            for (int32_t q = t0_begin[c]; q < t0_begin[c + 1]; ++q) t0_of_old[byc[q]] = (int32_t)c;
        }
    });
    P.old_of_new = byc;
    P.new_of_old.resize(n);
    parallel_chunks(n, 1 << 20, [&](int64_t, int64_t qb, int64_t qe) {
        for (int64_t q = qb; q < qe; ++q) P.new_of_old[byc[q]] = (int32_t)q;
    });

    timer.lap("tiling T0");
    // ---- tiles: runs and tile-local indices ------------------------------------------------------
    std::vector<int32_t> lidx[2];           // tile-local index of every particle (new numbering)
    const int32_t n_tiles[2] = {n_t0, tiling ? n_t1 : 0};
    {
        Tiling &A = P.T[0];
        A.tiles.resize(n_t0);
        lidx[0].resize(n);
        A.runs.resize(n_t0);
        parallel_chunks(n_t0, 1024, [&](int64_t, int64_t cb, int64_t ce) {
            for (int64_t c = cb; c < ce; ++c) {
                Tile &t = A.tiles[c];
                t = Tile();
                t.owner = P.owner_of_old[byc[t0_begin[c]]];
                t.run_begin = (int32_t)c; t.run_count = 1;
                t.n_local = t0_begin[c + 1] - t0_begin[c];
                A.runs[c] = {t0_begin[c], t.n_local};
                for (int32_t q = t0_begin[c]; q < t0_begin[c + 1]; ++q) lidx[0][q] = q - t0_begin[c];
            }
        });
        for (int32_t c = 0; c < n_t0; ++c) A.max_local = std::max(A.max_local, A.tiles[c].n_local);
        A.max_runs = 1;
    }
    if (tiling) {
        Tiling &B = P.T[1];
        B.tiles.resize(n_t1);
        lidx[1].resize(n);
        // segments = maximal ranges of the new numbering with equal (T0 tile, T1 tile)
        struct Seg { int32_t t1, start, len; };
        std::vector<Seg> segs;
        {
            constexpr int64_t kTilesPerChunk = 512;
            const int64_t nch = ((int64_t)n_t0 + kTilesPerChunk - 1) / kTilesPerChunk;
            std::vector<std::vector<Seg>> part((size_t)nch);
            parallel_chunks(n_t0, kTilesPerChunk, [&](int64_t ch, int64_t cb, int64_t ce) {
                std::vector<Seg> &out = part[(size_t)ch];
                for (int64_t c = cb; c < ce; ++c)           // a segment never spans two T0 tiles
                    for (int32_t q = t0_begin[c]; q < t0_begin[c + 1];) {
                        int32_t e = q + 1;
                        while (e < t0_begin[c + 1] && t1_of_old[byc[e]] == t1_of_old[byc[q]]) ++e;
                        out.push_back({t1_of_old[byc[q]], q, e - q});
                        q = e;
                    }
            });
            size_t total = 0;
            for (auto &v : part) total += v.size();
            segs.reserve(total);
            for (auto &v : part) segs.insert(segs.end(), v.begin(), v.end());
        }
        std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) { return a.t1 < b.t1; });
        size_t si = 0;
        for (int32_t c = 0; c < n_t1; ++c) {
            Tile &t = B.tiles[c];
            t = Tile();
            t.run_begin = (int32_t)B.runs.size();
            int32_t l = 0; int own = -2;
            while (si < segs.size() && segs[si].t1 == c) {
                B.runs.push_back({segs[si].start, segs[si].len});
                for (int32_t q = 0; q < segs[si].len; ++q) lidx[1][segs[si].start + q] = l + q;
                l += segs[si].len;
                int ow = P.owner_of_old[byc[segs[si].start]];
                own = own == -2 ? ow : (own == ow ? ow : -1);
                ++si;
            }
            t.run_count = (int32_t)B.runs.size() - t.run_begin;
            t.n_local = l; t.owner = own;
            if (t.run_count > kMaxTileRuns) throw std::runtime_error("a shifted tile is split into too many runs");
            B.max_local = std::max(B.max_local, l);
            B.max_runs = std::max(B.max_runs, t.run_count);
        }
    }

    timer.lap("runs");
    // ---- classify constraints --------------------------------------------------------------------
    // bit0: inside T0, bit1: inside T1
    std::vector<uint8_t> cls[3];
    for (int t = 0; t < 3; ++t) {
        cls[t].assign(C.count(t), 0);
        if (!tiling) continue;
        parallel_chunks(C.count(t), 1 << 20, [&](int64_t, int64_t kb, int64_t ke) {
            for (int64_t k = kb; k < ke; ++k) {
                const int32_t *v = C.idx(t, k);
                bool s0 = true, s1 = true;
                for (int a = 1; a < kVerts[t]; ++a) { s0 &= t0_of_old[v[a]] == t0_of_old[v[0]]; s1 &= t1_of_old[v[a]] == t1_of_old[v[0]]; }
                cls[t][k] = (uint8_t)((s0 ? 1 : 0) | (s1 ? 2 : 0));
            }
        });
    }

    timer.lap("classify");
    // ---- static split: S0 (run on T0 tiles) / S1 (run on T1 tiles) -------------------------------
    // own[t][k]: 0 -> S0, 1 -> S1, 2 -> global colours. Constraints inside only one tiling have no choice. The ones
    // inside both are labelled by alternating propagation: within one type and one direction bucket (distance
    // constraints: the dominant axis of the rest-pose edge), a free constraint that shares a particle with a
    // labelled one gets the opposite label, breadth first from the forced ones. Along a lattice row this alternates
    // S0/S1 spring by spring (the odd grid shift makes the forced springs of the two tilings agree with it), so each
    // side is a set of complete matchings; on irregular meshes it halves the constraint degree of every particle
    // per side, i.e. the number of rounds per tile.
    std::vector<uint8_t> own[3];
    // balanced extra lists (irregular meshes): per list the tile of its grid per particle, and the grid's shift; constraints assigned
    // to list e carry own code kOwnBalanced + e until the T2 layer e is built from them
    constexpr uint8_t kOwnBalanced = 20;
    std::vector<std::vector<int64_t>> bal_key;
    std::vector<double> bal_frac;
    {
        std::vector<std::vector<uint8_t>> bucket(3);
        std::vector<std::vector<int8_t>> label(3);      // -2: not tiled (global), -1: free, 0/1: assigned
        struct SplitTask { int t, b; };
        std::vector<SplitTask> split_tasks;
        for (int t = 0; t < 3; ++t) {
            own[t].assign(C.count(t), 2);
            if (!tiling) continue;
            const int64_t M = C.count(t);
            if (M == 0) continue;
            bucket[t].assign(M, 0);
            label[t].assign(M, -2);
            parallel_chunks(M, 1 << 20, [&](int64_t, int64_t kb, int64_t ke) {
                for (int64_t k = kb; k < ke; ++k) {
                    const uint8_t c = cls[t][k];
                    if (c == 0) continue;
                    label[t][k] = c == 3 ? -1 : (c == 1 ? 0 : 1);
                    if (t == 0) {
                        const int32_t *v = C.idx(t, k);
                        double best = -1; int ba = 0;
                        for (int a = 0; a < 3; ++a) {
                            double d = std::fabs((double)in.rest[3 * (int64_t)v[0] + a] - in.rest[3 * (int64_t)v[1] + a]);
                            if (d > best * (1 + 1e-9)) { best = d; ba = a; }
                        }
                        bucket[t][k] = (uint8_t)ba;
                    }
                }
            });
            for (int b = 0; b < (t == 0 ? 3 : 1); ++b) split_tasks.push_back({t, b});
        }
        // the (type, bucket) classes are independent of each other: each reads and writes the labels of its own constraints only
        // (the bucket is tested before the label is touched)
        parallel_chunks((int64_t)split_tasks.size(), 1, [&](int64_t ti, int64_t, int64_t) {
            const int t = split_tasks[(size_t)ti].t, b = split_tasks[(size_t)ti].b;
            const int nv = kVerts[t];
            const int64_t M = C.count(t);
            std::vector<int8_t> &lab = label[t];
            const std::vector<uint8_t> &bk = bucket[t];
            std::vector<int64_t> inc_off((size_t)n + 1, 0);
            std::vector<int32_t> inc, queue;
            for (int64_t k = 0; k < M; ++k) if (bk[k] == b && lab[k] != -2) for (int a = 0; a < nv; ++a) ++inc_off[C.idx(t, k)[a] + 1];
            for (int32_t p = 0; p < n; ++p) inc_off[p + 1] += inc_off[p];
            inc.resize(inc_off[n]);
            {
                std::vector<int64_t> cur(inc_off.begin(), inc_off.end() - 1);
                for (int64_t k = 0; k < M; ++k) if (bk[k] == b && lab[k] != -2) for (int a = 0; a < nv; ++a) inc[cur[C.idx(t, k)[a]]++] = (int32_t)k;
            }
            for (int64_t k = 0; k < M; ++k) if (bk[k] == b && lab[k] >= 0) queue.push_back((int32_t)k);
            size_t head = 0;
            int64_t next_seed = 0;
            for (;;) {
                while (head < queue.size()) {
                    const int32_t c = queue[head++];
                    const int32_t *v = C.idx(t, c);
                    for (int a = 0; a < nv; ++a)
                        for (int64_t q = inc_off[v[a]]; q < inc_off[v[a] + 1]; ++q) {
                            const int32_t c2 = inc[q];
                            if (lab[c2] == -1) { lab[c2] = (int8_t)(1 - lab[c]); queue.push_back(c2); }
                        }
                }
                // components without a forced member: seed the lowest unlabelled constraint with S0
                while (next_seed < M && !(bk[next_seed] == b && lab[next_seed] == -1)) ++next_seed;
                if (next_seed == M) break;
                lab[next_seed] = 0;
                queue.push_back((int32_t)next_seed);
            }
        });
        for (int t = 0; t < 3; ++t)
            if (!label[t].empty())
                parallel_chunks(C.count(t), 1 << 20, [&](int64_t, int64_t kb, int64_t ke) {
                    for (int64_t k = kb; k < ke; ++k) if (label[t][k] >= 0) own[t][k] = (uint8_t)label[t][k];
                });
        // Balanced extra lists. On an irregular mesh the constraints that cross the OTHER tiling's planes are forced into one list:
        // with two lists nearly every tile holds a particle with ~56 of the tile's constraints and needs that many groups; with a
        // third grid 37 of a hub particle's ~70 constraints are still forced into one of three lists (they cross planes of the
        // other two), and a tile's program is as long as its busiest particle's share. The first n_bal T2 layers are therefore
        // further GRIDS (each in the middle of the widest gap between the planes already in use) that carry a balanced share of
        // the mesh: a constraint may sit in any list whose tile holds all its particles, and moves wherever that strictly lowers
        // the largest per-list degree among its particles. With four lists only one constraint in sixty is still forced.
        // Only meshes with leftovers are touched: a structural lattice (none) keeps its two perfect lists.
        if (tiling && opts.third_tiling && opts.third_list) {
            bool any_left = false;
            for (int t = 0; t < 3 && !any_left; ++t) for (int64_t k = 0; k < C.count(t); ++k) if (own[t][k] == 2) { any_left = true; break; }
            if (any_left) {
                const int n_bal = std::min(std::max(opts.balanced_lists, 1), kMaxBalancedLists);
                bal_key.assign((size_t)n_bal, {});
                {
                    std::vector<double> pl = {0.0, shift_frac, 1.0};
                    for (int e = 0; e < n_bal; ++e) {
                        size_t g = 0;
                        for (size_t q = 1; q + 1 < pl.size(); ++q) if (pl[q + 1] - pl[q] > pl[g + 1] - pl[g] + 1e-12) g = q;
                        const double frac = pl[g] + 0.5 * (pl[g + 1] - pl[g]);
                        pl.insert(pl.begin() + (std::ptrdiff_t)g + 1, frac);
                        bal_frac.push_back(frac);
                    }
                }
                for (int e = 0; e < n_bal; ++e) {
                    // the grid's cells, over-full ones median-split like the cells of T0 and T1 (a mesh that fills only part of its
                    // bounding box has cells far above the average)
                    std::vector<int64_t> &cellm = bal_key[(size_t)e];
                    const double frac = bal_frac[(size_t)e];
                    cellm.assign((size_t)n, 0);
                    parallel_chunks(n, 1 << 18, [&](int64_t, int64_t pb, int64_t pe) {
                        for (int64_t q = pb; q < pe; ++q) {
                            int64_t s3[3];
                            for (int a = 0; a < 3; ++a) {
                                double r = (in.rest[3 * q + a] - org[a]) / cs;
                                s3[a] = std::min(std::max((int)std::floor(r - frac) + 1, 0), nc[a]);
                            }
                            cellm[q] = (s3[2] * (nc[1] + 1) + s3[1]) * (nc[0] + 1) + s3[0];
                        }
                    });
                    std::vector<int32_t> bym(n);
                    std::iota(bym.begin(), bym.end(), 0);
                    std::sort(bym.begin(), bym.end(), [&](int32_t a, int32_t b2) { return cellm[a] != cellm[b2] ? cellm[a] < cellm[b2] : a < b2; });
                    std::vector<int32_t> begins;
                    for (int32_t b2 = 0; b2 < n;) {
                        int32_t e2 = b2 + 1;
                        while (e2 < n && cellm[bym[e2]] == cellm[bym[b2]]) ++e2;
                        split_group(bym, b2, e2, cap, true, in.rest, begins);
                        b2 = e2;
                    }
                    begins.push_back(n);
                    for (size_t c = 0; c + 1 < begins.size(); ++c)
                        for (int32_t q = begins[c]; q < begins[c + 1]; ++q) cellm[bym[q]] = (int64_t)c;      // tile id of this grid
                }
                // lists: 0 = S0 (T0 tiles), 1 = S1 (T1 tiles), 2 + e = balanced list e. opt[t][k]: bit L set = the constraint's
                // particles share a tile of list L; cur[t][k]: the list it sits in (-1: inside none -- left for the later layers)
                const int n_lists = 2 + n_bal;
                std::vector<uint8_t> opt[3];
                std::vector<int8_t> cur[3];
                std::vector<std::vector<int32_t>> deg((size_t)n_lists, std::vector<int32_t>((size_t)n, 0));
                for (int t = 0; t < 3; ++t) {
                    opt[t].assign((size_t)C.count(t), 0);
                    cur[t].assign((size_t)C.count(t), -1);
                    parallel_chunks(C.count(t), 1 << 18, [&](int64_t, int64_t kb, int64_t ke) {
                        for (int64_t k = kb; k < ke; ++k) {
                            const int32_t *v = C.idx(t, k);
                            uint8_t o = cls[t][k];
                            for (int e = 0; e < n_bal; ++e) {
                                bool same = true;
                                for (int a = 1; a < kVerts[t]; ++a) same &= bal_key[(size_t)e][v[a]] == bal_key[(size_t)e][v[0]];
                                if (same) o |= (uint8_t)(4u << e);
                            }
                            opt[t][k] = o;
                            int c = own[t][k] <= 1 ? own[t][k] : -1;
                            if (c < 0) for (int e = 0; e < n_bal && c < 0; ++e) if (o & (4u << e)) c = 2 + e;
                            cur[t][k] = (int8_t)c;
                        }
                    });
                    for (int64_t k = 0; k < C.count(t); ++k)
                        if (cur[t][k] >= 0) { const int32_t *v = C.idx(t, k); for (int a = 0; a < kVerts[t]; ++a) ++deg[(size_t)cur[t][k]][v[a]]; }
                }
                for (int pass = 0; pass < 12; ++pass) {
                    int64_t moved = 0;
                    for (int t = 0; t < 3; ++t)
                        for (int64_t k = 0; k < C.count(t); ++k) {
                            const int a = cur[t][k];
                            if (a < 0) continue;
                            const int32_t *v = C.idx(t, k);
                            int32_t here = 0;
                            for (int q = 0; q < kVerts[t]; ++q) here = std::max(here, deg[(size_t)a][v[q]]);
                            int best = -1; int32_t best_there = INT32_MAX;
                            for (int L = 0; L < n_lists; ++L) {
                                if (L == a || !((opt[t][k] >> L) & 1u)) continue;
                                int32_t there = 0;
                                for (int q = 0; q < kVerts[t]; ++q) there = std::max(there, deg[(size_t)L][v[q]]);
                                if (there < best_there) { best_there = there; best = L; }
                            }
                            if (best >= 0 && best_there + 1 < here) {
                                for (int q = 0; q < kVerts[t]; ++q) { --deg[(size_t)a][v[q]]; ++deg[(size_t)best][v[q]]; }
                                cur[t][k] = (int8_t)best;
                                ++moved;
                            }
                        }
                    if (!moved) break;
                }
                // Leftovers (inside no list's tile: on the surrogate 250 of 1.4 M, nine in ten of them surface hinges) would cost a
                // cluster layer of their own -- one more launch per substep for a handful of constraints. Where the tiles of ONE
                // balanced list that hold a leftover's particles are small enough together (surface cells are under-full), merge
                // them into one sparse tile: tiles of a layer are particle-disjoint, so the union is a valid tile, its program is as
                // long as the longer of the two, and the leftover fits. (Sparse T2 tiles are explicit particle lists: nothing
                // requires a tile to be one grid cell.)
                if (opts.merge_tiles) {
                    std::vector<std::vector<int32_t>> parent((size_t)n_bal), tsize((size_t)n_bal), tcells((size_t)n_bal);
                    for (int e = 0; e < n_bal; ++e) {
                        int64_t nt = 0;
                        for (int32_t q = 0; q < n; ++q) nt = std::max(nt, bal_key[(size_t)e][q] + 1);
                        parent[(size_t)e].resize((size_t)nt); tsize[(size_t)e].assign((size_t)nt, 0); tcells[(size_t)e].assign((size_t)nt, 1);
                        std::iota(parent[(size_t)e].begin(), parent[(size_t)e].end(), 0);
                        for (int32_t q = 0; q < n; ++q) ++tsize[(size_t)e][(size_t)bal_key[(size_t)e][q]];
                    }
                    auto find = [&](int e, int32_t x) {
                        auto &pa = parent[(size_t)e];
                        while (pa[(size_t)x] != x) { pa[(size_t)x] = pa[(size_t)pa[(size_t)x]]; x = pa[(size_t)x]; }
                        return x;
                    };
                    int64_t merged = 0;
                    // two sweeps: first only unions that stay a small tile (512 particles); whatever is still left may then build a
                    // large tile (1024): one tile of the large kind costs less than a launch of its own for the last few constraints
                    for (int sweep = 0; sweep < 2; ++sweep)
                    for (int t = 0; t < 3; ++t)
                        for (int64_t k = 0; k < C.count(t); ++k) {
                            if (cur[t][k] >= 0) continue;
                            const int64_t cap_now = sweep == 0 ? kMergedTileCap : (int64_t)kMaxTileLocal;
                            const int32_t *v = C.idx(t, k);
                            int best = -1; int64_t best_total = INT64_MAX;
                            int32_t roots[4];
                            for (int e = 0; e < n_bal; ++e) {
                                int nr = 0; int64_t total = 0; int cells_in = 0;
                                for (int a = 0; a < kVerts[t]; ++a) {
                                    const int32_t r = find(e, (int32_t)bal_key[(size_t)e][v[a]]);
                                    bool seen = false;
                                    for (int q = 0; q < nr; ++q) seen |= roots[q] == r;
                                    if (!seen) { roots[nr++] = r; total += tsize[(size_t)e][(size_t)r]; cells_in += tcells[(size_t)e][(size_t)r]; }
                                }
                                // (a union of at most kMaxMergedCells original tiles: the leftovers this is for span neighbouring cells; long-range
                                // constraints must not chain the whole mesh into one tile -- they keep the cluster layers / global colours)
                                if (total <= cap_now && cells_in <= kMaxMergedCells && total < best_total) { best_total = total; best = e; }
                            }
                            if (best < 0) continue;
                            int32_t root = INT32_MAX;
                            for (int a = 0; a < kVerts[t]; ++a) root = std::min(root, find(best, (int32_t)bal_key[(size_t)best][v[a]]));
                            for (int a = 0; a < kVerts[t]; ++a) {
                                const int32_t r = find(best, (int32_t)bal_key[(size_t)best][v[a]]);
                                if (r != root) {
                                    parent[(size_t)best][(size_t)r] = root;
                                    tsize[(size_t)best][(size_t)root] += tsize[(size_t)best][(size_t)r];
                                    tcells[(size_t)best][(size_t)root] += tcells[(size_t)best][(size_t)r];
                                }
                            }
                            cur[t][k] = (int8_t)(2 + best);
                            for (int a = 0; a < kVerts[t]; ++a) ++deg[(size_t)(2 + best)][v[a]];
                            ++merged;
                        }
                    if (merged)
                        for (int e = 0; e < n_bal; ++e)
                            parallel_chunks(n, 1 << 18, [&](int64_t, int64_t pb, int64_t pe) {
                                for (int64_t q = pb; q < pe; ++q) {      // (read-only walk: find() above compresses paths, this one must not race)
                                    int32_t x = (int32_t)bal_key[(size_t)e][(size_t)q];
                                    while (parent[(size_t)e][(size_t)x] != x) x = parent[(size_t)e][(size_t)x];
                                    bal_key[(size_t)e][(size_t)q] = x;
                                }
                            });
                    if (timer.on) std::fprintf(stderr, "[plan] leftovers placed by merging tiles: %lld\n", (long long)merged);
                }
                if (timer.on) {
                    for (int L = 0; L < n_lists; ++L) {
                        int32_t mx = 0; std::vector<int32_t> forced((size_t)n, 0);
                        for (int32_t q = 0; q < n; ++q) mx = std::max(mx, deg[(size_t)L][q]);
                        for (int t = 0; t < 3; ++t) for (int64_t k = 0; k < C.count(t); ++k)
                            if (opt[t][k] == (1u << L)) { const int32_t *v = C.idx(t, k); for (int a = 0; a < kVerts[t]; ++a) ++forced[v[a]]; }
                        int32_t fm = 0; for (int32_t q = 0; q < n; ++q) fm = std::max(fm, forced[q]);
                        std::fprintf(stderr, "[plan] list %d: max degree %d, max forced degree %d\n", L, mx, fm);
                    }
                }
                for (int t = 0; t < 3; ++t)
                    for (int64_t k = 0; k < C.count(t); ++k) own[t][k] = cur[t][k] < 0 ? 2 : (cur[t][k] < 2 ? (uint8_t)cur[t][k] : (uint8_t)(kOwnBalanced + cur[t][k] - 2));
            }
        }
    }

    timer.lap("static split");
    // ---- tile programs ---------------------------------------------------------------------------
    // seq[tl] = (type,id) of the tiling's constraints in execution order; tile slices recorded in the tiles
    std::vector<uint8_t> seq_type[3];
    std::vector<int32_t> seq_id[3];
    std::vector<int64_t> seq_groups[3];   // group boundaries (end offsets) inside seq
    // programs of the tiles [tile_begin, tile_end) of tiling tl: the constraints with own code `code`, bucketed by tof
    // (tile of a particle, caller numbering), tile-local indices from lmap (new numbering)
    auto build_programs = [&](int tl, int32_t tile_begin, int32_t tile_end, const std::vector<int32_t> &tof,
                              const std::vector<int32_t> &lmap, uint8_t code) {
        if (tile_end <= tile_begin) return;
        Tiling &TT = P.T[tl];
        std::vector<int64_t> off[3];
        std::vector<int32_t> lst[3];
        const int32_t nt = tile_end - tile_begin;
        parallel_chunks(3, 1, [&](int64_t t, int64_t, int64_t) {         // bucket each type's constraints by tile
            auto &o = off[t];
            o.assign((size_t)nt + 1, 0);
            for (int64_t k = 0; k < C.count((int)t); ++k) if (own[t][k] == code) ++o[tof[C.idx((int)t, k)[0]] - tile_begin + 1];
            for (int32_t c = 0; c < nt; ++c) o[c + 1] += o[c];
            lst[t].resize(o[nt]);
            std::vector<int64_t> cur(o.begin(), o.end() - 1);
            for (int64_t k = 0; k < C.count((int)t); ++k) if (own[t][k] == code) lst[t][cur[tof[C.idx((int)t, k)[0]] - tile_begin]++] = (int32_t)k;
        });
        // Tiles are independent: chunks of tiles build their pieces of the tiling's arrays side by side, the pieces are
        // then laid end to end in tile order (the arrays come out exactly as a tile-by-tile loop would fill them).
        struct Piece {
            std::vector<uint32_t> rounds, t_dist, t_quad;
            std::vector<int32_t> t_dist_id, t_quad_id, seq_id;
            std::vector<uint8_t> t_quad_type, seq_type;
            std::vector<int64_t> seq_groups;
        };
        constexpr int64_t kTilesPerChunk = 64;
        const int64_t nch = ((int64_t)nt + kTilesPerChunk - 1) / kTilesPerChunk;
        std::vector<Piece> pieces((size_t)nch);
        parallel_chunks(nt, kTilesPerChunk, [&](int64_t ch, int64_t cb, int64_t ce) {
            Piece &Q = pieces[(size_t)ch];
            std::vector<Mask128> used((size_t)kMaxTileLocal);
            std::vector<int> col_all, col_t[3];
            std::vector<int32_t> lv[3];
            std::vector<std::vector<int32_t>> by[3];
            for (int64_t ci = cb; ci < ce; ++ci) {
                const int32_t c = tile_begin + (int32_t)ci;
                Tile &tile = TT.tiles[c];
                // offsets are relative to the piece until the pieces are placed
                tile.round_begin = (int32_t)Q.rounds.size();
                tile.d_begin = (int64_t)Q.t_dist.size(); tile.q_begin = (int64_t)Q.t_quad_id.size();
                tile.seq_begin = (int64_t)Q.seq_id.size();
                // the tile's constraints of every type, tile-local particle indices
                const int32_t *it[3];
                int64_t cnt[3];
                for (int t = 0; t < 3; ++t) {
                    it[t] = lst[t].data() + off[t][ci];
                    cnt[t] = off[t][ci + 1] - off[t][ci];
                    const int nv = kVerts[t];
                    lv[t].resize((size_t)cnt[t] * nv);
                    for (int64_t k = 0; k < cnt[t]; ++k) {
                        const int32_t *v = C.idx(t, it[t][k]);
                        for (int a = 0; a < nv; ++a) lv[t][k * nv + a] = lmap[P.new_of_old[v[a]]];
                    }
                }
                // Colours -> groups. A group is a set of constraints of the tile that share no particle: the kernel
                // projects a group's constraints concurrently, one barrier per group. With mixed groups the three types
                // are coloured TOGETHER (hinges first, then tets, then springs: the long projections get the low
                // colours), so a tile needs about as many groups as its busiest particle has constraints instead of the
                // sum of the per-type colour counts; otherwise each type is coloured on its own, springs first.
                // Either way a mesh with springs only gets the same groups.
                int ncol_t[3] = {0, 0, 0}, col_base[3] = {0, 0, 0}, ncol = 0;
                if (mixed_groups && (cnt[1] > 0 || cnt[2] > 0)) {
                    const int64_t total = cnt[0] + cnt[1] + cnt[2];
                    const int64_t b1 = cnt[2], b0 = cnt[2] + cnt[1];        // sequence: bending, volume, distance
                    ncol = greedy_colour(total, [&](int64_t k, int &nvo) {
                        if (k < b1) { nvo = 4; return (const int32_t *)lv[2].data() + (size_t)k * 4; }
                        if (k < b0) { nvo = 4; return (const int32_t *)lv[1].data() + (size_t)(k - b1) * 4; }
                        nvo = 2; return (const int32_t *)lv[0].data() + (size_t)(k - b0) * 2;
                    }, used, col_all);
                    col_t[2].assign(col_all.begin(), col_all.begin() + b1);
                    col_t[1].assign(col_all.begin() + b1, col_all.begin() + b0);
                    col_t[0].assign(col_all.begin() + b0, col_all.end());
                } else {
                    for (int t = 0; t < 3; ++t) {
                        if (cnt[t] == 0) continue;
                        const int nv = kVerts[t];
                        ncol_t[t] = greedy_colour(cnt[t], [&](int64_t k, int &nvo) { nvo = nv; return (const int32_t *)lv[t].data() + (size_t)k * nv; }, used, col_t[t]);
                    }
                    col_base[1] = ncol_t[0]; col_base[2] = ncol_t[0] + ncol_t[1];
                    ncol = ncol_t[0] + ncol_t[1] + ncol_t[2];
                }
                for (int t = 0; t < 3; ++t) {
                    by[t].assign((size_t)ncol, {});
                    for (int64_t k = 0; k < cnt[t]; ++k) by[t][(size_t)(col_base[t] + col_t[t][k])].push_back((int32_t)k);
                }
                for (int c = 0; c < ncol; ++c) {
                    size_t pieces = 0;
                    for (int t = 0; t < 3; ++t) {
                        std::vector<int32_t> &colv = by[t][(size_t)c];
                        const int nv = kVerts[t];
                        if (bank_aware && colv.size() > (size_t)kLdsGroup) {
                            // Lane order inside a colour is free (its constraints share no particle). The LDS serves a 16-byte
                            // gather or scatter for kLdsGroup lanes per cycle, conflict-free when their float4 indices differ
                            // modulo kLdsGroup: deal the constraints so that every aligned group of kLdsGroup lanes holds
                            // distinct first indices and, where the choice allows, distinct second indices (bank conflicts of
                            // the mid-tick kernel at 256^3: 21.2 M -> 9.5 M cycles per launch, SQ_LDS_BANK_CONFLICT).
                            std::vector<std::vector<int32_t>> bucket((size_t)kLdsGroup);
                            for (int32_t k : colv) bucket[lv[t][(size_t)k * nv] % kLdsGroup].push_back(k);
                            colv.clear();
                            for (bool any = true; any;) {
                                any = false;
                                uint32_t used_j = 0;
                                for (auto &bq : bucket) {
                                    if (bq.empty()) continue;
                                    any = true;
                                    size_t pick = bq.size() - 1;
                                    for (size_t d = 0; d < bq.size(); ++d) {
                                        const size_t q = bq.size() - 1 - d;
                                        if (!((used_j >> (lv[t][(size_t)bq[q] * nv + 1] % kLdsGroup)) & 1u)) { pick = q; break; }
                                    }
                                    used_j |= 1u << (lv[t][(size_t)bq[pick] * nv + 1] % kLdsGroup);
                                    colv.push_back(bq[pick]);
                                    bq.erase(bq.begin() + (std::ptrdiff_t)pick);
                                }
                            }
                        }
                        pieces = std::max(pieces, (colv.size() + kRoundThreads - 1) / kRoundThreads);
                    }
                    // a colour with more than kRoundThreads constraints of a type is cut into several groups
                    for (size_t piece = 0; piece < pieces; ++piece) {
                        uint32_t word = 0;
                        for (int t = 0; t < 3; ++t) {
                            const std::vector<int32_t> &colv = by[t][(size_t)c];
                            const size_t s0 = std::min(colv.size(), piece * kRoundThreads), s1 = std::min(colv.size(), s0 + kRoundThreads);
                            word |= (uint32_t)(s1 - s0) << (10 * t);
                            for (size_t q = s0; q < s1; ++q) {
                                const int32_t k = colv[q];
                                const int32_t *l = lv[t].data() + (size_t)k * kVerts[t];
                                if (t == 0) {
                                    Q.t_dist.push_back((uint32_t)l[0] | ((uint32_t)l[1] << 16));
                                    Q.t_dist_id.push_back(it[t][k]);
                                } else {
                                    Q.t_quad.push_back((uint32_t)l[0] | ((uint32_t)l[1] << 16));
                                    Q.t_quad.push_back((uint32_t)l[2] | ((uint32_t)l[3] << 16));
                                    Q.t_quad_id.push_back(it[t][k]);
                                    Q.t_quad_type.push_back((uint8_t)t);
                                }
                                Q.seq_type.push_back((uint8_t)t);
                                Q.seq_id.push_back(it[t][k]);
                            }
                        }
                        Q.rounds.push_back(word);
                        Q.seq_groups.push_back((int64_t)Q.seq_id.size());
                    }
                }
                tile.seq_end = (int64_t)Q.seq_id.size();
                tile.n_rounds = (int32_t)Q.rounds.size() - tile.round_begin;
                tile.d_end = (int64_t)Q.t_dist.size(); tile.q_end = (int64_t)Q.t_quad_id.size();
            }
        });
        // place the pieces
        std::vector<int64_t> r0((size_t)nch + 1), d0((size_t)nch + 1), q0((size_t)nch + 1), s0v((size_t)nch + 1), g0((size_t)nch + 1);
        r0[0] = (int64_t)TT.rounds.size(); d0[0] = (int64_t)TT.t_dist.size(); q0[0] = (int64_t)TT.t_quad_id.size();
        s0v[0] = (int64_t)seq_id[tl].size(); g0[0] = (int64_t)seq_groups[tl].size();
        for (int64_t ch = 0; ch < nch; ++ch) {
            const Piece &Q = pieces[(size_t)ch];
            r0[ch + 1] = r0[ch] + (int64_t)Q.rounds.size(); d0[ch + 1] = d0[ch] + (int64_t)Q.t_dist.size();
            q0[ch + 1] = q0[ch] + (int64_t)Q.t_quad_id.size(); s0v[ch + 1] = s0v[ch] + (int64_t)Q.seq_id.size();
            g0[ch + 1] = g0[ch] + (int64_t)Q.seq_groups.size();
        }
        if (r0[nch] > INT32_MAX) throw std::runtime_error("too many rounds");
        TT.rounds.resize((size_t)r0[nch]); TT.t_dist.resize((size_t)d0[nch]); TT.t_dist_id.resize((size_t)d0[nch]);
        TT.t_quad.resize((size_t)q0[nch] * 2); TT.t_quad_id.resize((size_t)q0[nch]); TT.t_quad_type.resize((size_t)q0[nch]);
        seq_type[tl].resize((size_t)s0v[nch]); seq_id[tl].resize((size_t)s0v[nch]); seq_groups[tl].resize((size_t)g0[nch]);
        parallel_chunks(nch, 1, [&](int64_t ch, int64_t, int64_t) {
            const Piece &Q = pieces[(size_t)ch];
            std::copy(Q.rounds.begin(), Q.rounds.end(), TT.rounds.begin() + r0[ch]);
            std::copy(Q.t_dist.begin(), Q.t_dist.end(), TT.t_dist.begin() + d0[ch]);
            std::copy(Q.t_dist_id.begin(), Q.t_dist_id.end(), TT.t_dist_id.begin() + d0[ch]);
            std::copy(Q.t_quad.begin(), Q.t_quad.end(), TT.t_quad.begin() + 2 * q0[ch]);
            std::copy(Q.t_quad_id.begin(), Q.t_quad_id.end(), TT.t_quad_id.begin() + q0[ch]);
            std::copy(Q.t_quad_type.begin(), Q.t_quad_type.end(), TT.t_quad_type.begin() + q0[ch]);
            std::copy(Q.seq_type.begin(), Q.seq_type.end(), seq_type[tl].begin() + s0v[ch]);
            std::copy(Q.seq_id.begin(), Q.seq_id.end(), seq_id[tl].begin() + s0v[ch]);
            for (size_t g = 0; g < Q.seq_groups.size(); ++g) seq_groups[tl][(size_t)g0[ch] + g] = Q.seq_groups[g] + s0v[ch];
            const int64_t cb = ch * kTilesPerChunk, ce = std::min<int64_t>(nt, cb + kTilesPerChunk);
            for (int64_t ci = cb; ci < ce; ++ci) {
                Tile &tile = TT.tiles[tile_begin + ci];
                tile.round_begin += (int32_t)r0[ch];
                tile.d_begin += d0[ch]; tile.d_end += d0[ch]; tile.q_begin += q0[ch]; tile.q_end += q0[ch];
                tile.seq_begin += s0v[ch]; tile.seq_end += s0v[ch];
            }
        });
    };
    if (n_tiles[0]) build_programs(0, 0, n_tiles[0], t0_of_old, lidx[0], 0);
    if (n_tiles[1]) build_programs(1, 0, n_tiles[1], t1_of_old, lidx[1], 1);

    // ---- third tiling T2, in layers ----------------------------------------------------------------
    // A constraint inside neither T0 nor T1 (it crosses a T0 plane on one axis and a T1 plane on another: about one in
    // ten on a tet mesh) would need a global colour, i.e. one tiny launch per colour and substep. Most of them fit a
    // cell of another shifted grid: such constraints are projected in LDS by one extra tile kernel per substep and
    // layer, on sparse tiles that hold just the particles they touch (explicit particle lists instead of runs). Up to
    // kMaxT2Layers grids with different shifts are tried in turn on what is still left. Like a T1 tile, a T2 tile that
    // spans ranks is projected redundantly by every rank that owns one of its particles, on ghosts refreshed just
    // before (one halo slot per layer, positions only): the plan does not depend on the partition.
    if (tiling && opts.third_tiling) {
        struct Cand { int64_t key; uint8_t type; int32_t id; };
        std::vector<Cand> cand;
        std::vector<int32_t> t2_of_old, lidx2, members, uf_parent, uf_size, uf_tile, uf_ptile;
        // layer shifts (fractions of a cell): first the middle of the widest gap between the T0 and T1 planes, then a
        // golden-ratio walk, skipping positions within 6 % of a cell of any plane already in use
        std::vector<double> planes = {0.0, shift_frac, 1.0};
        double next_frac = first_t2_frac;
        const int n_bal = (int)bal_key.size();
        for (int layer = 0; layer < kMaxT2Layers; ++layer) {
            const bool keyed = layer < n_bal;       // a balanced list: its grid and its constraints were chosen by the static split
            double frac = keyed ? bal_frac[(size_t)layer] : next_frac;
            for (int tries = 0; !keyed && tries < 32; ++tries) {
                bool close = false;
                for (double pl : planes) close |= std::fabs(frac - pl) < 0.06;
                if (!close) break;
                frac += 0.381966011250105; frac -= std::floor(frac);
            }
            planes.push_back(frac);
            if (!keyed || layer + 1 == n_bal) { next_frac = frac + 0.381966011250105; next_frac -= std::floor(next_frac); }
            auto cell2 = [&](int32_t q) {
                int64_t s3[3];
                for (int a = 0; a < 3; ++a) {
                    double r = (in.rest[3 * (int64_t)q + a] - org[a]) / cs;
                    s3[a] = std::min(std::max((int)std::floor(r - frac) + 1, 0), nc[a]);
                }
                return (s3[2] * (nc[1] + 1) + s3[1]) * (nc[0] + 1) + s3[0];
            };
            cand.clear();
            const uint8_t want = keyed ? (uint8_t)(kOwnBalanced + layer) : (uint8_t)2;
            int64_t left = 0;
            for (int t = 0; t < 3; ++t)
                for (int64_t k = 0; k < C.count(t); ++k) left += own[t][k] == want;
            if (left == 0) { if (keyed) continue; break; }
            // Few constraints left (they sit where the planes of the grids already tried cross): another grid would catch
            // only part of them and every further layer is one more launch per substep. Cluster layer instead: the
            // connected components of what is left become the sparse tiles -- a component shares no particle with any
            // other, so they all fit ONE layer; only a component of more than kMaxTileLocal particles is cut, and the
            // constraints across the cut wait for the next layer.
            const bool cluster = opts.cluster_layers && !keyed && layer > 0 && left <= std::max<int64_t>(4096, (P.m[0] + P.m[1] + P.m[2]) / 50);
            if (cluster) {
                std::vector<int32_t> &parent = uf_parent;
                if (parent.empty()) parent.assign((size_t)n, -1);        // -1: not touched in this layer
                std::vector<int32_t> touched;
                auto find = [&](int32_t x) {
                    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
                    return x;
                };
                for (int t = 0; t < 3; ++t)
                    for (int64_t k = 0; k < C.count(t); ++k) {
                        if (own[t][k] != 2) continue;
                        const int32_t *v = C.idx(t, k);
                        for (int a = 0; a < kVerts[t]; ++a) if (parent[v[a]] < 0) { parent[v[a]] = v[a]; touched.push_back(v[a]); }
                        int32_t r0 = find(v[0]);
                        for (int a = 1; a < kVerts[t]; ++a) {
                            const int32_t ra = find(v[a]);
                            if (ra != r0) { const int32_t lo = std::min(ra, r0), hi = std::max(ra, r0); parent[hi] = lo; r0 = lo; }
                        }
                    }
                // particles per component, and a tile per component (or per piece of an over-full one), in constraint order
                std::vector<int32_t> &csize = uf_size, &ctile = uf_tile, &ptile = uf_ptile;
                if (csize.empty()) { csize.assign((size_t)n, 0); ctile.assign((size_t)n, -1); ptile.assign((size_t)n, -1); }
                for (int32_t p : touched) ++csize[find(p)];
                int32_t n_new = 0;
                std::vector<int32_t> fill;                                  // particles in each new tile
                for (int t = 0; t < 3; ++t)
                    for (int64_t k = 0; k < C.count(t); ++k) {
                        if (own[t][k] != 2) continue;
                        const int32_t *v = C.idx(t, k);
                        const int32_t root = find(v[0]);
                        const int nv = kVerts[t];
                        int32_t tile = -1;
                        if (csize[root] <= kMaxTileLocal) {                 // the whole component is one tile
                            if (ctile[root] < 0) { ctile[root] = n_new++; fill.push_back(csize[root]); }
                            tile = ctile[root];
                        } else {                                            // over-full component: fill tiles greedily
                            int32_t seen = -1, fresh = 0; bool two = false;
                            for (int a = 0; a < nv; ++a) {
                                const int32_t pt = ptile[v[a]];
                                if (pt < 0) ++fresh; else if (seen < 0) seen = pt; else if (pt != seen) two = true;
                            }
                            if (two) continue;                              // bridges two tiles of this layer: next layer
                            if (seen < 0) { seen = ctile[root]; if (seen < 0 || fill[(size_t)seen] + fresh > kMaxTileLocal) { seen = n_new++; fill.push_back(0); ctile[root] = seen; } }
                            else if (fill[(size_t)seen] + fresh > kMaxTileLocal) continue;
                            tile = seen;
                            for (int a = 0; a < nv; ++a) if (ptile[v[a]] < 0) { ptile[v[a]] = tile; ++fill[(size_t)tile]; }
                        }
                        cand.push_back({(int64_t)tile, (uint8_t)t, (int32_t)k});
                    }
                for (int32_t p : touched) { csize[p] = 0; ctile[p] = -1; ptile[p] = -1; parent[p] = -1; }
            } else {
                for (int t = 0; t < 3; ++t)
                    for (int64_t k = 0; k < C.count(t); ++k) {
                        if (own[t][k] != want) continue;
                        const int32_t *v = C.idx(t, k);
                        const int64_t c0 = keyed ? bal_key[(size_t)layer][v[0]] : cell2(v[0]);
                        bool same = true;
                        for (int a = 1; a < kVerts[t]; ++a) same &= (keyed ? bal_key[(size_t)layer][v[a]] : cell2(v[a])) == c0;
                        if (same) cand.push_back({c0, (uint8_t)t, (int32_t)k});
                    }
            }
            if (cand.empty()) continue;
            std::stable_sort(cand.begin(), cand.end(), [](const Cand &x, const Cand &y) { return x.key < y.key; });
            t2_of_old.assign(n, -1);
            lidx2.assign(n, -1);
            Tiling &T2 = P.T[2];
            const int32_t tile_begin = (int32_t)T2.tiles.size();
            const uint8_t code = (uint8_t)(3 + layer);
            for (size_t b = 0; b < cand.size();) {
                size_t e = b + 1;
                while (e < cand.size() && cand[e].key == cand[b].key) ++e;
                members.clear();
                for (size_t q = b; q < e; ++q) {
                    const int32_t *v = C.idx(cand[q].type, cand[q].id);
                    for (int a = 0; a < kVerts[cand[q].type]; ++a) members.push_back(P.new_of_old[v[a]]);
                }
                std::sort(members.begin(), members.end());
                members.erase(std::unique(members.begin(), members.end()), members.end());
                if ((int)members.size() <= kMaxTileLocal) {      // an over-full cell keeps its constraints for the next layer / the global colours
                    Tile t = Tile();
                    t.owner = P.owner_of_old[P.old_of_new[members[0]]];
                    for (int32_t mq : members) if (P.owner_of_old[P.old_of_new[mq]] != t.owner) t.owner = -1;
                    t.run_begin = 0; t.run_count = 0;
                    t.n_local = (int32_t)members.size();
                    t.gather_begin = (int64_t)T2.gather.size();
                    const int32_t id = (int32_t)T2.tiles.size();
                    for (size_t q = 0; q < members.size(); ++q) {
                        T2.gather.push_back(members[q]);
                        lidx2[members[q]] = (int32_t)q;
                        t2_of_old[P.old_of_new[members[q]]] = id;
                    }
                    T2.max_local = std::max(T2.max_local, t.n_local);
                    T2.tiles.push_back(t);
                    for (size_t q = b; q < e; ++q) own[cand[q].type][cand[q].id] = code;
                } else if (keyed) {
                    for (size_t q = b; q < e; ++q) own[cand[q].type][cand[q].id] = 2;       // (back to the leftovers: later layers / global colours)
                }
                b = e;
            }
            const int32_t tile_end = (int32_t)T2.tiles.size();
            if (tile_end > tile_begin) {
                build_programs(2, tile_begin, tile_end, t2_of_old, lidx2, code);
                P.t2_layers.push_back({tile_begin, tile_end});
            }
        }
        for (int t = 0; t < 3; ++t) for (auto &o : own[t]) if (o > 3) o = 3;       // 3 = some T2 layer
    }

    timer.lap("tile programs");
    // ---- global colours for constraints inside neither tiling ------------------------------------
    {
        std::vector<Mask128> gused;
        std::vector<int32_t> left;
        std::vector<int> colr;
        for (int t = 0; t < 3; ++t) {
            left.clear();
            for (int64_t k = 0; k < C.count(t); ++k) if (own[t][k] == 2) left.push_back((int32_t)k);
            if (left.empty()) continue;
            if (gused.empty()) gused.assign(n, Mask128());
            const int nv = kVerts[t];
            int ncol = greedy_colour((int64_t)left.size(), [&](int64_t k, int &nvo) { nvo = nv; return C.idx(t, left[k]); }, gused, colr);
            size_t base = P.gcolours.size();
            P.gcolours.resize(base + ncol);
            for (int c = 0; c < ncol; ++c) P.gcolours[base + c].type = t;
            for (size_t k = 0; k < left.size(); ++k) {
                GColour &g = P.gcolours[base + colr[k]];
                g.ids.push_back(left[k]);
                const int32_t *v = C.idx(t, left[k]);
                for (int a = 1; a < nv; ++a) if (P.owner_of_old[v[a]] != P.owner_of_old[v[0]]) g.cut = true;
            }
            P.cons_in_global += (int64_t)left.size();
        }
    }
    P.cons_in_tiles = P.m[0] + P.m[1] + P.m[2] - P.cons_in_global;

    timer.lap("global colours");
    // ---- published orders per parity --------------------------------------------------------------
    // parity p: S_p on the tiles of T_p, S2 on the tiles of T2, the global colours, S_(1-p) on the tiles of T_(1-p)
    parallel_chunks(2, 1, [&](int64_t p64, int64_t, int64_t) {          // the two parities fill disjoint outputs
        const int p = (int)p64;
        auto &ot = P.order_type[p]; auto &oi = P.order_id[p];
        auto &tasks = P.task_off[p]; auto &groups = P.group_off[p];
        {
            const size_t total = (size_t)(P.m[0] + P.m[1] + P.m[2]);
            size_t n_tasks = 1, n_groups = 1;
            for (int tl = 0; tl < 3; ++tl) { n_tasks += P.T[tl].tiles.size(); n_groups += seq_groups[tl].size(); }
            for (const GColour &g : P.gcolours) { n_tasks += g.ids.size() / kRoundThreads + 1; n_groups += g.ids.size() / kRoundThreads + 1; }
            ot.reserve(total); oi.reserve(total); tasks.reserve(n_tasks); groups.reserve(n_groups);
        }
        tasks.push_back(0); groups.push_back(0);
        auto append_tiles = [&](int tl, int kind, int32_t tile_begin = 0, int32_t tile_end = -1, int layer = -1) {
            if (tile_end < 0) tile_end = (int32_t)P.T[tl].tiles.size();
            if (tile_end <= tile_begin) return;
            Phase ph; ph.kind = kind; ph.type = -1; ph.tiling = tl; ph.gcolour = -1; ph.halo_slot = -1; ph.layer = layer;
            ph.order_begin = (int64_t)oi.size(); ph.task_begin = (int64_t)tasks.size() - 1;
            // the tiles' constraints are one contiguous slice of the tiling's sequence
            const int64_t sb = P.T[tl].tiles[tile_begin].seq_begin, se = P.T[tl].tiles[tile_end - 1].seq_end;
            const int64_t base = (int64_t)oi.size() - sb;
            ot.insert(ot.end(), seq_type[tl].begin() + sb, seq_type[tl].begin() + se);
            oi.insert(oi.end(), seq_id[tl].begin() + sb, seq_id[tl].begin() + se);
            {
                const auto &sg = seq_groups[tl];      // ascending
                auto gb = std::upper_bound(sg.begin(), sg.end(), sb), ge = std::upper_bound(sg.begin(), sg.end(), se);
                for (auto it = gb; it != ge; ++it) groups.push_back(base + *it);
            }
            for (int32_t c = tile_begin; c < tile_end; ++c) {
                Tile &tile = P.T[tl].tiles[c];
                tile.order_begin[p] = base + tile.seq_begin; tile.order_end[p] = base + tile.seq_end;
                if (tile.seq_end > tile.seq_begin) tasks.push_back(base + tile.seq_end);
            }
            ph.order_end = (int64_t)oi.size(); ph.task_end = (int64_t)tasks.size() - 1;
            if (ph.order_end > ph.order_begin) P.phases[p].push_back(ph);
        };
        const int first = tiling ? p : 0;
        if (!P.T[first].tiles.empty()) append_tiles(first, 1);
        for (size_t ly = 0; ly < P.t2_layers.size(); ++ly) append_tiles(2, 3, P.t2_layers[ly].first, P.t2_layers[ly].second, (int)ly);
        for (size_t gc = 0; gc < P.gcolours.size(); ++gc) {
            const GColour &g = P.gcolours[gc];
            Phase ph; ph.kind = 0; ph.type = g.type; ph.tiling = -1; ph.gcolour = (int)gc;
            ph.halo_slot = g.cut ? 2 + (int)gc : -1;
            ph.order_begin = (int64_t)oi.size(); ph.task_begin = (int64_t)tasks.size() - 1;
            int cnt = 0;
            for (int32_t id : g.ids) {
                ot.push_back((uint8_t)g.type); oi.push_back(id);
                if (++cnt == kRoundThreads) { tasks.push_back((int64_t)oi.size()); groups.push_back((int64_t)oi.size()); cnt = 0; }
            }
            if (cnt) { tasks.push_back((int64_t)oi.size()); groups.push_back((int64_t)oi.size()); }
            ph.order_end = (int64_t)oi.size(); ph.task_end = (int64_t)tasks.size() - 1;
            P.phases[p].push_back(ph);
        }
        if (tiling && !P.T[1 - p].tiles.empty()) append_tiles(1 - p, 2);
        if ((int64_t)oi.size() != P.m[0] + P.m[1] + P.m[2]) throw std::runtime_error("planner lost constraints");
    });
    // T2 layers whose tiles span ranks need ghost positions: one halo slot per layer, after the global colours' slots
    if (opts.world > 1)
        for (size_t ly = 0; ly < P.t2_layers.size(); ++ly) {
            bool multi2 = false;
            for (int32_t c = P.t2_layers[ly].first; c < P.t2_layers[ly].second; ++c) multi2 |= P.T[2].tiles[c].owner < 0;
            if (multi2)
                for (int p = 0; p < 2; ++p)
                    for (Phase &ph : P.phases[p]) if (ph.kind == 3 && ph.layer == (int)ly) ph.halo_slot = 2 + (int)P.gcolours.size() + (int)ly;
        }
    // halo slot 1: T1 tiles with more than one owner need ghosts (positions and previous positions)
    if (tiling && opts.world > 1) {
        bool multi = false;
        for (const Tile &t : P.T[1].tiles) multi |= t.owner < 0;
        if (multi)
            for (int p = 0; p < 2; ++p)
                for (Phase &ph : P.phases[p]) if (ph.kind != 0 && ph.tiling == 1) ph.halo_slot = 1;
    }
    timer.lap("published orders");
}

void extract_local(const Plan &P, const Input &in, int rank, LocalPlan &L) {
    PlanTimer timer;
    L = LocalPlan();
    L.rank = rank; L.world = P.opts.world;
    const int32_t n = P.n;
    const int world = P.opts.world;
    Cons C{&in};
    auto owner_new = [&](int32_t nw) { return P.owner_of_old[P.old_of_new[nw]]; };
    int32_t ob = n, oe = n;
    for (int32_t q = 0; q < n; ++q) if (owner_new(q) == rank) { ob = q; break; }
    for (int32_t q = ob; q < n; ++q) if (owner_new(q) != rank) { oe = q; break; }
    if (ob == n) ob = oe = 0;
    L.n_owned = oe - ob;
    // (slot, consumer rank, particle) triples where the particle's owner differs from the consumer and
    // either side is this rank
    struct Need { int32_t slot, consumer, nw; };
    std::vector<Need> needs;
    L.halo.resize(2 + P.gcolours.size() + P.t2_layers.size());
    for (auto &h : L.halo) { h.send_idx.assign(world, {}); h.recv_idx.assign(world, {}); }
    for (int p = 0; p < 2; ++p) L.order_mask[p].assign(P.order_id[p].size(), 0);
    std::vector<int> owners;
    // tiles
    for (int tl = 0; tl < 3; ++tl) {
        const Tiling &TT = P.T[tl];
        for (int32_t c = 0; c < (int32_t)TT.tiles.size(); ++c) {
            const Tile &tile = TT.tiles[c];
            bool mine;
            if (tile.owner >= 0) {
                mine = tile.owner == rank;
            } else if (tl == 2) {
                int32_t layer_of_t2 = 0;
                while (layer_of_t2 + 1 < (int32_t)P.t2_layers.size() && c >= P.t2_layers[layer_of_t2].second) ++layer_of_t2;
                owners.clear();
                for (int32_t q = 0; q < tile.n_local; ++q) owners.push_back(owner_new(TT.gather[tile.gather_begin + q]));
                std::sort(owners.begin(), owners.end());
                owners.erase(std::unique(owners.begin(), owners.end()), owners.end());
                mine = std::binary_search(owners.begin(), owners.end(), rank);
                for (int32_t q = 0; q < tile.n_local; ++q) {
                    const int32_t nw = TT.gather[tile.gather_begin + q];
                    const int ow = owner_new(nw);
                    for (int cons : owners) {
                        if (cons == ow || (cons != rank && ow != rank)) continue;
                        needs.push_back({2 + (int32_t)P.gcolours.size() + layer_of_t2, cons, nw});
                    }
                }
            } else {
                owners.clear();
                for (int r = 0; r < tile.run_count; ++r) owners.push_back(owner_new(TT.runs[tile.run_begin + r].start));
                std::sort(owners.begin(), owners.end());
                owners.erase(std::unique(owners.begin(), owners.end()), owners.end());
                mine = std::binary_search(owners.begin(), owners.end(), rank);
                for (int r = 0; r < tile.run_count; ++r) {
                    const Run &rn = TT.runs[tile.run_begin + r];
                    int ow = owner_new(rn.start);
                    for (int cons : owners) {
                        if (cons == ow || (cons != rank && ow != rank)) continue;
                        for (int32_t q = 0; q < rn.len; ++q) needs.push_back({1, cons, rn.start + q});
                    }
                }
            }
            if (!mine) continue;
            L.T[tl].tile_ids.push_back(c);
            for (int p = 0; p < 2; ++p)
                for (int64_t k = tile.order_begin[p]; k < tile.order_end[p]; ++k) L.order_mask[p][k] = 1;
        }
    }
    // global colours: executed by every rank that owns one of the constraint's particles
    std::vector<std::vector<int32_t>> g_exec(P.gcolours.size());
    for (size_t gc = 0; gc < P.gcolours.size(); ++gc) {
        const GColour &g = P.gcolours[gc];
        const int nv = kVerts[g.type];
        for (int32_t id : g.ids) {
            const int32_t *v = C.idx(g.type, id);
            bool mine = false, multi = false;
            for (int a = 0; a < nv; ++a) { int ow = P.owner_of_old[v[a]]; mine |= ow == rank; multi |= ow != P.owner_of_old[v[0]]; }
            if (mine) g_exec[gc].push_back(id);
            if (multi)
                for (int a = 0; a < nv; ++a)
                    for (int b = 0; b < nv; ++b) {
                        int cons = P.owner_of_old[v[b]], ow = P.owner_of_old[v[a]];
                        if (cons == ow || (cons != rank && ow != rank)) continue;
                        needs.push_back({2 + (int32_t)gc, cons, P.new_of_old[v[a]]});
                    }
        }
    }
    for (int p = 0; p < 2; ++p)
        for (const Phase &ph : P.phases[p]) {
            if (ph.kind != 0) continue;
            // order entries of a global colour: mark the executed ones (ids are listed in the same order)
            const auto &ex = g_exec[ph.gcolour];
            size_t e = 0;
            for (int64_t k = ph.order_begin; k < ph.order_end; ++k)
                if (e < ex.size() && ex[e] == P.order_id[p][k]) { L.order_mask[p][k] = 1; ++e; }
        }
    std::sort(needs.begin(), needs.end(), [](const Need &a, const Need &b) {
        if (a.slot != b.slot) return a.slot < b.slot;
        if (a.consumer != b.consumer) return a.consumer < b.consumer;
        return a.nw < b.nw;
    });
    needs.erase(std::unique(needs.begin(), needs.end(), [](const Need &a, const Need &b) {
        return a.slot == b.slot && a.consumer == b.consumer && a.nw == b.nw; }), needs.end());
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
    for (const Need &nd : needs) {
        HaloSlot &H = L.halo[nd.slot];
        int ow = owner_new(nd.nw);
        if (nd.consumer == rank) H.recv_idx[ow].push_back(local_of_new[nd.nw]);
        else if (ow == rank) H.send_idx[nd.consumer].push_back(local_of_new[nd.nw]);
    }
    for (int tl = 0; tl < 3; ++tl) {
        LocalTiling &LT = L.T[tl];
        LT.run_begin.push_back(0);
        LT.gather_begin.push_back(0);
        for (int32_t c : LT.tile_ids) {
            const Tile &tile = P.T[tl].tiles[c];
            if (tl == 2) {      // sparse tile: explicit particle list instead of runs
                for (int32_t q = 0; q < tile.n_local; ++q) {
                    const int32_t li = local_of_new[P.T[2].gather[tile.gather_begin + q]];
                    if (li < 0) throw std::runtime_error("internal: T2 tile particle not resident");
                    LT.gather.push_back(li);
                }
                LT.gather_begin.push_back((int32_t)LT.gather.size());
                LT.run_begin.push_back((int32_t)LT.runs.size());
                continue;
            }
            for (int r = 0; r < tile.run_count; ++r) {
                const Run &rn = P.T[tl].runs[tile.run_begin + r];
                int32_t ls = local_of_new[rn.start];
                if (ls < 0) throw std::runtime_error("internal: tile run not resident");
                if (local_of_new[rn.start + rn.len - 1] != ls + rn.len - 1) throw std::runtime_error("internal: run split");
                LT.runs.push_back({ls, rn.len});
            }
            LT.run_begin.push_back((int32_t)LT.runs.size());
        }
    }
    // ---- pair hashes: what this rank and each peer must agree on (see LocalPlan::pair_hash) ---------------------
    {
        auto gid = [&](int32_t old) { return (uint64_t)(uint32_t)(in.global_id ? in.global_id[old] : old); };
        auto mix = [](uint64_t &h, uint64_t v) { h = (h ^ v) * 1099511628211ull; h ^= h >> 29; };
        const uint64_t kSeed = 1469598103934665603ull;
        std::vector<uint64_t> h_send((size_t)world, kSeed), h_recv((size_t)world, kSeed), h_tiles((size_t)world, kSeed);
        for (size_t slot = 0; slot < L.halo.size(); ++slot)
            for (int pr = 0; pr < world; ++pr) {
                const auto &sv = L.halo[slot].send_idx[(size_t)pr], &rv = L.halo[slot].recv_idx[(size_t)pr];
                if (sv.empty() && rv.empty()) continue;
                mix(h_send[(size_t)pr], 0x5e4d0000ull + slot); mix(h_recv[(size_t)pr], 0x5e4d0000ull + slot);
                for (int32_t li : sv) mix(h_send[(size_t)pr], gid(L.local_to_old[(size_t)li]));
                for (int32_t li : rv) mix(h_recv[(size_t)pr], gid(L.local_to_old[(size_t)li]));
            }
        // programs of the tiles this rank shares with a peer (T1 / T2 tiles that span ranks), in tile order
        for (int tl = 1; tl < 3; ++tl)
            for (int32_t c : L.T[tl].tile_ids) {
                const Tile &tile = P.T[tl].tiles[(size_t)c];
                if (tile.owner >= 0) continue;
                owners.clear();
                if (tl == 2) for (int32_t q = 0; q < tile.n_local; ++q) owners.push_back(owner_new(P.T[2].gather[(size_t)tile.gather_begin + q]));
                else for (int r = 0; r < tile.run_count; ++r) owners.push_back(owner_new(P.T[tl].runs[(size_t)tile.run_begin + r].start));
                std::sort(owners.begin(), owners.end());
                owners.erase(std::unique(owners.begin(), owners.end()), owners.end());
                uint64_t ht = kSeed;
                for (int64_t k = tile.order_begin[0]; k < tile.order_end[0]; ++k) {
                    const int t = P.order_type[0][(size_t)k];
                    const int32_t *v = C.idx(t, P.order_id[0][(size_t)k]);
                    mix(ht, (uint64_t)t);
                    for (int a = 0; a < kVerts[t]; ++a) mix(ht, gid(v[a]));
                }
                for (int ow : owners) if (ow != rank) mix(h_tiles[(size_t)ow], ht);
            }
        L.pair_hash.assign((size_t)world, 0);
        for (int pr = 0; pr < world; ++pr) {
            if (pr == rank) continue;
            uint64_t h = kSeed;
            // the lower rank's send lists first: rank a's (send, recv) must be rank b's (recv, send)
            mix(h, rank < pr ? h_send[(size_t)pr] : h_recv[(size_t)pr]);
            mix(h, rank < pr ? h_recv[(size_t)pr] : h_send[(size_t)pr]);
            mix(h, h_tiles[(size_t)pr]);
            L.pair_hash[(size_t)pr] = h;
        }
    }
    timer.lap("extract_local: tiles + halo");
    L.gcolours.resize(P.gcolours.size());
    for (size_t gc = 0; gc < P.gcolours.size(); ++gc) {
        const GColour &g = P.gcolours[gc];
        LocalGColour &LG = L.gcolours[gc];
        LG.type = g.type;
        const int nv = kVerts[g.type];
        for (int32_t id : g_exec[gc]) {
            const int32_t *v = C.idx(g.type, id);
            for (int a = 0; a < nv; ++a) {
                int32_t li = local_of_new[P.new_of_old[v[a]]];
                if (li < 0) throw std::runtime_error("internal: constraint particle not resident");
                LG.idx.push_back(li);
            }
            LG.id.push_back(id);
        }
    }
}

}  // namespace sbp
