// AddressSanitizer / UBSan driver for the host planner (softbodyunity_amd/csrc/plan.cpp): builds plans for a lattice, an
// irregular cloud with 4-vertex constraints and a few degenerate inputs, for several world sizes, and checks the basic
// partition invariants. CPU only (GPU sanitizers are not available on the pool); built and run by tests/test_sanitizers.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <vector>

#include "plan.hpp"

namespace {

struct Mesh {
    std::vector<float> rest;
    std::vector<int32_t> dist, vol, bend;
};

Mesh lattice(int n) {
    Mesh m;
    for (int z = 0; z < n; ++z) for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) { m.rest.push_back((float)x); m.rest.push_back((float)y); m.rest.push_back((float)z); }
    auto id = [n](int x, int y, int z) { return (z * n + y) * n + x; };
    for (int z = 0; z < n; ++z) for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
        if (x + 1 < n) { m.dist.push_back(id(x, y, z)); m.dist.push_back(id(x + 1, y, z)); }
        if (y + 1 < n) { m.dist.push_back(id(x, y, z)); m.dist.push_back(id(x, y + 1, z)); }
        if (z + 1 < n) { m.dist.push_back(id(x, y, z)); m.dist.push_back(id(x, y, z + 1)); }
    }
    return m;
}

Mesh cloud(int n, unsigned seed) {     // jittered points, each joined to a few near neighbours by index proximity
    Mesh m;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    const int side = (int)std::ceil(std::cbrt((double)n));
    for (int p = 0; p < n; ++p) {
        int x = p % side, y = (p / side) % side, z = p / (side * side);
        m.rest.push_back(x + 0.6f * u(rng)); m.rest.push_back(y + 0.6f * u(rng)); m.rest.push_back(z + 0.6f * u(rng));
    }
    auto ok = [n](int q) { return q >= 0 && q < n; };
    for (int p = 0; p < n; ++p) {
        const int nb[5] = {p + 1, p + side, p + side * side, p + side + 1, p - side + 1};
        for (int q : nb) if (ok(q) && q != p) { m.dist.push_back(p); m.dist.push_back(q); }
        if (ok(p + 1) && ok(p + side) && ok(p + side * side)) {
            m.vol.push_back(p); m.vol.push_back(p + 1); m.vol.push_back(p + side); m.vol.push_back(p + side * side);
            if (p % 3 == 0) { m.bend.push_back(p); m.bend.push_back(p + 1); m.bend.push_back(p + side); m.bend.push_back(p + side * side); }
        }
    }
    return m;
}

void check(const Mesh &m, int world, int tile, int partition = 0) {
    sbp::Input in{m.rest.data(), (int32_t)(m.rest.size() / 3), m.dist.data(), (int64_t)m.dist.size() / 2,
                  m.vol.data(), (int64_t)m.vol.size() / 4, m.bend.data(), (int64_t)m.bend.size() / 4};
    for (int rank = 0; rank < world; rank += (world > 4 ? 3 : 1)) {
        sbp::Opts o; o.rank = rank; o.world = world; o.tile_particles = tile; o.partition = partition;
        sbp::Plan P; sbp::LocalPlan L;
        sbp::build_plan(in, o, P);
        sbp::extract_local(P, in, rank, L);
        const int64_t total = in.m_d + in.m_v + in.m_b;
        for (int par = 0; par < 2; ++par)
            if ((int64_t)P.order_id[par].size() != total) throw std::runtime_error("order does not cover every constraint");
        std::vector<char> seen(in.n, 0);
        for (int32_t q : P.old_of_new) { if (q < 0 || q >= in.n || seen[q]) throw std::runtime_error("numbering is not a permutation"); seen[q] = 1; }
        if (L.n_owned < 0 || L.n_owned > (int64_t)L.local_to_old.size()) throw std::runtime_error("bad owned count");
        for (const auto &H : L.halo)
            if ((int)H.send_idx.size() != world || (int)H.recv_idx.size() != world) throw std::runtime_error("halo slot size");
    }
}

// An L-shaped part of the cloud (three quarters of the box empty in one corner region): fill < 0.8, so the fill-aware grid, the
// balanced extra lists and the tile merge of plan.cpp run; automatic partition => RCB.
Mesh l_shape(const Mesh &full) {
    const int32_t n = (int32_t)(full.rest.size() / 3);
    float hi[3] = {0, 0, 0};
    for (int32_t p = 0; p < n; ++p) for (int a = 0; a < 3; ++a) hi[a] = std::max(hi[a], full.rest[3 * p + a]);
    std::vector<int32_t> map((size_t)n, -1);
    Mesh m;
    for (int32_t p = 0; p < n; ++p) {
        const float *x = &full.rest[3 * (size_t)p];
        if (x[0] > 0.45f * hi[0] && x[1] > 0.45f * hi[1]) continue;         // cut a column out
        map[(size_t)p] = (int32_t)(m.rest.size() / 3);
        m.rest.insert(m.rest.end(), x, x + 3);
    }
    auto keep = [&](const std::vector<int32_t> &src, int nv, std::vector<int32_t> &dst) {
        for (size_t k = 0; k + nv <= src.size(); k += nv) {
            bool ok = true;
            for (int a = 0; a < nv; ++a) ok &= map[(size_t)src[k + a]] >= 0;
            if (ok) for (int a = 0; a < nv; ++a) dst.push_back(map[(size_t)src[k + a]]);
        }
    };
    keep(full.dist, 2, m.dist); keep(full.vol, 4, m.vol); keep(full.bend, 4, m.bend);
    return m;
}

// Sharded authoring: every rank plans the window rank_window() gives it; owned counts must add up to the whole mesh and the pair
// hashes must be symmetric.
void check_sharded(const Mesh &m, int world, int tile) {
    sbp::Input whole{m.rest.data(), (int32_t)(m.rest.size() / 3), m.dist.data(), (int64_t)m.dist.size() / 2, nullptr, 0, nullptr, 0};
    sbp::Domain dom;
    sbp::compute_domain(whole, dom);
    dom.set = true;
    std::vector<std::vector<uint64_t>> pair((size_t)world);
    int64_t owned_total = 0;
    for (int rank = 0; rank < world; ++rank) {
        sbp::Opts o; o.rank = rank; o.world = world; o.tile_particles = tile; o.domain = dom; o.partition = 1;
        int clo[3], chi[3]; double blo[3], bhi[3];
        sbp::rank_window(dom, o, clo, chi, blo, bhi);
        std::vector<int32_t> gid, map((size_t)whole.n, -1);
        Mesh w;
        for (int32_t p = 0; p < whole.n; ++p) {
            bool in = true;
            for (int a = 0; a < 3; ++a) in &= m.rest[3 * (size_t)p + a] >= blo[a] && m.rest[3 * (size_t)p + a] < bhi[a];
            if (!in) continue;
            map[(size_t)p] = (int32_t)gid.size(); gid.push_back(p);
            w.rest.insert(w.rest.end(), &m.rest[3 * (size_t)p], &m.rest[3 * (size_t)p] + 3);
        }
        for (size_t k = 0; k + 2 <= m.dist.size(); k += 2)
            if (map[(size_t)m.dist[k]] >= 0 && map[(size_t)m.dist[k + 1]] >= 0) { w.dist.push_back(map[(size_t)m.dist[k]]); w.dist.push_back(map[(size_t)m.dist[k + 1]]); }
        sbp::Input in{w.rest.data(), (int32_t)gid.size(), w.dist.data(), (int64_t)w.dist.size() / 2, nullptr, 0, nullptr, 0};
        in.global_id = gid.data();
        sbp::Plan P; sbp::LocalPlan L;
        sbp::build_plan(in, o, P);
        sbp::extract_local(P, in, rank, L);
        owned_total += L.n_owned;
        pair[(size_t)rank] = L.pair_hash;
    }
    if (owned_total != whole.n) throw std::runtime_error("sharded ranks do not own the whole mesh between them");
    for (int a = 0; a < world; ++a) for (int b = 0; b < world; ++b) if (a != b && pair[(size_t)a][(size_t)b] != pair[(size_t)b][(size_t)a]) throw std::runtime_error("pair hashes are not symmetric");
}

template <class F> void expect_throw(const char *what, F f) {
    try { f(); } catch (const std::exception &) { return; }
    std::fprintf(stderr, "expected an exception: %s\n", what);
    std::exit(2);
}

}  // namespace

int main() {
    const Mesh a = lattice(20), b = cloud(9000, 7), c = lattice(3);
    for (int tile : {512, 64, -1}) { check(a, 1, tile); check(b, 1, tile); }
    for (int world : {2, 3, 8}) { check(a, world, 64); check(b, world, 128); check(c, world, 512); }
    for (int world : {3, 8}) { check(a, world, 64, 2); check(b, world, 128, 2); }       // RCB forced
    {   // a mesh that fills its box unevenly: fill-aware grid, balanced lists, tile merge; automatic partition (RCB)
        const Mesh l = l_shape(b);
        check(l, 1, 128); check(l, 8, 128); check(l, 5, 64);
    }
    check_sharded(lattice(24), 8, 64); check_sharded(lattice(20), 3, 27);
    {   // a single particle, no constraints
        Mesh s; s.rest = {0.f, 0.f, 0.f};
        check(s, 1, 512); check(s, 2, 512);
    }
    Mesh bad = lattice(4);
    bad.dist[1] = 1000;
    expect_throw("index out of range", [&] { check(bad, 1, 512); });
    Mesh nan = lattice(4);
    nan.rest[5] = NAN;
    expect_throw("non-finite rest position", [&] { check(nan, 1, 512); });
    Mesh rep = lattice(4);
    rep.dist[1] = rep.dist[0];
    expect_throw("repeated particle", [&] { check(rep, 1, 512); });
    std::puts("SANITIZE OK");
    return 0;
}
