// AddressSanitizer / UBSan driver for the host planner (softbodyunity_amd/csrc/plan.cpp): builds plans for a lattice, an
// irregular cloud with 4-vertex constraints and a few degenerate inputs, for several world sizes, and checks the basic
// partition invariants. CPU only (GPU sanitizers are not available on the pool); built and run by tests/test_sanitizers.py.
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

void check(const Mesh &m, int world, int tile) {
    sbp::Input in{m.rest.data(), (int32_t)(m.rest.size() / 3), m.dist.data(), (int64_t)m.dist.size() / 2,
                  m.vol.data(), (int64_t)m.vol.size() / 4, m.bend.data(), (int64_t)m.bend.size() / 4};
    for (int rank = 0; rank < world; rank += (world > 4 ? 3 : 1)) {
        sbp::Opts o; o.rank = rank; o.world = world; o.tile_particles = tile;
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
