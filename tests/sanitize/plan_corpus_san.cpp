// AddressSanitizer / UBSan driver for the host planner over a CORPUS of meshes (tests/test_sanitizers.py writes it from tools/fuzz_plan.py's
// generator: particles on a point, a line, a plane, chains, complete graphs, hubs, isolated particles, no constraints, one or two particles,
// duplicate constraints, extreme scales, NaN / infinite positions; world up to 17, tile sizes down to 1, every partition -- and from
// tests/fuzz/fuzz_windows.py's: ranks' WINDOWS of lattice boxes with their domain and whole-mesh ids, sharded authoring). Each entry is
// planned for every rank; an exception is a refusal (counted), anything the sanitizers see is a failure. CPU only.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "plan.hpp"

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: plan_corpus_san <corpus>\n"); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("corpus"); return 2; }
    int planned = 0, refused = 0, entries = 0;
    for (;;) {
        int32_t h[12];
        if (std::fread(h, sizeof(int32_t), 12, f) != 12) break;
        const int32_t n = h[0], md = h[1], mv = h[2], mb = h[3], world = h[4], tile = h[5], partition = h[6];
        const int32_t window_rank = h[7];           // >= 0: the entry is ONE RANK'S WINDOW of a larger mesh (sharded authoring): domain + whole-mesh ids follow
        const int32_t dims[3] = {h[8], h[9], h[10]};
        double dom[9] = {0};                       // n_global, lo[3], hi[3], spacing, fill
        std::vector<int32_t> gid;
        if (window_rank >= 0) {
            gid.resize((size_t)n);
            if (std::fread(dom, sizeof(double), 9, f) != 9 || std::fread(gid.data(), sizeof(int32_t), gid.size(), f) != gid.size()) { std::fprintf(stderr, "truncated corpus\n"); return 2; }
        }
        std::vector<float> rest((size_t)3 * n);
        std::vector<int32_t> dist((size_t)2 * md), vol((size_t)4 * mv), bend((size_t)4 * mb);
        bool ok = std::fread(rest.data(), sizeof(float), rest.size(), f) == rest.size();
        ok = ok && std::fread(dist.data(), sizeof(int32_t), dist.size(), f) == dist.size();
        ok = ok && std::fread(vol.data(), sizeof(int32_t), vol.size(), f) == vol.size();
        ok = ok && std::fread(bend.data(), sizeof(int32_t), bend.size(), f) == bend.size();
        if (!ok) { std::fprintf(stderr, "truncated corpus\n"); return 2; }
        ++entries;
        sbp::Input in{rest.data(), n, dist.data(), md, vol.data(), mv, bend.data(), mb};
        if (window_rank >= 0) in.global_id = gid.data();
        for (int rank = (window_rank >= 0 ? window_rank : 0); rank < (window_rank >= 0 ? window_rank + 1 : world); ++rank) {
            sbp::Opts o; o.rank = rank; o.world = world; o.tile_particles = tile; o.partition = partition;
            for (int a = 0; a < 3; ++a) o.dims[a] = dims[a];
            if (window_rank >= 0) {
                o.domain.set = true; o.domain.n_global = (int64_t)dom[0]; o.domain.ell = dom[7]; o.domain.fill = dom[8];
                for (int a = 0; a < 3; ++a) { o.domain.lo[a] = dom[1 + a]; o.domain.hi[a] = dom[4 + a]; }
            }
            // (the ABI resolves the automatic tile size before it calls the planner: 0 -> 512, or 256 with 4-vertex constraints)
            if (o.tile_particles == 0) o.tile_particles = (mv + mb > 0) ? 256 : 512;
            try {
                sbp::Plan P; sbp::LocalPlan L;
                sbp::build_plan(in, o, P);
                sbp::extract_local(P, in, rank, L);
                if ((int64_t)P.order_id[0].size() != (int64_t)md + mv + mb) { std::fprintf(stderr, "entry %d: order does not cover every constraint\n", entries); return 1; }
                ++planned;
            } catch (const std::exception &) { ++refused; }
        }
    }
    std::fclose(f);
    std::printf("SANITIZE OK entries %d plans %d refused %d\n", entries, planned, refused);
    return 0;
}
