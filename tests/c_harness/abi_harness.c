/* Plain-C consumer of include/softbody.h + softbody_group.h / softbody_plan.h / softbody_debug.h (SURVEY.md §4 iv): proves the header compiles as C, the library links
 * from C, and the host-side entry points behave with plain pointers. With a GPU (argv[1] == "gpu") it also runs
 * the hot path once. Exit code 0 = all checks passed; prints one line per check. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "softbody.h"
#include "softbody_debug.h"
#include "softbody_group.h"
#include "softbody_plan.h"

#define CHECK(cond, what)                                        \
    do {                                                         \
        if (!(cond)) { printf("FAIL %s (last error: %s)\n", what, sb_last_error()); return 1; } \
        printf("ok   %s\n", what);                               \
    } while (0)

int main(int argc, char **argv) {
    const int n = 6, N = n * n * n;
    float *pos = malloc(sizeof(float) * 3 * N), *w = malloc(sizeof(float) * N);
    int32_t *ij = malloc(sizeof(int32_t) * 2 * 3 * n * n * (n - 1));
    float *rest = malloc(sizeof(float) * 3 * n * n * (n - 1));
    int m = 0;
    for (int z = 0; z < n; ++z) for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
        int p = (z * n + y) * n + x;
        pos[3 * p] = (float)x; pos[3 * p + 1] = (float)y; pos[3 * p + 2] = (float)z; w[p] = 1.0f;
    }
    for (int axis = 0; axis < 3; ++axis)
        for (int z = 0; z < n; ++z) for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
            int c[3] = {x, y, z};
            if (c[axis] + 1 >= n) continue;
            int p = (z * n + y) * n + x, off = axis == 0 ? 1 : (axis == 1 ? n : n * n);
            ij[2 * m] = p; ij[2 * m + 1] = p + off; rest[m] = 1.0f; ++m;
        }
    CHECK(sb_abi_version() == SB_ABI_VERSION, "sb_abi_version matches the header");
    sb_desc d; sb_desc_default(&d);
    CHECK(d.world == 1 && d.tile_particles == 0 && d.use_graph == 1, "sb_desc_default");
    /* planner: pure host code */
    sb_plan_opts o; memset(&o, 0, sizeof o); o.world = 1; o.tile_particles = 64;
    sb_plan *plan = NULL;
    CHECK(sb_plan_build(pos, N, ij, m, NULL, 0, NULL, 0, &o, &plan) == SB_OK && plan, "sb_plan_build");
    CHECK(sb_plan_order_count(plan) == m, "order covers every constraint");
    for (int parity = 0; parity < 2; ++parity) {
        uint8_t *t = malloc(m); int32_t *id = malloc(sizeof(int32_t) * m); char *seen = calloc(m, 1);
        CHECK(sb_plan_get_order(plan, parity, t, id) == SB_OK, "sb_plan_get_order");
        int ok = 1;
        for (int k = 0; k < m; ++k) { if (t[k] != 0 || id[k] < 0 || id[k] >= m || seen[id[k]]) ok = 0; else seen[id[k]] = 1; }
        CHECK(ok, "order is a permutation");
        free(t); free(id); free(seen);
    }
    CHECK(sb_plan_get_order(plan, 2, NULL, NULL) == SB_ERR_INVALID_ARG, "bad parity rejected");
    CHECK(sb_plan_destroy(plan) == SB_OK, "sb_plan_destroy");
    ij[1] = N + 5;
    CHECK(sb_plan_build(pos, N, ij, m, NULL, 0, NULL, 0, &o, &plan) == SB_ERR_INVALID_ARG, "out-of-range index rejected");
    CHECK(strstr(sb_last_error(), "out of range") != NULL, "sb_last_error explains");
    ij[1] = 1;
    CHECK(sb_create(NULL, NULL) == SB_ERR_INVALID_ARG, "null arguments rejected");
    sb_solver *s = NULL;
    int rc = sb_create(&d, &s);
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        CHECK(rc == SB_OK && s, "sb_create on the GPU");
        CHECK(sb_set_particles(s, pos, NULL, w, N) == SB_OK, "sb_set_particles");
        CHECK(sb_set_distance_constraints(s, ij, rest, m, 0.0f) == SB_OK, "sb_set_distance_constraints");
        CHECK(sb_finalize(s) == SB_OK, "sb_finalize");
        d.gravity[1] = 0.0f;
        CHECK(sb_step(s, 0.02f, 10) == SB_OK, "sb_step");
        float *out = malloc(sizeof(float) * 3 * N);
        CHECK(sb_get_positions(s, out, N) == SB_OK, "sb_get_positions");
        int finite = 1; for (int k = 0; k < 3 * N; ++k) if (!isfinite(out[k])) finite = 0;
        CHECK(finite && out[1] < pos[1], "positions are finite and the cube fell");
        /* the table validator and kinematic particles through the plain C ABI */
        sb_validate_report rep;
        CHECK(sb_debug_validate(s, 0, &rep) == SB_OK && rep.constraints_checked == m && rep.first_stage == -1, "sb_debug_validate: every constraint seen, clean");
        { int clean = 1; for (int k = 0; k < 6; ++k) if (rep.errors[k]) clean = 0; CHECK(clean, "no validator errors"); }
        int32_t free_id = 0; float target[3] = {0.0f, 9.0f, 0.0f};
        CHECK(sb_set_kinematic_positions(s, &free_id, target, 1) == SB_ERR_INVALID_ARG, "a free particle cannot be moved kinematically");
        CHECK(sb_set_kinematic_positions(s, NULL, NULL, 0) == SB_OK, "an empty kinematic list is accepted");
        CHECK(sb_destroy(s) == SB_OK, "sb_destroy");
        /* one process driving every rank (softbody_group.h): two ranks on this box's one device, the calling thread walking the tick
         * (SB_GROUP_WALK: no plugin thread), mailboxes connected by pointer. The gathered result must equal the single solver's above.
         * (two ranks on ONE device: the waiting kernel of one rank must not sit in front of the other's in a shared hardware queue:
         * the test starts this harness with GPU_MAX_HW_QUEUES set) */
        {
            sb_desc gd; sb_desc_default(&gd);
            gd.halo_transport = SB_TRANSPORT_PEER; gd.tile_particles = 64;
            int32_t devs[2] = {0, 0};
            sb_group *g = NULL;
            CHECK(sb_group_create(&gd, devs, 2, SB_GROUP_WALK, &g) == SB_OK && g, "sb_group_create (2 ranks, walk mode)");
            CHECK(sb_group_rank_count(g) == 2, "sb_group_rank_count");
            CHECK(sb_group_set_particles(g, pos, NULL, w, N) == SB_OK, "sb_group_set_particles");
            CHECK(sb_group_set_distance_constraints(g, ij, rest, m, 0.0f) == SB_OK, "sb_group_set_distance_constraints");
            CHECK(sb_group_step(g, 0.02f, 10) == SB_ERR_STATE, "sb_group_step before sb_group_finalize is refused");
            CHECK(sb_group_finalize(g) == SB_OK, "sb_group_finalize");
            CHECK(sb_group_step(g, 0.02f, 10) == SB_OK, "sb_group_step");
            float *gout = malloc(sizeof(float) * 3 * N);
            CHECK(sb_group_get_positions(g, gout, N) == SB_OK, "sb_group_get_positions (gathered, caller numbering)");
            /* the single solver above planned with the automatic tile size, this group with 64: same physics, another Gauss-Seidel order --
             * so compare against a single solver with the same tile size */
            sb_solver *one = NULL; sb_desc od; sb_desc_default(&od); od.tile_particles = 64;
            CHECK(sb_create(&od, &one) == SB_OK && sb_set_particles(one, pos, NULL, w, N) == SB_OK &&
                  sb_set_distance_constraints(one, ij, rest, m, 0.0f) == SB_OK && sb_finalize(one) == SB_OK && sb_step(one, 0.02f, 10) == SB_OK &&
                  sb_get_positions(one, out, N) == SB_OK, "the same mesh on one solver");
            CHECK(memcmp(gout, out, sizeof(float) * 3 * N) == 0, "two ranks behind the group == one solver, bit for bit");
            sb_solver *r1 = NULL; sb_stats st;
            CHECK(sb_group_get_rank(g, 1, &r1) == SB_OK && sb_get_stats(r1, &st) == SB_OK && st.n_particles_owned > 0 && st.n_particles_owned < N &&
                  st.n_particles_local > st.n_particles_owned, "rank 1 owns a part of the mesh and holds ghosts");
            CHECK(sb_destroy(one) == SB_OK && sb_group_destroy(g) == SB_OK, "sb_group_destroy");
            free(gout);
        }
        free(out);
    } else {
        CHECK(rc == SB_ERR_NO_DEVICE || rc == SB_OK, "sb_create without a GPU fails with SB_ERR_NO_DEVICE");
        if (rc == SB_OK) sb_destroy(s);
    }
    free(pos); free(w); free(ij); free(rest);
    printf("ALL OK\n");
    return 0;
}
