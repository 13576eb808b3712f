/* Plain-C consumer of include/softbody.h (SURVEY.md §4 iv): proves the header compiles as C, the library links
 * from C, and the host-side entry points behave with plain pointers. With a GPU (argv[1] == "gpu") it also runs
 * the hot path once. Exit code 0 = all checks passed; prints one line per check. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "softbody.h"

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
        free(out);
    } else {
        CHECK(rc == SB_ERR_NO_DEVICE || rc == SB_OK, "sb_create without a GPU fails with SB_ERR_NO_DEVICE");
        if (rc == SB_OK) sb_destroy(s);
    }
    free(pos); free(w); free(ij); free(rest);
    printf("ALL OK\n");
    return 0;
}
