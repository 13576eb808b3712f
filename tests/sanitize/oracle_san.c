/* AddressSanitizer / UBSan driver for the CPU oracle (test infrastructure exercising test infrastructure): a small
 * lattice with a few tets and hinges, sequential and task-parallel ticks, natural order and a permuted order. */
#include "../../oracle/oracle.c"

#include <stdio.h>

int main(void) {
    enum { N = 6, NP = N * N * N };
    static float x[3 * NP], v[3 * NP], w[NP], xprev[3 * NP];
    static int32_t ij[2 * 3 * NP]; static float rest[3 * NP];
    int m = 0;
    for (int z = 0; z < N; ++z) for (int y = 0; y < N; ++y) for (int xx = 0; xx < N; ++xx) {
        int p = (z * N + y) * N + xx;
        x[3 * p] = xx + 0.01f * (float)((p * 37) % 11); x[3 * p + 1] = (float)y; x[3 * p + 2] = z - 0.02f * (float)((p * 13) % 7);
        w[p] = (y == N - 1) ? 0.0f : 1.0f;
        if (xx + 1 < N) { ij[2 * m] = p; ij[2 * m + 1] = p + 1; rest[m++] = 1.0f; }
        if (y + 1 < N) { ij[2 * m] = p; ij[2 * m + 1] = p + N; rest[m++] = 1.0f; }
        if (z + 1 < N) { ij[2 * m] = p; ij[2 * m + 1] = p + N * N; rest[m++] = 1.0f; }
    }
    int32_t tets[4 * 3] = {0, 1, N, N * N, 1, 2, N + 1, N * N + 1, 7, 8, 7 + N, 7 + N * N};
    float r6[3] = {1.0f, 1.0f, 1.0f};
    int32_t hinges[4 * 2] = {0, 1, N, N * N, 2, 3, N + 2, N * N + 2};
    float hrest[4] = {0.0f, 1.0f, 0.0f, 1.0f};
    orc_constraints c = {ij, rest, m, tets, r6, 3, hinges, hrest, 2};
    orc_params p = {{0.0f, -9.81f, 0.0f}, 0.1f, {1e-7f, 1e-7f, 1e-5f}, {0.0f, 1.0f, 0.0f, -1.0f}, 1};
    const int64_t total = (int64_t)m + 3 + 2;
    /* natural order, sequential */
    orc_schedule nat[2] = {{NULL, NULL, NULL, 0, NULL}, {NULL, NULL, NULL, 0, NULL}};
    orc_step(x, v, w, xprev, NP, &c, nat, &p, 0.02f, 5);
    /* reversed order; one task per constraint, one phase per task (trivially independent) */
    uint8_t *ot = malloc((size_t)total); int32_t *oi = malloc((size_t)total * 4);
    int64_t *off = malloc((size_t)(total + 1) * 8);
    for (int64_t k = 0; k < total; ++k) {
        int64_t q = total - 1 - k;
        if (q < m) { ot[k] = 0; oi[k] = (int32_t)q; } else if (q < m + 3) { ot[k] = 1; oi[k] = (int32_t)(q - m); } else { ot[k] = 2; oi[k] = (int32_t)(q - m - 3); }
        off[k] = k;
    }
    off[total] = total;
    orc_schedule rev[2] = {{ot, oi, off, (int32_t)total, off}, {ot, oi, off, (int32_t)total, off}};
    orc_step(x, v, w, xprev, NP, &c, rev, &p, 0.02f, 3);
    orc_step_tasks(x, v, w, xprev, NP, &c, rev, &p, 0.02f, 4);
    for (int k = 0; k < 3 * NP; ++k) if (!(x[k] == x[k]) || !(v[k] == v[k])) { puts("NaN"); return 1; }
    free(ot); free(oi); free(off);
    puts("SANITIZE OK");
    return 0;
}
