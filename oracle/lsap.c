/* oracle/lsap.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the rectangular linear-sum-assignment solver behind
 * scipy.optimize.linear_sum_assignment, which the reference calls once per image at
 * training/hungarian_matcher.py:79.  scipy is a third-party dependency that is not vendored under
 * /root/reference (not even listed in requirements.txt); the container has scipy 1.15.3, whose
 * `_lsap` extension implements the modified Jonker-Volgenant shortest-augmenting-path algorithm
 * of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016,
 * on float64 costs.  Restated here from that published algorithm plus scipy's documented
 * behaviour:
 *   - a tall matrix (rows > cols) is solved on its transpose and the pairs are returned sorted by
 *     the original row index;
 *   - the "remaining columns" list is filled in reverse so a constant matrix gives the identity;
 *   - among equal shortest-path costs an unassigned column (a new sink) is preferred;
 *   - NaN or -inf entries are invalid (scipy raises ValueError), +inf is accepted, and a row whose
 *     every reachable entry is +inf makes the problem infeasible.
 * Pinned: tests/test_oracle_cpu.py checks this file bit-for-bit against scipy on the committed
 * vectors in tests/golden/lsap_cases.npz and on live random cases.
 *
 * The HIP kernel (self-driving-model_amd/csrc/lsap.hip) follows the same scan order, so its
 * indices are bit-exact with this file and with scipy, ties included.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LSAP_OK 0
#define LSAP_INFEASIBLE -1
#define LSAP_INVALID -2
#define LSAP_NOMEM -3

/* Solve for one nr x nc row-major float64 cost matrix.  Writes min(nr,nc) pairs to (row_idx,
 * col_idx), row_idx ascending.  Returns LSAP_OK or a negative code. */
int oracle_lsap_f64(const double *cost_in, int64_t nr, int64_t nc, int64_t *row_idx, int64_t *col_idx)
{
    if (nr == 0 || nc == 0) return LSAP_OK;
    const int transpose = nc < nr;
    double *cost = NULL;
    const double *C = cost_in;
    if (transpose) {
        cost = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
        if (!cost) return LSAP_NOMEM;
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) cost[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
        C = cost;
    }
    for (int64_t i = 0; i < nr * nc; ++i)
        if (C[i] != C[i] || C[i] == -INFINITY) { free(cost); return LSAP_INVALID; }

    double *u = (double *)calloc((size_t)nr, sizeof(double));
    double *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);      /* shortest path costs */
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    unsigned char *SR = (unsigned char *)malloc((size_t)nr);
    unsigned char *SC = (unsigned char *)malloc((size_t)nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int rc = LSAP_OK;
    if (!u || !v || !spc || !path || !col4row || !row4col || !SR || !SC || !remaining) { rc = LSAP_NOMEM; goto done; }
    for (int64_t j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int64_t i = 0; i < nr; ++i) col4row[i] = -1;

    for (int64_t cur = 0; cur < nr; ++cur) {
        /* shortest augmenting path from row `cur` */
        double min_val = 0.0;
        int64_t n_rem = nc;
        for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        memset(SR, 0, (size_t)nr);
        memset(SC, 0, (size_t)nc);
        for (int64_t j = 0; j < nc; ++j) spc[j] = INFINITY;
        int64_t sink = -1, i = cur;
        while (sink == -1) {
            int64_t index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int64_t it = 0; it < n_rem; ++it) {
                const int64_t j = remaining[it];
                const double r = min_val + C[i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == INFINITY) { rc = LSAP_INFEASIBLE; goto done; }
            const int64_t j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--n_rem];
        }
        /* dual update */
        u[cur] += min_val;
        for (int64_t r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
        for (int64_t j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        /* augment */
        int64_t j = sink;
        for (;;) {
            const int64_t r = path[j];
            row4col[j] = r;
            const int64_t t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
    if (transpose) {
        /* pairs (col4row[k], k) sorted by col4row[k]; row4col is its inverse on assigned columns */
        int64_t n = 0;
        for (int64_t j = 0; j < nc; ++j)
            if (row4col[j] != -1) { row_idx[n] = j; col_idx[n] = row4col[j]; ++n; }
    } else {
        for (int64_t r = 0; r < nr; ++r) { row_idx[r] = r; col_idx[r] = col4row[r]; }
    }
done:
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(SR); free(SC); free(remaining);
    return rc;
}

/* float32 entry: the reference hands scipy an fp32 matrix (hungarian_matcher.py:73-79); scipy
 * casts it to float64 before solving.  Same here. */
int oracle_lsap_f32(const float *cost, int64_t nr, int64_t nc, int64_t *row_idx, int64_t *col_idx)
{
    if (nr == 0 || nc == 0) return LSAP_OK;
    double *c = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
    if (!c) return LSAP_NOMEM;
    for (int64_t i = 0; i < nr * nc; ++i) c[i] = (double)cost[i];
    int rc = oracle_lsap_f64(c, nr, nc, row_idx, col_idx);
    free(c);
    return rc;
}
