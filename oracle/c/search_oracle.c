/*
 * search_oracle.c -- plain-C restatement of the cosine top-k definition (TEST INFRASTRUCTURE, see oracle/__init__.py).
 *
 * The reference has no search code; this follows the definition in
 * oracle/search_oracle.py independently of numpy:
 *     score(q, b) = (float) ( sum_d (double)q_d * (double)b_d  /  max(sqrt(sum_d q_d^2), 1e-12) )
 *     result      = the k best rows by (score descending, row index ascending)
 * Inputs are float32 (the caller upcasts float16 banks, which is exact).  Rows are visited in ascending index
 * order and a candidate only displaces the current k-th entry when its score is strictly greater, so equal
 * scores keep the lower index.
 *
 * build:  make -C oracle/c      (gcc -O2 -fopenmp -shared; see Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

int isc_oracle_cosine_topk(const float* bank, int64_t n, int d, const float* queries, int q, int k,
                           int64_t index_base, float* out_scores, int64_t* out_indices) {
    if (!bank || !queries || !out_scores || !out_indices || n <= 0 || d <= 0 || q <= 0 || k <= 0 || k > n) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int qi = 0; qi < q; ++qi) {
        const float* qv = queries + (int64_t)qi * d;
        double nq = 0.0;
        for (int j = 0; j < d; ++j) nq += (double)qv[j] * (double)qv[j];
        double denom = sqrt(nq);
        if (denom < 1e-12) denom = 1e-12;
        float* bs = out_scores + (int64_t)qi * k; /* kept sorted, best first */
        int64_t* bi = out_indices + (int64_t)qi * k;
        int have = 0;
        for (int64_t r = 0; r < n; ++r) {
            const float* bv = bank + r * d;
            double acc = 0.0;
            for (int j = 0; j < d; ++j) acc += (double)qv[j] * (double)bv[j];
            const float s = (float)(acc / denom);
            if (have < k || s > bs[have - 1]) {
                int pos = have < k ? have : k - 1;
                while (pos > 0 && s > bs[pos - 1]) {
                    bs[pos] = bs[pos - 1];
                    bi[pos] = bi[pos - 1];
                    --pos;
                }
                bs[pos] = s;
                bi[pos] = r + index_base;
                if (have < k) ++have;
            }
        }
    }
    return 0;
}
