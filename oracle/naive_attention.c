/*
 * oracle/naive_attention.c -- CPU restatement of the reference's naive attention.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker.  The product
 * path (cuda_flashattention_amd/, libfa2_mi355x.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  oracle/Makefile builds oracle/_ref/libref_naive.so straight
 * from the reference's own sources (src/util/naive_attention.h, src/00_naive_attention/
 * main.cpp) where they lie under /root/reference; tests/test_oracle_vs_ref.py asserts
 * that every "verbatim-order" function below is BIT-IDENTICAL to it, and
 * tests/golden/ holds vectors generated from that reference build (gen_golden.py).
 *
 * Two families:
 *   1. verbatim-order fp32 restatements (same operation order as the reference, so
 *      results are bit-identical; O(N) scratch instead of the reference's N x N heaps):
 *        oracle_naive_attention        <- src/00_naive_attention/main.cpp:8-38
 *        oracle_naive_forward_pass     <- src/util/naive_attention.h:7-61
 *        oracle_naive_attention_backward <- src/util/naive_attention.h:84-161
 *   2. scalable forms used at BASELINE sizes ([B][H][N][d], optional causal mask,
 *      double accumulation, O(N^2 d) backward dS = P o (dP - D) instead of the
 *      reference's O(N^3) Jacobian loop at naive_attention.h:130-140), validated
 *      against family 1 at small N:
 *        oracle_attention_forward_f64, oracle_attention_backward_f64,
 *        oracle_attention_forward_rows_f64 (a strided subset of query rows).
 *   3. ring-step restatement (resumable online-softmax state) following
 *      src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:67-137 and
 *      src/util/attention_helper.h:40-132 at whole-shard granularity:
 *        oracle_ring_step.
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Family 1: verbatim-order fp32                                              */
/* ------------------------------------------------------------------------- */

/* src/00_naive_attention/main.cpp:8-38: score = (sum_k q*k) / sqrtf(d); max seeded with
 * -FLT_MAX; un-normalised weighted sum of V, divided by the exp-sum at the end. */
void oracle_naive_attention(const float* Q, const float* K, const float* V, float* O,
                            int N, int d)
{
    float* w = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    const float root = sqrtf((float)d);
    for (int i = 0; i < N; ++i) {
        const float* q = Q + (size_t)i * d;
        float* o = O + (size_t)i * d;
        float top = -FLT_MAX;
        for (int j = 0; j < N; ++j) {
            const float* kr = K + (size_t)j * d;
            float acc = 0.0f;
            for (int c = 0; c < d; ++c) acc += q[c] * kr[c];
            acc /= root;
            w[j] = acc;
            if (acc > top) top = acc;
        }
        float denom = 0.0f;
        for (int j = 0; j < N; ++j) {
            w[j] = expf(w[j] - top);
            denom += w[j];
        }
        for (int c = 0; c < d; ++c) o[c] = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float* vr = V + (size_t)j * d;
            for (int c = 0; c < d; ++c) o[c] += w[j] * vr[c];
        }
        for (int c = 0; c < d; ++c) o[c] /= denom;
    }
    free(w);
}

/* One softmax row in the order of src/util/naive_attention.h:15-46 (forward) and
 * :91-109 (backward recomputation): s = (sum_k q*k)*scale; max seeded with -1e9f
 * (through double fmax, as the reference calls ::fmax on floats); p = expf(s-max);
 * running sum in j order; then p /= sum.  Returns max, writes sum. */
static float softmax_row_ref_order(const float* q, const float* K, int N, int d,
                                   float scale, float* p, float* sum_out)
{
    for (int j = 0; j < N; ++j) {
        const float* kr = K + (size_t)j * d;
        float acc = 0.0f;
        for (int c = 0; c < d; ++c) acc += q[c] * kr[c];
        p[j] = acc * scale;
    }
    float top = -1e9f;
    for (int j = 0; j < N; ++j) top = (float)fmax((double)top, (double)p[j]);
    float total = 0.0f;
    for (int j = 0; j < N; ++j) {
        p[j] = expf(p[j] - top);
        total += p[j];
    }
    *sum_out = total;
    return top;
}

/* src/util/naive_attention.h:7-61.  scale == 0 selects 1/sqrtf(d) (:9).  L may be NULL
 * (:41-42).  P is normalised BEFORE the PV product (:43-45). */
void oracle_naive_forward_pass(const float* Q, const float* K, const float* V,
                               float* O, float* L, int N, int d, float scale)
{
    if (scale == 0) scale = 1.0f / sqrtf((float)d);
    float* p = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; ++i) {
        float total;
        const float top = softmax_row_ref_order(Q + (size_t)i * d, K, N, d, scale, p, &total);
        if (L) L[i] = top + logf(total);
        for (int j = 0; j < N; ++j) p[j] /= total;
        float* o = O + (size_t)i * d;
        for (int c = 0; c < d; ++c) {
            float acc = 0.0f;
            for (int j = 0; j < N; ++j) acc += p[j] * V[(size_t)j * d + c];
            o[c] = acc;
        }
    }
    free(p);
}

/* src/util/naive_attention.h:84-161.  Like the reference it ignores O and L and
 * recomputes the softmax; dS uses the explicit softmax Jacobian, O(N^3) (:130-140).
 * Accumulation orders match: dV[j][c] and dK[j][c] sum over query rows i ascending
 * from 0.0f; dQ[i][c] sums over keys j ascending; dQ,dK are scaled after the sum. */
void oracle_naive_attention_backward(const float* Q, const float* K, const float* V,
                                     const float* O, const float* L, const float* dO,
                                     float* dQ, float* dK, float* dV,
                                     int N, int d, float scale)
{
    (void)O; (void)L;
    const size_t n = (size_t)(N > 0 ? N : 1);
    float* p  = (float*)malloc(sizeof(float) * n);
    float* dp = (float*)malloc(sizeof(float) * n);
    float* ds = (float*)malloc(sizeof(float) * n);
    for (size_t t = 0; t < (size_t)N * d; ++t) { dV[t] = 0.0f; dK[t] = 0.0f; }
    for (int i = 0; i < N; ++i) {
        float total;
        (void)softmax_row_ref_order(Q + (size_t)i * d, K, N, d, scale, p, &total);
        for (int j = 0; j < N; ++j) p[j] /= total;
        const float* g = dO + (size_t)i * d;
        /* dV += P^T dO  (:112-119) */
        for (int j = 0; j < N; ++j)
            for (int c = 0; c < d; ++c) dV[(size_t)j * d + c] += p[j] * g[c];
        /* dP = dO V^T  (:121-128) */
        for (int j = 0; j < N; ++j) {
            const float* vr = V + (size_t)j * d;
            float acc = 0.0f;
            for (int c = 0; c < d; ++c) acc += g[c] * vr[c];
            dp[j] = acc;
        }
        /* dS = dP * softmax Jacobian (:130-140) */
        for (int j = 0; j < N; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < N; ++k) {
                const float jac = (j == k) ? p[j] * (1 - p[k]) : -p[j] * p[k];
                acc += dp[k] * jac;
            }
            ds[j] = acc;
        }
        /* dQ = dS K * scale (:142-149) */
        for (int c = 0; c < d; ++c) {
            float acc = 0.0f;
            for (int j = 0; j < N; ++j) acc += ds[j] * K[(size_t)j * d + c];
            dQ[(size_t)i * d + c] = acc * scale;
        }
        /* dK = dS^T Q * scale, accumulated in query order (:150-157) */
        const float* q = Q + (size_t)i * d;
        for (int j = 0; j < N; ++j)
            for (int c = 0; c < d; ++c) dK[(size_t)j * d + c] += ds[j] * q[c];
    }
    for (size_t t = 0; t < (size_t)N * d; ++t) dK[t] *= scale;
    free(p); free(dp); free(ds);
}

/* ------------------------------------------------------------------------- */
/* Family 2: scalable forms, [B][H][N][d] slabs, double accumulation          */
/* ------------------------------------------------------------------------- */

/* One query row against keys [0, nk) of one head slab; double accumulation.
 * p[] receives the normalised probabilities; returns natural-log LSE. */
static double softmax_row_f64(const float* q, const float* K, int nk, int d, double scale,
                              double* p)
{
    double top = -INFINITY;
    for (int j = 0; j < nk; ++j) {
        const float* kr = K + (size_t)j * d;
        double acc = 0.0;
        for (int c = 0; c < d; ++c) acc += (double)q[c] * (double)kr[c];
        p[j] = acc * scale;
        if (p[j] > top) top = p[j];
    }
    double total = 0.0;
    for (int j = 0; j < nk; ++j) { p[j] = exp(p[j] - top); total += p[j]; }
    for (int j = 0; j < nk; ++j) p[j] /= total;
    return top + log(total);
}

/* Rows row0, row0+stride, ... (< N) of every (b,h) slab in [bh0, bh1).  Outputs are
 * written at the rows' natural positions in O [BH][N][d] / L [BH][N]; untouched rows keep
 * their previous content.  causal != 0 masks keys j > i. */
void oracle_attention_forward_rows_f64(const float* Q, const float* K, const float* V,
                                       float* O, float* L, int BH, int N, int d,
                                       float scale, int causal,
                                       int bh0, int bh1, int row0, int stride)
{
    if (scale == 0) scale = 1.0f / sqrtf((float)d);
    if (bh0 < 0) bh0 = 0;
    if (bh1 > BH) bh1 = BH;
    if (stride < 1) stride = 1;
#pragma omp parallel
    {
        double* p = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
        double* o = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
        const int per = (N - row0 + stride - 1) / stride;
        const long total = (long)(bh1 - bh0) * (per > 0 ? per : 0);
#pragma omp for schedule(dynamic, 8)
        for (long t = 0; t < total; ++t) {
            const int bh = bh0 + (int)(t / per);
            const int i = row0 + (int)(t % per) * stride;
            const size_t slab = (size_t)bh * N * d;
            const int nk = causal ? i + 1 : N;
            const double lse = softmax_row_f64(Q + slab + (size_t)i * d, K + slab, nk, d,
                                               (double)scale, p);
            for (int c = 0; c < d; ++c) o[c] = 0.0;
            for (int j = 0; j < nk; ++j) {
                const float* vr = V + slab + (size_t)j * d;
                const double w = p[j];
                for (int c = 0; c < d; ++c) o[c] += w * (double)vr[c];
            }
            for (int c = 0; c < d; ++c) O[slab + (size_t)i * d + c] = (float)o[c];
            if (L) L[(size_t)bh * N + i] = (float)lse;
        }
        free(p); free(o);
    }
}

void oracle_attention_forward_f64(const float* Q, const float* K, const float* V,
                                  float* O, float* L, int BH, int N, int d,
                                  float scale, int causal)
{
    oracle_attention_forward_rows_f64(Q, K, V, O, L, BH, N, d, scale, causal, 0, BH, 0, 1);
}

/* O(N^2 d) backward: D_i = sum_c dO_ic O_ic (with O recomputed from P, so the function,
 * like the reference's, does not depend on its O/L arguments), dS = P o (dP - D),
 * dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO.  One thread per slab. */
void oracle_attention_backward_f64(const float* Q, const float* K, const float* V,
                                   const float* dO, float* dQ, float* dK, float* dV,
                                   int BH, int N, int d, float scale, int causal)
{
    if (scale == 0) scale = 1.0f / sqrtf((float)d);
#pragma omp parallel for schedule(dynamic, 1)
    for (int bh = 0; bh < BH; ++bh) {
        const size_t slab = (size_t)bh * N * d;
        const size_t nd = (size_t)N * d;
        double* p   = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
        double* o   = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
        double* dq  = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
        double* aK  = (double*)calloc(nd > 0 ? nd : 1, sizeof(double));
        double* aV  = (double*)calloc(nd > 0 ? nd : 1, sizeof(double));
        for (int i = 0; i < N; ++i) {
            const int nk = causal ? i + 1 : N;
            const float* q = Q + slab + (size_t)i * d;
            const float* g = dO + slab + (size_t)i * d;
            (void)softmax_row_f64(q, K + slab, nk, d, (double)scale, p);
            for (int c = 0; c < d; ++c) o[c] = 0.0;
            for (int j = 0; j < nk; ++j) {
                const float* vr = V + slab + (size_t)j * d;
                for (int c = 0; c < d; ++c) o[c] += p[j] * (double)vr[c];
            }
            double D = 0.0;
            for (int c = 0; c < d; ++c) D += (double)g[c] * o[c];
            for (int c = 0; c < d; ++c) dq[c] = 0.0;
            for (int j = 0; j < nk; ++j) {
                const float* vr = V + slab + (size_t)j * d;
                const float* kr = K + slab + (size_t)j * d;
                double dp = 0.0;
                for (int c = 0; c < d; ++c) dp += (double)g[c] * (double)vr[c];
                const double ds = p[j] * (dp - D);
                double* ak = aK + (size_t)j * d;
                double* av = aV + (size_t)j * d;
                for (int c = 0; c < d; ++c) {
                    dq[c] += ds * (double)kr[c];
                    ak[c] += ds * (double)q[c];
                    av[c] += p[j] * (double)g[c];
                }
            }
            for (int c = 0; c < d; ++c) dQ[slab + (size_t)i * d + c] = (float)(dq[c] * scale);
        }
        for (size_t t = 0; t < nd; ++t) {
            dK[slab + t] = (float)(aK[t] * scale);
            dV[slab + t] = (float)aV[t];
        }
        free(p); free(o); free(dq); free(aK); free(aV);
    }
}

/* The same O(N^2 d) backward for ONE head slab [N][d], with the query rows split over the OpenMP threads (private
 * double partials of dK / dV, summed slice by slice at the end -- no critical section), so that a whole head at
 * N = 8192 .. 65536 is seconds, not minutes: what the at-size GPU parity tests compare a head of the backward with.
 * Same arithmetic as oracle_attention_backward_f64 (tests/test_oracle_golden.py checks them against each other). */
void oracle_attention_backward_head_f64(const float* Q, const float* K, const float* V,
                                        const float* dO, float* dQ, float* dK, float* dV,
                                        int N, int d, float scale, int causal)
{
    if (scale == 0) scale = 1.0f / sqrtf((float)d);
    const size_t nd = (size_t)N * d;
    int nthreads = 1;
    double** parts = NULL;
#pragma omp parallel
    {
#pragma omp single
        {
#ifdef _OPENMP
            extern int omp_get_num_threads(void);
            nthreads = omp_get_num_threads();
#endif
            parts = (double**)calloc((size_t)2 * nthreads, sizeof(double*));
        }
        int tid = 0;
#ifdef _OPENMP
        extern int omp_get_thread_num(void);
        tid = omp_get_thread_num();
#endif
        double* p  = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
        double* o  = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
        double* dq = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
        double* aK = (double*)calloc(nd > 0 ? nd : 1, sizeof(double));
        double* aV = (double*)calloc(nd > 0 ? nd : 1, sizeof(double));
        parts[2 * tid] = aK; parts[2 * tid + 1] = aV;
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < N; ++i) {
            const int nk = causal ? i + 1 : N;
            const float* q = Q + (size_t)i * d;
            const float* g = dO + (size_t)i * d;
            (void)softmax_row_f64(q, K, nk, d, (double)scale, p);
            for (int c = 0; c < d; ++c) o[c] = 0.0;
            for (int j = 0; j < nk; ++j) {
                const float* vr = V + (size_t)j * d;
                for (int c = 0; c < d; ++c) o[c] += p[j] * (double)vr[c];
            }
            double D = 0.0;
            for (int c = 0; c < d; ++c) D += (double)g[c] * o[c];
            for (int c = 0; c < d; ++c) dq[c] = 0.0;
            for (int j = 0; j < nk; ++j) {
                const float* vr = V + (size_t)j * d;
                const float* kr = K + (size_t)j * d;
                double dp = 0.0;
                for (int c = 0; c < d; ++c) dp += (double)g[c] * (double)vr[c];
                const double ds = p[j] * (dp - D);
                double* ak = aK + (size_t)j * d;
                double* av = aV + (size_t)j * d;
                for (int c = 0; c < d; ++c) {
                    dq[c] += ds * (double)kr[c];
                    ak[c] += ds * (double)q[c];
                    av[c] += p[j] * (double)g[c];
                }
            }
            for (int c = 0; c < d; ++c) dQ[(size_t)i * d + c] = (float)(dq[c] * scale);
        }
        /* implicit barrier above: every partial is complete; each thread now sums a slice of the elements */
#pragma omp for schedule(static)
        for (long t = 0; t < (long)nd; ++t) {
            double sk = 0.0, sv = 0.0;
            for (int w = 0; w < nthreads; ++w)
                if (parts[2 * w]) { sk += parts[2 * w][t]; sv += parts[2 * w + 1][t]; }
            dK[t] = (float)(sk * scale);
            dV[t] = (float)sv;
        }
        free(p); free(o); free(dq); free(aK); free(aV);
    }
    free(parts);
}

/* ------------------------------------------------------------------------- */
/* Family 3: one ring step (resumable online softmax)                         */
/* ------------------------------------------------------------------------- */

/* State per query row: O (un-normalised accumulator), l (running sum, kept in L like the
 * reference does until the last step), m (running max).  One call folds one resident
 * K/V shard of nk keys into the state of nq query rows, as
 * ring_attention_forward_kernel does per launch (ring_attention_kernel.cu:67-137) with
 * the update rule of attention_helper.h:76-110:  m' = max(m, rowmax), l = e^{m-m'} l +
 * sum e^{s-m'}, O = e^{m-m'} O + sum e^{s-m'} v.  On last != 0 it finalises:
 * O /= l, L = m + log l (:112-124).  The whole shard is treated as one tile (tiling only
 * changes rounding, not the value).  fp32 throughout, like the reference. */
void oracle_ring_step(const float* Q, const float* K, const float* V,
                      float* O, float* L, float* M,
                      int nq, int nk, int d, float scale, int last)
{
    float* s = (float*)malloc(sizeof(float) * (size_t)(nk > 0 ? nk : 1));
    for (int i = 0; i < nq; ++i) {
        const float* q = Q + (size_t)i * d;
        float* o = O + (size_t)i * d;
        float top = -INFINITY;
        for (int j = 0; j < nk; ++j) {
            const float* kr = K + (size_t)j * d;
            float acc = 0.0f;
            for (int c = 0; c < d; ++c) acc += q[c] * kr[c];
            s[j] = acc * scale;
            top = fmaxf(top, s[j]);
        }
        const float m_new = fmaxf(M[i], top);
        const float carry = expf(M[i] - m_new);
        float add = 0.0f;
        for (int c = 0; c < d; ++c) o[c] *= carry;
        for (int j = 0; j < nk; ++j) {
            const float w = expf(s[j] - m_new);
            add += w;
            const float* vr = V + (size_t)j * d;
            for (int c = 0; c < d; ++c) o[c] += w * vr[c];
        }
        float l = carry * L[i] + add;
        M[i] = m_new;
        if (last) {
            for (int c = 0; c < d; ++c) o[c] /= l;
            L[i] = m_new + logf(l);
        } else {
            L[i] = l;
        }
    }
    free(s);
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline: the naive forward + backward, one head per thread             */
/* ------------------------------------------------------------------------- */

/* What bench.py times beside the GPU: the reference's triple loops in its own arithmetic type (fp32,
 * sequential sums; naive_attention.h:15-58 for the forward, the O(N^2 d) form of :84-161 for the backward), ONE HEAD
 * PER THREAD as the survey prescribes: thread t owns head t % BH and works on `nblk` blocks of RB = 16 consecutive
 * query rows spread evenly over that head; its dK / dV sums are its own (nothing is shared, merged or locked), so the
 * timing is the loops' and nothing else's.  Per block: pass 1 forms the rows' probabilities (kept transposed,
 * pT[j][RB], one cache line per key), O and D; pass 2 walks the keys once and, per key, the block's rows -- the dK / dV
 * row of a key is touched once per block, not once per query row.  14 N d flops per query row: the row's share of the
 * 4 N^2 d + 10 N^2 d the bench counts.  Returns the number of query rows processed (all threads together);
 * O_rows / dQ_rows [threads][nblk*RB][d] and dK / dV [threads][N][d] are caller scratch (checked by the CPU tests). */
#define ORACLE_RB 16
long oracle_fwdbwd_heads_f32(const float* Q, const float* K, const float* V, const float* dO,
                             float* O_rows, float* dQ_rows, float* dK, float* dV,
                             int BH, int N, int d, float scale, int nblk, int threads)
{
    if (scale == 0) scale = 1.0f / sqrtf((float)d);
    const size_t nd = (size_t)N * d;
    const int blocks_in_head = N / ORACLE_RB;
    if (nblk > blocks_in_head) nblk = blocks_in_head;
    if (nblk < 1 || threads < 1) return 0;
#pragma omp parallel for schedule(static, 1) num_threads(threads)
    for (int t = 0; t < threads; ++t) {
        const size_t slab = (size_t)(t % BH) * nd;
        const float *Qh = Q + slab, *Kh = K + slab, *Vh = V + slab, *Gh = dO + slab;
        float* aK = dK + (size_t)t * nd;
        float* aV = dV + (size_t)t * nd;
        memset(aK, 0, nd * sizeof(float)); memset(aV, 0, nd * sizeof(float));
        float* pT = (float*)malloc(sizeof(float) * (size_t)N * ORACLE_RB);
        float* s  = (float*)malloc(sizeof(float) * (size_t)N);
        float D[ORACLE_RB];
        for (int b = 0; b < nblk; ++b) {
            const int r0 = (int)((long)b * blocks_in_head / nblk) * ORACLE_RB;
            float* Ob = O_rows + ((size_t)t * nblk + b) * ORACLE_RB * d;
            float* Qb = dQ_rows + ((size_t)t * nblk + b) * ORACLE_RB * d;
            for (int r = 0; r < ORACLE_RB; ++r) {                       /* pass 1 */
                const float* q = Qh + (size_t)(r0 + r) * d;
                const float* g = Gh + (size_t)(r0 + r) * d;
                float total;
                (void)softmax_row_ref_order(q, Kh, N, d, scale, s, &total);
                float* o = Ob + (size_t)r * d;
                for (int c = 0; c < d; ++c) o[c] = 0.0f;
                for (int j = 0; j < N; ++j) {
                    const float pj = s[j] / total;
                    pT[(size_t)j * ORACLE_RB + r] = pj;
                    const float* vr = Vh + (size_t)j * d;
                    for (int c = 0; c < d; ++c) o[c] += pj * vr[c];
                }
                float acc = 0.0f;
                for (int c = 0; c < d; ++c) acc += g[c] * o[c];
                D[r] = acc;
                float* dq = Qb + (size_t)r * d;
                for (int c = 0; c < d; ++c) dq[c] = 0.0f;
            }
            for (int j = 0; j < N; ++j) {                               /* pass 2 */
                const float* kr = Kh + (size_t)j * d;
                const float* vr = Vh + (size_t)j * d;
                float* ak = aK + (size_t)j * d;
                float* av = aV + (size_t)j * d;
                for (int r = 0; r < ORACLE_RB; ++r) {
                    const float* q = Qh + (size_t)(r0 + r) * d;
                    const float* g = Gh + (size_t)(r0 + r) * d;
                    float* dq = Qb + (size_t)r * d;
                    const float pj = pT[(size_t)j * ORACLE_RB + r];
                    float dp = 0.0f;
                    for (int c = 0; c < d; ++c) dp += g[c] * vr[c];
                    const float ds = pj * (dp - D[r]);
                    for (int c = 0; c < d; ++c) {
                        dq[c] += ds * kr[c];
                        ak[c] += ds * q[c];
                        av[c] += pj * g[c];
                    }
                }
            }
            for (int r = 0; r < ORACLE_RB; ++r)
                for (int c = 0; c < d; ++c) Qb[(size_t)r * d + c] *= scale;
        }
        for (size_t e = 0; e < nd; ++e) aK[e] *= scale;
        free(pT); free(s);
    }
    return (long)threads * nblk * ORACLE_RB;
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline sample: forward + backward work of a strided set of query rows */
/* ------------------------------------------------------------------------- */

/* For query rows row0, row0+stride, ... of ONE head slab [N][d]: the full naive forward of
 * the row (naive_attention.h:15-58 in fp32, the reference's arithmetic type) followed by
 * that row's share of the backward in the O(N^2 d) form (dP, dS = P o (dP - D), its dQ row
 * and its contributions to dK and dV): 14 N d flops per row, i.e. exactly the row's share
 * of the 4 N^2 d + 10 N^2 d the bench counts.  Threads split the rows and keep private dK/dV
 * partials (summed at the end).  Used by bench.py to time the CPU path beside the GPU. */
void oracle_fwdbwd_rows_f32(const float* Q, const float* K, const float* V, const float* dO,
                            float* O_rows, float* dQ_rows, float* dK, float* dV,
                            int N, int d, float scale, int row0, int stride)
{
    if (scale == 0) scale = 1.0f / sqrtf((float)d);
    if (stride < 1) stride = 1;
    const int nrows = row0 < N ? (N - row0 + stride - 1) / stride : 0;
    const size_t nd = (size_t)N * d;
    for (size_t t = 0; t < nd; ++t) { dK[t] = 0.0f; dV[t] = 0.0f; }
#pragma omp parallel
    {
        float* p  = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
        float* aK = (float*)calloc(nd > 0 ? nd : 1, sizeof(float));
        float* aV = (float*)calloc(nd > 0 ? nd : 1, sizeof(float));
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < nrows; ++t) {
            const int i = row0 + t * stride;
            const float* q = Q + (size_t)i * d;
            const float* g = dO + (size_t)i * d;
            float total;
            (void)softmax_row_ref_order(q, K, N, d, scale, p, &total);
            for (int j = 0; j < N; ++j) p[j] /= total;
            float* o = O_rows + (size_t)t * d;
            for (int c = 0; c < d; ++c) o[c] = 0.0f;
            for (int j = 0; j < N; ++j) {
                const float* vr = V + (size_t)j * d;
                for (int c = 0; c < d; ++c) o[c] += p[j] * vr[c];
            }
            float D = 0.0f;
            for (int c = 0; c < d; ++c) D += g[c] * o[c];
            float* dq = dQ_rows + (size_t)t * d;
            for (int c = 0; c < d; ++c) dq[c] = 0.0f;
            for (int j = 0; j < N; ++j) {
                const float* vr = V + (size_t)j * d;
                const float* kr = K + (size_t)j * d;
                float dp = 0.0f;
                for (int c = 0; c < d; ++c) dp += g[c] * vr[c];
                const float ds = p[j] * (dp - D);
                float* ak = aK + (size_t)j * d;
                float* av = aV + (size_t)j * d;
                for (int c = 0; c < d; ++c) {
                    dq[c] += ds * kr[c];
                    ak[c] += ds * q[c];
                    av[c] += p[j] * g[c];
                }
            }
            for (int c = 0; c < d; ++c) dq[c] *= scale;
        }
#pragma omp critical
        {
            for (size_t t = 0; t < nd; ++t) { dK[t] += aK[t] * scale; dV[t] += aV[t]; }
        }
        free(p); free(aK); free(aV);
    }
}
