/* TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the reference DWFST loss hot path.
 *
 * Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * load this file (built as oracle/liboracle.so by oracle/Makefile), and only as the checker /
 * the reported CPU baseline ("kind": "port").
 *
 * Restates (reference files under /root/reference):
 *   waterfall/graph.py:61-66      lattice = intersect_dense(graph, dense); tot = get_tot_scores(log)
 *   k2 (third-party, not vendored, version unpinned; README.md:23)  semantics per SURVEY.md 8(0):
 *       alpha_0[s] = 0 if s == 0 else -inf
 *       alpha_{t+1}[d] = (+)_{a: dst=d, label>=0} alpha_t[src] + fl32(w + lp[t,label])
 *       tot = (+)_{a: label=-1} alpha_T[src] + w
 *       posterior(t,a) = exp(alpha_t[src] + x + beta_{t+1}[dst] - tot);  d tot/d lp[t,v] = sum of
 *       posteriors of arcs with label v.
 *   utils/lattice_analysis.py:129,133  tropical total + one-best (ties -> lowest arc index).
 * Parity: pinned by tests/test_oracle.py against the numpy oracle, which is pinned against golden
 * vectors produced by the reference's own manual_ctc code.  k2's output_beam pruning arithmetic is
 * parity unpinned (k2 absent).
 *
 * The same arithmetic is used for fp32 (ACC = float; what graph.py:83 use_double_scores=False
 * asks of k2) and fp64 accumulators.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t num_states;
    int32_t num_arcs;
    const int32_t *src, *dst, *label;
    const float *score;
} ograph;

#define DEFINE_FB(NAME, ACC, EXPF, LOGF)                                                          \
    /* work: 2*(T+1)*S ACC values. grad may be NULL (forward only). grad[t*ldg + v] += gscale*post */ \
    static ACC NAME(const ograph *g, const float *lp, int64_t ldlp, int T, float *grad,           \
                    int64_t ldg, float gscale, ACC *work) {                                       \
        const int S = g->num_states, A = g->num_arcs;                                             \
        ACC *alpha = work, *beta = work + (size_t)(T + 1) * S;                                    \
        ACC *m = (ACC *)malloc(sizeof(ACC) * S), *s = (ACC *)malloc(sizeof(ACC) * S);             \
        const ACC NI = (ACC)(-INFINITY);                                                          \
        for (int i = 0; i < S; ++i) alpha[i] = NI;                                                \
        alpha[0] = 0;                                                                             \
        for (int t = 0; t < T; ++t) {                                                             \
            const ACC *ac = alpha + (size_t)t * S;                                                \
            ACC *an = alpha + (size_t)(t + 1) * S;                                                \
            const float *e = lp + (size_t)t * ldlp;                                               \
            for (int i = 0; i < S; ++i) { m[i] = NI; s[i] = 0; }                                  \
            for (int a = 0; a < A; ++a) {                                                         \
                if (g->label[a] < 0) continue;                                                    \
                ACC v = ac[g->src[a]] + (ACC)(float)(g->score[a] + e[g->label[a]]);               \
                if (v > m[g->dst[a]]) m[g->dst[a]] = v;                                           \
            }                                                                                     \
            for (int a = 0; a < A; ++a) {                                                         \
                if (g->label[a] < 0) continue;                                                    \
                int d = g->dst[a];                                                                \
                if (m[d] == NI) continue;                                                         \
                ACC v = ac[g->src[a]] + (ACC)(float)(g->score[a] + e[g->label[a]]);               \
                s[d] += EXPF(v - m[d]);                                                           \
            }                                                                                     \
            for (int i = 0; i < S; ++i) an[i] = (m[i] == NI) ? NI : m[i] + LOGF(s[i]);            \
        }                                                                                         \
        ACC tm = NI, ts = 0;                                                                      \
        const ACC *aT = alpha + (size_t)T * S;                                                    \
        for (int a = 0; a < A; ++a)                                                               \
            if (g->label[a] < 0) { ACC v = aT[g->src[a]] + (ACC)g->score[a]; if (v > tm) tm = v; } \
        if (tm != NI)                                                                             \
            for (int a = 0; a < A; ++a)                                                           \
                if (g->label[a] < 0) ts += EXPF(aT[g->src[a]] + (ACC)g->score[a] - tm);           \
        ACC tot = (tm == NI) ? NI : tm + LOGF(ts);                                                \
        if (grad && tot != NI) {                                                                  \
            ACC *bT = beta + (size_t)T * S;                                                       \
            for (int i = 0; i < S; ++i) { m[i] = NI; s[i] = 0; }                                  \
            for (int a = 0; a < A; ++a)                                                           \
                if (g->label[a] < 0 && (ACC)g->score[a] > m[g->src[a]]) m[g->src[a]] = g->score[a]; \
            for (int a = 0; a < A; ++a)                                                           \
                if (g->label[a] < 0) s[g->src[a]] += EXPF((ACC)g->score[a] - m[g->src[a]]);       \
            for (int i = 0; i < S; ++i) bT[i] = (m[i] == NI) ? NI : m[i] + LOGF(s[i]);            \
            for (int t = T - 1; t >= 0; --t) {                                                    \
                const ACC *bn = beta + (size_t)(t + 1) * S;                                       \
                ACC *bc = beta + (size_t)t * S;                                                   \
                const ACC *ac = alpha + (size_t)t * S;                                            \
                const float *e = lp + (size_t)t * ldlp;                                           \
                float *gr = grad + (size_t)t * ldg;                                               \
                for (int i = 0; i < S; ++i) { m[i] = NI; s[i] = 0; }                              \
                for (int a = 0; a < A; ++a) {                                                     \
                    if (g->label[a] < 0) continue;                                                \
                    ACC v = (ACC)(float)(g->score[a] + e[g->label[a]]) + bn[g->dst[a]];           \
                    if (v > m[g->src[a]]) m[g->src[a]] = v;                                       \
                }                                                                                 \
                for (int a = 0; a < A; ++a) {                                                     \
                    if (g->label[a] < 0) continue;                                                \
                    int q = g->src[a];                                                            \
                    if (m[q] == NI) continue;                                                     \
                    ACC v = (ACC)(float)(g->score[a] + e[g->label[a]]) + bn[g->dst[a]];           \
                    s[q] += EXPF(v - m[q]);                                                       \
                    ACC p = ac[q] + v - tot;                                                      \
                    gr[g->label[a]] += gscale * (float)EXPF(p);                                   \
                }                                                                                 \
                for (int i = 0; i < S; ++i) bc[i] = (m[i] == NI) ? NI : m[i] + LOGF(s[i]);        \
            }                                                                                     \
        }                                                                                         \
        free(m); free(s);                                                                         \
        return tot;                                                                               \
    }

DEFINE_FB(fb_f32, float, expf, logf)
DEFINE_FB(fb_f64, double, exp, log)

/* Batch entry: graphs given as concatenated arc arrays with per-utterance offsets.
 * arc_off[B+1], nstates[B]; arcs local state ids.  seg[B*3] = (seq, start, dur).
 * tot[B] (fp32 of -loss sign: tot itself); grad [Bseq,T,V] must be zero-filled by the caller when
 * want_grad; gscale[B] multiplies posteriors (pass -1 for d loss/d lp with reduction='sum').
 * Returns 0. use_double: accumulate in fp64.  nthreads: OpenMP threads over utterances.      */
int oracle_graphloss_batch(int B, const int32_t *arc_off, const int32_t *nstates, const int32_t *src,
                           const int32_t *dst, const int32_t *label, const float *score,
                           const float *lp, int64_t strideB, int64_t strideT, int V,
                           const int32_t *seg, const float *gscale, int use_double, int want_grad,
                           int nthreads, float *tot, float *grad) {
    (void)V;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 0; i < B; ++i) {
        ograph g;
        g.num_states = nstates[i];
        g.num_arcs = arc_off[i + 1] - arc_off[i];
        g.src = src + arc_off[i]; g.dst = dst + arc_off[i];
        g.label = label + arc_off[i]; g.score = score + arc_off[i];
        int b = seg[3 * i], st = seg[3 * i + 1], T = seg[3 * i + 2];
        const float *lpb = lp + (size_t)b * strideB + (size_t)st * strideT;
        float *gb = want_grad ? grad + (size_t)b * strideB + (size_t)st * strideT : NULL;
        size_t n = (size_t)2 * (T + 1) * g.num_states;
        if (use_double) {
            double *w = (double *)malloc(sizeof(double) * n);
            tot[i] = (float)fb_f64(&g, lpb, strideT, T, gb, strideT, gscale ? gscale[i] : 1.f, w);
            free(w);
        } else {
            float *w = (float *)malloc(sizeof(float) * n);
            tot[i] = fb_f32(&g, lpb, strideT, T, gb, strideT, gscale ? gscale[i] : 1.f, w);
            free(w);
        }
    }
    return 0;
}

/* The log-semiring totals of oracle_graphloss_batch(use_double = 1) WITHOUT the final rounding to fp32: what
 * lattice.get_tot_scores(log_semiring=True, use_double_scores=True) returns (utils/lattice_analysis.py:128,148).    */
int oracle_tot_scores_f64(int B, const int32_t *arc_off, const int32_t *nstates, const int32_t *src,
                          const int32_t *dst, const int32_t *label, const float *score,
                          const float *lp, int64_t strideB, int64_t strideT,
                          const int32_t *seg, int nthreads, double *tot) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 0; i < B; ++i) {
        ograph g;
        g.num_states = nstates[i];
        g.num_arcs = arc_off[i + 1] - arc_off[i];
        g.src = src + arc_off[i]; g.dst = dst + arc_off[i];
        g.label = label + arc_off[i]; g.score = score + arc_off[i];
        int b = seg[3 * i], st = seg[3 * i + 1], T = seg[3 * i + 2];
        const float *lpb = lp + (size_t)b * strideB + (size_t)st * strideT;
        double *w = (double *)malloc(sizeof(double) * (size_t)2 * (T + 1) * g.num_states);
        tot[i] = fb_f64(&g, lpb, strideT, T, NULL, strideT, 1.f, w);
        free(w);
    }
    return 0;
}

/* Tropical best path, fp32 arithmetic v = fl(alpha[src] + fl(w + lp)), ties -> lowest arc index
 * (see oracle/dwfst_oracle.py viterbi()).  path[T+1] receives arc ids local to the graph.
 * Returns the score; -inf when infeasible (path untouched).                                    */
float oracle_viterbi_f32(int S, int A, const int32_t *src, const int32_t *dst, const int32_t *label,
                         const float *score, const float *lp, int64_t ldlp, int T, int32_t *path) {
    float *al = (float *)malloc(sizeof(float) * S), *an = (float *)malloc(sizeof(float) * S);
    int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * S);
    for (int i = 0; i < S; ++i) al[i] = -INFINITY;
    al[0] = 0;
    for (int t = 0; t < T; ++t) {
        const float *e = lp + (size_t)t * ldlp;
        int32_t *b = bp + (size_t)t * S;
        for (int i = 0; i < S; ++i) { an[i] = -INFINITY; b[i] = -1; }
        for (int a = 0; a < A; ++a) {
            if (label[a] < 0) continue;
            float v = al[src[a]] + (float)(score[a] + e[label[a]]);
            if (v > an[dst[a]]) { an[dst[a]] = v; b[dst[a]] = a; }   /* strict > keeps lowest index */
        }
        float *tmp = al; al = an; an = tmp;
    }
    float best = -INFINITY; int fa = -1;
    for (int a = 0; a < A; ++a)
        if (label[a] < 0) { float v = al[src[a]] + score[a]; if (v > best) { best = v; fa = a; } }
    if (fa >= 0) {
        path[T] = fa;
        int s = src[fa];
        for (int t = T - 1; t >= 0; --t) { int a = bp[(size_t)t * S + s]; path[t] = a; s = src[a]; }
    }
    free(al); free(an); free(bp);
    return best;
}
