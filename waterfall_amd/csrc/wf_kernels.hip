// HIP kernels (gfx950 / CDNA4) and C-ABI compute entry points of libwfdwfst.so.
//
// What runs on the GPU for one wf_fb_logsum() call (reference: waterfall/graph.py:61-66 and its
// autograd backward, i.e. k2.intersect_dense + get_tot_scores + their backward passes):
//
//   1. gather_kernel    E[i,t,k] = log_probs[b, start+t, label_k]  for graphs that use few labels
//                       (numerators); graphs that use most labels (denominators) read log_probs
//                       rows directly.  Fully parallel, HBM/L2 bound.
//   2. sweep_kernel     one workgroup per (utterance, direction).  Frame-synchronous recursion with
//                       alpha (or beta) double-buffered in LDS, the CSR arc arrays staged in LDS
//                       when they fit, emission rows prefetched a chunk of frames ahead through
//                       registers into an LDS ring, lane-per-state log-sum-exp for ordinary states
//                       and a wavefront-shuffle + LDS block reduction for high fan-in states.
//                       Forward and backward sweeps are independent and run concurrently; every
//                       alpha_t / beta_t row is streamed to the workspace.  Latency bound (T
//                       dependent steps), no lattice is ever materialised.
//   3. post_kernel      one workgroup per (sequence, frame): arc posteriors
//                       exp(alpha_t[src] + x + beta_{t+1}[dst] - tot) reduced per label in a fixed
//                       order (no float atomics => bit-stable), then the whole gradient row is
//                       written (zeros + K values for numerators).  HBM-write bound.
//
// wf_viterbi(): sweep_kernel in the tropical semiring storing one back-pointer per lattice state,
// then backtrace_kernel (one lane per utterance).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/wf_dwfst.h"
#include "wf_internal.h"

int wf_fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return wf_fail(WF_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

constexpr int kMaxThreads = 1024;
constexpr int kPF = 8;        // emission prefetch registers per thread
constexpr int kMaxChunk = 32; // frames per emission chunk
constexpr int kMaxHeavy = 16; // heavy states reduced per cooperative round
constexpr size_t kLdsBudget = 160 * 1024;  // one workgroup may own the CU's whole LDS (MI355X: 160 KiB)

enum SweepMode { M_FB = 0, M_TOT_LOG = 1, M_TOT_MAX = 2, M_VITERBI = 3 };

// ------------------------------------------------------------------------------------------
// semiring arithmetic
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ double fexp(double x) { return exp(x); }
__device__ __forceinline__ double flog(double x) { return log(x); }

template <typename Acc> __device__ __forceinline__ Acc neg_inf() { return (Acc)(-INFINITY); }

// running log-sum-exp state: m = running max (Acc), s = sum of exp(v - m) (Sum).
// Acc = double, Sum = float is the default log-semiring arithmetic: alpha/beta and every max are
// kept in fp64, only the *differences* v - m (small numbers) go through the fp32 v_exp/v_log units.
// Plain fp32 alpha (what k2 does with use_double_scores=False) drifts by ~3e-3 in the posterior
// exponent over T = 500 frames, which breaks the 1e-3 gradient tolerance.
template <typename Acc, typename Sum> __device__ __forceinline__ void lse_push(Acc& m, Sum& s, Acc v) {
    Sum d = (v == neg_inf<Acc>()) ? neg_inf<Sum>() : (Sum)(v - m);  // m == -inf, v finite -> +inf
    Sum e = fexp(-fabs(d));
    s = d > 0 ? s * e + (Sum)1 : s + e;
    m = fmax(m, v);
}
template <typename Acc, typename Sum> __device__ __forceinline__ void lse_merge(Acc& m, Sum& s, Acc m2, Sum s2) {
    Acc mn = fmax(m, m2);
    if (mn == neg_inf<Acc>()) { m = mn; s = 0; return; }
    s = s * fexp((Sum)(m - mn)) + s2 * fexp((Sum)(m2 - mn));
    m = mn;
}
template <typename Acc, typename Sum> __device__ __forceinline__ Acc lse_value(Acc m, Sum s) {
    return (m == neg_inf<Acc>()) ? m : m + (Acc)flog(s);
}
// tropical: (value, arc) with ties -> lowest arc index
template <typename Acc> __device__ __forceinline__ void max_push(Acc& m, int& arc, Acc v, int a) {
    if (v > m || (v == m && v != neg_inf<Acc>() && a < arc)) { m = v; arc = a; }
}

template <typename T> __device__ __forceinline__ T shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }

struct SweepArgs {
    const int32_t* blob;
    const int64_t* utt;
    const float* lp;
    int64_t strideB, strideT;
    const float* E;
    void* alpha;   // arena of Acc
    void* beta;
    void* tot;     // float[B] (double[B] for tot-only + double acc)
    double* totd;  // workspace copy of tot in fp64 for post_kernel (M_FB), may be null
    double* aoff;  // per-frame offsets of the renormalised fp32 alpha rows (register sweep, M_FB)
    double* boff;
    double* tot2;  // tot in log2 units (quad sweep -> post kernels)
    int32_t* bp;   // viterbi back-pointers, arena int32: [utt](T x S)
    int32_t* fin_arc;  // viterbi: best final arc per utterance [B]
    int chunk;     // frames per emission chunk (ESTAGE)
    int kpad;      // LDS emission row pitch
    int smax;      // LDS alpha pitch (max states)
    int arcs_lds_words;  // words reserved for the LDS copy of the CSR arrays
    int hcap;      // heavy states reduced per cooperative round (sizes the LDS reduction scratch)
    int tot_double;
    int ablate;    // measurement aid (WF_SWEEP_PROFILE=1 sets bit 8): workgroup 0 prints its cycles per frame step
};

// arc accessors: CSR arrays either in LDS (packed 16+16) or in global memory
struct ArcsLds {
    const int32_t* off;
    const int32_t* pk;
    const float* w;
    __device__ __forceinline__ int begin(int s) const { return off[s]; }
    __device__ __forceinline__ void get(int a, int& oth, int& col, float& ww) const {
        int p = pk[a];
        oth = p & 0xffff;
        col = (p >> 16) & 0xffff;
        ww = w[a];
    }
};
struct ArcsGlobal {
    const int32_t* off;
    const int32_t* oth_;
    const int32_t* col_;
    const float* w;
    __device__ __forceinline__ int begin(int s) const { return off[s]; }
    __device__ __forceinline__ void get(int a, int& oth, int& col, float& ww) const {
        oth = oth_[a];
        col = col_[a];
        ww = w[a];
    }
};

// One frame step over the states owned by this thread + cooperative heavy states.
// cur/nxt: LDS alpha buffers; erow(col) returns the emission of column col for this frame.
template <typename Acc, typename Sum, bool TROP, typename Arcs, typename ERow>
__device__ __forceinline__ void sweep_step(const Arcs& arcs, const ERow& erow, const Acc* cur, Acc* nxt,
                                           int S, int nheavy, int hcap, const int32_t* heavy, Acc* red,
                                           int* red_arc, const int32_t* arc_id, int32_t* bp_row,
                                           Acc* store_row) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    // ---- heavy states: every thread takes a strided slice of the arc list
    for (int h0 = 0; h0 < nheavy; h0 += hcap) {
        const int hn = min(hcap, nheavy - h0);
        for (int hi = 0; hi < hn; ++hi) {
            const int h = heavy[h0 + hi];
            const int a0 = arcs.begin(h), a1 = arcs.begin(h + 1);
            Acc m = neg_inf<Acc>();
            Sum s = 0;
            int barc = 0x7fffffff;
            for (int a = a0 + tid; a < a1; a += nt) {
                int oth, col; float w;
                arcs.get(a, oth, col, w);
                Acc v = cur[oth] + (Acc)(w + erow(col));
                if (TROP) max_push(m, barc, v, arc_id ? arc_id[a] : a);
                else lse_push(m, s, v);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                Acc m2 = shfl_xor(m, o);
                if (TROP) { int b2 = shfl_xor(barc, o); max_push(m, barc, m2, b2); }
                else { Sum s2 = shfl_xor(s, o); lse_merge(m, s, m2, s2); }
            }
            if (lane == 0) {
                red[(hi * 16 + wave) * 2] = m;
                red[(hi * 16 + wave) * 2 + 1] = (Acc)s;
                if (TROP) red_arc[hi * 16 + wave] = barc;
            }
        }
        __syncthreads();
        if (tid < hn) {
            Acc m = red[(tid * 16) * 2];
            Sum s = (Sum)red[(tid * 16) * 2 + 1];
            int barc = TROP ? red_arc[tid * 16] : 0;
            for (int w = 1; w < nw; ++w) {
                Acc m2 = red[(tid * 16 + w) * 2];
                Sum s2 = (Sum)red[(tid * 16 + w) * 2 + 1];
                if (TROP) max_push(m, barc, m2, red_arc[tid * 16 + w]);
                else lse_merge(m, s, m2, s2);
            }
            const int h = heavy[h0 + tid];
            Acc val = TROP ? m : lse_value(m, s);
            nxt[h] = val;
            if (store_row) store_row[h] = val;
            if (TROP && bp_row) bp_row[h] = (m == neg_inf<Acc>()) ? -1 : barc;
        }
        if (h0 + hcap < nheavy) __syncthreads();
    }
    // ---- ordinary states: one lane per state
    for (int sidx = tid; sidx < S; sidx += nt) {
        const int a0 = arcs.begin(sidx), a1 = arcs.begin(sidx + 1);
        if (a1 - a0 > WF_HEAVY_DEG) continue;
        Acc m = neg_inf<Acc>();
        Sum s = 0;
        int barc = 0x7fffffff;
        for (int a = a0; a < a1; ++a) {
            int oth, col; float w;
            arcs.get(a, oth, col, w);
            Acc v = cur[oth] + (Acc)(w + erow(col));
            if (TROP) max_push(m, barc, v, arc_id ? arc_id[a] : a);
            else lse_push(m, s, v);
        }
        Acc val = TROP ? m : lse_value(m, s);
        nxt[sidx] = val;
        if (store_row) store_row[sidx] = val;
        if (TROP && bp_row) bp_row[sidx] = (m == neg_inf<Acc>()) ? -1 : barc;
    }
}

struct ERowLds {
    const float* p;
    __device__ __forceinline__ float operator()(int col) const { return p[col]; }
};
struct ERowGlobal {
    const float* p;
    __device__ __forceinline__ float operator()(int col) const { return p[col]; }
};

template <typename Acc, typename Sum, int MODE, bool ARCS_LDS, bool ESTAGE>
__global__ void __launch_bounds__(kMaxThreads) sweep_kernel(SweepArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool TROP = (MODE == M_TOT_MAX || MODE == M_VITERBI);
    constexpr int NDIR = (MODE == M_FB) ? 2 : 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int u = blockIdx.x / NDIR, dir = blockIdx.x % NDIR;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = (int)U[WF_U_FSA], T = (int)U[WF_U_T];
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = I[WF_I_S], A = I[WF_I_A], K = I[WF_I_K], direct = I[WF_I_DIRECT];
    const int Kd = max(K, 1);

    // ---- LDS carve-up
    Acc* buf0 = (Acc*)smem;
    Acc* buf1 = buf0 + p.smax;
    Acc* red = buf1 + p.smax;                       // [hcap][16][2]
    int* red_arc = (int*)(red + p.hcap * 16 * 2);   // [hcap][16]
    float* ebuf = (float*)(red_arc + p.hcap * 16);  // [2][chunk*kpad]
    int32_t* larcs = (int32_t*)(ebuf + (ESTAGE ? 2 * p.chunk * p.kpad : 0));

    const int32_t* g_off = blob + I[dir ? WF_I_OUT_OFF : WF_I_IN_OFF];
    const int32_t* g_oth = blob + I[dir ? WF_I_OUT_DST : WF_I_IN_SRC];
    const int32_t* g_col = blob + I[dir ? WF_I_OUT_COL : WF_I_IN_COL];
    const float* g_w = (const float*)(blob + I[dir ? WF_I_OUT_W : WF_I_IN_W]);
    const int nheavy = I[dir ? WF_I_NHEAVY_OUT : WF_I_NHEAVY_IN];
    const int32_t* heavy = blob + I[dir ? WF_I_HEAVY_OUT : WF_I_HEAVY_IN];
    const int32_t* arc_id = (MODE == M_VITERBI) ? blob + I[WF_I_IN_ARC] : nullptr;

    if (ARCS_LDS) {
        int32_t* l_off = larcs;
        int32_t* l_pk = l_off + ((S + 1 + 3) & ~3);
        float* l_w = (float*)(l_pk + ((A + 3) & ~3));
        for (int i = tid; i <= S; i += nt) l_off[i] = g_off[i];
        for (int i = tid; i < A; i += nt) {
            l_pk[i] = (g_oth[i] & 0xffff) | (g_col[i] << 16);
            l_w[i] = g_w[i];
        }
    }

    // emission rows: compact -> gathered arena (pitch K); direct -> log_probs (pitch strideT)
    const float* ebase;
    int64_t epitch;
    if (direct) {
        ebase = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
        epitch = p.strideT;
    } else {
        ebase = p.E + U[WF_U_E];
        epitch = K;
    }
    auto frame_of = [&](int step) { return dir ? (T - 1 - step) : step; };

    Acc* arena = (Acc*)(dir ? p.beta : p.alpha) + U[dir ? WF_U_BETA : WF_U_ALPHA];
    Acc* cur = buf0;
    Acc* nxt = buf1;

    // ---- initial vector: alpha_0 = delta(start) ; beta_T = final weights
    {
        const float* fb = (const float*)(blob + I[TROP ? WF_I_FINB_MAX : WF_I_FINB_LOG]);
        for (int s = tid; s < S; s += nt) {
            Acc v = dir ? (Acc)fb[s] : (s == 0 ? (Acc)0 : neg_inf<Acc>());
            cur[s] = v;
            if (MODE == M_FB) arena[(int64_t)(dir ? T : 0) * S + s] = v;
        }
    }

    // ---- emission prefetch state
    const int C = p.chunk;
    float pre[kPF];
    auto issue_chunk = [&](int c) {
        if (!ESTAGE) return;
        const int step0 = c * C;
#pragma unroll
        for (int r = 0; r < kPF; ++r) {
            const int j = tid + r * nt;
            const int so = j / Kd, col = j - so * Kd;
            const int step = step0 + so;
            pre[r] = (so < C && step < T && col < K) ? ebase[(int64_t)frame_of(step) * epitch + col] : 0.f;
        }
    };
    auto commit_chunk = [&](int c) {
        if (!ESTAGE) return;
        float* dstb = ebuf + (c & 1) * C * p.kpad;
#pragma unroll
        for (int r = 0; r < kPF; ++r) {
            const int j = tid + r * nt;
            const int so = j / Kd, col = j - so * Kd;
            if (so < C && col < K) dstb[so * p.kpad + col] = pre[r];
        }
    };
    issue_chunk(0);
    commit_chunk(0);
    __syncthreads();

    const int nchunks = ESTAGE ? (T + C - 1) / C : 1;
    const int steps_per_chunk = ESTAGE ? C : T;
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) issue_chunk(c + 1);
        const int step_end = min(T, (c + 1) * steps_per_chunk);
        for (int step = c * steps_per_chunk; step < step_end; ++step) {
            const int f = frame_of(step);
            // row written this step: forward -> alpha_{f+1}; backward -> beta_f
            Acc* store_row = (MODE == M_FB) ? arena + (int64_t)(dir ? f : f + 1) * S : nullptr;
            int32_t* bp_row = (MODE == M_VITERBI) ? p.bp + U[WF_U_ALPHA] + (int64_t)f * S : nullptr;
            if (ESTAGE) {
                ERowLds er{ebuf + (c & 1) * C * p.kpad + (step - c * C) * p.kpad};
                if (ARCS_LDS) {
                    ArcsLds arcs{larcs, larcs + ((S + 1 + 3) & ~3),
                                 (const float*)(larcs + ((S + 1 + 3) & ~3) + ((A + 3) & ~3))};
                    sweep_step<Acc, Sum, TROP>(arcs, er, cur, nxt, S, nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row);
                } else {
                    ArcsGlobal arcs{g_off, g_oth, g_col, g_w};
                    sweep_step<Acc, Sum, TROP>(arcs, er, cur, nxt, S, nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row);
                }
            } else {
                ERowGlobal er{ebase + (int64_t)f * epitch};
                if (ARCS_LDS) {
                    ArcsLds arcs{larcs, larcs + ((S + 1 + 3) & ~3),
                                 (const float*)(larcs + ((S + 1 + 3) & ~3) + ((A + 3) & ~3))};
                    sweep_step<Acc, Sum, TROP>(arcs, er, cur, nxt, S, nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row);
                } else {
                    ArcsGlobal arcs{g_off, g_oth, g_col, g_w};
                    sweep_step<Acc, Sum, TROP>(arcs, er, cur, nxt, S, nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row);
                }
            }
            if (step + 1 == step_end && c + 1 < nchunks) commit_chunk(c + 1);
            __syncthreads();
            Acc* t_ = cur; cur = nxt; nxt = t_;
        }
    }

    // ---- forward: total score over final arcs (block reduction)
    if (dir == 0) {
        const int NF = I[WF_I_NF];
        const int32_t* fsrc = blob + I[WF_I_FIN_SRC];
        const float* fw = (const float*)(blob + I[WF_I_FIN_W]);
        const int32_t* farc = blob + I[WF_I_FIN_ARC];
        Acc m = neg_inf<Acc>();
        Sum s = 0;
        int barc = 0x7fffffff;
        for (int i = tid; i < NF; i += nt) {
            Acc v = cur[fsrc[i]] + (Acc)fw[i];
            if (TROP) max_push(m, barc, v, farc[i]);
            else lse_push(m, s, v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Acc m2 = shfl_xor(m, o);
            if (TROP) { int b2 = shfl_xor(barc, o); max_push(m, barc, m2, b2); }
            else { Sum s2 = shfl_xor(s, o); lse_merge(m, s, m2, s2); }
        }
        const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
        if (lane == 0) { red[wave * 2] = m; red[wave * 2 + 1] = (Acc)s; if (TROP) red_arc[wave] = barc; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) {
                if (TROP) max_push(m, barc, red[w * 2], red_arc[w]);
                else lse_merge(m, s, red[w * 2], (Sum)red[w * 2 + 1]);
            }
            Acc val = TROP ? m : lse_value(m, s);
            if (p.tot_double) ((double*)p.tot)[u] = (double)val;
            else ((float*)p.tot)[u] = (float)val;
            if (MODE == M_FB && p.totd) p.totd[u] = (double)val;
            if (MODE == M_VITERBI) p.fin_arc[u] = (m == neg_inf<Acc>()) ? -1 : barc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// register-resident "quad" sweep: the fast path for graphs whose states have at most WF_REG_DEG (4)
// arcs on the swept side (every numerator topo o transcript) plus at most one
// high-degree state (state 0 of a topology used as denominator).
//
// Measured on MI355X: a lone wavefront issues about one instruction every 4-5 cycles whatever its
// kind, so the latency of a frame step is (instructions per wave) x 4 + LDS round trip + barrier.
// The step is therefore spread as thinly as possible: FOUR LANES PER STATE, one arc per lane.  The
// graph does not change from frame to frame, so each lane keeps its arc's (src | column, weight) in
// VGPRs; a step is one ds_read of alpha[src], one fma/add, a 2-step DPP quad max, v_exp, a 2-step DPP
// quad sum, v_log, one ds_write by the quad's first lane, one s_barrier, and one coalesced store of
// the alpha row.  Row / emission pointers advance incrementally in SGPRs.  Emissions are prefetched
// kPD frames ahead by the lane that will use them into a register ring (the frame loop is unrolled
// kPD times so the ring is statically indexed).
//
// Log semiring: fp32 in the log2 domain (v_exp/v_log are base 2) with renormalisation: LDS and the
// stored rows hold alpha~ = alpha*log2(e) - c, where the workgroup-uniform fp64 offset c advances
// once per group of kPD frames by the maximum of the last row, so the values that matter stay
// within ~1e2 of zero and fp32 keeps ~1e-5 absolute precision (plain fp32 alpha of magnitude ~2500
// drifts by 3e-3 over T = 500: DESIGN.md section 2).  Dead states hold -1e30 instead of -inf, which
// removes every special case from the step (no NaN can arise).  c is stored once per group.
// Tropical semiring: the same kernel with exact fp32 add/max in natural units (no scaling, no
// renormalisation), ties -> lowest arc index, one back-pointer per lattice state.
// ------------------------------------------------------------------------------------------
constexpr int kPD = 8;
constexpr float kBig = 1e30f;
constexpr float kDead = -1e29f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr double kLn2 = 0.6931471805599453;

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xf, false);
}
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E: every lane has a source, all four lanes get the result
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, dpp_f<0xB1, 0xf>(v, v));
    return fmaxf(v, dpp_f<0x4E, 0xf>(v, v));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_f<0xB1, 0xf>(v, v);
    return v + dpp_f<0x4E, 0xf>(v, v);
}
// full-wave reductions on the DPP network (row_shr:4, row_shr:8, row_bcast:15, row_bcast:31); the
// result lands in lane 63 and is broadcast through an SGPR.  Lanes without a source keep `old`.
__device__ __forceinline__ float wave_max(float v) {
    v = quad_max(v);
    v = fmaxf(v, dpp_f<0x114, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x118, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = quad_sum(v);
    v += dpp_f<0x114, 0xf>(0.f, v);
    v += dpp_f<0x118, 0xf>(0.f, v);
    v += dpp_f<0x142, 0xa>(0.f, v);
    v += dpp_f<0x143, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<0xB1, 0xf>(v, v));
    v = min(v, dpp_i<0x4E, 0xf>(v, v));
    v = min(v, dpp_i<0x114, 0xf>(v, v));
    v = min(v, dpp_i<0x118, 0xf>(v, v));
    v = min(v, dpp_i<0x142, 0xa>(v, v));
    v = min(v, dpp_i<0x143, 0xc>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

template <int MODE, int R, int HD>
__global__ void __launch_bounds__(kMaxThreads) sweep_quad_kernel(SweepArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool TROP = (MODE == M_TOT_MAX || MODE == M_VITERBI);
    constexpr bool LOGS = !TROP;
    constexpr int NDIR = (MODE == M_FB) ? 2 : 1;
    constexpr int HDN = HD > 0 ? HD : 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int slot = tid & 3, quad = tid >> 2, nq = nt >> 2;
    const int u = blockIdx.x / NDIR, dir = blockIdx.x % NDIR;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = (int)U[WF_U_FSA], T = (int)U[WF_U_T];
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = I[WF_I_S], A = I[WF_I_A], K = I[WF_I_K], direct = I[WF_I_DIRECT];
    const float escale = LOGS ? kLog2e : 1.f;

    float* cur = (float*)smem;
    float* nxt = cur + p.smax;
    float* red = nxt + p.smax;             // [16][2]
    int* red_arc = (int*)(red + 32);       // [16]
    float* rmax = (float*)(red_arc + 16);  // [16] per-wave maxima for the renormalisation

    const int32_t* g_off = blob + I[dir ? WF_I_OUT_OFF : WF_I_IN_OFF];
    const int32_t* g_oth = blob + I[dir ? WF_I_OUT_DST : WF_I_IN_SRC];
    const int32_t* g_col = blob + I[dir ? WF_I_OUT_COL : WF_I_IN_COL];
    const float* g_w = (const float*)(blob + I[dir ? WF_I_OUT_W : WF_I_IN_W]);
    const int32_t* arc_id = blob + I[WF_I_IN_ARC];
    const bool has_over = HD > 0 && I[dir ? WF_I_NOVER_OUT : WF_I_NOVER_IN] > 0;
    const int over = has_over ? I[dir ? WF_I_OVER_OUT : WF_I_OVER_IN] : -1;
    const bool has_arcs = A > 0;

    // ---- this lane's arc of each of its states (padding: weight -1e30, source = the state itself)
    int pk[R];
    float w[R];
    int aid[R];
    bool writer[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int sidx = quad + r * nq;
        int a0 = 0, deg = 0;
        if (sidx < S && sidx != over) { a0 = g_off[sidx]; deg = g_off[sidx + 1] - a0; }
        const bool valid = slot < deg;
        const int a = a0 + slot;
        pk[r] = valid ? ((g_oth[a] & 0xffff) | (g_col[a] << 16)) : (sidx < S ? sidx : 0);
        w[r] = valid ? fmaxf(g_w[a] * escale, -kBig) : -kBig;
        aid[r] = (valid && MODE == M_VITERBI) ? arc_id[a] : 0x7fffffff;
        writer[r] = slot == 0 && sidx < S && sidx != over;
    }
    int hpk[HDN];
    float hw[HDN];
    int haid[HDN];
    {
        int h0 = 0, h1 = 0;
        if (has_over) { h0 = g_off[over]; h1 = g_off[over + 1]; }
#pragma unroll
        for (int j = 0; j < HDN; ++j) {
            const int a = h0 + tid + j * nt;
            const bool valid = HD > 0 && a < h1;
            hpk[j] = valid ? ((g_oth[a] & 0xffff) | (g_col[a] << 16)) : 0;
            hw[j] = valid ? fmaxf(g_w[a] * escale, -kBig) : -kBig;
            haid[j] = (valid && MODE == M_VITERBI) ? arc_id[a] : 0x7fffffff;
        }
    }

    const float* ebase;
    int64_t epitch;
    if (direct) {
        ebase = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
        epitch = p.strideT;
    } else {
        ebase = p.E + U[WF_U_E];
        epitch = K;
    }
    // Loads are unconditional (a select around a load makes hipcc branch and drain vmcnt per element):
    // steps past the end re-read the last frame, and a graph without arcs / an empty segment reads
    // the blob header through a zero pitch.
    if (!has_arcs || T == 0) { ebase = (const float*)blob; epitch = 0; }
    const int last_step = max(T - 1, 0);
    const int64_t estep = dir ? -epitch : epitch;
    const float* eload = ebase + (int64_t)(dir ? last_step : 0) * epitch;   // row of load index 0

    float* arena = (float*)(dir ? p.beta : p.alpha) + U[dir ? WF_U_BETA : WF_U_ALPHA];
    double* goff = (MODE == M_FB) ? (dir ? p.boff : p.aoff) + U[WF_U_OFFS] : nullptr;
    double coff = 0.0;   // current offset c (log2 units)

    {
        const float* fb = (const float*)(blob + I[TROP ? WF_I_FINB_MAX : WF_I_FINB_LOG]);
        for (int s = tid; s < S; s += nt) {
            float v;
            if (dir) { const float f = fb[s]; v = (f == -INFINITY) ? -kBig : f * escale; }
            else v = (s == 0) ? 0.f : -kBig;
            cur[s] = v;
            if (MODE == M_FB) arena[(int64_t)(dir ? T : 0) * S + s] = v;
        }
    }
    // row written by step 0: forward alpha_1, backward beta_{T-1}; then +-S per step
    float* srow = arena + (int64_t)(dir ? (T - 1) : 1) * S;
    const int64_t sstep = dir ? -(int64_t)S : (int64_t)S;
    int32_t* bprow = (MODE == M_VITERBI) ? p.bp + U[WF_U_ALPHA] : nullptr;

    // ---- emission ring
    float er[kPD][R];
    float her[kPD][HDN];
#define WF_LOAD_FRAME(SLOT)                                                                \
    do {                                                                                   \
        _Pragma("unroll") for (int r = 0; r < R; ++r) er[SLOT][r] = eload[(pk[r] >> 16) & 0xffff]; \
        if (HD > 0) {                                                                      \
            _Pragma("unroll") for (int j = 0; j < HDN; ++j) her[SLOT][j] = eload[(hpk[j] >> 16) & 0xffff]; \
        }                                                                                  \
    } while (0)
#pragma unroll
    for (int k = 0; k < kPD; ++k) {
        WF_LOAD_FRAME(k);
        if (k < last_step) eload += estep;
    }
    __syncthreads();

    // one frame step using (and then refilling) ring slot k; k is a constant after unrolling
    auto do_step = [&](const int step, const int k) __attribute__((always_inline)) {
        // consume this frame's emissions (x = the lattice arc score) before the ring slot is refilled,
        // so that the refill lands in the same registers
        float x[R], hx[HDN];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[r] = LOGS ? fmaf(er[k][r], kLog2e, w[r]) : w[r] + er[k][r];
            asm volatile("" : "+v"(x[r]));   // pin: must not sink below the refill
        }
#pragma unroll
        for (int j = 0; j < HDN; ++j) {
            hx[j] = LOGS ? fmaf(her[k][j], kLog2e, hw[j]) : hw[j] + her[k][j];
            asm volatile("" : "+v"(hx[j]));
        }
        // keep the scheduler from hoisting the refill above the reads of the old values: that would
        // need a second register set and a copy + vmcnt(0) on the loop back-edge
        __builtin_amdgcn_sched_barrier(0);
        WF_LOAD_FRAME(k);
        if (step + kPD < last_step) eload += estep;
        if (LOGS && k == 0) {
            if (step >= kPD) {
                // advance the offset by the maximum of the row finished at the end of the previous group
                float m = rmax[0];
                for (int i = 1; i < nw; ++i) m = fmaxf(m, rmax[i]);
                if (m < kDead) m = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) x[r] -= m;
#pragma unroll
                for (int j = 0; j < HDN; ++j) hx[j] -= m;
                coff += (double)m;
            }
            if (MODE == M_FB && tid == 0) goff[step / kPD] = coff;
        }
        float rowmax = -kBig;
        // ---- the high-degree state: one arc (or HD arcs) per lane, wave (DPP) + block (LDS) reduction
        if (HD > 0 && has_over) {
            float hv[HDN];
            float m = -INFINITY;
            int barc = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < HDN; ++j) {
                hv[j] = cur[hpk[j] & 0xffff] + hx[j];
                if (TROP) { if (hv[j] > m || (hv[j] == m && haid[j] < barc)) { m = hv[j]; barc = haid[j]; } }
                else m = fmaxf(m, hv[j]);
            }
            const float M = wave_max(m);
            float ssum = 0.f;
            if (TROP) {
                barc = wave_min_i(m == M ? barc : 0x7fffffff);
            } else {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < HDN; ++j) sm += __builtin_amdgcn_exp2f(hv[j] - M);
                ssum = wave_sum(sm);
            }
            float val = TROP ? M : M + __builtin_amdgcn_logf(ssum);
            if (nw > 1) {
                if (lane == 0) { red[wave * 2] = M; red[wave * 2 + 1] = ssum; if (TROP) red_arc[wave] = barc; }
                __syncthreads();
                if (tid == 0) {
                    float Mx = red[0];
                    for (int i = 1; i < nw; ++i) Mx = fmaxf(Mx, red[i * 2]);
                    if (TROP) {
                        int b = 0x7fffffff;
                        for (int i = 0; i < nw; ++i) if (red[i * 2] == Mx) b = min(b, red_arc[i]);
                        barc = b;
                        val = Mx;
                    } else {
                        float st = 0.f;
                        for (int i = 0; i < nw; ++i) st += red[i * 2 + 1] * __builtin_amdgcn_exp2f(red[i * 2] - Mx);
                        val = Mx + __builtin_amdgcn_logf(st);
                    }
                }
            }
            if (tid == 0) {
                nxt[over] = val;
                if (MODE == M_FB) srow[over] = val;
                if (MODE == M_VITERBI) bprow[over] = (val < kDead) ? -1 : barc;
                rowmax = val;
            }
        }
        // ---- ordinary states: one quad per state
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int sidx = quad + r * nq;
            const float v = cur[pk[r] & 0xffff] + x[r];
            float val;
            int barc = aid[r];
            if (TROP) {
                // (max value, lowest arc id) over the quad: two symmetric exchange steps
                float m = v;
                {
                    const float v2 = dpp_f<0xB1, 0xf>(m, m);
                    const int a2 = dpp_i<0xB1, 0xf>(barc, barc);
                    if (v2 > m || (v2 == m && a2 < barc)) { m = v2; barc = a2; }
                }
                {
                    const float v2 = dpp_f<0x4E, 0xf>(m, m);
                    const int a2 = dpp_i<0x4E, 0xf>(barc, barc);
                    if (v2 > m || (v2 == m && a2 < barc)) { m = v2; barc = a2; }
                }
                val = m;
            } else {
                const float m = quad_max(v);
                const float sm = quad_sum(__builtin_amdgcn_exp2f(v - m));
                val = m + __builtin_amdgcn_logf(sm);
            }
            if (writer[r]) {
                nxt[sidx] = val;
                if (MODE == M_FB) srow[sidx] = val;
                if (MODE == M_VITERBI) bprow[sidx] = (val < kDead) ? -1 : barc;
            }
            if (LOGS) rowmax = fmaxf(rowmax, val);
        }
        if (LOGS && k == kPD - 1) {
            const float wm = wave_max(rowmax);
            if (lane == 0) rmax[wave] = wm;
        }
        srow += sstep;
        if (MODE == M_VITERBI) bprow += S;
        __syncthreads();
        float* t_ = cur; cur = nxt; nxt = t_;
    };
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    int t0 = 0;
    for (; t0 + kPD <= T; t0 += kPD) {   // full groups: no per-step guard, the ring stays in place
#pragma unroll
        for (int k = 0; k < kPD; ++k) do_step(t0 + k, k);
    }
#pragma unroll
    for (int k = 0; k < kPD; ++k)
        if (t0 + k < T) do_step(t0 + k, k);
#undef WF_LOAD_FRAME
    if ((p.ablate & 8) && blockIdx.x == 0 && tid == 0) {
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        printf("sweep_quad: T=%d nt=%d cycles/step=%.1f ns/step=%.1f clock=%.2f GHz\n", T, nt,
               (double)(clk1 - clk0) / max(T, 1), (double)(rt1 - rt0) * 10.0 / max(T, 1),
               (double)(clk1 - clk0) / ((double)(rt1 - rt0) * 10.0));
    }

    // ---- forward: total score over final arcs (two block reductions; once per kernel)
    if (dir == 0) {
        const int NF = I[WF_I_NF];
        const int32_t* fsrc = blob + I[WF_I_FIN_SRC];
        const float* fw = (const float*)(blob + I[WF_I_FIN_W]);
        const int32_t* farc = blob + I[WF_I_FIN_ARC];
        float m = -INFINITY;
        int barc = 0x7fffffff;
        for (int i = tid; i < NF; i += nt) {
            const float f = fw[i];
            const float v = cur[fsrc[i]] + ((f == -INFINITY) ? -kBig : f * escale);
            if (v > m || (v == m && farc[i] < barc)) { m = v; barc = farc[i]; }
        }
        float M = wave_max(m);
        if (lane == 0) red[wave] = M;
        __syncthreads();
        M = red[0];
        for (int i = 1; i < nw; ++i) M = fmaxf(M, red[i]);
        __syncthreads();
        if (TROP) {
            const int b = wave_min_i(m == M ? barc : 0x7fffffff);
            if (lane == 0) red_arc[wave] = b;
        } else {
            float sm = 0.f;
            for (int i = tid; i < NF; i += nt) {
                const float f = fw[i];
                sm += __builtin_amdgcn_exp2f(cur[fsrc[i]] + ((f == -INFINITY) ? -kBig : f * escale) - M);
            }
            sm = wave_sum(sm);
            if (lane == 0) red[wave] = sm;
        }
        __syncthreads();
        if (tid == 0) {
            const bool dead = !(M > kDead);   // also true for NF == 0 (M = -inf)
            double out;
            if (TROP) {
                int b = red_arc[0];
                for (int i = 1; i < nw; ++i) b = min(b, red_arc[i]);
                out = dead ? (double)-INFINITY : (double)M;
                if (MODE == M_VITERBI) p.fin_arc[u] = dead ? -1 : b;
            } else {
                float st = red[0];
                for (int i = 1; i < nw; ++i) st += red[i];
                const double tot2 = coff + (double)M + (double)__builtin_amdgcn_logf(st);
                out = dead ? (double)-INFINITY : tot2 * kLn2;
                if (MODE == M_FB && p.tot2) p.tot2[u] = dead ? (double)-INFINITY : tot2;
            }
            if (p.tot_double) ((double*)p.tot)[u] = out;
            else ((float*)p.tot)[u] = (float)out;
            if (MODE == M_FB && p.totd) p.totd[u] = out;
        }
    }
}

// ------------------------------------------------------------------------------------------
// emission gather (compact graphs)
// ------------------------------------------------------------------------------------------
struct GatherArgs {
    const int32_t* blob;
    const int64_t* utt;
    const float* lp;
    int64_t strideB, strideT;
    float* E;
    int B;
};

// grid.x = row blocks, grid.y = utterance; each thread one (row, column) element
__global__ void __launch_bounds__(256) gather_kernel(GatherArgs p) {
    const int u = blockIdx.y;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    if (I[WF_I_DIRECT]) return;
    const int K = I[WF_I_K], T = (int)U[WF_U_T];
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const float* src = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
    float* dst = p.E + U[WF_U_E];
    const int64_t n = (int64_t)T * K;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(j / K), k = (int)(j - (int64_t)t * K);
        dst[j] = src[(int64_t)t * p.strideT + labels[k]];
    }
}

// ------------------------------------------------------------------------------------------
// posterior / gradient rows
// ------------------------------------------------------------------------------------------
struct PostArgs {
    const int32_t* blob;
    const int64_t* utt;
    const int32_t* seq2utt;
    const float* gscale;  // [B]
    const float* lp;
    int64_t strideB, strideT;
    const float* E;
    const double* alpha;
    const double* beta;
    const double* totd;
    const double* tot2;   // tot in log2 units (quad sweep), valid when ab_float
    float* grad;
    int Tmax, V, smax;
    int ecap;       // LDS floats reserved for the staged emission row (compact graphs), 0 = none
    int acap;       // LDS floats reserved for per-arc posteriors (arc-parallel mode), 0 = column loop
    int skip_zero;  // rows were zero-filled by zero_kernel already: only scatter the K values
    int fr;         // frames per workgroup
    int ab_float;   // alpha/beta rows are renormalised fp32 (log2 units) + per-group fp64 offsets (quad sweep)
    const double* aoff;
    const double* boff;
};

// Quad-sweep rows are stored relative to a per-group offset: alpha row t (t >= 1) was produced by
// forward step t-1, beta row t+1 (< T) by backward step T-2-t; rows 0 / T are the unscaled initial
// vectors.  Returns off_a(t) + off_b(t+1) in log2 units.
__device__ __forceinline__ double ab_offsets(const double* aoffs, const double* boffs, int t, int T) {
    const double ga = t > 0 ? aoffs[(t - 1) / kPD] : 0.0;
    const double gb = (t + 1 < T) ? boffs[(T - 2 - t) / kPD] : 0.0;
    return ga + gb;
}

// zero n floats starting at p (4-byte aligned) with 16-byte stores where alignment allows
__device__ __forceinline__ void zero_span(float* p, int n, int tid, int nt) {
    int head = (int)(((16u - (unsigned)((uintptr_t)p & 15u)) & 15u) >> 2);
    head = min(head, n);
    if (tid < head) p[tid] = 0.f;
    float4* q = (float4*)(p + head);
    const int n4 = (n - head) >> 2;
    for (int i = tid; i < n4; i += nt) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int t0 = head + n4 * 4;
    if (tid < n - t0) p[t0 + tid] = 0.f;
}

// Zero-fill of the gradient box, launched on a side stream so that it overlaps the latency-bound
// sweeps (which leave most of the chip idle).  grid = (Tmax, N).
struct ZeroArgs {
    float* grad;
    int64_t strideB, strideT;
    int V;
};
// contiguous box: grid-stride 16-byte non-temporal stores (streaming: do not displace the gathered
// emissions and the graph from L2 while the sweeps are running)
__global__ void __launch_bounds__(256) zero_flat_kernel(float4* p, size_t n4, float* tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        __builtin_nontemporal_store(z, (v4f*)p + i);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

__global__ void __launch_bounds__(256) zero_kernel(ZeroArgs p) {
    zero_span(p.grad + (int64_t)blockIdx.y * p.strideB + (int64_t)blockIdx.x * p.strideT, p.V, threadIdx.x, blockDim.x);
}

// p.fr consecutive frames are handled by one workgroup.  Measured on cfg3: fr = 1 with 64-lane
// workgroups (16 000 independent waves, 2 residency rounds) beats fr = 8 (each frame costs a chain of
// dependent L2 loads that only parallelism across workgroups hides).

// SMALL: S <= blockDim and the staged emission row fits one element per lane, so the next frame's
// alpha / beta / emission values are prefetched into registers while the current frame is reduced.
template <bool SMALL>
__global__ void __launch_bounds__(256) post_kernel(PostArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double wmax[4];
    float* la = (float*)smem;   // alpha_t[s] - a_ref
    float* lb = la + p.smax;    // beta_{t+1}[s] - (tot - a_ref)
    float* le = lb + p.smax;    // emission row (compact graphs)
    float* lpost = le + p.ecap; // per-arc posteriors (arc-parallel mode)
    const int tid = threadIdx.x, nt = blockDim.x;
    const int f0 = blockIdx.x * p.fr, q = blockIdx.y;
    const int f1 = min(f0 + p.fr, p.Tmax);
    float* gbase = p.grad + (int64_t)q * p.strideB;
    const int V = p.V;
    const int u = p.seq2utt[q];
    const bool has = u >= 0;
    const int64_t* U = p.utt + (int64_t)(has ? u : 0) * WF_U_WORDS;
    const int start = (int)U[WF_U_START], T = (int)U[WF_U_T];
    const double tot = has ? p.totd[u] : 0.0;
    if (!has || tot == (double)-INFINITY || f0 >= start + T || f1 <= start) {
        if (!p.skip_zero)
            for (int f = f0; f < f1; ++f) zero_span(gbase + (int64_t)f * p.strideT, V, tid, nt);
        return;
    }
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = I[WF_I_S], A = I[WF_I_A], K = I[WF_I_K], direct = I[WF_I_DIRECT];
    const double* abase = p.alpha + U[WF_U_ALPHA];
    const double* bbase = p.beta + U[WF_U_BETA];
    const float* abase_f = (const float*)p.alpha + U[WF_U_ALPHA];
    const float* bbase_f = (const float*)p.beta + U[WF_U_BETA];
    const bool abf = p.ab_float != 0;
    // posterior = exp2(xs * (alpha + beta) + es * (w + e)): quad-sweep rows are already in log2 units
    const float xs = abf ? 1.f : kLog2e, es = kLog2e;
    const double* aoffs = p.aoff + U[WF_U_OFFS];
    const double* boffs = p.boff + U[WF_U_OFFS];
    auto ldA = [&](int64_t i) { return abf ? (double)abase_f[i] : abase[i]; };
    auto ldB = [&](int64_t i) { return abf ? (double)bbase_f[i] : bbase[i]; };
    const float* ebase = direct ? p.lp + (int64_t)q * p.strideB + (int64_t)start * p.strideT : p.E + U[WF_U_E];
    const int64_t epitch = direct ? p.strideT : K;
    const bool stage_e = !direct && K <= p.ecap;
    const int32_t* loff = p.blob + I[WF_I_LAB_OFF];
    const int32_t* lsrc = p.blob + I[WF_I_LAB_SRC];
    const int32_t* ldst = p.blob + I[WF_I_LAB_DST];
    const int32_t* lcol = p.blob + I[WF_I_LAB_COL];
    const float* lw = (const float*)(p.blob + I[WF_I_LAB_W]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const float gs = p.gscale ? p.gscale[u] : 1.f;
    const bool arcpar = A <= p.acap && I[WF_I_MAXLAB_DEG] > 8;
    const int nwv = nt >> 6;

    // SMALL: register prefetch of frame t's values (one alpha, one beta, one emission per lane)
    double pa = (double)-INFINITY, pb = (double)-INFINITY;
    float pe = 0.f;
    auto prefetch = [&](int t) {
        if (!SMALL || t < 0 || t >= T) return;
        if (tid < S) { pa = ldA((int64_t)t * S + tid); pb = ldB((int64_t)(t + 1) * S + tid); }
        if (stage_e && tid < K) pe = ebase[(int64_t)t * epitch + tid];
    };
    prefetch(max(f0 - start, 0));

    for (int f = f0; f < f1; ++f) {
        float* grow = gbase + (int64_t)f * p.strideT;
        const int t = f - start;
        if (t < 0 || t >= T) {   // uniform
            if (!p.skip_zero) zero_span(grow, V, tid, nt);
            continue;
        }
        const int64_t arow = (int64_t)t * S, brow = (int64_t)(t + 1) * S;
        const float* erow = ebase + (int64_t)t * epitch;
        // reference point: the best alpha of this frame, so that both LDS vectors hold small numbers
        double mx = (double)-INFINITY;
        double ca = pa, cb = pb;
        if (SMALL) {
            mx = ca;
            if (stage_e && tid < K) le[tid] = pe;
            prefetch(t + 1 < T && f + 1 < f1 ? t + 1 : -1);
        } else {
            if (!abf)
                for (int s = tid; s < S; s += nt) mx = fmax(mx, ldA(arow + s));
            if (stage_e)
                for (int k = tid; k < K; k += nt) le[k] = erow[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, shfl_xor(mx, o));
        if ((tid & 63) == 0) wmax[tid >> 6] = mx;
        if (!p.skip_zero) {
            if (!direct) zero_span(grow, V, tid, nt);
            else if (K < V) zero_span(grow + K, V - K, tid, nt);
        }
        __syncthreads();
        double a_ref = wmax[0];
        for (int x = 1; x < nwv; ++x) a_ref = fmax(a_ref, wmax[x]);
        double b_ref = tot - a_ref;
        if (abf) {   // rows are already relative to their offsets: fold (off_a + off_b - tot) into alpha
            a_ref = -(ab_offsets(aoffs, boffs, t, T) - p.tot2[u]);
            b_ref = 0.0;
        }
        if (SMALL) {
            if (tid < S) { la[tid] = (float)(ca - a_ref); lb[tid] = (float)(cb - b_ref); }
        } else {
            for (int s = tid; s < S; s += nt) {
                la[s] = (float)(ldA(arow + s) - a_ref);
                lb[s] = (float)(ldB(brow + s) - b_ref);
            }
        }
        __syncthreads();
        if (arcpar) {
            // arc-parallel: a column shared by many arcs (the blank of a CTC numerator sits on ~40% of
            // the arcs) would serialise one lane; compute one posterior per lane, then add per column in
            // arc order (fixed order => bit-stable, no atomics)
            for (int a = tid; a < A; a += nt) {
                const int c = lcol[a];
                const float e = stage_e ? le[c] : erow[c];
                lpost[a] = __builtin_amdgcn_exp2f((la[lsrc[a]] + lb[ldst[a]]) * xs + (lw[a] + e) * es);
            }
            __syncthreads();
            for (int k = tid; k < K; k += nt) {
                const int a0 = loff[k], a1 = loff[k + 1];
                float acc = 0.f;
                for (int a = a0; a < a1; ++a) acc += lpost[a];
                grow[direct ? k : labels[k]] = gs * acc;
            }
        } else {
            for (int k = tid; k < K; k += nt) {
                const int a0 = loff[k], a1 = loff[k + 1];
                const float e = stage_e ? le[k] : erow[k];
                float acc = 0.f;
                for (int a = a0; a < a1; ++a) {
                    acc += __builtin_amdgcn_exp2f((la[lsrc[a]] + lb[ldst[a]]) * xs + (lw[a] + e) * es);   // 0 for dead states
                }
                grow[direct ? k : labels[k]] = gs * acc;
            }
        }
        __syncthreads();   // la / lb / le / lpost are reused by the next frame
    }
}

// One wavefront per gradient row for small compact graphs (numerators): S, A, K <= 64 * kPS.
// Everything the row needs is fetched in two batches of independent loads (header, then alpha /
// beta / emissions / arc descriptors together), so a wave pays ~3 memory latencies in all and the
// 16 000 rows of a batch run as 16 000 independent waves.  Rows are pre-zeroed (skip_zero).
constexpr int kPS = 8;

__global__ void __launch_bounds__(64) post_small_kernel(PostArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* la = (float*)smem;
    float* lb = la + p.smax;
    float* le = lb + p.smax;
    float* lpost = le + p.ecap;
    const int lane = threadIdx.x;
    const int f = blockIdx.x, q = blockIdx.y;
    const int u = p.seq2utt[q];
    if (u < 0) return;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int start = (int)U[WF_U_START], T = (int)U[WF_U_T];
    const int t = f - start;
    const double tot = p.totd[u];
    if (t < 0 || t >= T || tot == (double)-INFINITY) return;
    const int S = (int)U[WF_U_S], A = (int)U[WF_U_A], K = (int)U[WF_U_K];
    const float* arow = (const float*)p.alpha + U[WF_U_ALPHA] + (int64_t)t * S;
    const float* brow = (const float*)p.beta + U[WF_U_BETA] + (int64_t)(t + 1) * S;
    const float* erow = p.E + U[WF_U_E] + (int64_t)t * K;
    const float cst = (float)(ab_offsets(p.aoff + U[WF_U_OFFS], p.boff + U[WF_U_OFFS], t, T) - p.tot2[u]);
    const int32_t* loff = p.blob + U[WF_U_LAB_OFF];
    const int32_t* lsrc = p.blob + U[WF_U_LAB_SRC];
    const int32_t* ldst = p.blob + U[WF_U_LAB_DST];
    const int32_t* lcol = p.blob + U[WF_U_LAB_COL];
    const float* lw = (const float*)(p.blob + U[WF_U_LAB_W]);
    const int32_t* labels = p.blob + U[WF_U_LABELS];
    const float gs = p.gscale ? p.gscale[u] : 1.f;
    float* grow = p.grad + (int64_t)q * p.strideB + (int64_t)f * p.strideT;

    // ---- batch 2: all independent loads (indices clamped so that no load is conditional)
    float av[kPS], bv[kPS];
    float ev[kPS];
    int asrc[kPS], adst[kPS], acol[kPS];
    float aw[kPS];
    int k0[2], k1[2], klab[2];
#pragma unroll
    for (int i = 0; i < kPS; ++i) {
        const int s = min(lane + i * 64, S - 1);
        av[i] = arow[s];
        bv[i] = brow[s];
        ev[i] = erow[min(lane + i * 64, K - 1)];
        const int a = min(lane + i * 64, A - 1);
        asrc[i] = lsrc[a]; adst[i] = ldst[a]; acol[i] = lcol[a]; aw[i] = lw[a];
    }
#pragma unroll
    for (int i = 0; i < kPS; ++i) {
        const int s = lane + i * 64;
        if (s < S) { la[s] = av[i] + cst; lb[s] = bv[i]; }
        if (s < K) le[s] = ev[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPS; ++i) {
        const int a = lane + i * 64;
        if (a < A) lpost[a] = __builtin_amdgcn_exp2f(la[asrc[i]] + (aw[i] + le[acol[i]]) * kLog2e + lb[adst[i]]);
    }
    __syncthreads();
    // per-column sums in arc order (bit-stable)
    for (int k = lane; k < K; k += 64) {
        const int a0 = loff[k], a1 = loff[k + 1];
        const int lab = labels[k];
        float acc = 0.f;
        for (int a = a0; a < a1; ++a) acc += lpost[a];
        grow[lab] = gs * acc;
    }
}

// ------------------------------------------------------------------------------------------
// Viterbi back-trace: one lane per utterance
// ------------------------------------------------------------------------------------------
struct TraceArgs {
    const int32_t* blob;
    const int64_t* utt;
    const int32_t* bp;
    const int32_t* fin_arc;
    const int32_t* arcs_src;  // unused (src looked up through the blob)
    int32_t* best_arcs;
    int B, Tmax;
};

__global__ void backtrace_kernel(TraceArgs p) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= p.B) return;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = I[WF_I_S], T = (int)U[WF_U_T], base = I[WF_I_ARC_BASE];
    int32_t* out = p.best_arcs + (int64_t)u * (p.Tmax + 1);
    for (int i = 0; i <= p.Tmax; ++i) out[i] = -1;
    const int fa = p.fin_arc[u];
    if (fa < 0) return;
    const int32_t* asrc = p.blob + I[WF_I_ARC_SRC];  // src state of every original arc (local id)
    out[T] = base + fa;
    int s = asrc[fa];
    const int32_t* bp = p.bp + U[WF_U_ALPHA];
    for (int t = T - 1; t >= 0; --t) {
        const int a = bp[(int64_t)t * S + s];
        out[t] = base + a;
        s = asrc[a];
    }
}

// ------------------------------------------------------------------------------------------
// host side: workspace planning and launches
// ------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Plan {
    int B = 0;
    size_t off_utt = 0, off_seq2utt = 0, off_gscale = 0, off_totd = 0, off_tot2 = 0, off_aoff = 0, off_boff = 0, off_E = 0, off_alpha = 0, off_beta = 0,
           off_bp = 0, off_finarc = 0, total = 0;
    std::vector<int64_t> utt;  // B * WF_U_WORDS
    int smax = 0, kmax_compact = 0, kmax = 0, amax = 0, hmax = 0, maxlabdeg = 0;
    bool any_compact = false, all_compact = true, any_heavy = false, arcs_lds_ok = true;
    // register-resident sweep eligibility (per direction)
    bool reg_ok_in = true, reg_ok_out = true;
    int over_deg_in = 0, over_deg_out = 0;
    int64_t n_E = 0, n_ab = 0;
    int Tlong = 0;
};

int check_blob(const void* graph_host) {
    if (!graph_host) return wf_fail(WF_EINVAL, "null prepared graph");
    const int32_t* H = (const int32_t*)graph_host;
    if (H[WF_H_MAGIC] != WF_BLOB_MAGIC) return wf_fail(WF_EINVAL, "prepared graph: bad magic");
    return WF_OK;
}

// mode: 0 = forward-backward (alpha+beta arenas), 1 = tot only, 2 = viterbi (bp arena)
int make_plan(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int B, int V,
              int num_seqs, size_t acc_bytes, int mode, Plan* P) {
    int rc = check_blob(graph_host);
    if (rc) return rc;
    if (B < 1 || !seg || V < 1) return wf_fail(WF_EINVAL, "bad B/seg/V");
    const int32_t* H = (const int32_t*)graph_host;
    const int G = H[WF_H_NUM_FSAS];
    if (!utt2fsa && G != B) return wf_fail(WF_EINVAL, "graph has %d FSAs but %d segments (pass utt2fsa to share graphs)", G, B);
    P->B = B;
    P->utt.assign((size_t)B * WF_U_WORDS, 0);
    int64_t nE = 0, nab = 0, row0 = 0;
    for (int i = 0; i < B; ++i) {
        const int g = utt2fsa ? utt2fsa[i] : i;
        if (g < 0 || g >= G) return wf_fail(WF_ERANGE, "utt2fsa[%d] = %d out of range", i, g);
        const int32_t* I = H + WF_HDR_WORDS + (size_t)g * WF_INFO_WORDS;
        if (I[WF_I_MAX_LABEL] >= V) return wf_fail(WF_ERANGE, "fsa %d uses label %d but log_probs has V=%d columns", g, I[WF_I_MAX_LABEL], V);
        const int b = seg[3 * i], st = seg[3 * i + 1], T = seg[3 * i + 2];
        if (b < 0 || st < 0 || T < 0 || (num_seqs > 0 && b >= num_seqs)) return wf_fail(WF_ERANGE, "segment %d = (%d,%d,%d) out of range", i, b, st, T);
        int64_t* U = &P->utt[(size_t)i * WF_U_WORDS];
        U[WF_U_FSA] = g; U[WF_U_SEQ] = b; U[WF_U_START] = st; U[WF_U_T] = T;
        U[WF_U_ALPHA] = nab; U[WF_U_BETA] = nab; U[WF_U_E] = nE; U[WF_U_ROW0] = row0;
        U[WF_U_OFFS] = row0 + i;   // sum of (T+1) over the utterances before this one
        U[WF_U_S] = I[WF_I_S]; U[WF_U_A] = I[WF_I_A]; U[WF_U_K] = I[WF_I_K]; U[WF_U_DIRECT] = I[WF_I_DIRECT];
        U[WF_U_MAXLABDEG] = I[WF_I_MAXLAB_DEG];
        U[WF_U_LAB_OFF] = I[WF_I_LAB_OFF]; U[WF_U_LAB_SRC] = I[WF_I_LAB_SRC]; U[WF_U_LAB_DST] = I[WF_I_LAB_DST];
        U[WF_U_LAB_COL] = I[WF_I_LAB_COL]; U[WF_U_LAB_W] = I[WF_I_LAB_W]; U[WF_U_LABELS] = I[WF_I_LABELS];
        nab += (int64_t)(T + 1) * I[WF_I_S];
        if (I[WF_I_DIRECT]) P->all_compact = false;
        if (!I[WF_I_DIRECT]) { nE += (int64_t)T * I[WF_I_K]; P->any_compact = true; P->kmax_compact = std::max(P->kmax_compact, (int)I[WF_I_K]); }
        P->kmax = std::max(P->kmax, (int)I[WF_I_K]);
        P->smax = std::max(P->smax, (int)I[WF_I_S]);
        P->amax = std::max(P->amax, (int)I[WF_I_A]);
        P->maxlabdeg = std::max(P->maxlabdeg, (int)I[WF_I_MAXLAB_DEG]);
        P->Tlong = std::max(P->Tlong, T);
        if (I[WF_I_NHEAVY_IN] || I[WF_I_NHEAVY_OUT]) P->any_heavy = true;
        P->hmax = std::max(P->hmax, std::max((int)I[WF_I_NHEAVY_IN], (int)I[WF_I_NHEAVY_OUT]));
        if (I[WF_I_S] > 65535 || I[WF_I_K] > 65535) P->arcs_lds_ok = false;
        if (I[WF_I_NOVER_IN] > 1) P->reg_ok_in = false;
        if (I[WF_I_NOVER_OUT] > 1) P->reg_ok_out = false;
        if (I[WF_I_NOVER_IN] == 1) P->over_deg_in = std::max(P->over_deg_in, (int)I[WF_I_MAXIN]);
        if (I[WF_I_NOVER_OUT] == 1) P->over_deg_out = std::max(P->over_deg_out, (int)I[WF_I_MAXOUT]);
        row0 += T;
    }
    P->n_E = nE; P->n_ab = nab;
    size_t o = 0;
    P->off_utt = o; o = align_up(o + sizeof(int64_t) * WF_U_WORDS * B, 256);
    P->off_seq2utt = o; o = align_up(o + sizeof(int32_t) * std::max(num_seqs, 1), 256);
    P->off_gscale = o; o = align_up(o + sizeof(float) * B, 256);
    P->off_finarc = o; o = align_up(o + sizeof(int32_t) * B, 256);
    P->off_totd = o; o = align_up(o + sizeof(double) * B, 256);
    P->off_tot2 = o; o = align_up(o + sizeof(double) * B, 256);
    P->off_E = o; o = align_up(o + sizeof(float) * (size_t)nE, 256);
    if (mode == 0) {
        P->off_alpha = o; o = align_up(o + acc_bytes * (size_t)nab, 256);
        P->off_beta = o; o = align_up(o + acc_bytes * (size_t)nab, 256);
        P->off_aoff = o; o = align_up(o + sizeof(double) * (size_t)(row0 + B), 256);
        P->off_boff = o; o = align_up(o + sizeof(double) * (size_t)(row0 + B), 256);
    } else if (mode == 2) {
        P->off_bp = o; o = align_up(o + sizeof(int32_t) * (size_t)nab, 256);
    }
    P->total = o;
    return WF_OK;
}

// optional per-kernel timing (bench.py's roofline leg): HIP events recorded on the caller's stream
// around each kernel of the last wf_fb_logsum / wf_fb_backward call of this thread.
struct Timing {
    bool on = false;
    hipEvent_t ev[7] = {};   // t0 | gather | sweep | (post t0) | post | zero t0 | zero t1
    bool made = false;
    bool have_fwd = false, have_post = false, have_zero = false;
};
thread_local Timing g_tm;
int tm_record(int i, hipStream_t st) {
    if (!g_tm.on) return WF_OK;
    if (!g_tm.made) {
        for (auto& e : g_tm.ev) HIP_TRY(hipEventCreate(&e));
        g_tm.made = true;
    }
    HIP_TRY(hipEventRecord(g_tm.ev[i], st));
    return WF_OK;
}

// side stream for work that can overlap the latency-bound sweeps (one per device, per host thread)
struct Side {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
thread_local Side g_side[16];
int side_get(Side** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return wf_fail(WF_EUNSUPPORTED, "device index %d", dev);
    Side& s = g_side[dev];
    if (!s.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
    }
    *out = &s;
    return WF_OK;
}

struct SweepCfg {
    int threads, chunk, kpad, smax, arcs_words, hcap;
    bool arcs_lds, estage;
    size_t lds;
};

SweepCfg sweep_config(const Plan& P, size_t acc_bytes) {
    SweepCfg c{};
    c.threads = std::min(kMaxThreads, std::max(64, (int)align_up(P.smax, 64)));
    c.smax = (int)align_up(P.smax, 4);
    c.hcap = std::max(1, std::min(kMaxHeavy, P.hmax));
    size_t base = 2 * (size_t)c.smax * acc_bytes + (size_t)c.hcap * 16 * 2 * acc_bytes + (size_t)c.hcap * 16 * 4;
    // emission staging
    const int kmax = std::max(P.kmax, 1);
    c.kpad = kmax;
    c.estage = kmax <= kPF * c.threads;
    c.chunk = 0;
    if (c.estage) {
        c.chunk = std::min(kMaxChunk, (kPF * c.threads) / kmax);
        while (c.chunk > 1 && base + 2 * (size_t)c.chunk * c.kpad * 4 > kLdsBudget / 2) c.chunk /= 2;
        if (base + 2 * (size_t)c.chunk * c.kpad * 4 > kLdsBudget) { c.estage = false; c.chunk = 0; }
    }
    size_t e_bytes = c.estage ? 2 * (size_t)c.chunk * c.kpad * 4 : 0;
    c.arcs_words = (int)(align_up(P.smax + 1, 4) + 2 * align_up(P.amax, 4));
    c.arcs_lds = P.arcs_lds_ok && base + e_bytes + (size_t)c.arcs_words * 4 <= kLdsBudget;
    if (!c.arcs_lds) c.arcs_words = 0;
    c.lds = base + e_bytes + (size_t)c.arcs_words * 4;
    return c;
}

template <typename Acc, typename Sum, int MODE>
int launch_sweep(const SweepCfg& c, const SweepArgs& a, int B, hipStream_t st) {
    const int ndir = (MODE == M_FB) ? 2 : 1;
    dim3 grid(B * ndir), block(c.threads);
    if (c.lds > kLdsBudget)
        return wf_fail(WF_EUNSUPPORTED, "graph too large for the LDS-resident sweep: %d states need %zu B of LDS (> %zu)",
                       c.smax, c.lds, kLdsBudget);
#define WF_LAUNCH(AL, ES)                                                                            \
    do {                                                                                             \
        auto k = sweep_kernel<Acc, Sum, MODE, AL, ES>;                                                    \
        if (c.lds > 48 * 1024)                                                                       \
            HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds)); \
        hipLaunchKernelGGL(k, grid, block, c.lds, st, a);                                            \
    } while (0)
    if (c.arcs_lds && c.estage) WF_LAUNCH(true, true);
    else if (c.arcs_lds) WF_LAUNCH(true, false);
    else if (c.estage) WF_LAUNCH(false, true);
    else WF_LAUNCH(false, false);
#undef WF_LAUNCH
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// ---- register-resident quad sweep: eligibility and launch
struct RegCfg {
    bool ok = false;
    int threads = 0, R = 1, HD = 0, smax = 0;
    size_t lds = 0;
};

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

RegCfg reg_config(const Plan& P, bool both_dirs) {
    RegCfg c;
    static const int no_reg = env_int("WF_NO_REG_SWEEP", 0);
    if (no_reg) return c;
    if (!P.arcs_lds_ok || P.smax > kMaxThreads || !P.reg_ok_in || (both_dirs && !P.reg_ok_out)) return c;
    // four lanes per state; R states per quad once 4*S exceeds the workgroup
    c.R = P.smax * 4 <= kMaxThreads ? 1 : (P.smax * 2 <= kMaxThreads ? 2 : 4);
    static const int force_r = env_int("WF_QUAD_R", 0);
    if (force_r == 2 || force_r == 4) c.R = std::max(c.R, force_r);
    c.threads = std::max(64, (int)align_up(4 * (size_t)((P.smax + c.R - 1) / c.R), 64));
    const int od = std::max(P.over_deg_in, both_dirs ? P.over_deg_out : 0);
    c.HD = (od + c.threads - 1) / c.threads;
    if (c.HD > 2 || (c.R == 4 && c.HD > 0)) return c;   // R = 4 with a cooperative state would spill VGPRs
    c.smax = (int)align_up(P.smax, 4);
    c.lds = 2 * (size_t)c.smax * 4 + 32 * 4 + 64 + 64;
    c.ok = true;
    return c;
}

template <int MODE>
int launch_sweep_quad(const RegCfg& c, SweepArgs a, int B, hipStream_t st) {
    const int ndir = (MODE == M_FB) ? 2 : 1;
    dim3 grid(B * ndir), block(c.threads);
    a.smax = c.smax;
    static const int prof = env_int("WF_SWEEP_PROFILE", 0);
    a.ablate = prof ? 8 : 0;
#define WF_LAUNCH_QUAD(RR, HH) hipLaunchKernelGGL((sweep_quad_kernel<MODE, RR, HH>), grid, block, c.lds, st, a)
#define WF_LAUNCH_QUAD_R(RR)                       \
    do {                                           \
        if (c.HD == 0) WF_LAUNCH_QUAD(RR, 0);      \
        else if (c.HD == 1) WF_LAUNCH_QUAD(RR, 1); \
        else WF_LAUNCH_QUAD(RR, 2);                \
    } while (0)
    if (c.R == 1) WF_LAUNCH_QUAD_R(1);
    else if (c.R == 2) WF_LAUNCH_QUAD_R(2);
    else WF_LAUNCH_QUAD_R(4);
#undef WF_LAUNCH_QUAD_R
#undef WF_LAUNCH_QUAD
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

}  // namespace

extern "C" int wf_fb_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg,
                                     int32_t B, int32_t V, uint32_t flags, int32_t need_grad, size_t* bytes) {
    if (!bytes) return wf_fail(WF_EINVAL, "null bytes");
    Plan P;
    const size_t ab = 8;  // log semiring: alpha/beta are always kept in fp64
    int maxseq = 0;
    if (seg) for (int i = 0; i < B; ++i) maxseq = std::max(maxseq, seg[3 * i] + 1);
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, maxseq, ab, (need_grad || (flags & WF_KEEP_AB)) ? 0 : 1, &P);
    if (rc) return rc;
    *bytes = P.total + 256;
    return WF_OK;
}

extern "C" int wf_viterbi_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg,
                                          int32_t B, int32_t V, uint32_t flags, size_t* bytes) {
    if (!bytes) return wf_fail(WF_EINVAL, "null bytes");
    Plan P;
    const size_t ab = (flags & WF_DOUBLE_ACC) ? 8 : 4;
    int maxseq = 0;
    if (seg) for (int i = 0; i < B; ++i) maxseq = std::max(maxseq, seg[3 * i] + 1);
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, maxseq, ab, 2, &P);
    if (rc) return rc;
    *bytes = P.total + 256;
    return WF_OK;
}

namespace {
// copies the per-call tables into the workspace and launches the gather when needed
int upload_tables(const Plan& P, unsigned char* ws, const int32_t* seq2utt, int num_seqs, hipStream_t st) {
    HIP_TRY(hipMemcpyAsync(ws + P.off_utt, P.utt.data(), sizeof(int64_t) * P.utt.size(), hipMemcpyHostToDevice, st));
    if (seq2utt) HIP_TRY(hipMemcpyAsync(ws + P.off_seq2utt, seq2utt, sizeof(int32_t) * num_seqs, hipMemcpyHostToDevice, st));
    return WF_OK;
}

int build_seq2utt(const int32_t* seg, int B, int num_seqs, std::vector<int32_t>* m) {
    m->assign(num_seqs, -1);
    for (int i = 0; i < B; ++i) {
        if ((*m)[seg[3 * i]] >= 0) return wf_fail(WF_EUNSUPPORTED, "sequence %d appears in more than one segment", seg[3 * i]);
        (*m)[seg[3 * i]] = i;
    }
    return WF_OK;
}

int launch_gather(const Plan& P, const void* graph_dev, unsigned char* ws, const float* lp,
                  int64_t strideB, int64_t strideT, hipStream_t st) {
    if (!P.any_compact) return WF_OK;
    GatherArgs g{(const int32_t*)graph_dev, (const int64_t*)(ws + P.off_utt), lp, strideB, strideT,
                 (float*)(ws + P.off_E), P.B};
    int64_t per = (int64_t)P.Tlong * P.kmax_compact;
    int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (per + 255) / 256), 4096);
    hipLaunchKernelGGL(gather_kernel, dim3(gx, P.B), dim3(256), 0, st, g);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

int launch_post(const Plan& P, const void* graph_dev, unsigned char* ws, const float* lp,
                int64_t strideB, int64_t strideT, int num_seqs, int Tmax, int V, const float* gscale_dev,
                float* grad, bool skip_zero, bool ab_float, hipStream_t st) {
    PostArgs q{};
    q.ab_float = ab_float ? 1 : 0;
    q.aoff = (const double*)(ws + P.off_aoff); q.boff = (const double*)(ws + P.off_boff);
    q.blob = (const int32_t*)graph_dev; q.utt = (const int64_t*)(ws + P.off_utt);
    q.seq2utt = (const int32_t*)(ws + P.off_seq2utt);
    q.gscale = gscale_dev;
    q.lp = lp; q.strideB = strideB; q.strideT = strideT; q.E = (const float*)(ws + P.off_E);
    q.alpha = (const double*)(ws + P.off_alpha); q.beta = (const double*)(ws + P.off_beta);
    q.totd = (const double*)(ws + P.off_totd); q.tot2 = (const double*)(ws + P.off_tot2); q.grad = grad; q.Tmax = Tmax; q.V = V;
    q.smax = (int)align_up(P.smax, 4);
    q.ecap = (P.any_compact && P.kmax_compact <= 8192) ? (int)align_up(P.kmax_compact, 4) : 0;
    q.acap = (P.maxlabdeg > 8 && P.amax <= 8192) ? (int)align_up(P.amax, 4) : 0;
    q.skip_zero = skip_zero ? 1 : 0;
    size_t lds = (2 * (size_t)q.smax + q.ecap + q.acap) * 4;
    if (ab_float && skip_zero && !getenv("WF_NO_POST_SMALL") && P.all_compact && P.amax >= 1 && P.amax <= 64 * kPS && P.smax <= 64 * kPS &&
        P.kmax_compact <= 64 * kPS && q.ecap > 0) {
        q.acap = (int)align_up(P.amax, 4);
        size_t lds2 = (2 * (size_t)q.smax + q.ecap + q.acap) * 4;
        hipLaunchKernelGGL(post_small_kernel, dim3(Tmax, num_seqs), dim3(64), lds2, st, q);
        HIP_TRY(hipGetLastError());
        return WF_OK;
    }
    const bool tiny_rows = skip_zero || V <= 1024;   // a 64-lane workgroup would be slow at zero-filling long rows
    const int threads = (P.amax <= 1024 && P.smax <= 512 && tiny_rows) ? 64 : 256;
    q.fr = 1;
    dim3 grid((Tmax + q.fr - 1) / q.fr, num_seqs), block(threads);
    const bool small = P.smax <= threads && (q.ecap == 0 || P.kmax_compact <= threads);
    if (small) {
        hipLaunchKernelGGL(post_kernel<true>, grid, block, lds, st, q);
    } else {
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)post_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(post_kernel<false>, grid, block, lds, st, q);
    }
    HIP_TRY(hipGetLastError());
    return WF_OK;
}
}  // namespace

extern "C" int wf_fb_logsum(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                            const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                            int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, float output_beam,
                            uint32_t flags, const float* grad_scale, float* tot, float* grad,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !log_probs || !tot || !workspace) return wf_fail(WF_EINVAL, "wf_fb_logsum: null pointer");
    if (output_beam > 0 && !std::isinf(output_beam))
        return wf_fail(WF_EUNSUPPORTED, "wf_fb_logsum: output_beam pruning is not implemented yet; pass <= 0 or inf for the exact sum");
    hipStream_t st = (hipStream_t)stream;
    const bool dbl = flags & WF_DOUBLE_ACC;
    const bool need_ab = grad || (flags & WF_KEEP_AB);
    const size_t ab = 8;
    Plan P;
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, ab, need_ab ? 0 : 1, &P);
    if (rc) return rc;
    for (int i = 0; i < B; ++i)
        if (seg[3 * i + 1] + seg[3 * i + 2] > Tmax) return wf_fail(WF_ERANGE, "segment %d exceeds Tmax=%d", i, Tmax);
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    std::vector<int32_t> seq2utt;
    if (need_ab) { rc = build_seq2utt(seg, B, num_seqs, &seq2utt); if (rc) return rc; }
    rc = upload_tables(P, ws, need_ab ? seq2utt.data() : nullptr, num_seqs, st);
    if (rc) return rc;
    g_tm.have_fwd = g_tm.have_post = g_tm.have_zero = false;
    // Fork: zero-fill the gradient box on the side stream while the sweeps (which occupy only 2B
    // workgroups) run; post_kernel then only scatters the K posteriors per row.
    const bool overlap = grad && P.any_compact && !getenv("WF_NO_OVERLAP");
    Side* side = nullptr;
    if ((rc = tm_record(0, st))) return rc;
    rc = launch_gather(P, graph_dev, ws, log_probs, strideB, strideT, st);
    if (rc) return rc;
    if ((rc = tm_record(1, st))) return rc;
    if (overlap) {
        // forked after the gather: both are HBM-bound and would only slow each other down
        if ((rc = side_get(&side))) return rc;
        HIP_TRY(hipEventRecord(side->fork, st));
        HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
        if ((rc = tm_record(5, side->stream))) return rc;
        if (strideT == V && strideB == (int64_t)Tmax * V) {
            const size_t n = (size_t)num_seqs * Tmax * V;
            if (((uintptr_t)grad & 15) == 0 && !getenv("WF_ZERO_MEMSET")) {
                const size_t n4 = n / 4;
                // Size the grid so that the fill ends about when the sweeps do (~0.28 us per frame step,
                // ~80 GB/s of streaming stores per workgroup): more workgroups than that only steal
                // issue slots and memory-queue space from the latency-bound sweep workgroups
                // (measured on cfg3: 32-64 blocks -> sweep 140 us; 1024 blocks -> 172 us).
                int want = (int)((double)n * 4.0 / (std::max(P.Tlong, 1) * 0.28e-6) / 80e9) + 1;
                want = std::max(32, std::min(want, 2048));
                if (getenv("WF_ZERO_BLOCKS")) want = atoi(getenv("WF_ZERO_BLOCKS"));
                const int blocks = (int)std::min<size_t>((n4 + 255) / 256, (size_t)want);
                hipLaunchKernelGGL(zero_flat_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, side->stream,
                                   (float4*)grad, n4, grad + n4 * 4, (int)(n - n4 * 4));
                HIP_TRY(hipGetLastError());
            } else {
                HIP_TRY(hipMemsetAsync(grad, 0, sizeof(float) * n, side->stream));
            }
        } else {
            ZeroArgs z{grad, strideB, strideT, V};
            hipLaunchKernelGGL(zero_kernel, dim3(Tmax, num_seqs), dim3(256), 0, side->stream, z);
            HIP_TRY(hipGetLastError());
        }
        if ((rc = tm_record(6, side->stream))) return rc;
        g_tm.have_zero = g_tm.on;
        HIP_TRY(hipEventRecord(side->join, side->stream));
    }
    SweepCfg c = sweep_config(P, ab);
    SweepArgs a{};
    a.blob = (const int32_t*)graph_dev; a.utt = (const int64_t*)(ws + P.off_utt);
    a.lp = log_probs; a.strideB = strideB; a.strideT = strideT;
    a.E = (const float*)(ws + P.off_E);
    a.alpha = ws + P.off_alpha; a.beta = ws + P.off_beta; a.tot = tot; a.bp = nullptr; a.fin_arc = nullptr;
    a.totd = (double*)(ws + P.off_totd);
    a.chunk = c.chunk; a.kpad = c.kpad; a.smax = c.smax; a.arcs_lds_words = c.arcs_words; a.hcap = c.hcap; a.tot_double = 0;
    a.aoff = (double*)(ws + P.off_aoff); a.boff = (double*)(ws + P.off_boff);
    a.tot2 = (double*)(ws + P.off_tot2);
    RegCfg rc_ = dbl ? RegCfg() : reg_config(P, need_ab);
    if (rc_.ok) rc = need_ab ? launch_sweep_quad<M_FB>(rc_, a, B, st) : launch_sweep_quad<M_TOT_LOG>(rc_, a, B, st);
    else if (need_ab) rc = dbl ? launch_sweep<double, double, M_FB>(c, a, B, st) : launch_sweep<double, float, M_FB>(c, a, B, st);
    else rc = dbl ? launch_sweep<double, double, M_TOT_LOG>(c, a, B, st) : launch_sweep<double, float, M_TOT_LOG>(c, a, B, st);
    if (rc) return rc;
    if ((rc = tm_record(2, st))) return rc;
    g_tm.have_fwd = g_tm.on;
    if (grad) {
        if (overlap) HIP_TRY(hipStreamWaitEvent(st, side->join, 0));
        if ((rc = tm_record(3, st))) return rc;
        rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, overlap, rc_.ok, st);
        if (rc) return rc;
        if ((rc = tm_record(4, st))) return rc;
        g_tm.have_post = g_tm.on;
    }
    return WF_OK;
}

extern "C" int wf_timing_enable(int on) { g_tm.on = on != 0; g_tm.have_fwd = g_tm.have_post = g_tm.have_zero = false; return WF_OK; }

// ms[0..3] = gather, sweep, post, zero-fill kernel time of the last timed call(s); -1 = not measured.
// Synchronises on the recorded events (a measurement aid, not part of the hot path).
extern "C" int wf_timing_read(float* ms) {
    if (!ms) return wf_fail(WF_EINVAL, "wf_timing_read: null");
    ms[0] = ms[1] = ms[2] = ms[3] = -1.f;
    if (g_tm.have_fwd) {
        HIP_TRY(hipEventSynchronize(g_tm.ev[2]));
        HIP_TRY(hipEventElapsedTime(&ms[0], g_tm.ev[0], g_tm.ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms[1], g_tm.ev[1], g_tm.ev[2]));
    }
    if (g_tm.have_post) {
        HIP_TRY(hipEventSynchronize(g_tm.ev[4]));
        HIP_TRY(hipEventElapsedTime(&ms[2], g_tm.ev[3], g_tm.ev[4]));
    }
    if (g_tm.have_zero) {
        HIP_TRY(hipEventSynchronize(g_tm.ev[6]));
        HIP_TRY(hipEventElapsedTime(&ms[3], g_tm.ev[5], g_tm.ev[6]));
    }
    return WF_OK;
}

extern "C" int wf_fb_backward(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                              const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                              int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, uint32_t flags,
                              const float* grad_scale, const float* tot, float* grad,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !log_probs || !tot || !grad || !workspace) return wf_fail(WF_EINVAL, "wf_fb_backward: null pointer");
    const bool dbl = flags & WF_DOUBLE_ACC;
    Plan P;
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, 0, &P);
    if (rc) return rc;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    (void)dbl; (void)tot;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = tm_record(3, st))) return rc;
    const bool ab_float = !dbl && reg_config(P, true).ok;   // same decision as the forward call made
    rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, false, ab_float, st);
    if (rc) return rc;
    if ((rc = tm_record(4, st))) return rc;
    g_tm.have_post = g_tm.on;
    return WF_OK;
}

extern "C" int wf_tot_scores(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                             const float* log_probs, int64_t strideB, int64_t strideT, int32_t V,
                             const int32_t* seg, int32_t B, uint32_t flags, void* tot,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !log_probs || !tot || !workspace) return wf_fail(WF_EINVAL, "wf_tot_scores: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool dbl = flags & WF_DOUBLE_ACC;
    const size_t ab = (dbl || !(flags & WF_TROPICAL)) ? 8 : 4;
    Plan P;
    int maxseq = 0;
    if (seg) for (int i = 0; i < B; ++i) maxseq = std::max(maxseq, seg[3 * i] + 1);
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, maxseq, ab, 1, &P);
    if (rc) return rc;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    rc = upload_tables(P, ws, nullptr, 0, st);
    if (rc) return rc;
    rc = launch_gather(P, graph_dev, ws, log_probs, strideB, strideT, st);
    if (rc) return rc;
    SweepCfg c = sweep_config(P, ab);
    SweepArgs a{};
    a.blob = (const int32_t*)graph_dev; a.utt = (const int64_t*)(ws + P.off_utt);
    a.lp = log_probs; a.strideB = strideB; a.strideT = strideT; a.E = (const float*)(ws + P.off_E);
    a.tot = tot; a.chunk = c.chunk; a.kpad = c.kpad; a.smax = c.smax; a.arcs_lds_words = c.arcs_words; a.hcap = c.hcap;
    a.tot_double = dbl ? 1 : 0;
    RegCfg rc_ = dbl ? RegCfg() : reg_config(P, false);
    if (rc_.ok) return (flags & WF_TROPICAL) ? launch_sweep_quad<M_TOT_MAX>(rc_, a, B, st)
                                             : launch_sweep_quad<M_TOT_LOG>(rc_, a, B, st);
    if (flags & WF_TROPICAL) return dbl ? launch_sweep<double, double, M_TOT_MAX>(c, a, B, st) : launch_sweep<float, float, M_TOT_MAX>(c, a, B, st);
    return dbl ? launch_sweep<double, double, M_TOT_LOG>(c, a, B, st) : launch_sweep<double, float, M_TOT_LOG>(c, a, B, st);
}

extern "C" int wf_viterbi(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                          const float* log_probs, int64_t strideB, int64_t strideT, int32_t V,
                          const int32_t* seg, int32_t B, int32_t Tmax, uint32_t flags, float* best_score,
                          int32_t* best_arcs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !log_probs || !best_score || !best_arcs || !workspace) return wf_fail(WF_EINVAL, "wf_viterbi: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool dbl = flags & WF_DOUBLE_ACC;
    const size_t ab = dbl ? 8 : 4;
    Plan P;
    int maxseq = 0;
    for (int i = 0; i < B; ++i) {
        maxseq = std::max(maxseq, seg[3 * i] + 1);
        if (seg[3 * i + 2] > Tmax) return wf_fail(WF_ERANGE, "segment %d longer than Tmax=%d", i, Tmax);
    }
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, maxseq, ab, 2, &P);
    if (rc) return rc;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    rc = upload_tables(P, ws, nullptr, 0, st);
    if (rc) return rc;
    rc = launch_gather(P, graph_dev, ws, log_probs, strideB, strideT, st);
    if (rc) return rc;
    SweepCfg c = sweep_config(P, ab);
    SweepArgs a{};
    a.blob = (const int32_t*)graph_dev; a.utt = (const int64_t*)(ws + P.off_utt);
    a.lp = log_probs; a.strideB = strideB; a.strideT = strideT; a.E = (const float*)(ws + P.off_E);
    a.tot = best_score; a.bp = (int32_t*)(ws + P.off_bp); a.fin_arc = (int32_t*)(ws + P.off_finarc);
    a.chunk = c.chunk; a.kpad = c.kpad; a.smax = c.smax; a.arcs_lds_words = c.arcs_words; a.hcap = c.hcap; a.tot_double = 0;
    RegCfg rc_ = dbl ? RegCfg() : reg_config(P, false);
    if (rc_.ok) rc = launch_sweep_quad<M_VITERBI>(rc_, a, B, st);
    else rc = dbl ? launch_sweep<double, double, M_VITERBI>(c, a, B, st) : launch_sweep<float, float, M_VITERBI>(c, a, B, st);
    if (rc) return rc;
    TraceArgs t{a.blob, a.utt, a.bp, a.fin_arc, nullptr, best_arcs, B, Tmax};
    hipLaunchKernelGGL(backtrace_kernel, dim3((B + 63) / 64), dim3(64), 0, st, t);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}
