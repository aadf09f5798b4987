// HIP kernels (gfx950 / CDNA4) and C-ABI compute entry points of libwfdwfst.so.
//
// What runs on the GPU for one wf_fb_logsum() call with a gradient (reference: waterfall/graph.py:61-66
// and its autograd backward, i.e. k2.intersect_dense + get_tot_scores + their backward passes); see
// DESIGN.md section 4 for the measurements behind each choice:
//
//   1. gather_lin_kernel   P[i,t,k] = exp(log_probs[b, start+t, label_k] - c_t) in fp64 (c_t = the row's
//                          maximum) and the raw values E[i,t,k].  Fully parallel.
//   2. sweep_band_kernel   (graphs with a banded state order: every topo o transcript numerator) or
//      sweep_lin_kernel    (other graphs of <= 1024 states): the forward and the backward recursion in
//                          the PROBABILITY domain (fp64, rescaled by powers of two), one workgroup per
//                          (utterance, direction), every alpha_t / beta_t row streamed to the workspace.
//                          Latency bound (T dependent steps); no lattice is ever materialised.
//      zero_flat_kernel    beside them on a side stream: zero-fill of the gradient box (HBM-write bound).
//   3. check_kernel        sum_s alpha_t[s] * beta_t[s] = tot on every row, else the utterance is flagged
//   4. sweep_kernel        (log semiring, fp64 accumulators) re-runs flagged utterances: exits at once otherwise
//   5. post_small_kernel / post_kernel   arc posteriors reduced per label in a fixed order (no float
//                          atomics => bit-stable), K values scattered into the zeroed gradient row.
//
// Graphs too large for 2./3. and WF_DOUBLE_ACC take sweep_kernel directly (frame-synchronous log-semiring
// recursion with alpha double-buffered in LDS, CSR arcs staged in LDS when they fit, emission rows
// prefetched through an LDS ring, block reduction for high fan-in states).
// Forward-only calls and wf_viterbi(): sweep_quad_kernel / sweep_kernel in the log or tropical semiring
// (one back-pointer per lattice state), then backtrace_kernel (one lane per utterance).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include <type_traits>
#include <utility>

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

enum SweepMode { M_FB = 0, M_TOT_LOG = 1, M_TOT_MAX = 2, M_VITERBI = 3, M_FB_MAX = 4 };   // M_FB_MAX: tropical, both directions, rows kept

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
    double* tot2;  // tot in log2 units (linear sweep -> check / post kernels)
    const double* P;      // linear sweep: emission probabilities exp(lp - c_t), fp64, [utt](T x K)
    const double* cmax;   // linear sweep: c_t per (utterance, frame) row, natural log units
    double* ctot2;        // linear sweep: sum_t c_t in log2 units, per utterance
    const int32_t* flag;  // fallback sweep: only utterances with flag[u] != 0 are processed
    // output_beam pruning: tropical alpha^ / beta^ rows of the first pass, the tropical best score per
    // utterance, the beam; the masked log-semiring pass keeps lattice arc (t, a) iff
    // alpha^_t[src] + x + beta^_{t+1}[dst] >= best - beam and leaves thr[u] = best - beam for the post kernel
    const float* mA;
    const float* mB;
    const float* best;
    float* thr;
    float beam;
    int32_t* bp;   // viterbi back-pointers, arena int32: [utt](T x S)
    int32_t* fin_arc;  // viterbi: best final arc per utterance [B]
    int chunk;     // frames per emission chunk (ESTAGE)
    int kpad;      // LDS emission row pitch
    int smax;      // LDS alpha pitch (max states)
    int arcs_lds_words;  // words reserved for the LDS copy of the CSR arrays
    int hcap;      // heavy states reduced per cooperative round (sizes the LDS reduction scratch)
    int tot_double;
    int32_t* flag_clear;  // sweep_hub_kernel (no gather kernel in front of it): the utterance's fallback flag, cleared
    double* cmax_w;       // sweep_hub_kernel: the per-frame normalisers c_t (all 0) for the kernels that read them
    // sweep_band_kernel: workgroups >= nsweep of the launch zero-fill the gradient box (n4 float4 + a tail) instead of
    // sweeping -- the fill that otherwise runs beside the sweep on a side stream, without the fork / join events
    // (two dependent queue packets each).  nsweep == gridDim.x: no fill attached.
    float4* zp;
    size_t zn4;
    float* ztail;
    int zntail;
    int nsweep;
    // sweep_hub_kernel, split ring (rows too long for three whole LDS buffers): pitch of the three buffers of a row's
    // front part and of the two buffers of its back part, in floats; 0: `chunk` whole-row buffers of pitch kpad
    int hub_kpf, hub_kpb;
    // sweep_hub_kernel in segments (fb_logsum_impl: the posterior kernel of the frames both sweeps have passed runs beside
    // the sweeps' next segment): this launch runs steps [hub_steps(T, seg_n0), hub_steps(T, seg_n1)) of seg_den parts; the
    // state that is not in the rows (the hub's fp64 value, the cumulative shift, the next shift) waits in hubsave
    int seg_n0, seg_n1, seg_den;
    double* hubsave;      // [2B][4]
    // banded path with output_beam pruning: keep-mask words [2 directions][bmask_rows][64] (band_mask_kernel -> the MASK
    // instantiations of sweep_band_kernel), one row per (utterance, frame) in WF_U_ROW0 order
    unsigned long long* bmask;
    int64_t bmask_rows;
    // measurement aid (wf_timing_enable): {earliest start, latest end} over the workgroups of a sweep_band_kernel launch on
    // the 100 MHz constant clock (s_memrealtime) -- the launch's own duration on the device, free of event / profiler overhead
    unsigned long long* tspan;
    int ablate;    // measurement aids, never set by the product path: bit 8 (WF_SWEEP_PROFILE=1) workgroup 0 prints its cycles per
                   // frame step; sweep_kernel, WF_SWEEP_ABLATE: 16 no register copy of streamed arcs, 32 no row stores, 64 no heavy
                   // states, 128 no light states (32-128 give wrong results: timing only); sweep_hub_kernel (tools/abl_hub.sh):
                   // 1 row stores all to row 0, 2 no loads, 4 no per-frame barrier (wrong results: timing only), 32 4-byte
                   // LDS-DMA pieces only, 64 the general frame body (both with unchanged results)
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
template <typename Acc, typename Sum, bool TROP, bool MASK, typename Arcs, typename ERow>
__device__ __forceinline__ void sweep_step(const Arcs& arcs, const ERow& erow, const Acc* cur, Acc* nxt,
                                           int S, int nheavy, int hcap, const int32_t* heavy, Acc* red,
                                           int* red_arc, const int32_t* arc_id, int32_t* bp_row,
                                           Acc* store_row, const float* mrow_a = nullptr, const float* mrow_b = nullptr,
                                           float thr = 0.f, int dir = 0, bool skip_light = false) {
    // MASK: lattice arc kept iff alpha^_t[src] + x + beta^_{t+1}[dst] >= thr (mrow_a = alpha^_t, mrow_b = beta^_{t+1};
    // `oth` is the arc's source in the forward sweep and its destination in the backward sweep)
    auto kept = [&](int st, int oth, float x) {
        const float v = dir ? mrow_a[st] + x + mrow_b[oth] : mrow_a[oth] + x + mrow_b[st];
        return v >= thr;
    };
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
            // kHU arcs per lane and round: their descriptor loads, then their emission gathers, are independent and
            // in flight together (one arc at a time costs two dependent L2 round trips per arc: with the 15 001-arc hub
            // of the BPE-5k denominators that was the whole frame step)
            constexpr int kHU = 8;
            for (int ab = a0 + tid; ab < a1; ab += nt * kHU) {
                int oth[kHU], col[kHU];
                float x[kHU];
#pragma unroll
                for (int j = 0; j < kHU; ++j) {
                    float w;
                    arcs.get(min(ab + j * nt, a1 - 1), oth[j], col[j], w);
                    x[j] = w;
                }
#pragma unroll
                for (int j = 0; j < kHU; ++j) x[j] += erow(col[j]);
#pragma unroll
                for (int j = 0; j < kHU; ++j) {
                    const int a = ab + j * nt;
                    if (a >= a1) break;
                    Acc v = cur[oth[j]] + (Acc)x[j];
                    if (MASK && !kept(h, oth[j], x[j])) v = neg_inf<Acc>();
                    if (TROP) max_push(m, barc, v, arc_id ? arc_id[a] : a);
                    else lse_push(m, s, v);
                }
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
    if (skip_light) return;   // the caller runs sweep_light_cached instead
    for (int sidx = tid; sidx < S; sidx += nt) {
        const int a0 = arcs.begin(sidx), a1 = arcs.begin(sidx + 1);
        if (a1 - a0 > WF_HEAVY_DEG) continue;
        Acc m = neg_inf<Acc>();
        Sum s = 0;
        int barc = 0x7fffffff;
        for (int a = a0; a < a1; ++a) {
            int oth, col; float w;
            arcs.get(a, oth, col, w);
            const float x = w + erow(col);
            Acc v = cur[oth] + (Acc)x;
            if (MASK && !kept(sidx, oth, x)) v = neg_inf<Acc>();
            if (TROP) max_push(m, barc, v, arc_id ? arc_id[a] : a);
            else lse_push(m, s, v);
        }
        Acc val = TROP ? m : lse_value(m, s);
        nxt[sidx] = val;
        if (store_row) store_row[sidx] = val;
        if (TROP && bp_row) bp_row[sidx] = (m == neg_inf<Acc>()) ? -1 : barc;
    }
}

// Arcs streamed from memory (graphs whose CSR does not fit the LDS next to the alpha buffers, e.g. the
// BPE-5k denominators with 10 002 states): a thread owns the same states in every frame, so the first
// kCDeg arcs of each of its (up to kCSlot) states are loaded ONCE and stay in registers; the per-frame
// loop then has no dependent chain of L2 round trips (offsets -> arc -> alpha / emission) left, only the
// independent LDS / emission gathers.  Measured on cfg4 (mmictc denominator): see DESIGN.md.
constexpr int kCSlot = 10, kCDeg = 2, kCHalf = 5;
struct ArcCache {
    unsigned mine;           // bit i: slot i is a state this thread writes (not past S, not a heavy state)
    unsigned more;           // bit i: that state has more than kCDeg arcs (the rest is read from memory)
    int pk[kCSlot][kCDeg];   // other state | column << 16 ; (0, weight -inf) for the slots beyond the degree
    float w[kCSlot][kCDeg];
};

template <typename Acc, typename Sum, typename ERow>
__device__ __forceinline__ void sweep_light_cached(const ArcCache& c, const ArcsGlobal& arcs, const ERow& erow, const Acc* cur,
                                                   Acc* nxt, Acc* store_row) {
    const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int h = 0; h < kCSlot; h += kCHalf) {
        float x[kCHalf][kCDeg];   // the emission gathers of half of the slots in flight at once
#pragma unroll
        for (int i = 0; i < kCHalf; ++i)
#pragma unroll
            for (int d = 0; d < kCDeg; ++d) x[i][d] = c.w[h + i][d] + erow((c.pk[h + i][d] >> 16) & 0xffff);
#pragma unroll
        for (int i = 0; i < kCHalf; ++i) {
            if (!((c.mine >> (h + i)) & 1)) continue;
            const int sidx = tid + (h + i) * nt;
            Acc m = neg_inf<Acc>();
            Sum s = 0;
#pragma unroll
            for (int d = 0; d < kCDeg; ++d) lse_push(m, s, cur[c.pk[h + i][d] & 0xffff] + (Acc)x[i][d]);
            if ((c.more >> (h + i)) & 1) {
                const int a0 = arcs.begin(sidx), a1 = arcs.begin(sidx + 1);
                for (int a = a0 + kCDeg; a < a1; ++a) {
                    int oth, col; float w;
                    arcs.get(a, oth, col, w);
                    lse_push(m, s, cur[oth] + (Acc)(w + erow(col)));
                }
            }
            const Acc val = lse_value(m, s);
            nxt[sidx] = val;
            if (store_row) store_row[sidx] = val;
        }
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

template <typename Acc, typename Sum, int MODE, bool ARCS_LDS, bool ESTAGE, bool MASK = false>
__global__ void __launch_bounds__(kMaxThreads) sweep_kernel(SweepArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool TROP = (MODE == M_TOT_MAX || MODE == M_VITERBI || MODE == M_FB_MAX);
    constexpr bool ROWS = (MODE == M_FB || MODE == M_FB_MAX);   // both directions, every row kept
    constexpr int NDIR = ROWS ? 2 : 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int u = blockIdx.x / NDIR, dir = blockIdx.x % NDIR;
    if (p.flag && !p.flag[u]) return;   // fallback pass: only the utterances the check flagged
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
            if (ROWS) arena[(int64_t)(dir ? T : 0) * S + s] = v;
        }
    }

    // ---- register copy of this thread's light arcs (arcs streamed from memory, log semiring, no mask)
    constexpr bool CACHE = !ARCS_LDS && !ESTAGE && !TROP && !MASK && sizeof(Sum) == 4;   // (the others have no registers to spare)
    const bool use_cache = CACHE && S <= kCSlot * nt && K < 65536 && S < 65536 && !(p.ablate & 16);
    ArcCache ac;
    ac.mine = ac.more = 0;
    if (use_cache) {
#pragma unroll
        for (int i = 0; i < kCSlot; ++i) {
            const int sidx = tid + i * nt;
            int a0 = 0, deg = -1;
            if (sidx < S) { a0 = g_off[sidx]; deg = g_off[sidx + 1] - a0; }
            if (deg > WF_HEAVY_DEG) deg = -1;
            if (deg >= 0) ac.mine |= 1u << i;
            if (deg > kCDeg) ac.more |= 1u << i;
#pragma unroll
            for (int d = 0; d < kCDeg; ++d) {
                const bool on = d < deg;
                ac.pk[i][d] = on ? (g_oth[a0 + d] | (g_col[a0 + d] << 16)) : 0;
                ac.w[i][d] = on ? g_w[a0 + d] : -INFINITY;
            }
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
            Acc* store_row = (ROWS && !(p.ablate & 32)) ? arena + (int64_t)(dir ? f : f + 1) * S : nullptr;
            const float* mra = MASK ? p.mA + U[WF_U_ALPHA] + (int64_t)f * S : nullptr;
            const float* mrb = MASK ? p.mB + U[WF_U_BETA] + (int64_t)(f + 1) * S : nullptr;
            const float mthr = MASK ? p.best[u] - p.beam : 0.f;
            int32_t* bp_row = (MODE == M_VITERBI) ? p.bp + U[WF_U_ALPHA] + (int64_t)f * S : nullptr;
            if (ESTAGE) {
                ERowLds er{ebuf + (c & 1) * C * p.kpad + (step - c * C) * p.kpad};
                if (ARCS_LDS) {
                    ArcsLds arcs{larcs, larcs + ((S + 1 + 3) & ~3),
                                 (const float*)(larcs + ((S + 1 + 3) & ~3) + ((A + 3) & ~3))};
                    sweep_step<Acc, Sum, TROP, MASK>(arcs, er, cur, nxt, S, nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row, mra, mrb, mthr, dir);
                } else {
                    ArcsGlobal arcs{g_off, g_oth, g_col, g_w};
                    sweep_step<Acc, Sum, TROP, MASK>(arcs, er, cur, nxt, S, (p.ablate & 64) ? 0 : nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row, mra, mrb, mthr, dir, use_cache);
                    if (CACHE) { if (use_cache && !(p.ablate & 128)) sweep_light_cached<Acc, Sum>(ac, arcs, er, cur, nxt, store_row); }
                }
            } else {
                ERowGlobal er{ebase + (int64_t)f * epitch};
                if (ARCS_LDS) {
                    ArcsLds arcs{larcs, larcs + ((S + 1 + 3) & ~3),
                                 (const float*)(larcs + ((S + 1 + 3) & ~3) + ((A + 3) & ~3))};
                    sweep_step<Acc, Sum, TROP, MASK>(arcs, er, cur, nxt, S, nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row, mra, mrb, mthr, dir);
                } else {
                    ArcsGlobal arcs{g_off, g_oth, g_col, g_w};
                    sweep_step<Acc, Sum, TROP, MASK>(arcs, er, cur, nxt, S, (p.ablate & 64) ? 0 : nheavy, p.hcap, heavy, red, red_arc, arc_id, bp_row, store_row, mra, mrb, mthr, dir, use_cache);
                    if (CACHE) { if (use_cache && !(p.ablate & 128)) sweep_light_cached<Acc, Sum>(ac, arcs, er, cur, nxt, store_row); }
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
            if ((MODE == M_FB || MODE == M_TOT_LOG) && p.totd) p.totd[u] = (double)val;
            if (MASK && p.thr) p.thr[u] = p.best[u] - p.beam;
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
    static_assert(MODE != M_FB, "forward-only kernel: the forward-backward pass is sweep_lin_kernel / sweep_kernel");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool TROP = (MODE == M_TOT_MAX || MODE == M_VITERBI);
    constexpr bool LOGS = !TROP;
    constexpr int HDN = HD > 0 ? HD : 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int slot = tid & 3, quad = tid >> 2, nq = nt >> 2;
    const int u = blockIdx.x;
    constexpr int dir = 0;
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

    double coff = 0.0;   // current offset c (log2 units)

    {
        const float* fb = (const float*)(blob + I[TROP ? WF_I_FINB_MAX : WF_I_FINB_LOG]);
        for (int s = tid; s < S; s += nt) {
            float v;
            if (dir) { const float f = fb[s]; v = (f == -INFINITY) ? -kBig : f * escale; }
            else v = (s == 0) ? 0.f : -kBig;
            cur[s] = v;
        }
    }
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
                if (MODE == M_VITERBI) bprow[sidx] = (val < kDead) ? -1 : barc;
            }
            if (LOGS) rowmax = fmaxf(rowmax, val);
        }
        if (LOGS && k == kPD - 1) {
            const float wm = wave_max(rowmax);
            if (lane == 0) rmax[wave] = wm;
        }
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
            }
            if (p.tot_double) ((double*)p.tot)[u] = out;
            else ((float*)p.tot)[u] = (float)out;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Linear-domain forward-backward sweep: the fast path of wf_fb_logsum (loss + gradient).
//
// Measured on MI355X (tools/ubench/lat.hip): one frame step costs ~140 cycles for the LDS
// read -> write -> barrier round trip alone, and every *dependent* VALU instruction adds 8-16 cycles
// (4-cycle issue per wave64 instruction, times the waves sharing a SIMD).  A log-semiring step is a
// chain of ~10 dependent operations including two transcendentals (add, max, max, sub, exp, add,
// add, log, add); the same step in the probability domain is one multiply and three FMAs:
//
//      alpha^_{t+1}[d] = sum_a  alpha^_t[src(a)] * ( exp(w_a) * P_t[col(a)] ),   P_t[k] = exp(lp[t,k] - c_t)
//
// with everything except the FMA chain off the critical path (the products exp(w)*P for the next
// frame are formed while the LDS reads of this frame are in flight).  fp64 carries the dynamic
// range: c_t = max_k lp[t, label_k] takes out the per-frame scale (gather_lin_kernel) and every kLD
// frames the row is multiplied by a power of two that steers its maximum to 2^800, so a state
// underflows only when it is ~1800 bits (1250 nats) below the best state of its row.  That cannot
// be told from a dead state inside the sweep, so check_kernel verifies afterwards that
// sum_s alpha_t[s] * beta_t[s] = tot on every row; an utterance that fails (or totals to zero
// without being structurally infeasible) is re-run by the log-semiring sweep_kernel (fallback).
// Rows are stored as they are (scaled fp64) plus one cumulative shift per group of kLD frames; the
// post kernels take log2 of them.  lane = state, arcs of the state (<= WF_REG_DEG) in VGPRs, at
// most one high-degree state reduced by the whole workgroup, as in sweep_quad_kernel.
// ------------------------------------------------------------------------------------------
// emission ring depth = frames per rescale group: 8 where the register budget allows it (workgroups of
// <= 256 lanes); every step waits for the vector-memory operations issued more than LD-1 steps
// earlier (ring loads AND row stores share vmcnt), so LD steps must cover a store acknowledge
// (600-950 cycles, MI355X_MICROARCH.md).
constexpr int kLinBias = 800;          // the row maximum is steered to 2^kLinBias
constexpr double kLog2eD = 1.4426950408889634074;

__device__ __forceinline__ int f64_exp_field(double v) { return (__double2hiint(v) >> 20) & 0x7ff; }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_d(double v) {   // lanes without a source receive 0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0xB1, 0xf>(v);
    v += dpp_d<0x4E, 0xf>(v);
    v += dpp_d<0x114, 0xf>(v);
    v += dpp_d<0x118, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_i<0xB1, 0xf>(v, v));
    v = max(v, dpp_i<0x4E, 0xf>(v, v));
    v = max(v, dpp_i<0x114, 0xf>(v, v));
    v = max(v, dpp_i<0x118, 0xf>(v, v));
    v = max(v, dpp_i<0x142, 0xa>(v, v));
    v = max(v, dpp_i<0x143, 0xc>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
// index of the cumulative shift carried by the row that sweep step s produced (s < 0: the initial row)
// lag: steps between a rescale decision and the first row that carries it (sweep_lin_kernel 1, sweep_band_kernel 0)
__device__ __forceinline__ int lin_shift_index(int s, int grp, int lag) { return s < lag ? 0 : (s - lag) / grp + 1; }

// workgroup-uniform values that were read through vector loads: move them to SGPRs so that loop
// counters and row pointers are advanced by the scalar unit
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffff));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | lo;
}

// lane `srclane`'s copy of v (all lanes execute it; ds_bpermute addresses lanes in bytes)
__device__ __forceinline__ double bperm_d(double v, int srclane) {
    const int lo = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// WV (graphs of at most 64 states, one wavefront: the small denominators, e.g. cfg2's 60-state 2state topology): the
// state vector never leaves the registers -- a lane fetches its source states' values with ds_bpermute instead of an
// LDS write -> barrier -> read round trip, and the exponent bookkeeping stays in the wave.
template <int HD, int NTMAX, bool WV = false>
__global__ void __launch_bounds__(NTMAX) sweep_lin_kernel(SweepArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int D = WF_REG_DEG;
    constexpr int kLD = NTMAX <= 256 ? 8 : 4;
    constexpr int HDN = HD > 0 ? HD : 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int u = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = uni((int)U[WF_U_FSA]), T = uni((int)U[WF_U_T]);
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = uni(I[WF_I_S]), A = uni(I[WF_I_A]), K = uni(I[WF_I_K]);
    if (T < uni(I[WF_I_MINLEN])) {   // structurally infeasible (models.py:305-313 mask_inf): no sweep needed
        if (dir == 0 && tid == 0) {
            ((float*)p.tot)[u] = -INFINITY;
            p.totd[u] = (double)-INFINITY;
            p.tot2[u] = (double)-INFINITY;
            p.ctot2[u] = 0.0;
        }
        return;
    }

    double* cur = (double*)smem;
    double* nxt = cur + p.smax;
    double* red = nxt + p.smax;          // [16]
    int* rexp = (int*)(red + 16);        // [16] per-wave maxima of the exponent field (unused waves: 0)

    const int32_t* g_off = blob + uni(I[dir ? WF_I_OUT_OFF : WF_I_IN_OFF]);
    const int32_t* g_oth = blob + uni(I[dir ? WF_I_OUT_DST : WF_I_IN_SRC]);
    const int32_t* g_col = blob + uni(I[dir ? WF_I_OUT_COL : WF_I_IN_COL]);
    const float* g_w = (const float*)(blob + uni(I[dir ? WF_I_OUT_W : WF_I_IN_W]));
    const bool has_over = HD > 0 && uni(I[dir ? WF_I_NOVER_OUT : WF_I_NOVER_IN]) > 0;
    const int over = has_over ? uni(I[dir ? WF_I_OVER_OUT : WF_I_OVER_IN]) : -1;
    const bool mine = tid < S && tid != over;   // this lane owns state `tid`

    // ---- this lane's arcs (padding: probability 0 from state 0, column 0)
    int src[D], col[D];
    double Wt[D];
    {
        int a0 = 0, deg = 0;
        if (mine) { a0 = g_off[tid]; deg = g_off[tid + 1] - a0; }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const bool valid = i < deg;
            src[i] = valid ? g_oth[a0 + i] : 0;
            col[i] = valid ? g_col[a0 + i] : 0;
            Wt[i] = valid ? exp((double)g_w[a0 + i]) : 0.0;
        }
    }
    int hsrc[HDN], hcol[HDN];
    double hW[HDN];
    {
        int h0 = 0, h1 = 0;
        if (has_over) { h0 = g_off[over]; h1 = g_off[over + 1]; }
#pragma unroll
        for (int j = 0; j < HDN; ++j) {
            const int a = h0 + tid + j * nt;
            const bool valid = HD > 0 && a < h1;
            hsrc[j] = valid ? g_oth[a] : 0;
            hcol[j] = valid ? g_col[a] : 0;
            hW[j] = valid ? exp((double)g_w[a]) : 0.0;
        }
    }

    // Loads are unconditional: load index n reads frame min(n, T-1) (time-reversed for beta); a graph
    // without arcs or an empty segment reads the utterance table (small integers = finite tiny
    // doubles) through a zero pitch.
    const double* pbase = p.P + uni(U[WF_U_P]);
    int64_t ppitch = uni((int)U[WF_U_PP]);
    if (A == 0 || T == 0) { pbase = (const double*)U; ppitch = 0; }
    const int last_step = max(T - 1, 0);
    const int64_t pstep = dir ? -ppitch : ppitch;
    const double* pload = pbase + (int64_t)(dir ? last_step : 0) * ppitch;   // row of load index 0

    double* arena = (double*)(dir ? p.beta : p.alpha) + uni(U[dir ? WF_U_BETA : WF_U_ALPHA]);
    double* goff = (dir ? p.boff : p.aoff) + uni(U[WF_U_OFFS]);

    // ---- initial vector (alpha_0 = delta(start); beta_T = final weights) and its exponent maximum
    double av = 0.0;   // WV: this lane's state value
    int wv_emax = 0;   // WV: exponent maximum of the row that ended the previous group
    {
        const float* fb = (const float*)(blob + uni(I[WF_I_FINB_LOG]));
        int emax = 0;
        for (int s = tid; s < S; s += nt) {
            const double v = dir ? exp((double)fb[s]) : (s == 0 ? 1.0 : 0.0);
            if (WV) av = v; else cur[s] = v;
            arena[(int64_t)(dir ? T : 0) * S + s] = v;
            emax = max(emax, f64_exp_field(v));
        }
        emax = wave_max_i(emax);
        if (WV) {
            wv_emax = emax;
        } else {
            if (tid < 16) rexp[tid] = 0;
            __syncthreads();
            if (lane == 0) rexp[wave] = emax;
        }
        if (tid == 0) goff[0] = 0.0;
    }
    double* srow = arena + (int64_t)(dir ? (T - 1) : 1) * S;   // row written by step 0, then +-S per step
    const int64_t sstep = dir ? -(int64_t)S : (int64_t)S;

    // ---- emission ring: slot (step % kLD) holds the probabilities of frame step+1 while step runs
    double pr[kLD][D], hpr[kLD][HDN];
    double qc[D], qn[D], hqc[HDN], hqn[HDN];   // arc probabilities of this step / the next one
#define WF_LOAD_ROW(DST, HDST)                                                      \
    do {                                                                            \
        _Pragma("unroll") for (int i = 0; i < D; ++i) DST[i] = pload[col[i]];       \
        if (HD > 0) {                                                               \
            _Pragma("unroll") for (int j = 0; j < HDN; ++j) HDST[j] = pload[hcol[j]]; \
        }                                                                           \
    } while (0)
    WF_LOAD_ROW(qc, hqc);   // load index 0 = frame 0
    if (0 < last_step) pload += pstep;
#pragma unroll
    for (int k = 0; k < kLD; ++k) {   // load indices 1 .. kLD
        WF_LOAD_ROW(pr[k], hpr[k]);
        if (k + 1 < last_step) pload += pstep;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) qc[i] *= Wt[i];
#pragma unroll
    for (int j = 0; j < HDN; ++j) hqc[j] *= hW[j];
    double cum = 0.0, cum_last = 0.0;   // cumulative power-of-two shift (log2 units) carried by the rows
    if (!WV) __syncthreads();

    auto do_step = [&](const int step, const int k) __attribute__((always_inline)) {
        // ---- critical path, part 1: issue the LDS reads of the source states' alpha
        double a[D], ha[HDN];
#pragma unroll
        for (int i = 0; i < D; ++i) a[i] = WV ? bperm_d(av, src[i]) : cur[src[i]];
#pragma unroll
        for (int j = 0; j < HDN; ++j) ha[j] = (HD > 0) ? (WV ? bperm_d(av, hsrc[j]) : cur[hsrc[j]]) : 0.0;
        int4 rx[4];
        if (k == 0 && !WV) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rx[i] = ((const int4*)rexp)[i];
        }
        // ---- in the shadow of the LDS latency: next step's arc probabilities from ring slot k, which
        // is then refilled (load index step+1+kLD).  The values are pinned before the refill so that it
        // lands in the same registers (otherwise hipcc rotates the ring with copies behind a vmcnt(0)).
#pragma unroll
        for (int i = 0; i < D; ++i) { qn[i] = Wt[i] * pr[k][i]; asm volatile("" : "+v"(qn[i])); }
#pragma unroll
        for (int j = 0; j < HDN; ++j) { hqn[j] = hW[j] * hpr[k][j]; asm volatile("" : "+v"(hqn[j])); }
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.ablate & 2)) WF_LOAD_ROW(pr[k], hpr[k]);
        if (step + 1 + kLD < last_step) pload += pstep;
        cum_last = cum;
        if (k == 0 && !(p.ablate & 4)) {
            // rescale decision from the row finished at the end of the previous group
            int emax;
            if (WV) {
                emax = wv_emax;
            } else {
                int4 m4 = rx[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) { m4.x = max(m4.x, rx[i].x); m4.y = max(m4.y, rx[i].y); m4.z = max(m4.z, rx[i].z); m4.w = max(m4.w, rx[i].w); }
                emax = max(max(m4.x, m4.y), max(m4.z, m4.w));
            }
            const int sh = emax > 0 ? (1023 + kLinBias) - emax : 0;
#pragma unroll
            for (int i = 0; i < D; ++i) qn[i] = ldexp(qn[i], sh);
#pragma unroll
            for (int j = 0; j < HDN; ++j) hqn[j] = ldexp(hqn[j], sh);
            cum += (double)sh;
            if (tid == 0) goff[step / kLD + 1] = cum;
        }
        // ---- critical path, part 2
        int emine = 0;
        if (HD > 0 && has_over) {
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < HDN; ++j) part = fma(ha[j], hqc[j], part);
            double tot_h = wave_sum_d(part);
            if (!WV && nw > 1) {
                if (lane == 0) red[wave] = tot_h;
                __syncthreads();
                if (tid == 0) {
                    tot_h = red[0];
                    for (int i = 1; i < nw; ++i) tot_h += red[i];
                }
            }
            if (WV) {
                if (tid == over) { av = tot_h; srow[over] = tot_h; emine = f64_exp_field(tot_h); }
            } else if (tid == 0) {
                nxt[over] = tot_h;
                srow[over] = tot_h;
                emine = f64_exp_field(tot_h);
            }
        }
        double v = a[0] * qc[0];
#pragma unroll
        for (int i = 1; i < D; ++i) v = fma(a[i], qc[i], v);
        if (mine) {
            if (WV) av = v; else nxt[tid] = v;
            if (!(p.ablate & 1)) srow[tid] = v;
            emine = max(emine, f64_exp_field(v));
        }
        if (k == kLD - 1 && !(p.ablate & 4)) {
            const int em = wave_max_i(emine);
            if (WV) wv_emax = em;
            else if (lane == 0) rexp[wave] = em;
        }
        srow += sstep;
#pragma unroll
        for (int i = 0; i < D; ++i) qc[i] = qn[i];
#pragma unroll
        for (int j = 0; j < HDN; ++j) hqc[j] = hqn[j];
        if (!WV) {
            __syncthreads();
            double* t_ = cur; cur = nxt; nxt = t_;
        }
    };
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
    int t0 = 0;
    for (; t0 + kLD <= T; t0 += kLD) {
#pragma unroll
        for (int k = 0; k < kLD; ++k) do_step(t0 + k, k);
    }
#pragma unroll
    for (int k = 0; k < kLD; ++k)
        if (t0 + k < T) do_step(t0 + k, k);
#undef WF_LOAD_ROW
    if ((p.ablate & 8) && blockIdx.x == 0 && tid == 0)
        printf("sweep_lin: T=%d nt=%d cycles/step=%.1f\n", T, nt, (double)(__builtin_amdgcn_s_memtime() - clk0) / max(T, 1));

    // ---- forward workgroup: total score = log(sum over final arcs) - shifts + sum_t c_t
    if (dir == 0) {
        const int NF = I[WF_I_NF];
        const int32_t* fsrc = blob + I[WF_I_FIN_SRC];
        const float* fw = (const float*)(blob + I[WF_I_FIN_W]);
        double part = 0.0, csum = 0.0;
        if (WV) {   // every lane takes part in the exchange; uniform trip count
            for (int i0 = 0; i0 < NF; i0 += 64) {
                const int i = i0 + lane;
                const double a = bperm_d(av, i < NF ? fsrc[i] : 0);
                if (i < NF) part = fma(a, exp((double)fw[i]), part);
            }
        } else {
            for (int i = tid; i < NF; i += nt) part = fma(cur[fsrc[i]], exp((double)fw[i]), part);
        }
        const double* crow = p.cmax + U[WF_U_ROW0];
        for (int t = tid; t < T; t += nt) csum += crow[t];
        part = wave_sum_d(part);
        csum = wave_sum_d(csum);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        // two rounds through the 16 reduction slots: final-arc sums, then the c_t sums
        double fsum = 0.0;
        if (tid == 0) for (int i = 0; i < nw; ++i) fsum += red[i];
        __syncthreads();
        if (lane == 0) red[wave] = csum;
        __syncthreads();
        if (tid == 0) {
            double ct = 0.0;
            for (int i = 0; i < nw; ++i) ct += red[i];
            const double ctot2 = ct * kLog2eD;
            const bool dead = !(fsum > 0.0);
            const double tot2 = dead ? (double)-INFINITY : log2(fsum) - cum_last + ctot2;
            const double tot = dead ? (double)-INFINITY : tot2 * kLn2;
            p.tot2[u] = tot2;
            p.ctot2[u] = ctot2;
            p.totd[u] = tot;
            ((float*)p.tot)[u] = (float)tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Linear-domain sweep for graphs of up to kLBSlot * 1024 states whose arcs do not fit the LDS (the BPE-5k
// denominators of cfg4: S = 10 002, A = 30 002, a hub state with fan-in 15 001).  One workgroup of 1024 lanes
// per (utterance, direction); a lane owns the states tid + i * 1024 in every frame and keeps their first two arcs
// in registers (every light state of the 11 topologies has <= 2-3 arcs), as it does its slice of the hub's arc
// list: nothing of the graph is re-read.  "Weights elided" graphs only (every arc weight 0, all training
// topologies), so a frame step is one fp64 FMA per arc on alpha_t[src] and P_t[col], both in the LDS:
//  * the state vector ping-pongs between two **fp32** LDS buffers (two fp64 ones would be the whole LDS at
//    S = 10 002), steered to 2^kLBBias every frame from the exponent maximum of the previous row; sums are formed
//    in fp64 and the rows that leave for the workspace are the unrounded fp64 sums times 2^(kLinBias - kLBBias), i.e.
//    exactly the rows sweep_lin_kernel would store.  A state more than ~2^-226 below its row's maximum flushes to
//    zero; like every underflow of the linear sweeps this is caught by the row check (fallback), not assumed away.
//  * the fp32 emission row of the NEXT frame is loaded (coalesced, from gather_lin_kernel's P32 rows) into
//    registers at the top of a step and stored to the LDS behind the step's first barrier: no step waits for memory.
// Two barriers per frame (the hub's partial sums meet at the first).  Shifts: group size 1, lag 0 in
// lin_shift_index terms; tot2 / ctot2 and the row check / fallback are those of sweep_lin_kernel.
// ------------------------------------------------------------------------------------------
constexpr int kLBSlot = 10, kLBDeg = 2, kLBHeavy = 64, kLBHub = 16;
constexpr int kLBBias = 100;   // fp32 LDS copy of a row: maximum steered to 2^kLBBias

__global__ void __launch_bounds__(kMaxThreads) sweep_linbig_kernel(SweepArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int u = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = uni((int)U[WF_U_FSA]), T = uni((int)U[WF_U_T]);
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = uni(I[WF_I_S]), A = uni(I[WF_I_A]), K = uni(I[WF_I_K]);
    if (T < uni(I[WF_I_MINLEN])) {   // structurally infeasible: no sweep needed
        if (dir == 0 && tid == 0) {
            ((float*)p.tot)[u] = -INFINITY;
            p.totd[u] = (double)-INFINITY;
            p.tot2[u] = (double)-INFINITY;
            p.ctot2[u] = 0.0;
        }
        return;
    }
    const int spad = p.smax + 4;
    double* red = (double*)smem;                  // [kLBHeavy][16] per-wave partial sums of the heavy states
    int* rexp = (int*)(red + kLBHeavy * 16);      // [2][16] per-wave maxima of the exponent field, by step parity
    float* buf0 = (float*)(rexp + 32);            // [spad] x 2: slot S holds 0 (the source of padding arcs)
    float* buf1 = buf0 + spad;
    float* pl = buf1 + spad;                      // [kpad] emission probabilities of the current frame

    const int32_t* g_off = blob + uni(I[dir ? WF_I_OUT_OFF : WF_I_IN_OFF]);
    const int32_t* g_oth = blob + uni(I[dir ? WF_I_OUT_DST : WF_I_IN_SRC]);
    const int32_t* g_col = blob + uni(I[dir ? WF_I_OUT_COL : WF_I_IN_COL]);
    const int nheavy = uni(I[dir ? WF_I_NHEAVY_OUT : WF_I_NHEAVY_IN]);
    const int32_t* heavy = blob + uni(I[dir ? WF_I_HEAVY_OUT : WF_I_HEAVY_IN]);

    // ---- this lane's states: first kLBDeg arcs in registers (padding: source S = 0.0, column 0)
    unsigned mine = 0, more = 0;
    int pk[kLBSlot][kLBDeg];
#pragma unroll
    for (int i = 0; i < kLBSlot; ++i) {
        const int sidx = tid + i * nt;
        int a0 = 0, deg = -1;
        if (sidx < S) { a0 = g_off[sidx]; deg = g_off[sidx + 1] - a0; }
        if (deg > WF_HEAVY_DEG) deg = -1;
        if (deg >= 0) mine |= 1u << i;
        if (deg > kLBDeg) more |= 1u << i;
#pragma unroll
        for (int d = 0; d < kLBDeg; ++d) pk[i][d] = d < deg ? (g_oth[a0 + d] | (g_col[a0 + d] << 16)) : S;
    }
    // ---- the first heavy state's arcs, a strided slice per lane (hubs beyond kLBHub * 1024 arcs: the rest is streamed)
    int hpk[kLBHub];
    {
        int h0 = 0, h1 = 0;
        if (nheavy > 0) { h0 = g_off[heavy[0]]; h1 = g_off[heavy[0] + 1]; }
#pragma unroll
        for (int j = 0; j < kLBHub; ++j) {
            const int a = h0 + tid + j * nt;
            hpk[j] = a < h1 ? (g_oth[a] | (g_col[a] << 16)) : S;
        }
    }

    const float* pbase = (const float*)p.P + uni(U[WF_U_P]);   // fp32 rows (gather_lin_kernel, P32)
    int64_t ppitch = uni((int)U[WF_U_PP]);
    if (A == 0 || T == 0) { pbase = (const float*)U; ppitch = 0; }
    double* arena = (double*)(dir ? p.beta : p.alpha) + uni(U[dir ? WF_U_BETA : WF_U_ALPHA]);
    double* goff = (dir ? p.boff : p.aoff) + uni(U[WF_U_OFFS]);

    // ---- initial vector (stored unscaled, shift 0) and its exponent maximum
    {
        const float* fb = (const float*)(blob + uni(I[WF_I_FINB_LOG]));
        int emax = 0;
        for (int s = tid; s < S; s += nt) {
            const double v = dir ? exp((double)fb[s]) : (s == 0 ? 1.0 : 0.0);
            buf0[s] = (float)v;
            arena[(int64_t)(dir ? T : 0) * S + s] = v;
            emax = max(emax, f64_exp_field(v));
        }
        if (tid == 0) { buf0[S] = 0.f; buf1[S] = 0.f; goff[0] = 0.0; }
        if (T > 0) {   // emission row of the first step
            const float* prow0 = pbase + (int64_t)(dir ? T - 1 : 0) * ppitch;
            for (int k = tid; k < K; k += nt) pl[k] = prow0[k];
        }
        emax = wave_max_i(emax);
        if (tid < 32) rexp[tid] = 0;
        __syncthreads();
        if (lane == 0) rexp[wave] = emax;
        __syncthreads();
    }
    // the LDS copy of row r is the true scaled row times 2^-lds_less: 0 for the initial row, kLinBias - kLBBias after
    double cum = 0.0;   // cumulative shift of the rows in the workspace
    int lds_less = 0;
    float* cur = buf0;
    float* nxt = buf1;
    for (int step = 0; step < T; ++step) {
        const int f = dir ? T - 1 - step : step;
        double* srow = arena + (int64_t)(dir ? f : f + 1) * S;
        int* rx_in = rexp + (step & 1) * 16;
        int* rx_out = rexp + ((step + 1) & 1) * 16;
        // the descriptors are loop-invariant: without this, hipcc hoists the two LDS byte addresses of every arc out of
        // the frame loop (72 more live registers) and spills
#pragma unroll
        for (int i = 0; i < kLBSlot; ++i)
#pragma unroll
            for (int d = 0; d < kLBDeg; ++d) asm volatile("" : "+v"(pk[i][d]));
#pragma unroll
        for (int j = 0; j < kLBHub; ++j) asm volatile("" : "+v"(hpk[j]));
        // next step's emission row: coalesced loads now, into the LDS behind this step's first barrier.  Uniform base
        // + lane offset (no per-slot address registers); slots past K read on into the arena, which is padded by
        // kLBSlot * 1024 floats for this (make_plan), and are never stored
        float pn[kLBSlot];
        {
            const int fn = step + 1 < T ? (dir ? f - 1 : f + 1) : f;
            const float* prown = pbase + (int64_t)fn * ppitch;
#pragma unroll
            for (int i = 0; i < kLBSlot; ++i) pn[i] = (prown + i * nt)[(unsigned)tid];
        }
        // rescale decided from the previous row's exponent maximum (exponent fields of the LDS-scale values)
        int sh;
        {
            int4 m4 = ((const int4*)rx_in)[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                const int4 r = ((const int4*)rx_in)[i];
                m4.x = max(m4.x, r.x); m4.y = max(m4.y, r.y); m4.z = max(m4.z, r.z); m4.w = max(m4.w, r.w);
            }
            const int emax = max(max(m4.x, m4.y), max(m4.z, m4.w));
            sh = emax > 0 ? (1023 + kLBBias) - emax : 0;   // brings the LDS-scale row to 2^kLBBias
        }
        // workspace row = LDS-scale row * 2^(kLinBias - kLBBias); its cumulative shift:
        cum += (double)(sh + (kLinBias - kLBBias) - lds_less);
        lds_less = kLinBias - kLBBias;
        if (tid == 0) goff[step + 1] = cum;
        if (lane == 0) rx_out[wave] = 0;
        int emine = 0;
        // ---- this lane's light states
#pragma unroll
        for (int i = 0; i < kLBSlot; ++i) {
            double v = (double)cur[pk[i][0] & 0xffff] * (double)pl[(pk[i][0] >> 16) & 0xffff];
#pragma unroll
            for (int d = 1; d < kLBDeg; ++d) v = fma((double)cur[pk[i][d] & 0xffff], (double)pl[(pk[i][d] >> 16) & 0xffff], v);
            if ((more >> i) & 1) {
                const int32_t* go = g_off + i * nt;
                for (int a = go[(unsigned)tid] + kLBDeg; a < go[(unsigned)tid + 1]; ++a) v = fma((double)cur[g_oth[a]], (double)pl[g_col[a]], v);
            }
            if ((mine >> i) & 1) {
                v = ldexp(v, sh);
                (nxt + i * nt)[(unsigned)tid] = (float)v;
                (srow + i * nt)[(unsigned)tid] = ldexp(v, kLinBias - kLBBias);
                emine = max(emine, f64_exp_field(v));
            }
        }
        // ---- heavy states: every lane a strided slice of the arc list
        for (int hi = 0; hi < nheavy; ++hi) {
            const int hs = heavy[hi];
            const int a0 = g_off[hs], a1 = g_off[hs + 1];
            double part = 0.0;
            int ab = a0 + tid;
            if (hi == 0) {   // descriptors in registers
#pragma unroll
                for (int j = 0; j < kLBHub; ++j) part = fma((double)cur[hpk[j] & 0xffff], (double)pl[(hpk[j] >> 16) & 0xffff], part);
                ab += kLBHub * nt;
            }
            for (; ab < a1; ab += nt * 4) {
                int so[4], co[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int a = ab + j * nt;
                    const bool on = a < a1;
                    const int ac = on ? a : a1 - 1;
                    so[j] = on ? g_oth[ac] : S;
                    co[j] = g_col[ac];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) part = fma((double)cur[so[j]], (double)pl[co[j]], part);
            }
            part = wave_sum_d(part);
            if (lane == 0) red[hi * 16 + wave] = part;
        }
        __syncthreads();   // partial sums complete; every read of pl is done
#pragma unroll
        for (int i = 0; i < kLBSlot; ++i)
            if (tid < K - i * nt) (pl + i * nt)[(unsigned)tid] = pn[i];
        if (tid < nheavy) {
            double v = red[tid * 16];
            for (int w = 1; w < nw; ++w) v += red[tid * 16 + w];
            v = ldexp(v, sh);
            const int hs = heavy[tid];
            nxt[hs] = (float)v;
            srow[hs] = ldexp(v, kLinBias - kLBBias);
            emine = max(emine, f64_exp_field(v));
        }
        emine = wave_max_i(emine);
        if (lane == 0) rx_out[wave] = emine;
        __syncthreads();
        float* t_ = cur; cur = nxt; nxt = t_;
    }

    // ---- forward workgroup: total score = log(sum over final arcs) - shifts + sum_t c_t
    if (dir == 0) {
        const int NF = I[WF_I_NF];
        const int32_t* fsrc = blob + I[WF_I_FIN_SRC];
        const float* fw = (const float*)(blob + I[WF_I_FIN_W]);
        double part = 0.0, csum = 0.0;
        for (int i = tid; i < NF; i += nt) part = fma((double)cur[fsrc[i]], exp((double)fw[i]), part);
        const double* crow = p.cmax + U[WF_U_ROW0];
        for (int t = tid; t < T; t += nt) csum += crow[t];
        part = wave_sum_d(part);
        csum = wave_sum_d(csum);
        if (lane == 0) { red[wave] = part; red[16 + wave] = csum; }
        __syncthreads();
        if (tid == 0) {
            double fsum = 0.0, ct = 0.0;
            for (int i = 0; i < nw; ++i) { fsum += red[i]; ct += red[16 + i]; }
            const double ctot2 = ct * kLog2eD;
            const bool dead = !(fsum > 0.0);
            const double tot2 = dead ? (double)-INFINITY : log2(fsum) - (cum - (double)lds_less) + ctot2;
            const double tot = dead ? (double)-INFINITY : tot2 * kLn2;
            p.tot2[u] = tot2;
            p.ctot2[u] = ctot2;
            p.totd[u] = tot;
            ((float*)p.tot)[u] = (float)tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Hub sweep (sweep_hub_kernel): the probability-domain forward / backward recursion of a TOPOLOGY USED AS DENOMINATOR over
// a large table (cfg4: mmictc / 3state-skip at P = 5000, S = 10 002, fan-in(0) = 10 001 - 15 001), in the hub form that
// wf_graph_prepare detects (wf_internal.h WF_I_HUB_*): state 0 is the hub, every other state belongs to one of NG groups
// of G consecutive states (one group per phone) that talk only to the hub and among themselves, with the same arc pattern
// in every group and labels at a regular stride.
//
// One 1024-lane workgroup per (utterance, direction).  A lane OWNS groups (tid + k * 1024): their G state values stay in
// fp64 registers for the whole sweep, their emissions are read straight from the log_probs row (adjacent lanes: adjacent
// columns; no gather kernel, no probability arena) two frames ahead, exp'd by the lane, and combined through the group's
// small transition pattern (coefficient table in scalar registers).  The ONLY cross-lane traffic of a frame is the hub:
// each lane's contribution is summed with a DPP wave reduction and 16 LDS words, one barrier per frame -- against
// 72 scattered LDS reads per lane and frame in sweep_linbig_kernel (5.7 us per frame step -> see DESIGN.md).
// Rows leave as fp64 scaled probabilities with one power-of-two shift per frame, the format of sweep_linbig_kernel
// (group 1, lag 0; c_t = 0: the emissions are not normalised per frame), so the row check, the fallback and the
// posterior kernels are shared.
// ------------------------------------------------------------------------------------------
constexpr int kHubCompute = 768;   // lanes that own groups (12 wavefronts); the other 4 wavefronts of the workgroup load
constexpr int kHubSlots = 7;       // groups per compute lane: NG <= 7 * 768
constexpr int kHubLoad = kMaxThreads - kHubCompute;
constexpr size_t kHubLdsFixed = 32 * 8 + 32 * 4;   // LDS in front of the row buffers: partial sums and exponents of the hub

// The arc patterns of the reference's two-state-per-phone topologies (utils/get_token_fst_*_k2.py) in hub form, as compile-
// time entry lists "new value of i += P[class c] * value of j" (forward direction; i, j: 0 = x, 1 = x + 1, 2 = the hub),
// encoded i * 16 + j * 4 + c, every arc count 1.  With the pattern known at compile time a group's step is one fp64 FMA
// per arc and one conversion per class; any other pattern (PAT = -1) evaluates the whole coefficient table (3x the work).
constexpr int kHubPatMax = 8;
constexpr int hub_e(int i, int j, int c) { return i * 16 + j * 4 + c; }
struct HubPattern { int n; int e[kHubPatMax]; };
constexpr HubPattern kHubPatterns[] = {
    {6, {hub_e(0, 2, 0), hub_e(1, 0, 1), hub_e(1, 1, 1), hub_e(2, 0, 1), hub_e(2, 1, 1), hub_e(2, 2, 0)}},                  // 0 mmictc
    {7, {hub_e(0, 2, 0), hub_e(0, 0, 0), hub_e(1, 0, 1), hub_e(1, 1, 1), hub_e(2, 0, 1), hub_e(2, 1, 1), hub_e(2, 2, 0)}},  // 1 mmictc-1
    {6, {hub_e(0, 2, 0), hub_e(0, 0, 0), hub_e(1, 0, 1), hub_e(1, 1, 1), hub_e(2, 0, 1), hub_e(2, 1, 1)}},                  // 2 2state
    {5, {hub_e(0, 2, 0), hub_e(1, 0, 1), hub_e(1, 1, 1), hub_e(2, 0, 1), hub_e(2, 1, 1)}},                                  // 3 2state-1
    {5, {hub_e(0, 2, 0), hub_e(1, 0, 1), hub_e(1, 1, 1), hub_e(2, 0, 2), hub_e(2, 1, 2)}},                                  // 4 3state-skip
};
constexpr int kHubNumPatterns = 5;

// the G values of one group as one store (rows are 4-byte aligned; the groups start at state 1)
template <int G> struct __attribute__((packed, aligned(4))) HubRow { float v[G]; };

// Split ring of sweep_hub_kernel: the slots below *ks read their emissions from the front part of a row (columns below
// *kf, three LDS buffers of pitch kpf), the others from its back part (two buffers of pitch kpb).  Needs the tokens of a
// group numbered consecutively (column of group g, class c: c0 + g * cs + c), so that a slot's columns are one run.
// The largest split whose parts fit their buffers (64 spare floats each: DMA pieces are whole, the last word is the
// pad); *ks = 0: none.  Host (sizing the buffers) and device (finding the split again) share this function.
__host__ __device__ inline void hub_split(int kpf, int kpb, int K, int c0, int cs, int* ks, int* kf) {
    *ks = 0; *kf = 0;
    for (int k = kHubSlots - 1; k >= 1; --k) {
        const long long f = (long long)c0 + (long long)kHubCompute * k * cs;
        if (f >= K) continue;
        if (((f + 63) / 64) * 64 + 64 <= kpf && ((K - f + 63) / 64) * 64 + 64 <= kpb) { *ks = k; *kf = (int)f; return; }
    }
}

// PAT: index into kHubPatterns (G = 2), or -1: the pattern is only known at run time.
// Segments of the hub sweep: after n of den parts a sweep has done hub_steps frame steps (utterances shorter than
// kHubSegMinT are not cut: their first segment is the whole sweep).  Frame t's posteriors need alpha row t (forward step
// t - 1) and beta row t + 1 (backward step T - 2 - t): with both sweeps s steps in, frames T - s - 1 .. s are ready.
constexpr int kHubSegMinT = 32;
__host__ __device__ inline int hub_steps(int T, int n, int den) {
    if (n <= 0) return 0;
    return (n >= den || T < kHubSegMinT) ? T : (int)((long long)T * n / den);
}
__device__ inline bool hub_seg_has(int T, int t, int n, int den) {
    if (n <= 0) return false;
    const int s_ = hub_steps(T, n, den);
    return t >= T - s_ - 1 && t <= s_;
}

template <int G, int NC, int PAT>
__global__ void __launch_bounds__(kMaxThreads) sweep_hub_kernel(SweepArgs p) {
    constexpr int W = G + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    double* red = (double*)smem;          // [2][16] per-wave partial sums of the hub, by step parity
    int* rexp = (int*)(red + 32);         // [2][16] per-wave exponent maxima
    float* pbuf = (float*)(rexp + 32);    // [nb][kp] log_probs rows of nb consecutive frames (frame f in buffer f % nb)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int u = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = uni((int)U[WF_U_FSA]), T = uni((int)U[WF_U_T]);
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = uni(I[WF_I_S]), K = uni(I[WF_I_K]);
    const bool segmented = p.seg_den > 0;
    const int step0 = segmented ? hub_steps(T, p.seg_n0, p.seg_den) : 0, step1 = segmented ? hub_steps(T, p.seg_n1, p.seg_den) : T;
    const bool first_seg = !segmented || p.seg_n0 == 0;
    if (!first_seg && step0 >= step1) return;   // (an utterance that is not cut: its first segment did everything)
    if (first_seg && dir == 0 && tid == 0) p.flag_clear[u] = 0;
    if (T < uni(I[WF_I_MINLEN])) {   // structurally infeasible: no sweep needed
        if (first_seg && dir == 0 && tid == 0) {
            ((float*)p.tot)[u] = -INFINITY;
            p.totd[u] = (double)-INFINITY;
            p.tot2[u] = (double)-INFINITY;
            p.ctot2[u] = 0.0;
        }
        return;
    }
    const int kp = p.kpad;   // pitch of one probability row in the LDS (>= K)
    const float* lpbase = p.lp + uni(U[WF_U_SEQ]) * p.strideB + uni(U[WF_U_START]) * p.strideT;
    if (tid < 32) { red[tid] = 0.0; rexp[tid] = 0; }

    if (wave >= kHubCompute / 64) {
        // =============================== loader wavefronts ===============================
        // Only loads here, only stores in the compute wavefronts: loads and stores share one in-order counter (vmcnt), so
        // a wavefront that does both waits for its earlier row stores to be acknowledged (~2 us under this load) whenever
        // it needs a loaded value -- measured: 5.2 us per frame step with everything in one wavefront.
        // The log_probs row of a frame goes straight into the LDS (LDS-DMA: 64 consecutive floats per wave-instruction, no
        // registers); with three row buffers frame s + 2 is requested while frame s is being used, and its arrival is
        // awaited a whole frame step later.  Raw s_barrier: __syncthreads() would drain the DMAs in flight.
        const int lw = wave - kHubCompute / 64;
        const int nb = p.chunk;   // whole-row buffers in the LDS (2 or 3)
        // A run of n floats (row columns lo .. lo + n - 1) into an LDS buffer: 16-byte pieces (gfx950: 64 lanes x 4 floats
        // per wave-instruction) over its whole 256-float blocks, 4-byte pieces over the rest -- a quarter of the DMA
        // instructions, and a frame's batch stays far below the 63 requests that vmcnt can count, so that two batches can
        // be in flight.  nreq_of: requests per run of this wavefront.
        const bool only4 = p.ablate & 32;   // (timing aid: 4-byte pieces only)
        auto nreq_of = [&](const int n) {
            const int n16 = only4 ? 0 : n >> 8, nchunk = (n + 63) >> 6;
            int r = 0;
            for (int c = lw; c < n16; c += kHubLoad / 64) ++r;
            for (int c = 4 * n16 + lw; c < nchunk; c += kHubLoad / 64) ++r;
            return uni(r);
        };
        auto issue_run = [&](const float* src, float* dst, const int n) __attribute__((always_inline)) {
            if (p.ablate & 2) return;   // (timing aid: no loads)
            const int n16 = only4 ? 0 : n >> 8, nchunk = (n + 63) >> 6;
            for (int c = lw; c < n16; c += kHubLoad / 64)
                __builtin_amdgcn_global_load_lds(src + 256 * c + 4 * lane, (__attribute__((address_space(3))) float*)(dst + 256 * c), 16, 0, 0);
            for (int c = 4 * n16 + lw; c < nchunk; c += kHubLoad / 64)
                __builtin_amdgcn_global_load_lds(src + min(64 * c + lane, n - 1), (__attribute__((address_space(3))) float*)(dst + 64 * c), 4, 0, 0);
        };
        auto rowp = [&](const int frame) { return lpbase + (int64_t)(dir ? T - 1 - frame : frame) * p.strideT; };
        // waits until at most the newest n requests are outstanding: loads complete in order, so everything requested
        // before them has landed (s_waitcnt takes an immediate: one case per count; vmcnt's low four bits)
        auto wait_but = [&](const int n) __attribute__((always_inline)) {
            switch ((p.ablate & 2) ? 0 : n) {
#define WF_CASE(n) case n: __builtin_amdgcn_s_waitcnt(0x0f70 | n); break;
                WF_CASE(1) WF_CASE(2) WF_CASE(3) WF_CASE(4) WF_CASE(5) WF_CASE(6) WF_CASE(7) WF_CASE(8)
                WF_CASE(9) WF_CASE(10) WF_CASE(11) WF_CASE(12) WF_CASE(13) WF_CASE(14) WF_CASE(15)
#undef WF_CASE
                default: __builtin_amdgcn_s_waitcnt(0x0f70);   // (0, or more than the field holds: everything)
            }
        };
        if (p.hub_kpf > 0) {
            // split ring: the front part of frame s + 2 (three buffers) and the back part of frame s + 1 (two buffers) are
            // requested at the start of step s; the barrier that ends step s waits for all but the former -- most of a
            // row has two frame steps to land, and the batch that has one is small
            int Kf, ksplit;
            hub_split(p.hub_kpf, p.hub_kpb, K, uni(blob[uni(I[WF_I_HUB_C0])]), uni(I[WF_I_HUB_CS]), &ksplit, &Kf);
            float* fbuf = pbuf;
            float* bbuf = pbuf + 3 * (size_t)p.hub_kpf;
            const int nf = nreq_of(Kf);
            // (buffers by the step's number inside this launch's segment: the compute wavefronts count from 0 too)
            auto front = [&](const int fr) __attribute__((always_inline)) { issue_run(rowp(fr), fbuf + (size_t)((fr - step0) % 3) * p.hub_kpf, Kf); };
            auto back = [&](const int fr) __attribute__((always_inline)) { issue_run(rowp(fr) + Kf, bbuf + (size_t)((fr - step0) & 1) * p.hub_kpb, K - Kf); };
            if (step1 > step0) { front(step0); back(step0); }
            if (step1 > step0 + 1) { front(step0 + 1); wait_but(nf); } else __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();         // the initial row is stored, the segment's first frame is staged
            for (int step = step0; step < step1; ++step) {
                if (step + 1 < step1) back(step + 1);
                if (step + 2 < step1) { front(step + 2); wait_but(nf); }
                else __builtin_amdgcn_s_waitcnt(0x0f70);
                if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();
            }
            return;
        }
        const int nreq = nreq_of(K);
        auto issue = [&](const int frame) __attribute__((always_inline)) { issue_run(rowp(frame), pbuf + (size_t)((frame - step0) % nb) * kp, K); };
        if (step1 > step0) issue(step0);
        if (nb == 3) {
            // three row buffers: frame s + 2 is requested at the START of step s (its buffer, frame s - 1's, is free since the
            // last barrier), frame s + 1 -- requested a whole step earlier -- is awaited before the barrier that ends step s:
            // a request has two frame steps to land
            if (step1 > step0 + 1) { issue(step0 + 1); wait_but(nreq); } else __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();         // the initial row is stored, the segment's first frame is staged
            for (int step = step0; step < step1; ++step) {
                if (step + 2 < step1) { issue(step + 2); wait_but(nreq); }
                else __builtin_amdgcn_s_waitcnt(0x0f70);
                if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();
            }
        } else {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();
            for (int step = step0; step < step1; ++step) {
                if (step + 1 < step1) issue(step + 1);
                __builtin_amdgcn_s_waitcnt(0x0f70);
                if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();
            }
        }
        return;
    }
    // =============================== compute wavefronts ===============================
    const int NG = uni(I[WF_I_HUB_NG]), cs = uni(I[WF_I_HUB_CS]), NX = uni(I[WF_I_HUB_NX]);
    const int32_t* c0p = blob + uni(I[WF_I_HUB_C0]);
    const float* cfp = (const float*)(blob + uni(I[WF_I_HUB_COEF]));
    const int32_t* xcol = blob + uni(I[WF_I_HUB_XCOL]);
    // coefficient of class c in "new value of i += coefficient * value of j" (i, j in 0..G, G = the hub): the arcs j -> i in
    // the forward sweep, i -> j in the backward sweep.  Uniform values: they live in scalar registers.
    float cf[W][W][NC];
    unsigned present = 0;
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int e = dir ? (j * W + i) : (i * W + j);
                cf[i][j][c] = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, cfp[e * NC + c])));
                if (cf[i][j][c] != 0.f) present |= 1u << (i * W + j);
            }
    present = (unsigned)uni((int)present);
    int c0[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) c0[c] = uni(c0p[c]);
    const int nslot = (NG + kHubCompute - 1) / kHubCompute;
    // Rows leave as fp32 (same element offsets as the fp64 arenas of the other sweeps: the first half of the region): the
    // values of a row are kept within [2^-64, 2^16) of 1 by the per-frame shift, states more than 2^-126 below the row's
    // maximum are lost to fp32 -- a row that loses mass that way fails the row check and the utterance is redone.
    float* arena = (float*)(dir ? p.beta : p.alpha) + uni(U[dir ? WF_U_BETA : WF_U_ALPHA]);
    double* goff = (dir ? p.boff : p.aoff) + uni(U[WF_U_OFFS]);
    const float finw = __builtin_bit_cast(float, uni(I[WF_I_HUB_FINW]));
    const int xc = NX > 0 ? xcol[min(tid, NX - 1)] : 0;

    // ---- initial row: alpha_0 = delta(hub); beta_T = the hub's final weight
    // (the groups' values are fp32 registers: what the rows hold; the hub's value and its sum are fp64)
    float a[kHubSlots][G];
#pragma unroll
    for (int k = 0; k < kHubSlots; ++k)
#pragma unroll
        for (int i = 0; i < G; ++i) a[k][i] = 0.f;
    double h = dir ? exp((double)finw) : 1.0;
    double cum = 0.0;
    int sh0 = 0;
    if (first_seg) {
        float* row = arena + (int64_t)(dir ? T : 0) * S;
        for (int s2 = tid; s2 < S; s2 += kHubCompute) row[s2] = s2 == 0 ? (float)h : 0.f;
        if (tid == 0) goff[0] = 0.0;
        if (dir == 0) {   // c_t = 0 for the kernels that read the per-frame normaliser
            double* crow = p.cmax_w + uni(U[WF_U_ROW0]);
            for (int t = tid; t < T; t += kHubCompute) crow[t] = 0.0;
        }
    } else {
        // a later segment: the groups' values are the last row the segment before wrote (fp32 registers left as they were),
        // the rest waits in hubsave
        const float* row = arena + (int64_t)(dir ? T - step0 : step0) * S;
#pragma unroll
        for (int k = 0; k < kHubSlots; ++k) {
            const int g = tid + k * kHubCompute;
#pragma unroll
            for (int i = 0; i < G; ++i) a[k][i] = g < NG ? row[1 + g * G + i] : 0.f;
        }
        const double* sv = p.hubsave + (size_t)(2 * u + dir) * 4;
        h = sv[0];
        cum = sv[1];
        sh0 = uni((int)sv[2]);
    }
    // LDS offsets of this lane's emissions inside a row buffer, per owned group and class; a lane without a group in its last
    // slot reads the buffer's pad word (-inf: probability 0), so that its values stay 0 with no masking in the frame loop
    const int nb = p.chunk;
    // split ring (see the loader wavefronts): slots >= ksplit read columns >= Kf from the back buffers
    int ksplit = kHubSlots, Kf = 0;
    const int kpf = p.hub_kpf > 0 ? p.hub_kpf : kp, kpb = p.hub_kpb;
    if (p.hub_kpf > 0) hub_split(p.hub_kpf, p.hub_kpb, K, c0[0], cs, &ksplit, &Kf);
    const int nfb = p.hub_kpf > 0 ? 3 : nb;
    const float* bbuf = pbuf + 3 * (size_t)kpf - Kf;   // (indexed by column)
    int poff[kHubSlots][NC];
#pragma unroll
    for (int k = 0; k < kHubSlots; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int g = tid + k * kHubCompute;
            poff[k][c] = g < NG ? c0[c] + g * cs : (k < ksplit ? kpf - 1 : Kf + kpb - 1);
        }
    // (the DMA pieces end before a buffer's pad word)
    if (tid < nfb) pbuf[(size_t)tid * kpf + kpf - 1] = -INFINITY;
    if (p.hub_kpf > 0 && tid < 2) pbuf[3 * (size_t)kpf + (size_t)tid * kpb + kpb - 1] = -INFINITY;
    __syncthreads();   // the segment's first frame is staged
    int sh_end = 0;
    // the frame loop, once per direction at compile time (TR: the backward sweep uses the transposed pattern)
    auto frames = [&](const auto tr) __attribute__((always_inline)) {
        constexpr bool TR = decltype(tr)::value;
        int sh = sh0, ib = 0;
        for (int step = step0; step < step1; ++step) {
            const float* pc = pbuf + (size_t)ib * kpf;
            const float* pcb = bbuf + (size_t)((step - step0) & 1) * kpb;   // (read by no slot unless the ring is split)
            ib = ib + 1 == nfb ? 0 : ib + 1;
            const int f = TR ? T - 1 - step : step;
            float* srow = arena + ((p.ablate & 1) ? (int64_t)0 : (int64_t)(TR ? f : f + 1) * S);   // (ablate: timing aid)
            cum += (double)sh;
            if (tid == 0) goff[step + 1] = cum;
            double part = 0.0;      // this lane's share of the hub's new value
            double selfd = 0.0;     // sum of the probabilities on hub self-loops (times the hub's value below)
            int emine = 0;
            const int shc = min(max(sh, -16), 64);   // rides in the exp2 argument (fp32: no underflow for lp > -43, no overflow)
            const int shr = sh - shc;                // the rest, by ldexp (uniform, rare: the first frames)
            const float shf = (float)shc;
            // Fast frame body (known pattern, the usual case shr == 0, the slot count NS a compile-time constant): ONE basic
            // block for all slots -- every emission is requested from the LDS before the first is used, no uniform branch
            // per slot (`k < nslot`, `shr`) and no exec mask but on the last slot's store (slots below it are full in
            // every lane), so the scheduler overlaps the slots' LDS reads and exp2s.  All of a group's arithmetic is fp32
            // (the state values are fp32 registers and leave as they are: no conversion on the way in or out -- the
            // conversions were four of a slot's six quarter-rate instructions); a lane's share of the hub's sum goes to
            // fp64 once per frame.  The row's exponent comes from the stored values' bit patterns (positive floats order
            // like unsigned integers: one max per slot).
            auto fast = [&](const auto nst) __attribute__((always_inline)) {
                constexpr int NS = decltype(nst)::value;
                constexpr HubPattern pat = kHubPatterns[PAT >= 0 ? PAT : 0];
                float lv[NS][NC];
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int c = 0; c < NC; ++c) lv[k][c] = (k < ksplit ? pc : pcb)[poff[k][c]];
                const float h32 = (float)h;
                float part32 = 0.f, self32 = 0.f;
                unsigned eb = 0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    float pr[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) pr[c] = __builtin_amdgcn_exp2f(fmaf(lv[k][c], kLog2e, shf));
                    float nv[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) nv[i] = 0.f;
#pragma unroll
                    for (int e = 0; e < pat.n; ++e) {
                        const int ii = pat.e[e] >> 4, jj = (pat.e[e] >> 2) & 3, c = pat.e[e] & 3;
                        const int i = TR ? jj : ii, j = TR ? ii : jj;
                        if (i == G && j == G) self32 += pr[c < NC ? c : 0];
                        else {
                            const float vj = j == G ? h32 : a[k][j < G ? j : 0];
                            if (i == G) part32 = fmaf(pr[c < NC ? c : 0], vj, part32);
                            else nv[i < G ? i : 0] = fmaf(pr[c < NC ? c : 0], vj, nv[i < G ? i : 0]);
                        }
                    }
                    HubRow<G> v;
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        a[k][i] = nv[i];
                        v.v[i] = nv[i];
                        eb = max(eb, __builtin_bit_cast(unsigned, nv[i]));
                    }
                const int g = tid + k * kHubCompute;
                    if (k < NS - 1 || g < NG) {
                        // (streaming: the rows are read again only by the posterior kernel, after the whole sweep)
                        if constexpr (G == 2) {
                            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                            f2u w2 = {v.v[0], v.v[1]};
                            __builtin_nontemporal_store(w2, (f2u*)(srow + 1 + g * G));
                        } else {
#pragma unroll
                            for (int i = 0; i < G; ++i) __builtin_nontemporal_store(v.v[i], srow + 1 + g * G + i);
                        }
                    }
                }
                part += (double)part32;
                selfd += (double)self32;
                const int e32 = (int)(eb >> 23);
                emine = e32 > 0 ? e32 + (1023 - 127) : 0;
            };
            bool done = false;
            if (PAT >= 0 && shr == 0 && !(p.ablate & 64)) {
                done = true;
                switch (nslot) {
                    case 1: fast(std::integral_constant<int, 1>{}); break;
                    case 2: fast(std::integral_constant<int, 2>{}); break;
                    case 3: fast(std::integral_constant<int, 3>{}); break;
                    case 4: fast(std::integral_constant<int, 4>{}); break;
                    case 5: fast(std::integral_constant<int, 5>{}); break;
                    case 6: fast(std::integral_constant<int, 6>{}); break;
                    case 7: fast(std::integral_constant<int, 7>{}); break;
                    default: done = false;
                }
            }
            static_assert(kHubSlots == 7, "the switch above lists the slot counts");
#pragma unroll
            for (int k = 0; k < kHubSlots; ++k) {
                if (k < nslot && !done) {
                    // exp(lp) * 2^shf: the row's rescaling rides in the exponent argument
                    float pr[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) pr[c] = __builtin_amdgcn_exp2f(fmaf((k < ksplit ? pc : pcb)[poff[k][c]], kLog2e, shf));
                    double nv[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) nv[i] = 0.0;
                    if (PAT >= 0) {
                        constexpr HubPattern pat = kHubPatterns[PAT >= 0 ? PAT : 0];
                        double prd[NC];
#pragma unroll
                        for (int c = 0; c < NC; ++c) prd[c] = (double)pr[c];
#pragma unroll
                        for (int e = 0; e < pat.n; ++e) {
                            const int ii = pat.e[e] >> 4, jj = (pat.e[e] >> 2) & 3, c = pat.e[e] & 3;
                            const int i = TR ? jj : ii, j = TR ? ii : jj;
                            if (i == G && j == G) selfd += prd[c < NC ? c : 0];
                            else {
                                const double vj = j == G ? h : (double)a[k][j < G ? j : 0];
                                if (i == G) part = fma(prd[c < NC ? c : 0], vj, part);
                                else nv[i < G ? i : 0] = fma(prd[c < NC ? c : 0], vj, nv[i < G ? i : 0]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < W; ++i)
#pragma unroll
                            for (int j = 0; j < W; ++j) {
                                if (present & (1u << (i * W + j))) {   // uniform
                                    float cp = cf[i][j][0] * pr[0];
#pragma unroll
                                    for (int c = 1; c < NC; ++c) cp = fmaf(cf[i][j][c], pr[c], cp);
                                    if (i == G && j == G) selfd += (double)cp;
                                    else {
                                        const double vj = j == G ? h : (double)a[k][j < G ? j : 0];
                                        if (i == G) part = fma((double)cp, vj, part);
                                        else nv[i < G ? i : 0] = fma((double)cp, vj, nv[i < G ? i : 0]);
                                    }
                                }
                            }
                    }
                    if (shr) {   // a real branch (the empty asm keeps the compiler from turning it into an unconditional ldexp + select)
                        asm volatile("");
#pragma unroll
                        for (int i = 0; i < G; ++i) nv[i] = ldexp(nv[i], shr);
                    }
                    HubRow<G> v;
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        a[k][i] = v.v[i] = (float)nv[i];
                        emine = max(emine, f64_exp_field(nv[i]));
                    }
                    const int g = tid + k * kHubCompute;
                    if (g < NG) *(HubRow<G>*)(srow + 1 + g * G) = v;
                }
            }
            if (tid < NX) selfd += (double)__builtin_amdgcn_exp2f(fmaf((xc < Kf || ksplit == kHubSlots ? pc : pcb)[xc], kLog2e, shf));
            for (int e = tid + kHubCompute; e < NX; e += kHubCompute) {   // (more hub self-loop labels than compute lanes: no reference topology has them)
                const int c2 = xcol[e];
                selfd += (double)__builtin_amdgcn_exp2f(fmaf((c2 < Kf || ksplit == kHubSlots ? pc : pcb)[c2], kLog2e, shf));
            }
            part = fma(h, selfd, part);
            if (shr) { asm volatile(""); part = ldexp(part, shr); }
            part = wave_sum_d(part);
            emine = wave_max_i(emine);
            const int par = (step & 1) * 16;
            if (lane == 0) { red[par + wave] = part; rexp[par + wave] = emine; }
            if (!(p.ablate & 4)) __syncthreads();   // (ablate: timing aid)
            static_assert(kHubCompute / 64 == 12, "the reduction below sums twelve partial values");
            double r[12];
            int ex[12];
#pragma unroll
            for (int w = 0; w < 12; ++w) { r[w] = red[par + w]; ex[w] = rexp[par + w]; }
            // a fixed tree (the same in every lane: the hub's value is uniform), four additions deep
            const double hn = (((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))) + ((r[8] + r[9]) + (r[10] + r[11]));
            int em = max(max(max(ex[0], ex[1]), max(ex[2], ex[3])), max(max(ex[4], ex[5]), max(ex[6], ex[7])));
            em = max(em, max(max(ex[8], ex[9]), max(ex[10], ex[11])));
            h = hn;
            em = max(em, f64_exp_field(hn));
            sh = uni(em > 0 ? 1023 - em : 0);   // next row: maximum back to [1, 2)
            if (tid == 0) srow[0] = (float)hn;
        }
        sh_end = sh;
    };
    if (dir) frames(std::true_type{}); else frames(std::false_type{});
    if (step1 < T) {   // the next segment continues from here
        if (tid == 0) {
            double* sv = p.hubsave + (size_t)(2 * u + dir) * 4;
            sv[0] = h; sv[1] = cum; sv[2] = (double)sh_end;
        }
        return;
    }
    if (dir == 0 && tid == 0) {
        const double fsum = h * exp((double)finw);
        const bool dead = !(fsum > 0.0);
        const double tot2 = dead ? (double)-INFINITY : log2(fsum) - cum;
        const double tot = dead ? (double)-INFINITY : tot2 * kLn2;
        // (a cut utterance: the posterior kernels have been normalising with hub_tot_kernel's tot2 since the sweeps met;
        // it stays -- unless this end of the sweep says that nothing gets through at all)
        if (first_seg || dead) { p.tot2[u] = tot2; p.ctot2[u] = 0.0; }
        p.totd[u] = tot;
        ((float*)p.tot)[u] = (float)tot;
    }
}

// The total of a cut hub sweep as soon as both directions have passed the middle frame: sum_s alpha_t[s] beta_t[s] (every
// path crosses exactly one state at time t), from the fp32 rows and their cumulative shifts.  One workgroup per utterance.
struct HubTotArgs {
    const int32_t* blob;
    const int64_t* utt;
    const void* alpha;
    const void* beta;
    const double* aoff;
    const double* boff;
    double* tot2;
    double* ctot2;
    int seg_n, seg_den;
};
__global__ void __launch_bounds__(256) hub_tot_kernel(HubTotArgs p) {
    __shared__ double s_part[4];
    const int u = blockIdx.x, tid = threadIdx.x;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int T = (int)U[WF_U_T], S = I[WF_I_S];
    if (T < kHubSegMinT || T < I[WF_I_MINLEN]) return;   // not cut (its first segment wrote the total) / infeasible
    const int t = T / 2;   // hub_steps(T, seg_n) >= t >= T - hub_steps(T, seg_n): seg_n / seg_den >= 1/2 + 1/kHubSegMinT
    const float* arow = (const float*)p.alpha + U[WF_U_ALPHA] + (int64_t)t * S;
    const float* brow = (const float*)p.beta + U[WF_U_BETA] + (int64_t)t * S;
    // (the hub and the groups' states: what the sweeps write -- a row's slot of the super-final state is never touched)
    const int nrow = min(S, 1 + I[WF_I_HUB_NG] * I[WF_I_HUB_G]);
    double acc = 0.0;
    for (int s2 = tid; s2 < nrow; s2 += 256) acc = fma((double)arow[s2], (double)brow[s2], acc);
    acc = wave_sum_d(acc);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        const double sum = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        // alpha row t came from forward step t - 1, beta row t from backward step T - 1 - t (one shift per step)
        const double ga = p.aoff[U[WF_U_OFFS] + lin_shift_index(t - 1, 1, 0)], gb = p.boff[U[WF_U_OFFS] + lin_shift_index(T - 1 - t, 1, 0)];
        p.tot2[u] = sum > 0.0 ? log2(sum) - ga - gb : (double)-INFINITY;
        p.ctot2[u] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// Banded wave sweep: the linear-domain forward-backward recursion of a numerator graph with NO LDS
// round trip of the state vector and NO barrier on the critical path.
//
// Every topo o transcript graph has a topological order of its states in which each arc goes from
// position p - d to position p with 0 <= d <= BW <= 4 (wf_graph.cpp, WF_I_BAND_*).  ONE wavefront (the
// chain wave) then carries the whole state vector in registers, R consecutive positions per lane:
// the sources of a state are in the same lane or in lane - 1, fetched with a wave_shr:1 DPP move, and
// a frame step is BW DPP moves + R*(BW+1) fp64 FMAs.  The arc probabilities come from emission rows
// that a second wavefront (the stager) copies, coalesced and two chunks of C frames ahead, from the
// workspace into an LDS double buffer; the chain wave reads P_t[column] of its R*(BW+1) cells from
// there one step ahead of their use.  Chain and stager meet at one barrier per chunk.  The alpha rows
// leave the chain wave as plain stores (never waited for).  The backward sweep is the same code on
// the reversed order.  Rescaling (once per chunk, decided by the stager from the exponents of the
// previous chunk's last row) and the row check / fallback are as for sweep_lin_kernel.
// Measured (cfg3, MI355X): 190 cycles per frame step against 440 for sweep_lin_kernel.
// ------------------------------------------------------------------------------------------
constexpr int kBandKP = 256;   // row pitch of the staged emission rows; a banded graph uses < kBandKP label columns (two 128-column pieces per row)
constexpr int kBandC = 16;     // frames per chunk

// wave_shr:1 of a double: lane i receives lane i-1's value, lane 0 receives 0 (bound_ctrl)
__device__ __forceinline__ double wave_shr1_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr size_t kBandPBytes = (size_t)kBandC * kBandKP * 8;          // one chunk of staged emission rows
constexpr size_t kBandLdsUsed = 2 * kBandPBytes + 2 * 64 * 4 + 16;    // + per-lane exponents (x2) + shift words
// Requested: more than half of the CU's 160 KiB, so that no two sweep workgroups share a CU (two chain waves
// on one SIMD would halve each other's issue rate; there are only 2B workgroups for 256 CUs)
constexpr size_t kBandMBytes = (size_t)kBandC * 64 * 8;               // one chunk of staged keep-mask words (MASK)
constexpr size_t kBandMaskOff = (kBandLdsUsed + 15) & ~(size_t)15;
constexpr size_t kBandLdsMask = kBandMaskOff + 2 * kBandMBytes;
constexpr size_t kBandLds = kBandLdsMask > 84 * 1024 ? kBandLdsMask : 84 * 1024;
constexpr int kBandMaskKMax = kBandKP - 4;   // MASK: columns shifted by two, the pair store of an odd K's pad column included

// MM: arcs per (position, distance) cell (1: deterministic topologies); WE: every arc weight is 0 and
// MM == 1, so a cell's probability is the staged emission itself (no multiply)
// GEN (R = BW = MM = 1): ANY graph of at most 64 non-final states whose states have <= WF_REG_DEG arcs, but for one
// state of up to 64 (the small denominators: cfg2's 2state topology, S = 60, hub fan-in 59).  lane = state; a lane
// fetches its sources' values with ds_bpermute instead of the DPP shift, the hub's arcs sit one per lane and are
// summed with the DPP tree.  Stager, chunking, rescaling and the total are the banded kernel's.
// MASK (output_beam pruning, see trop_band_kernel / band_mask_kernel below): the stager also stages one 64-bit word per
// lane and frame whose bit j*MM + m says whether the lattice arc of cell (j, m) survives the beam; a pruned cell reads
// the zero that sits at offset 0 of every staged row (its LDS address is ANDed with the sign-extended bit: two VALU
// instructions per cell, off the dependent chain), so the staged columns start 16 bytes into the row.
template <int R, int BW, int MM, bool WE, bool GEN = false, bool MASK = false>
__global__ void __launch_bounds__(128) sweep_band_kernel(SweepArgs p) {
    constexpr int GD = WF_REG_DEG;
    constexpr int NQ = GEN ? GD + 1 : R * (BW + 1), C = kBandC;
    constexpr unsigned ZSH = MASK ? 16u : 0u;   // byte offset of column 0 inside a staged row
    static_assert(NQ * MM <= 64, "one 64-bit mask word per lane and frame");
    static_assert(MM >= 1 && MM <= WF_BAND_MAX_M && (!WE || MM == 1), "cell capacity");
    static_assert(BW <= R, "sources must sit in the same lane or the previous one");
    static_assert(!GEN || (R == 1 && BW == 1 && MM == 1), "general-graph mode: one state per lane");
    extern __shared__ __align__(16) unsigned char smem[];
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    if (p.tspan && threadIdx.x == 0) atomicMin(p.tspan, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    // LDS: staged emission rows [2][C][kBandKP] x 8 B | 2 x 64 exponent words | 2 shift words
    int* ebuf = (int*)(smem + 2 * kBandPBytes);
    int* shw = ebuf + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= p.nsweep) {
        // fill role: these workgroups are dispatched after the 2B sweep workgroups and, like them, own their CU (the
        // kernel's LDS request), so no fill wave competes with a chain wave for issue slots
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        const size_t stride = (size_t)(gridDim.x - p.nsweep) * 128;
        for (size_t i = (size_t)(blockIdx.x - p.nsweep) * 128 + tid; i < p.zn4; i += stride)
            __builtin_nontemporal_store(z, (v4f*)p.zp + i);
        if ((int)blockIdx.x == p.nsweep && tid < p.zntail) p.ztail[tid] = 0.f;
        if (p.tspan) {
            __builtin_amdgcn_s_waitcnt(0);   // this workgroup's stores have been acknowledged
            if (tid == 0) atomicMax(p.tspan + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
        }
        return;
    }
    const int u = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = uni((int)U[WF_U_FSA]), T = uni((int)U[WF_U_T]);
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = uni(I[WF_I_S]), A = uni(I[WF_I_A]), K = uni(I[WF_I_K]);
    if (T < uni(I[WF_I_MINLEN])) {   // structurally infeasible: no sweep needed
        if (dir == 0 && tid == 0) {
            ((float*)p.tot)[u] = -INFINITY;
            p.totd[u] = (double)-INFINITY;
            p.tot2[u] = (double)-INFINITY;
            p.ctot2[u] = 0.0;
        }
        return;
    }
    const int Sn = S - 1;   // positions: every state but the super-final one
    const int nc = (T + C - 1) / C;
    if (tid < 2) shw[tid] = tid == 0 ? kLinBias : 0;   // chunk 0: the initial vector's maximum is 1; chunk 1: nothing measured yet

    if (wave == 0) {
        // =============================== chain wave ===============================
        const int gbw = uni(I[WF_I_BAND_BW]), gm = uni(I[WF_I_BAND_M]);
        const int32_t* order = blob + uni(I[WF_I_BAND_ORDER]);
        const int32_t* tab = blob + uni(I[dir ? WF_I_BANDB : WF_I_BANDF]);
        const float* fb = (const float*)(blob + uni(I[WF_I_FINB_LOG]));
        double* arena = (double*)(dir ? p.beta : p.alpha) + uni(U[dir ? WF_U_BETA : WF_U_ALPHA]);
        double* goff = (dir ? p.boff : p.aoff) + uni(U[WF_U_OFFS]);
        double a[R];
        int sidx[R];   // column of the state inside an alpha row; padding positions write the (unused)
                       // super-final column, so that no store needs an exec mask
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int pp = lane * R + r;
            int st = -1;
            if (pp < Sn) st = GEN ? pp : order[dir ? Sn - 1 - pp : pp];
            a[r] = st < 0 ? 0.0 : (dir ? exp((double)fb[st]) : (st == 0 ? 1.0 : 0.0));
            sidx[r] = st < 0 ? S - 1 : st;
            arena[(int64_t)(dir ? T : 0) * S + sidx[r]] = a[r];   // alpha_0 / beta_T
        }
        // this lane's cells: byte offset of the arc's column inside a staged row (empty cell: column K,
        // which the stager keeps at 0) and exp(weight)
        unsigned coff[NQ][MM];
        double cw[NQ][MM];
        int gsrc[NQ];   // GEN: the lane (= state) an arc comes from; slot GD is this lane's arc of the hub state
        int over = -1;
        if (GEN) {
            const int32_t* g_off = blob + uni(I[dir ? WF_I_OUT_OFF : WF_I_IN_OFF]);
            const int32_t* g_oth = blob + uni(I[dir ? WF_I_OUT_DST : WF_I_IN_SRC]);
            const int32_t* g_col = blob + uni(I[dir ? WF_I_OUT_COL : WF_I_IN_COL]);
            const float* g_w = (const float*)(blob + uni(I[dir ? WF_I_OUT_W : WF_I_IN_W]));
            if (uni(I[dir ? WF_I_NOVER_OUT : WF_I_NOVER_IN]) > 0) over = uni(I[dir ? WF_I_OVER_OUT : WF_I_OVER_IN]);
            int a0 = 0, deg = 0;
            if (lane < Sn && lane != over) { a0 = g_off[lane]; deg = g_off[lane + 1] - a0; }
            int h0 = 0, h1 = 0;
            if (over >= 0) { h0 = g_off[over]; h1 = g_off[over + 1]; }
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int arc = j < GD ? a0 + j : h0 + lane;
                const bool on = j < GD ? j < deg : h0 + lane < h1;
                gsrc[j] = on ? g_oth[arc] : 0;
                coff[j][0] = MASK ? (on ? (unsigned)g_col[arc] * 8u + ZSH : 0u) : (unsigned)(on ? g_col[arc] : K) * 8u;
                cw[j][0] = on ? exp((double)g_w[arc]) : 0.0;
            }
        }
#pragma unroll
        for (int j = 0; j < NQ && !GEN; ++j) {
            const int r = j / (BW + 1), d = j % (BW + 1);
            const int pp = lane * R + r;
            const bool cell = pp < Sn && d <= gbw;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                int col = K;
                float w = 0.f;
                if (cell && m < gm) {
                    const int32_t* ent = tab + (((size_t)pp * (gbw + 1) + d) * gm + m) * 2;
                    w = __int_as_float(ent[1]);
                    col = (w == -INFINITY) ? K : ent[0];   // an unused slot of the cell
                    if (w == -INFINITY) w = 0.f;
                }
                coff[j][m] = MASK ? (col == K ? 0u : (unsigned)col * 8u + ZSH) : (unsigned)col * 8u;
                cw[j][m] = exp((double)w);
            }
        }
        if (lane == 0) goff[0] = 0.0;
        double cum = 0.0;   // cumulative power-of-two shift (log2 units) carried by the rows
        // row written by step 0: alpha_1 / beta_{T-1}; then +-S doubles per step
        double* srow = arena + (int64_t)(dir ? (T - 1) : 1) * S;
        const int64_t sstep = dir ? -(int64_t)S : (int64_t)S;
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
        // q[j] = probability of cell j in the frame at row k of the staged chunk
        // mb: the chunk's staged mask words (MASK), row k = this frame's
        auto load_q = [&](const unsigned char* pb, const unsigned char* mb, const int k, double (&q)[NQ]) __attribute__((always_inline)) {
            int mlo = 0, mhi = 0;
            if (MASK) {
                mlo = *(const int*)(mb + (size_t)k * 512 + lane * 8);
                if (NQ * MM > 32) mhi = *(const int*)(mb + (size_t)k * 512 + lane * 8 + 4);
            }
            // address of cell (j, m)'s value: a pruned cell reads the zero at offset 0 of the row.  MASK: the row's offset
            // from the start of the LDS is a multiple of the row size, so it is OR-ed in (v_bfe_i32 + v_and_or_b32: two
            // instructions per cell; an add behind the AND would be a third)
            const unsigned rowoff = (unsigned)(pb - smem);
            auto val = [&](const int j, const int m) __attribute__((always_inline)) -> double {
                if (!MASK) return *(const double*)(pb + (size_t)k * kBandKP * 8 + coff[j][m]);
                const int bit = j * MM + m;
                const unsigned o = (coff[j][m] & (unsigned)__builtin_amdgcn_sbfe(bit < 32 ? mlo : mhi, bit & 31, 1)) | rowoff;
                return *(const double*)(smem + o + (size_t)k * kBandKP * 8);
            };
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (WE) {
                    q[j] = val(j, 0);
                } else {
                    double v = cw[j][0] * val(j, 0);
#pragma unroll
                    for (int m = 1; m < MM; ++m) v = fma(cw[j][m], val(j, m), v);
                    q[j] = v;
                }
            }
        };
        auto step = [&](const double (&q)[NQ]) __attribute__((always_inline)) {
            if (GEN) {
                double v = bperm_d(a[0], gsrc[0]) * q[0];
#pragma unroll
                for (int j = 1; j < GD; ++j) v = fma(bperm_d(a[0], gsrc[j]), q[j], v);
                const double hub = wave_sum_d(bperm_d(a[0], gsrc[GD]) * q[GD]);   // 0 without a hub state
                if (lane == over) v = hub;
                a[0] = v;
                srow[sidx[0]] = v;
                srow += sstep;
                return;
            }
            double nb[BW];   // nb[j]: the state at position lane*R - 1 - j (slot R-1-j of lane - 1)
#pragma unroll
            for (int j = 0; j < BW; ++j) nb[j] = wave_shr1_d(a[R - 1 - j]);
            double an[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double v = 0.0;
#pragma unroll
                for (int d = 0; d <= BW; ++d) {
                    const double sv = (r - d >= 0) ? a[r - d >= 0 ? r - d : 0] : nb[d - r - 1 >= 0 ? d - r - 1 : 0];
                    v = (d == 0) ? sv * q[r * (BW + 1)] : fma(sv, q[r * (BW + 1) + d], v);
                }
                an[r] = v;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a[r] = an[r];
                srow[sidx[r]] = an[r];
            }
            srow += sstep;
        };
        auto open_chunk = [&](const int c) __attribute__((always_inline)) {
            // rescale by the power of two the stager derived from the row two chunks back
            const int sh = uni(shw[c & 1]);
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = ldexp(a[r], sh);
            cum += (double)sh;
            if (lane == 0) goff[c + 1] = cum;
        };
        const int nfull = T / C;
        for (int c = 0; c < nfull; ++c) {   // full chunks: straight-line steps, probabilities read one step ahead
            __syncthreads();   // the rows of chunk c are staged
            const unsigned char* pb = smem + (size_t)(c & 1) * kBandPBytes;
            const unsigned char* mb = smem + kBandMaskOff + (size_t)(c & 1) * kBandMBytes;
            double q0[NQ], q1[NQ];
            load_q(pb, mb, 0, q0);
            open_chunk(c);
#pragma unroll
            for (int k = 0; k < C; k += 2) {
                load_q(pb, mb, k + 1, q1);
                step(q0);
                if (k + 2 < C) load_q(pb, mb, k + 2, q0);
                step(q1);
            }
            {   // this lane's largest exponent, for the stager's next rescale decision
                int e = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) e = max(e, f64_exp_field(a[r]));
                ebuf[(c & 1) * 64 + lane] = e;
            }
        }
        if (nfull < nc) {   // the last, partial chunk
            __syncthreads();
            const unsigned char* pb = smem + (size_t)(nfull & 1) * kBandPBytes;
            const unsigned char* mb = smem + kBandMaskOff + (size_t)(nfull & 1) * kBandMBytes;
            open_chunk(nfull);
            const int nk = T - nfull * C;
            for (int k = 0; k < nk; ++k) {
                double q0[NQ];
                load_q(pb + (size_t)k * kBandKP * 8, mb + (size_t)k * 512, 0, q0);
                step(q0);
            }
        }
        __syncthreads();
        if ((p.ablate & 8) && (blockIdx.x == 0 || (p.ablate & 16)) && lane == 0)
            printf("sweep_band<%d,%d,%d,%d> wg %d: T=%d cycles/step=%.1f setup=%llu cycles total=%llu\n", R, BW, MM, (int)WE, (int)blockIdx.x, T, (double)(__builtin_amdgcn_s_memtime() - clk0) / max(T, 1), clk0 - tstart, __builtin_amdgcn_s_memtime() - tstart);
        if (dir == 0) {
            // total = sum over positions of alpha_T * exp(final weight), then undo the scaling
            double part = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int pp = lane * R + r;
                if (pp < Sn) part = fma(a[r], exp((double)fb[GEN ? pp : order[pp]]), part);
            }
            double csum = 0.0;
            const double* crow = p.cmax + uni(U[WF_U_ROW0]);
            for (int t = lane; t < T; t += 64) csum += crow[t];
            const double fsum = wave_sum_d(part);
            const double ctot2 = wave_sum_d(csum) * kLog2eD;
            if (lane == 0) {
                const bool dead = !(fsum > 0.0);
                const double tot2 = dead ? (double)-INFINITY : log2(fsum) - cum + ctot2;
                const double tot = dead ? (double)-INFINITY : tot2 * kLn2;
                p.tot2[u] = tot2;
                p.ctot2[u] = ctot2;
                p.totd[u] = tot;
                ((float*)p.tot)[u] = (float)tot;
            }
        }
    } else {
        // =============================== stager wave ===============================
        // a graph without arcs or an empty segment has no emission rows: read the utterance table (small
        // integers = finite tiny doubles) through a zero pitch instead, so that every load stays unconditional
        const bool has_p = A > 0 && T > 0;
        const double* pbase = has_p ? p.P + uni(U[WF_U_P]) : (const double*)U;
        const int64_t ppitch = has_p ? uni((int)U[WF_U_PP]) : 0;
        const int last_step = max(T - 1, 0);
        constexpr int KVMAX = (kBandKP + 127) / 128;   // 128-column pieces per row: one 16-byte load per lane each
        int lcol[KVMAX];   // first of this lane's two columns of each piece, clamped into the row: every load is
                           // unconditional (a load under a run-time condition makes hipcc branch around it and
                           // drain vmcnt per element)
#pragma unroll
        for (int v = 0; v < KVMAX; ++v) lcol[v] = has_p ? min(128 * v + 2 * lane, (int)ppitch - 2) : 0;   // even: 16-byte aligned; the pad column of an odd K is readable
        typedef double d2 __attribute__((ext_vector_type(2)));
        unsigned loff[KVMAX];   // byte offset of the lane's pair inside a row
#pragma unroll
        for (int v = 0; v < KVMAX; ++v) loff[v] = (unsigned)lcol[v] * 8u;
        // Two register sets: the rows of a chunk are loaded two chunks before they are staged (the measured
        // load-to-use latency here is ~2 us, more than one chunk of the chain wave's time).
        d2 preA[C][KVMAX], preB[C][KVMAX];
        // MASK: this lane's keep-mask word of every frame of the chunk rides along (band_mask_kernel wrote them, one row
        // of 64 words per frame and direction); without rows to read (no arcs / empty segment) the table again
        constexpr int CM = MASK ? C : 1;
        unsigned long long mkA[CM], mkB[CM];
        const unsigned long long* mbase = nullptr;
        if (MASK) mbase = has_p ? p.bmask + ((int64_t)dir * p.bmask_rows + uni(U[WF_U_ROW0])) * 64 + lane : (const unsigned long long*)U;
        const int64_t mpitch = has_p ? 64 : 0;
        auto issue = [&](const int c, d2 (&pre)[C][KVMAX], unsigned long long (&mk)[CM]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int s = min(c * C + k, last_step);   // frames past the end repeat the last one
                const char* row = (const char*)(pbase + (int64_t)(dir ? T - 1 - s : s) * ppitch);   // scalar
#pragma unroll
                for (int v = 0; v < KVMAX; ++v) pre[k][v] = *(const d2*)(row + loff[v]);
                if (MASK) mk[k] = mbase[(int64_t)(dir ? T - 1 - s : s) * mpitch];
            }
        };
        // registers -> staged rows of chunk c (column K of every row is the zero that empty cells read)
        auto stage = [&](const int c, const d2 (&pre)[C][KVMAX], const unsigned long long (&mk)[CM]) __attribute__((always_inline)) {
            unsigned char* pb = smem + (size_t)(c & 1) * kBandPBytes;
            // whole pairs (16-byte writes; an odd K's last pair carries the unused pad column), piece by piece so
            // that the exec mask changes once per piece; then the zero column, which overwrites that pad
            // (MASK: the columns start ZSH bytes into the row and the zero sits at offset 0)
#pragma unroll
            for (int v = 0; v < KVMAX; ++v) {
                if (128 * v + 2 * lane < K) {   // not a clamped duplicate
#pragma unroll
                    for (int k = 0; k < C; ++k) *(d2*)(pb + (size_t)k * kBandKP * 8 + ZSH + loff[v]) = pre[k][v];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < C; ++k) *(double*)(pb + ((size_t)k * kBandKP + (MASK ? 0 : K)) * 8) = 0.0;
            }
            if (MASK) {
                unsigned char* mb = smem + kBandMaskOff + (size_t)(c & 1) * kBandMBytes;
#pragma unroll
                for (int k = 0; k < C; ++k) *(unsigned long long*)(mb + (size_t)k * 512 + lane * 8) = mk[k];
            }
        };
        auto decide = [&](const int c) __attribute__((always_inline)) {
            // ebuf holds the exponents at the end of chunk c-1, before the shift shw[c&1] that chunk c
            // applies: decide the shift of chunk c+1 so that the maximum returns to 2^kLinBias
            const int e = wave_max_i(ebuf[((c - 1) & 1) * 64 + lane]);
            const int sh_c = uni(shw[c & 1]);
            const int sh = e > 0 ? (1023 + kLinBias) - (e + sh_c) : 0;
            if (lane == 0) shw[(c + 1) & 1] = sh;
        };
        // Schedule (iteration c runs between barrier c and barrier c+1, beside the chain's chunk c): stage(c+1)
        // from the register set of its parity (loaded during iteration c-2), then issue(c+3) into the same set.
        // Even chunks use set A, odd chunks set B.
        if (nc > 0) {
            issue(0, preA, mkA);
            if (nc > 1) issue(1, preB, mkB);
            stage(0, preA, mkA);
            if (nc > 2) issue(2, preA, mkA);
        }
        unsigned long long t_bar = 0, t_stage = 0, t_issue = 0;
        for (int c = 0; c <= nc; c += 2) {
            const unsigned long long ta = __builtin_amdgcn_s_memtime();
            __syncthreads();   // barrier c (c even)
            const unsigned long long tb = __builtin_amdgcn_s_memtime();
            t_bar += tb - ta;
            if (c + 1 < nc) {
                stage(c + 1, preB, mkB);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                const unsigned long long tc = __builtin_amdgcn_s_memtime();
                if (c + 3 < nc) issue(c + 3, preB, mkB);
                const unsigned long long td = __builtin_amdgcn_s_memtime();
                t_stage += tc - tb; t_issue += td - tc;
            }
            if (c >= 1) decide(c);
            if (c + 1 <= nc) {
                __syncthreads();   // barrier c+1
                if (c + 2 < nc) {
                    stage(c + 2, preA, mkA);
                    if (c + 4 < nc) issue(c + 4, preA, mkA);
                }
                decide(c + 1);
            }
        }
        if ((p.ablate & 8) && blockIdx.x == 0 && lane == 0)
            printf("stager (even chunks): barrier-wait=%.0f stage=%.0f issue=%.0f cycles\n", 2.0 * t_bar / max(nc, 1), 2.0 * t_stage / max(nc, 1), 2.0 * t_issue / max(nc, 1));
    }
    if (p.tspan && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);   // the chain wave's row stores have been acknowledged
        atomicMax(p.tspan + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ------------------------------------------------------------------------------------------
// output_beam pruning on the banded path (k2.intersect_dense(..., output_beam), waterfall/graph.py:61-62,81; rule:
// SURVEY 8(0), k2's own arithmetic parity unpinned).  Three launches in front of / around the probability-domain sweep:
//   trop_band_kernel   tropical forward and backward sweep of the banded (or <= 64-state general) graph, the register
//                      layout of sweep_band_kernel with fp32 add / max instead of fp64 FMAs and no rescaling; every
//                      alpha^_t / beta^_t row leaves as fp32, best[u] = the tropical total
//   band_mask_kernel   fully parallel: per (frame, lane) one 64-bit word per direction whose bit j*MM + m says whether the
//                      lattice arc of the lane's cell (j, m) satisfies alpha^_t[src] + fl32(w + lp) + beta^_{t+1}[dst] >=
//                      best - output_beam
//   sweep_band_kernel<..., MASK>   the probability-domain sweep over the kept arcs only
// post_small_kernel<., MASK> evaluates the same expression per arc from the same fp32 rows, so sweep and posteriors agree
// on every arc.  max is exact and order-free, so the rows equal sweep_kernel<M_FB_MAX>'s bit for bit.
// ------------------------------------------------------------------------------------------
struct BandCell { int src, dst, oth, col; float w; bool on; };

// Cell (j, m) of lane `lane` in direction dir, as sweep_band_kernel lays the arcs out: banded graphs j = r*(BW+1) + d (the
// arc into position lane*R + r from d positions back; backward: the reversed order); GEN: j < WF_REG_DEG the lane's (=
// state's) own arcs, j == WF_REG_DEG its share of the one high-degree state's arcs.  oth = the state the sweep reads.
template <int R, int BW, int MM, bool GEN>
__device__ __forceinline__ BandCell band_cell(const int32_t* blob, const int32_t* I, const int dir, const int lane, const int j, const int m) {
    BandCell c{0, 0, 0, 0, 0.f, false};
    const int Sn = I[WF_I_S] - 1;
    if (GEN) {
        constexpr int GD = WF_REG_DEG;
        const int32_t* g_off = blob + I[dir ? WF_I_OUT_OFF : WF_I_IN_OFF];
        const int32_t* g_oth = blob + I[dir ? WF_I_OUT_DST : WF_I_IN_SRC];
        const int32_t* g_col = blob + I[dir ? WF_I_OUT_COL : WF_I_IN_COL];
        const float* g_w = (const float*)(blob + I[dir ? WF_I_OUT_W : WF_I_IN_W]);
        const int over = I[dir ? WF_I_NOVER_OUT : WF_I_NOVER_IN] > 0 ? I[dir ? WF_I_OVER_OUT : WF_I_OVER_IN] : -1;
        int own = lane, arc = 0;
        bool on = false;
        if (j < GD) {
            if (lane < Sn && lane != over) { arc = g_off[lane] + j; on = arc < g_off[lane + 1]; }
        } else if (over >= 0) {
            own = over;
            arc = g_off[over] + lane;
            on = arc < g_off[over + 1];
        }
        if (on) {
            c.on = true; c.oth = g_oth[arc]; c.col = g_col[arc]; c.w = g_w[arc];
            c.src = dir ? own : c.oth; c.dst = dir ? c.oth : own;
        }
        return c;
    }
    const int gbw = I[WF_I_BAND_BW], gm = I[WF_I_BAND_M];
    const int r = j / (BW + 1), d = j % (BW + 1);
    const int pp = lane * R + r;
    if (pp < Sn && d <= gbw && m < gm && pp - d >= 0) {
        const int32_t* order = blob + I[WF_I_BAND_ORDER];
        const int32_t* ent = blob + I[dir ? WF_I_BANDB : WF_I_BANDF] + (((size_t)pp * (gbw + 1) + d) * gm + m) * 2;
        const float w = __int_as_float(ent[1]);
        if (w != -INFINITY) {   // (an unused slot of the cell)
            const int own = order[dir ? Sn - 1 - pp : pp];
            c.on = true; c.col = ent[0]; c.w = w; c.oth = order[dir ? Sn - 1 - (pp - d) : pp - d];
            c.src = dir ? own : c.oth; c.dst = dir ? c.oth : own;
        }
    }
    return c;
}

// wave_shr:1 of a float: lane i receives lane i-1's value, lane 0 keeps `fill`
__device__ __forceinline__ float wave_shr1_f(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float bperm_f(float v, int srclane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srclane << 2, __builtin_bit_cast(int, v)));
}

constexpr size_t kTropEBytes = (size_t)kBandC * kBandKP * 4;   // one chunk of staged emission rows (fp32)

// WE: every arc weight is 0 (all training topologies): the staged emission is the arc score itself
template <int R, int BW, int MM, bool WE, bool GEN = false>
__global__ void __launch_bounds__(128) trop_band_kernel(SweepArgs p) {
    constexpr int GD = WF_REG_DEG;
    constexpr int NQ = GEN ? GD + 1 : R * (BW + 1), C = kBandC;
    extern __shared__ __align__(16) unsigned char smem[];   // staged emission rows [2][C][kBandKP] fp32; column K = -inf
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int fsa = uni((int)U[WF_U_FSA]), T = uni((int)U[WF_U_T]);
    const int32_t* blob = p.blob;
    const int32_t* I = blob + WF_HDR_WORDS + fsa * WF_INFO_WORDS;
    const int S = uni(I[WF_I_S]), A = uni(I[WF_I_A]), K = uni(I[WF_I_K]);
    if (T < uni(I[WF_I_MINLEN])) {   // structurally infeasible: no rows, nothing is read from them either
        if (dir == 0 && tid == 0) ((float*)p.tot)[u] = -INFINITY;
        return;
    }
    const int Sn = S - 1;
    const int nc = (T + C - 1) / C;
    if (wave == 0) {
        // =============================== chain wave ===============================
        const int32_t* order = blob + uni(I[WF_I_BAND_ORDER]);
        const float* fbm = (const float*)(blob + uni(I[WF_I_FINB_MAX]));
        float* arena = (float*)(dir ? p.beta : p.alpha) + uni(U[dir ? WF_U_BETA : WF_U_ALPHA]);
        float a[R];
        int sidx[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int pp = lane * R + r;
            int st = -1;
            if (pp < Sn) st = GEN ? pp : order[dir ? Sn - 1 - pp : pp];
            a[r] = st < 0 ? -INFINITY : (dir ? fbm[st] : (st == 0 ? 0.f : -INFINITY));
            sidx[r] = st < 0 ? S - 1 : st;
            arena[(int64_t)(dir ? T : 0) * S + sidx[r]] = a[r];   // alpha^_0 / beta^_T
        }
        unsigned coff[NQ][MM];
        float cw[NQ][MM];
        int gsrc[NQ];
        int over = -1;
        if (GEN && uni(I[dir ? WF_I_NOVER_OUT : WF_I_NOVER_IN]) > 0) over = uni(I[dir ? WF_I_OVER_OUT : WF_I_OVER_IN]);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const BandCell c = band_cell<R, BW, MM, GEN>(blob, I, dir, lane, j, m);
                coff[j][m] = (unsigned)(c.on ? c.col : K) * 4u;   // an empty cell reads column K = -inf
                cw[j][m] = c.on ? c.w : 0.f;
                if (m == 0) gsrc[j] = c.on ? c.oth : 0;
            }
        }
        float* srow = arena + (int64_t)(dir ? (T - 1) : 1) * S;
        const int64_t sstep = dir ? -(int64_t)S : (int64_t)S;
        // x[j][m] = fl32(w + lp) of cell (j, m) in the frame at row k of the staged chunk
        auto load_x = [&](const unsigned char* pb, const int k, float (&x)[NQ][MM]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NQ; ++j)
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    const float e = *(const float*)(pb + (size_t)k * kBandKP * 4 + coff[j][m]);
                    x[j][m] = WE ? e : cw[j][m] + e;
                }
        };
        auto step = [&](const float (&x)[NQ][MM]) __attribute__((always_inline)) {
            if (GEN) {
                float v = bperm_f(a[0], gsrc[0]) + x[0][0];
#pragma unroll
                for (int j = 1; j < GD; ++j) v = fmaxf(v, bperm_f(a[0], gsrc[j]) + x[j][0]);
                const float hub = wave_max(bperm_f(a[0], gsrc[GD]) + x[GD][0]);
                if (lane == over) v = hub;
                a[0] = v;
                srow[sidx[0]] = v;
                srow += sstep;
                return;
            }
            float nb[BW];
#pragma unroll
            for (int j = 0; j < BW; ++j) nb[j] = wave_shr1_f(a[R - 1 - j], -INFINITY);
            float an[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                constexpr int NT = (BW + 1) * MM;
                float t[NT < 3 ? 3 : NT] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int d = 0; d <= BW; ++d) {
                    const float sv = (r - d >= 0) ? a[r - d >= 0 ? r - d : 0] : nb[d - r - 1 >= 0 ? d - r - 1 : 0];
#pragma unroll
                    for (int m = 0; m < MM; ++m) t[d * MM + m] = sv + x[r * (BW + 1) + d][m];
                }
                // a tree of three-input maxima (v_max3_f32)
                float v = fmaxf(fmaxf(t[0], t[1]), t[2]);
#pragma unroll
                for (int i = 3; i + 1 < NT; i += 2) v = fmaxf(fmaxf(v, t[i]), t[i + 1]);
                if ((NT & 1) == 0) v = fmaxf(v, t[NT - 1]);
                an[r] = v;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a[r] = an[r];
                srow[sidx[r]] = an[r];
            }
            srow += sstep;
        };
        const int nfull = T / C;
        for (int c = 0; c < nfull; ++c) {
            __syncthreads();   // the rows of chunk c are staged
            const unsigned char* pb = smem + (size_t)(c & 1) * kTropEBytes;
            float x0[NQ][MM], x1[NQ][MM];
            load_x(pb, 0, x0);
#pragma unroll
            for (int k = 0; k < C; k += 2) {
                // (the scheduler otherwise sinks every LDS read to its use and waits for each in turn)
                load_x(pb, k + 1, x1);
                __builtin_amdgcn_sched_barrier(0);
                step(x0);
                __builtin_amdgcn_sched_barrier(0);
                if (k + 2 < C) load_x(pb, k + 2, x0);
                __builtin_amdgcn_sched_barrier(0);
                step(x1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nfull < nc) {
            __syncthreads();
            const unsigned char* pb = smem + (size_t)(nfull & 1) * kTropEBytes;
            const int nk = T - nfull * C;
            for (int k = 0; k < nk; ++k) {
                float x0[NQ][MM];
                load_x(pb + (size_t)k * kBandKP * 4, 0, x0);
                step(x0);
            }
        }
        __syncthreads();
        if (dir == 0) {   // best = max over the final arcs of alpha^_T[src] + w
            float part = -INFINITY;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int pp = lane * R + r;
                if (pp < Sn) part = fmaxf(part, a[r] + fbm[GEN ? pp : order[pp]]);
            }
            const float best = wave_max(part);
            if (lane == 0) ((float*)p.tot)[u] = best;
        }
    } else {
        // =============================== stager wave ===============================
        const bool has_e = A > 0 && T > 0;
        const int direct = uni(I[WF_I_DIRECT]);
        const float* ebase = direct ? p.lp + uni(U[WF_U_SEQ]) * p.strideB + uni(U[WF_U_START]) * p.strideT : p.E + uni(U[WF_U_E]);
        int64_t epitch = direct ? p.strideT : (int64_t)K;
        if (!has_e) { ebase = (const float*)U; epitch = 0; }   // every load stays unconditional (values unused)
        const int last_step = max(T - 1, 0);
        constexpr int KV = kBandKP / 64;
        int lcol[KV];
#pragma unroll
        for (int v = 0; v < KV; ++v) lcol[v] = has_e ? min(64 * v + lane, K - 1) : 0;
        float preA[C][KV], preB[C][KV];
        auto issue = [&](const int c, float (&pre)[C][KV]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int s = min(c * C + k, last_step);
                const float* row = ebase + (int64_t)(dir ? T - 1 - s : s) * epitch;
#pragma unroll
                for (int v = 0; v < KV; ++v) pre[k][v] = row[lcol[v]];
            }
        };
        auto stage = [&](const int c, const float (&pre)[C][KV]) __attribute__((always_inline)) {
            unsigned char* pb = smem + (size_t)(c & 1) * kTropEBytes;
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                if (64 * v + lane < K) {
#pragma unroll
                    for (int k = 0; k < C; ++k) *(float*)(pb + ((size_t)k * kBandKP + 64 * v + lane) * 4) = pre[k][v];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < C; ++k) *(float*)(pb + ((size_t)k * kBandKP + K) * 4) = -INFINITY;
            }
        };
        // the schedule of sweep_band_kernel's stager: stage(c+1) between barriers c and c+1 from the register set of its
        // parity (loaded two chunks earlier), then issue(c+3) into the same set
        if (nc > 0) {
            issue(0, preA);
            if (nc > 1) issue(1, preB);
            stage(0, preA);
            if (nc > 2) issue(2, preA);
        }
        for (int c = 0; c <= nc; c += 2) {
            __syncthreads();   // barrier c (c even)
            if (c + 1 < nc) {
                stage(c + 1, preB);
                if (c + 3 < nc) issue(c + 3, preB);
            }
            if (c + 1 <= nc) {
                __syncthreads();   // barrier c+1
                if (c + 2 < nc) {
                    stage(c + 2, preA);
                    if (c + 4 < nc) issue(c + 4, preA);
                }
            }
        }
    }
}

struct BandMaskArgs {
    const int32_t* blob;
    const int64_t* utt;
    const float* lp;
    int64_t strideB, strideT;
    const float* E;
    const float* mA;     // tropical alpha^ rows, fp32, element offsets of the alpha arena
    const float* mB;
    const float* best;   // [B]
    float beam;
    float* thr;          // [B] out: best - beam (post kernels, rescue)
    unsigned long long* mask;   // [2][rows][64]
    int64_t rows;
    int sp, kp;          // LDS pitches of a staged state row / emission row (floats)
};

constexpr int kMaskFR = 8;   // frames per wavefront

// One wavefront per kMaskFR frames of an utterance.  The three rows a frame's keep test reads (alpha^_t, beta^_{t+1}, the
// emissions) are brought into the LDS with coalesced loads, all frames' loads in flight together, and the per-cell
// gathers then run against the LDS (gathers from memory: 54 instructions of 64 scattered lanes per frame, bound by the
// texture addresser at ~16 cycles each -- 40 us on cfg3 against 8).
template <int R, int BW, int MM, bool GEN = false>
__global__ void __launch_bounds__(64) band_mask_kernel(BandMaskArgs p) {
    constexpr int NQ = GEN ? WF_REG_DEG + 1 : R * (BW + 1);
    extern __shared__ __align__(16) unsigned char smem[];   // [kMaskFR][2 * sp + kp] floats
    float* sh = (float*)smem;
    const int lane = threadIdx.x;
    const int t0 = blockIdx.x * kMaskFR, u = blockIdx.y;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int T = (int)U[WF_U_T];
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const float thr = p.best[u] - p.beam;
    if (t0 == 0 && lane == 0) p.thr[u] = thr;
    if (t0 >= T || T < I[WF_I_MINLEN]) return;
    const int S = I[WF_I_S], K = I[WF_I_K], direct = I[WF_I_DIRECT];
    const int sp = p.sp, pitch = 2 * p.sp + p.kp;
    const float* ebase = direct ? p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT : p.E + U[WF_U_E];
    const int64_t epitch = direct ? p.strideT : (int64_t)K;
    const float* abase = p.mA + U[WF_U_ALPHA];
    const float* bbase = p.mB + U[WF_U_BETA];
    const int nf = min(kMaskFR, T - t0);
    // ---- rows -> LDS (loads unconditional: indices clamped into the row, frames past the end repeat the last one)
    constexpr int NS = (64 * WF_BAND_MAX_R + 1 + 63) / 64, NK = (kBandKP + 63) / 64;
    if (K > 0) {
#pragma unroll
        for (int f = 0; f < kMaskFR; ++f) {
            const int t = t0 + min(f, nf - 1);
            float va[NS], vb[NS], ve[NK];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s_ = min(lane + 64 * i, S - 1);
                va[i] = abase[(int64_t)t * S + s_];
                vb[i] = bbase[(int64_t)(t + 1) * S + s_];
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) ve[i] = ebase[(int64_t)t * epitch + min(lane + 64 * i, K - 1)];
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (lane + 64 * i < S) { sh[f * pitch + lane + 64 * i] = va[i]; sh[f * pitch + sp + lane + 64 * i] = vb[i]; }
#pragma unroll
            for (int i = 0; i < NK; ++i)
                if (lane + 64 * i < K) sh[f * pitch + 2 * sp + lane + 64 * i] = ve[i];
        }
    }
    __builtin_amdgcn_wave_barrier();   // one wavefront: its LDS operations execute in order
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        int src[NQ][MM], dst[NQ][MM], col[NQ][MM];
        float w[NQ][MM];
        unsigned long long on = 0;
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const BandCell c = band_cell<R, BW, MM, GEN>(p.blob, I, dir, lane, j, m);
                src[j][m] = c.src; dst[j][m] = sp + c.dst; col[j][m] = 2 * sp + c.col; w[j][m] = c.w;   // an empty cell: state 0, column 0
                if (c.on) on |= 1ull << (j * MM + m);
            }
        unsigned long long* out = p.mask + ((int64_t)dir * p.rows + U[WF_U_ROW0]) * 64 + lane;
        for (int f = 0; f < nf; ++f) {
            const float* row = sh + f * pitch;
            unsigned long long keep = 0;
#pragma unroll
            for (int j = 0; j < NQ; ++j)
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    const float x = w[j][m] + row[col[j][m]];
                    if (row[src[j][m]] + x + row[dst[j][m]] >= thr) keep |= 1ull << (j * MM + m);
                }
            out[(int64_t)(t0 + f) * 64] = keep & on;
        }
    }
}

// ------------------------------------------------------------------------------------------
// check_kernel: one wavefront per (utterance, row t in 0..T).  Every path crosses exactly one state
// at time t, so sum_s alpha_t[s] * beta_t[s] = tot on every row; a row of the linear sweep that lost
// mass to underflow (or overflowed) breaks the identity.  Sets flag[u] for the fallback sweep.
// ------------------------------------------------------------------------------------------
constexpr float kCheckTol = 1e-4f;

struct CheckArgs {
    const int32_t* blob;
    const int64_t* utt;
    const double* alpha;
    const double* beta;
    const double* aoff;
    const double* boff;
    const double* tot2;
    const double* ctot2;
    int32_t* flag;
    int grp, lag;   // frames per rescale group of the sweep that wrote the rows; see lin_shift_index
    int force;      // test hook (WF_FORCE_FALLBACK): flag every utterance
    int rows_f32;   // the rows are fp32 (hub sweep), not fp64
};

// log2 of a non-negative double as (integer part exactly, fraction in fp32): log2(0) = -inf
__device__ __forceinline__ void log2_split(double v, double& e, float& f) {
    e = (double)__builtin_amdgcn_frexp_exp(v);
    f = __builtin_amdgcn_logf((float)__builtin_amdgcn_frexp_mant(v));
}

// kCheckFR consecutive rows per wavefront, their loads in flight together (one row per wave is a chain of dependent
// round trips -- utterance table, shifts, rows -- on 16 000 waves)
constexpr int kCheckFR = 4;

__global__ void __launch_bounds__(64) check_kernel(CheckArgs p) {
    const int lane = threadIdx.x;
    const int t0 = blockIdx.x * kCheckFR, u = blockIdx.y;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int T = (int)U[WF_U_T];
    if (t0 > T) return;
    if (p.force) { if (t0 == 0 && lane == 0) p.flag[u] = 1; return; }
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = (int)U[WF_U_S];
    const double tot2 = p.tot2[u];
    if (tot2 == (double)-INFINITY) {
        // zero total: fine when no path of this length exists, otherwise the sweep underflowed
        if (t0 == 0 && lane == 0 && T >= I[WF_I_MINLEN]) p.flag[u] = 1;
        return;
    }
    const double* abase = p.alpha + U[WF_U_ALPHA];
    const double* bbase = p.beta + U[WF_U_BETA];
    const float* abase32 = (const float*)p.alpha + U[WF_U_ALPHA];   // rows_f32 (sweep_hub_kernel): fp32 rows, same element offsets
    const float* bbase32 = (const float*)p.beta + U[WF_U_BETA];
    const double c0 = p.ctot2[u] - tot2;
    double cst[kCheckFR];
    float sum[kCheckFR];
#pragma unroll
    for (int r = 0; r < kCheckFR; ++r) {
        const int t = min(t0 + r, T);   // rows past T repeat row T (not judged)
        // alpha row t came from forward step t-1, beta row t from backward step T-1-t
        cst[r] = c0 - p.aoff[U[WF_U_OFFS] + lin_shift_index(t - 1, p.grp, p.lag)] -
                 p.boff[U[WF_U_OFFS] + lin_shift_index(T - 1 - t, p.grp, p.lag)];
        sum[r] = 0.f;
    }
    for (int s = lane; s < S - 1; s += 64) {   // the super-final state carries no mass at t <= T
        double a[kCheckFR], b[kCheckFR];
#pragma unroll
        for (int r = 0; r < kCheckFR; ++r) {
            const int64_t row = (int64_t)min(t0 + r, T) * S;
            a[r] = p.rows_f32 ? (double)abase32[row + s] : abase[row + s];
            b[r] = p.rows_f32 ? (double)bbase32[row + s] : bbase[row + s];
        }
#pragma unroll
        for (int r = 0; r < kCheckFR; ++r) {
            double ea, eb;
            float fa, fb;
            log2_split(a[r], ea, fa);
            log2_split(b[r], eb, fb);
            sum[r] += __builtin_amdgcn_exp2f((float)(ea + eb + cst[r]) + (fa + fb));   // exp2(-inf) = 0 for dead states
        }
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < kCheckFR; ++r) {
        const float v = wave_sum(sum[r]);
        if (t0 + r <= T && !(fabsf(v - 1.f) <= kCheckTol)) bad = true;
    }
    if (lane == 0 && bad) p.flag[u] = 1;
}

// ------------------------------------------------------------------------------------------
// State posteriors (waterfall/manual_ctc/prior.py:4-55 compute_align): out[u, t, s] = log of the
// probability that utterance u's path is in state s after consuming frame t,
// alpha_{t+1}[s] + beta_{t+1}[s] - tot; -inf for dead states, padding states, frames >= T and
// infeasible utterances.  One wavefront per (frame, utterance); reads the rows a wf_fb_logsum(WF_KEEP_AB)
// call left in the workspace, in whichever format the sweep (or the fallback) wrote them.
// ------------------------------------------------------------------------------------------
struct StatePostArgs {
    const int64_t* utt;
    const double* alpha;
    const double* beta;
    const double* aoff;
    const double* boff;
    const double* tot2;
    const double* ctot2;
    const double* totd;
    const int32_t* flag;
    int lin, grp, lag;
    float* out;
    int64_t strideB, strideT;
    int Smax, Tmax;
    int rows_f32;   // linear-domain rows are fp32 (hub sweep)
};

__global__ void __launch_bounds__(64) state_post_kernel(StatePostArgs p) {
    const int lane = threadIdx.x;
    const int t = blockIdx.x, u = blockIdx.y;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int T = (int)U[WF_U_T], S = (int)U[WF_U_S];
    float* orow = p.out + (int64_t)u * p.strideB + (int64_t)t * p.strideT;
    const double tot = p.totd[u];
    if (t >= T || tot == (double)-INFINITY) {
        for (int s = lane; s < p.Smax; s += 64) orow[s] = -INFINITY;
        return;
    }
    const int r = t + 1;   // row index: after frame t
    const double* arow = p.alpha + U[WF_U_ALPHA] + (int64_t)r * S;
    const double* brow = p.beta + U[WF_U_BETA] + (int64_t)r * S;
    const bool lin = p.lin && !p.flag[u];
    const float* arow32 = (const float*)p.alpha + U[WF_U_ALPHA] + (int64_t)r * S;
    const float* brow32 = (const float*)p.beta + U[WF_U_BETA] + (int64_t)r * S;
    const double cst = lin ? p.ctot2[u] - p.tot2[u] - p.aoff[U[WF_U_OFFS] + lin_shift_index(r - 1, p.grp, p.lag)] -
                                 p.boff[U[WF_U_OFFS] + lin_shift_index(T - 1 - r, p.grp, p.lag)]
                           : 0.0;
    for (int s = lane; s < p.Smax; s += 64) {
        float v = -INFINITY;
        if (s < S - 1) {
            if (lin) {
                double ea, eb;
                float fa, fb;
                log2_split(p.rows_f32 ? (double)arow32[s] : arow[s], ea, fa);
                log2_split(p.rows_f32 ? (double)brow32[s] : brow[s], eb, fb);
                v = ((float)(ea + eb + cst) + (fa + fb)) * 0.6931471805599453f;
            } else {
                v = (float)(arow[s] + brow[s] - tot);
            }
        }
        orow[s] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Big-graph sweep (graphs whose state vector does not fit the LDS: decoding graphs, SURVEY cfg5):
// one launch per frame over ALL states of ALL utterances, alpha ping-ponged in HBM/L2, the incoming-arc
// CSR streamed from memory (it is the dominant traffic: 16 B per arc and frame; utterances that share
// a graph re-read it from L2 / Infinity Cache).  One lane per state; states with more than
// WF_HEAVY_DEG incoming arcs get a workgroup each.  Forward only: total scores (log or tropical) and
// Viterbi back-pointers.  The launch sequence is fixed per call, so a caller may capture it in a HIP graph.
// ------------------------------------------------------------------------------------------
struct BigArgs {
    const int32_t* blob;
    const int64_t* utt;
    const float* lp;
    int64_t strideB, strideT;
    const float* E;
    void* buf;          // forward-only calls: [2][nstates] of Acc, ping-pong
    void* alpha;        // forward-backward calls: the (T+1) x S row arenas (every row kept), else null
    void* beta;
    int64_t nstates;
    int32_t* bp;        // [utt](T x S) back-pointers (Viterbi) or null
    int32_t* fin_arc;
    void* tot;
    double* totd;
    int tot_double;
    int t;              // frame of this launch
    int nlb;            // light-state blocks per utterance (max over graphs)
    int shared_S;       // != 0: alpha is in the shared-graph layout [parity][group][state][kNB] (big_step_shared_kernel)
    int shared_par;     //       and every utterance's alpha_T is in buffer shared_par
    int shared_groups;
    // output_beam pruning (MASK): tropical alpha / beta rows of a first pass, its best scores, the beam; lattice arc
    // (t, a) is kept iff mA_t[src] + (w + e) + mB_{t+1}[dst] >= best - beam (the rule of sweep_kernel's masked pass)
    const float* mA;
    const float* mB;
    const float* best;
    float* thr;
    float beam;
};

template <typename Acc, bool TROP>
__global__ void __launch_bounds__(256) big_init_kernel(BigArgs p) {
    const int u = blockIdx.y;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = (int)U[WF_U_S], T = (int)U[WF_U_T];
    Acc* a0 = p.alpha ? (Acc*)p.alpha + U[WF_U_ALPHA] : (Acc*)p.buf + U[WF_U_SOFF];
    Acc* bT = p.beta ? (Acc*)p.beta + U[WF_U_BETA] + (int64_t)T * S : nullptr;
    const float* fb = (const float*)(p.blob + I[TROP ? WF_I_FINB_MAX : WF_I_FINB_LOG]);
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
        a0[s] = s == 0 ? (Acc)0 : neg_inf<Acc>();
        if (bT) bT[s] = (Acc)fb[s];
    }
}

// DIR 0: alpha_{t+1} from alpha_t over the incoming arcs; DIR 1: beta_t from beta_{t+1} over the outgoing arcs
template <typename Acc, typename Sum, bool TROP, int DIR, bool MASK = false>
__global__ void __launch_bounds__(256) big_step_kernel(BigArgs p) {
    __shared__ double redv[8];   // (m, s) pairs of 4 waves; Acc <= double
    __shared__ int reda[4];
    const int u = blockIdx.y, tid = threadIdx.x;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int T = (int)U[WF_U_T];
    if (p.t >= T) return;   // ragged batch: this utterance is shorter than the frame of this launch
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = I[WF_I_S], K = I[WF_I_K];
    const int32_t* off = p.blob + I[DIR ? WF_I_OUT_OFF : WF_I_IN_OFF];
    const int32_t* oth = p.blob + I[DIR ? WF_I_OUT_DST : WF_I_IN_SRC];
    const int32_t* col = p.blob + I[DIR ? WF_I_OUT_COL : WF_I_IN_COL];
    const float* w = (const float*)(p.blob + I[DIR ? WF_I_OUT_W : WF_I_IN_W]);
    const int32_t* aid = p.blob + I[WF_I_IN_ARC];
    const Acc* cur;
    Acc* nxt;
    if (DIR) {
        cur = (const Acc*)p.beta + U[WF_U_BETA] + (int64_t)(p.t + 1) * S;
        nxt = (Acc*)p.beta + U[WF_U_BETA] + (int64_t)p.t * S;
    } else if (p.alpha) {
        cur = (const Acc*)p.alpha + U[WF_U_ALPHA] + (int64_t)p.t * S;
        nxt = (Acc*)p.alpha + U[WF_U_ALPHA] + (int64_t)(p.t + 1) * S;
    } else {
        cur = (const Acc*)p.buf + (int64_t)(p.t & 1) * p.nstates + U[WF_U_SOFF];
        nxt = (Acc*)p.buf + (int64_t)((p.t + 1) & 1) * p.nstates + U[WF_U_SOFF];
    }
    const float* erow = I[WF_I_DIRECT] ? p.lp + U[WF_U_SEQ] * p.strideB + (U[WF_U_START] + p.t) * p.strideT
                                        : p.E + U[WF_U_E] + (int64_t)p.t * K;
    int32_t* bprow = (!DIR && p.bp) ? p.bp + U[WF_U_ALPHA] + (int64_t)p.t * S : nullptr;
    const float* mra = MASK ? p.mA + U[WF_U_ALPHA] + (int64_t)p.t * S : nullptr;
    const float* mrb = MASK ? p.mB + U[WF_U_BETA] + (int64_t)(p.t + 1) * S : nullptr;
    const float mthr = MASK ? p.best[u] - p.beam : 0.f;
    // (s: the state this lane / workgroup reduces into; o: the arc's other end -- its source going forward)
    auto kept = [&](int s, int o, float we) { return !MASK || mra[DIR ? s : o] + we + mrb[DIR ? o : s] >= mthr; };
    if ((int)blockIdx.x < p.nlb) {
        const int s = blockIdx.x * 256 + tid;
        if (s >= S) return;
        const int a0 = off[s], a1 = off[s + 1];
        if (a1 - a0 > WF_HEAVY_DEG) return;   // a workgroup of its own, below
        Acc m = neg_inf<Acc>();
        Sum sm = 0;
        int barc = 0x7fffffff;
        for (int a = a0; a < a1; ++a) {
            if (MASK && !kept(s, oth[a], w[a] + erow[col[a]])) continue;
            const Acc v = cur[oth[a]] + (Acc)(w[a] + erow[col[a]]);
            if (TROP) max_push(m, barc, v, DIR ? 0 : aid[a]);
            else lse_push(m, sm, v);
        }
        nxt[s] = TROP ? m : lse_value(m, sm);
        if (bprow) bprow[s] = (m == neg_inf<Acc>()) ? -1 : barc;
    } else {
        const int h = blockIdx.x - p.nlb;
        if (h >= I[DIR ? WF_I_NHEAVY_OUT : WF_I_NHEAVY_IN]) return;
        const int s = (p.blob + I[DIR ? WF_I_HEAVY_OUT : WF_I_HEAVY_IN])[h];
        const int a0 = off[s], a1 = off[s + 1];
        Acc m = neg_inf<Acc>();
        Sum sm = 0;
        int barc = 0x7fffffff;
        for (int a = a0 + tid; a < a1; a += 256) {
            if (MASK && !kept(s, oth[a], w[a] + erow[col[a]])) continue;
            const Acc v = cur[oth[a]] + (Acc)(w[a] + erow[col[a]]);
            if (TROP) max_push(m, barc, v, DIR ? 0 : aid[a]);
            else lse_push(m, sm, v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const Acc m2 = shfl_xor(m, o);
            if (TROP) { const int b2 = shfl_xor(barc, o); max_push(m, barc, m2, b2); }
            else { const Sum s2 = shfl_xor(sm, o); lse_merge(m, sm, m2, s2); }
        }
        if ((tid & 63) == 0) { redv[(tid >> 6) * 2] = (double)m; redv[(tid >> 6) * 2 + 1] = (double)sm; reda[tid >> 6] = barc; }
        __syncthreads();
        if (tid == 0) {
            for (int x = 1; x < 4; ++x) {
                if (TROP) max_push(m, barc, (Acc)redv[x * 2], reda[x]);
                else lse_merge(m, sm, (Acc)redv[x * 2], (Sum)redv[x * 2 + 1]);
            }
            nxt[s] = TROP ? m : lse_value(m, sm);
            if (bprow) bprow[s] = (m == neg_inf<Acc>()) ? -1 : barc;
        }
    }
}

// ---- the same, for a batch that shares ONE graph (a decoding graph: lattice_analysis.py:123-125, cfg5) ------
// kNB utterances advance together: a lane owns one state for all of them, so every arc descriptor (16 B) is read
// once per frame instead of once per utterance and frame, and the gathers alpha_t[src] / e_t[col] fetch one
// full 32-byte sector (kNB floats) each instead of 4 useful bytes of it.  Layouts: alpha [state][kNB],
// emissions transposed once per call to Et[frame][column][kNB] (big_tr_kernel).  Utterances past their last
// frame carry their alpha forward unchanged, so after the last launch every utterance's alpha_T sits in buffer
// Tlong & 1.  Arithmetic and arc order per utterance are those of big_step_kernel (bit-identical results).
constexpr int kNB = 8;

struct BigSharedArgs {
    const int32_t* blob;
    const int64_t* utt;
    const float* lp;
    int64_t strideB, strideT;
    const float* E;
    float* Et;          // [group][Tlong][K][kNB]
    void* buf;          // [2][group][S][kNB] of Acc
    int32_t* bp;
    int B, Tlong, S, K, t, nlb;
};

__global__ void __launch_bounds__(256) big_tr_kernel(BigSharedArgs p) {
    const int t = blockIdx.x, g = blockIdx.y;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)p.utt[WF_U_FSA] * WF_INFO_WORDS;
    const int K = p.K, direct = I[WF_I_DIRECT];
    const float* row[kNB];
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        const int u = g * kNB + j;
        row[j] = nullptr;
        if (u < p.B) {
            const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
            if (t < (int)U[WF_U_T])
                row[j] = direct ? p.lp + U[WF_U_SEQ] * p.strideB + (U[WF_U_START] + t) * p.strideT : p.E + U[WF_U_E] + (int64_t)t * K;
        }
    }
    float4* out = (float4*)(p.Et + ((int64_t)g * p.Tlong + t) * K * kNB);
    for (int k = threadIdx.x; k < K; k += 256) {
        float v[kNB];
#pragma unroll
        for (int j = 0; j < kNB; ++j) v[j] = row[j] ? row[j][k] : 0.f;
        out[2 * k] = make_float4(v[0], v[1], v[2], v[3]);
        out[2 * k + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

template <typename Acc>
__global__ void __launch_bounds__(256) big_init_shared_kernel(BigSharedArgs p) {
    Acc* a0 = (Acc*)p.buf + (int64_t)blockIdx.y * p.S * kNB;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)p.S * kNB; i += (int64_t)gridDim.x * 256)
        a0[i] = i < kNB ? (Acc)0 : neg_inf<Acc>();
}

template <typename Acc> __device__ __forceinline__ void load8(const Acc* p, Acc* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<double>(const double* p, double* v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double2 a = ((const double2*)p)[j]; v[2 * j] = a.x; v[2 * j + 1] = a.y; }
}

template <typename Acc, typename Sum, bool TROP>
__global__ void __launch_bounds__(256) big_step_shared_kernel(BigSharedArgs p) {
    __shared__ double redv[4][kNB][2];
    __shared__ int reda[4][kNB];
    const int g = blockIdx.y, tid = threadIdx.x;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)p.utt[WF_U_FSA] * WF_INFO_WORDS;
    const int S = p.S;
    const int32_t* off = p.blob + I[WF_I_IN_OFF];
    const int32_t* src = p.blob + I[WF_I_IN_SRC];
    const int32_t* col = p.blob + I[WF_I_IN_COL];
    const float* w = (const float*)(p.blob + I[WF_I_IN_W]);
    const int32_t* aid = p.blob + I[WF_I_IN_ARC];
    const int64_t gstride = (int64_t)(gridDim.y) * S * kNB;
    const Acc* cur = (const Acc*)p.buf + (int64_t)(p.t & 1) * gstride + (int64_t)g * S * kNB;
    Acc* nxt = (Acc*)p.buf + (int64_t)((p.t + 1) & 1) * gstride + (int64_t)g * S * kNB;
    const float* et = p.Et + ((int64_t)g * p.Tlong + p.t) * p.K * kNB;
    // which of the group's utterances still have a frame t (uniform), and where their back-pointer rows are
    bool act[kNB];
    int32_t* bprow[kNB];
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        const int u = g * kNB + j;
        act[j] = false;
        bprow[j] = nullptr;
        if (u < p.B) {
            const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
            act[j] = p.t < (int)U[WF_U_T];
            if (p.bp && act[j]) bprow[j] = p.bp + U[WF_U_ALPHA] + (int64_t)p.t * S;
        }
    }
    Acc m[kNB];
    Sum sm[kNB];
    int barc[kNB];
#pragma unroll
    for (int j = 0; j < kNB; ++j) { m[j] = neg_inf<Acc>(); sm[j] = 0; barc[j] = 0x7fffffff; }
    auto push_arc = [&](int a) {
        Acc c[kNB];
        float e[kNB];
        load8<Acc>(cur + (int64_t)src[a] * kNB, c);
        load8<float>(et + (int64_t)col[a] * kNB, e);
        const float wa = w[a];
        const int id = TROP ? aid[a] : 0;
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
            const Acc v = c[j] + (Acc)(wa + e[j]);
            if (TROP) max_push(m[j], barc[j], v, id);
            else lse_push(m[j], sm[j], v);
        }
    };
    auto finish = [&](int s) {
        Acc keep[kNB];
        load8<Acc>(cur + (int64_t)s * kNB, keep);
        Acc out[kNB];
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
            out[j] = act[j] ? (TROP ? m[j] : lse_value(m[j], sm[j])) : keep[j];
            if (bprow[j]) __builtin_nontemporal_store((m[j] == neg_inf<Acc>()) ? -1 : barc[j], bprow[j] + s);   // (read once, by the back-trace)
        }
#pragma unroll
        for (int j = 0; j < kNB; ++j) nxt[(int64_t)s * kNB + j] = out[j];
    };
    if ((int)blockIdx.x < p.nlb) {
        const int s = blockIdx.x * 256 + tid;
        if (s >= S) return;
        const int a0 = off[s], a1 = off[s + 1];
        if (a1 - a0 > WF_HEAVY_DEG) return;
        // four arcs per round: their descriptor loads, then their alpha / emission gathers, are in flight together
        // (one arc at a time is three dependent memory round trips per arc, and a lane has only ~3 arcs to hide them with)
        for (int ab = a0; ab < a1; ab += 4) {
            int sj[4], cj[4], idj[4];
            float wj[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int a = min(ab + i, a1 - 1);
                sj[i] = src[a]; cj[i] = col[a]; wj[i] = w[a]; idj[i] = TROP ? aid[a] : 0;
            }
            Acc c[4][kNB];
            float e[4][kNB];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                load8<Acc>(cur + (int64_t)sj[i] * kNB, c[i]);
                load8<float>(et + (int64_t)cj[i] * kNB, e[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ab + i >= a1) break;
#pragma unroll
                for (int j = 0; j < kNB; ++j) {
                    const Acc v = c[i][j] + (Acc)(wj[i] + e[i][j]);
                    if (TROP) max_push(m[j], barc[j], v, idj[i]);
                    else lse_push(m[j], sm[j], v);
                }
            }
        }
        finish(s);
    } else {
        const int h = blockIdx.x - p.nlb;
        if (h >= I[WF_I_NHEAVY_IN]) return;
        const int s = (p.blob + I[WF_I_HEAVY_IN])[h];
        const int a0 = off[s], a1 = off[s + 1];
        for (int a = a0 + tid; a < a1; a += 256) push_arc(a);
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const Acc m2 = shfl_xor(m[j], o);
                if (TROP) { const int b2 = shfl_xor(barc[j], o); max_push(m[j], barc[j], m2, b2); }
                else { const Sum s2 = shfl_xor(sm[j], o); lse_merge(m[j], sm[j], m2, s2); }
            }
            if ((tid & 63) == 0) { redv[tid >> 6][j][0] = (double)m[j]; redv[tid >> 6][j][1] = (double)sm[j]; reda[tid >> 6][j] = barc[j]; }
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < kNB; ++j)
                for (int x = 1; x < 4; ++x) {
                    if (TROP) max_push(m[j], barc[j], (Acc)redv[x][j][0], reda[x][j]);
                    else lse_merge(m[j], sm[j], (Acc)redv[x][j][0], (Sum)redv[x][j][1]);
                }
            finish(s);
        }
    }
}

// total score over the final arcs, one workgroup per utterance
template <typename Acc, typename Sum, bool TROP>
__global__ void __launch_bounds__(256) big_final_kernel(BigArgs p) {
    __shared__ double redv[8];
    __shared__ int reda[4];
    const int u = blockIdx.x, tid = threadIdx.x;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int T = (int)U[WF_U_T];
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const Acc* cur = p.alpha ? (const Acc*)p.alpha + U[WF_U_ALPHA] + (int64_t)T * I[WF_I_S]
                             : (const Acc*)p.buf + (int64_t)(T & 1) * p.nstates + U[WF_U_SOFF];
    int64_t cstride = 1;
    if (p.shared_S) {
        cur = (const Acc*)p.buf + ((int64_t)p.shared_par * p.shared_groups + u / 8) * p.shared_S * 8 + (u & 7);
        cstride = 8;
    }
    const int NF = I[WF_I_NF];
    const int32_t* fsrc = p.blob + I[WF_I_FIN_SRC];
    const float* fw = (const float*)(p.blob + I[WF_I_FIN_W]);
    const int32_t* farc = p.blob + I[WF_I_FIN_ARC];
    Acc m = neg_inf<Acc>();
    Sum sm = 0;
    int barc = 0x7fffffff;
    for (int i = tid; i < NF; i += 256) {
        const Acc v = cur[fsrc[i] * cstride] + (Acc)fw[i];
        if (TROP) max_push(m, barc, v, farc[i]);
        else lse_push(m, sm, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const Acc m2 = shfl_xor(m, o);
        if (TROP) { const int b2 = shfl_xor(barc, o); max_push(m, barc, m2, b2); }
        else { const Sum s2 = shfl_xor(sm, o); lse_merge(m, sm, m2, s2); }
    }
    if ((tid & 63) == 0) { redv[(tid >> 6) * 2] = (double)m; redv[(tid >> 6) * 2 + 1] = (double)sm; reda[tid >> 6] = barc; }
    __syncthreads();
    if (tid == 0) {
        for (int x = 1; x < 4; ++x) {
            if (TROP) max_push(m, barc, (Acc)redv[x * 2], reda[x]);
            else lse_merge(m, sm, (Acc)redv[x * 2], (Sum)redv[x * 2 + 1]);
        }
        const Acc val = TROP ? m : lse_value(m, sm);
        if (p.tot_double) ((double*)p.tot)[u] = (double)val;
        else ((float*)p.tot)[u] = (float)val;
        if (p.totd) p.totd[u] = (double)val;
        if (p.fin_arc) p.fin_arc[u] = (m == neg_inf<Acc>()) ? -1 : barc;
        if (p.thr) p.thr[u] = p.best[u] - p.beam;
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

// Emission stage of the linear sweep: one wavefront per (frame, utterance).
//   c_t = max_k lp[t, label_k]      P[t,k] = exp(lp[t, label_k] - c_t)  (fp64: range, not precision)
// Compact graphs also keep the raw gathered values E[t,k] (read by the post kernels and by the
// fallback sweep).  Also clears the utterance's fallback flag.
constexpr int kGV = 8;   // values a lane keeps in registers between the two passes over the row

struct GatherLinArgs {
    const int32_t* blob;
    const int64_t* utt;
    const float* lp;
    int64_t strideB, strideT;
    float* E;
    double* P;
    double* cmax;
    int32_t* flag;
    float* P32;   // != null (sweep_linbig_kernel): the probabilities are written here as fp32 (same offsets and pitch), not to P
};

__global__ void __launch_bounds__(64) gather_lin_kernel(GatherLinArgs p) {
    const int lane = threadIdx.x;
    const int t = blockIdx.x, u = blockIdx.y;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    if (t == 0 && lane == 0) p.flag[u] = 0;
    const int T = (int)U[WF_U_T];
    if (t >= T) return;
    const int K = (int)U[WF_U_K], direct = (int)U[WF_U_DIRECT];
    const int32_t* labels = p.blob + U[WF_U_LABELS];
    const float* row = p.lp + U[WF_U_SEQ] * p.strideB + (U[WF_U_START] + t) * p.strideT;
    float v[kGV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kGV; ++i) {
        const int k = lane + 64 * i;
        v[i] = (k < K) ? row[direct ? k : labels[k]] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    for (int k = lane + 64 * kGV; k < K; k += 64) m = fmaxf(m, row[direct ? k : labels[k]]);
    m = wave_max(m);
    if (!(m > -INFINITY)) m = 0.f;   // every label impossible in this frame: P = 0
    float* erow = p.E + U[WF_U_E] + (int64_t)t * K;
    if (p.P32) {   // fp32 rows: ratios below 2^-126 of the frame's best label flush to 0 (the row check catches what that breaks)
        float* prow = p.P32 + U[WF_U_P] + (int64_t)t * U[WF_U_PP];
#pragma unroll
        for (int i = 0; i < kGV; ++i) {
            const int k = lane + 64 * i;
            if (k < K) {
                prow[k] = __builtin_amdgcn_exp2f((v[i] - m) * kLog2e);
                if (!direct) erow[k] = v[i];
            }
        }
        for (int k = lane + 64 * kGV; k < K; k += 64) {
            const float x = row[direct ? k : labels[k]];
            prow[k] = __builtin_amdgcn_exp2f((x - m) * kLog2e);
            if (!direct) erow[k] = x;
        }
        if (lane == 0) p.cmax[U[WF_U_ROW0] + t] = (double)m;
        return;
    }
    double* prow = p.P + U[WF_U_P] + (int64_t)t * U[WF_U_PP];
#pragma unroll
    for (int i = 0; i < kGV; ++i) {
        const int k = lane + 64 * i;
        if (k < K) {
            prow[k] = exp((double)v[i] - (double)m);
            if (!direct) erow[k] = v[i];
        }
    }
    for (int k = lane + 64 * kGV; k < K; k += 64) {
        const float x = row[direct ? k : labels[k]];
        prow[k] = exp((double)x - (double)m);
        if (!direct) erow[k] = x;
    }
    if (lane == 0) p.cmax[U[WF_U_ROW0] + t] = (double)m;
}

// Rows of at most 128 columns (numerators: K ~ L + 1): kGFR frames per wavefront, the gathers of all of them in
// flight together.  One frame per wave is a chain of three dependent memory round trips (utterance table -> label ->
// log-prob) on 16 000 waves that need two residency rounds; this is one round with the chains overlapped.
constexpr int kGFR = 4;

// TAB: the per-call utterance table arrives in the kernel arguments (int32 words, B <= kTabMaxB) instead of through a
// host-to-device copy in front of this kernel (3.7 us + a launch gap on the critical path of a 100 us step): every
// workgroup reads its entry from the argument segment, and the workgroups of frame block 0 write the table (and the
// sequence -> utterance map) into the workspace, where the kernels that follow find it as usual.
constexpr int kTabMaxB = 40;
struct UttTab32 { int32_t w[kTabMaxB * WF_U_WORDS]; };
struct GatherLinTabArgs {
    GatherLinArgs g;
    int64_t* utt_out;
    int32_t* seq2utt_out;
    UttTab32 tab;
};
static_assert(sizeof(GatherLinTabArgs) <= 4096, "kernel arguments are limited to 4 KB");

// NK: columns per lane (2: rows of at most 128 columns; 4: at most 256 -- cfg4's numerators have K = 2 L + 1 = 129)
template <bool TAB, int NK, typename Args>
__device__ __forceinline__ void gather_lin_small_body(const Args& a) {
    const GatherLinArgs& p = [&]() -> const GatherLinArgs& { if constexpr (TAB) return a.g; else return a; }();
    const int lane = threadIdx.x;
    const int t0 = blockIdx.x * kGFR, u = blockIdx.y;
    auto U = [&](int i) -> int64_t {
        if constexpr (TAB) return (int64_t)a.tab.w[u * WF_U_WORDS + i];
        else return p.utt[(int64_t)u * WF_U_WORDS + i];
    };
    if constexpr (TAB) {
        if (t0 == 0) {
            if (lane < WF_U_WORDS) a.utt_out[(int64_t)u * WF_U_WORDS + lane] = (int64_t)a.tab.w[u * WF_U_WORDS + lane];
            if (lane == 0 && a.seq2utt_out) a.seq2utt_out[U(WF_U_SEQ)] = u;
        }
    }
    if (t0 == 0 && lane == 0) p.flag[u] = 0;
    const int T = (int)U(WF_U_T);
    if (t0 >= T) return;
    const int K = (int)U(WF_U_K), direct = (int)U(WF_U_DIRECT);
    const int32_t* labels = p.blob + U(WF_U_LABELS);
    const float* row0 = p.lp + U(WF_U_SEQ) * p.strideB + (U(WF_U_START) + t0) * p.strideT;
    int col[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int k = lane + 64 * j;
        col[j] = k < K ? (direct ? k : labels[k]) : 0;
    }
    float v[kGFR][NK];
#pragma unroll
    for (int r = 0; r < kGFR; ++r) {
        const float* row = row0 + (int64_t)min(r, T - 1 - t0) * p.strideT;   // frames past T re-read the last one (not stored)
#pragma unroll
        for (int j = 0; j < NK; ++j) v[r][j] = row[col[j]];
    }
    const int64_t pp = U(WF_U_PP);
    double* prow = p.P + U(WF_U_P) + (int64_t)t0 * pp;
    float* erow = p.E + U(WF_U_E) + (int64_t)t0 * K;
    double* crow = p.cmax + U(WF_U_ROW0) + t0;
#pragma unroll
    for (int r = 0; r < kGFR; ++r) {
        if (t0 + r >= T) break;
        float x[NK], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NK; ++j) { x[j] = lane + 64 * j < K ? v[r][j] : -INFINITY; mx = fmaxf(mx, x[j]); }
        float m = wave_max(mx);
        if (!(m > -INFINITY)) m = 0.f;   // every label impossible in this frame: P = 0
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const int k = lane + 64 * j;
            if (k < K) { prow[r * pp + k] = exp((double)x[j] - (double)m); if (!direct) erow[r * K + k] = x[j]; }
        }
        if (lane == 0) crow[r] = (double)m;
    }
}

__global__ void __launch_bounds__(64) gather_lin_small_kernel(GatherLinArgs p) { gather_lin_small_body<false, 2>(p); }
__global__ void __launch_bounds__(64) gather_lin_small_tab_kernel(GatherLinTabArgs p) { gather_lin_small_body<true, 2>(p); }
__global__ void __launch_bounds__(64) gather_lin_small4_kernel(GatherLinArgs p) { gather_lin_small_body<false, 4>(p); }
__global__ void __launch_bounds__(64) gather_lin_small4_tab_kernel(GatherLinTabArgs p) { gather_lin_small_body<true, 4>(p); }

// ------------------------------------------------------------------------------------------
// posterior / gradient rows
// ------------------------------------------------------------------------------------------
constexpr int kPostHeavyBig = 16;   // post_big_kernel: arcs per column from which a wavefront sums the column

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
    const double* tot2;   // linear sweep: tot in log2 units
    const double* ctot2;  // linear sweep: sum_t c_t in log2 units
    const double* cmax;   // linear sweep: c_t per row (natural units)
    const int32_t* flag;  // linear sweep: utterances re-run by the fallback (their rows are log-domain fp64)
    float* grad;
    int Tmax, V, smax;
    int ecap;       // LDS floats reserved for the staged emission row (compact graphs), 0 = none
    int acap;       // LDS floats reserved for per-arc posteriors (arc-parallel mode), 0 = column loop
    int skip_zero;  // rows were zero-filled by zero_kernel already: only scatter the K values
    int fr;         // frames per workgroup
    int lin;        // != 0: the call used sweep_lin_kernel (rows are scaled probabilities, fp64, + per-group
                    // shifts); the value is the sweep's frames per rescale group
    int lag;        // see lin_shift_index
    int prof;       // measurement aid (WF_POST_PROFILE=1): one wave of post_small_kernel prints its cycle counts
    const float* mA;   // output_beam pruning: tropical alpha^ / beta^ rows and the per-utterance threshold (null: no pruning)
    const float* mB;
    const float* thr;
    const double* aoff;
    const double* boff;
    // post_small_kernel with the row check folded in (no check_kernel): pass 0 = every utterance's rows are the linear
    // sweep's; each frame's arc posteriors must sum to 1 (every path takes exactly one arc in frame t), else flag_w[u] is
    // set and rescue_kernel redoes the utterance afterwards.  pass -1: check_kernel + fallback ran before this launch
    // (flag tells which rows are which).
    int pass;
    int32_t* flag_w;
    int force;  // test hook (WF_FORCE_FALLBACK): pass 0 flags every utterance
    int accum;  // WF_ACCUMULATE (post_small_kernel): the posteriors are added to what the rows hold
    // post_hub_kernel beside a cut hub sweep: this launch computes the frames that are ready after hseg_n of hseg_den parts
    // of both sweeps and were not after hseg_prev (0: none before); the last launch (hseg_n == hseg_den) also zeroes the rows
    // nobody computes.  hseg_den == 0: every frame.
    int hseg_n, hseg_prev, hseg_den;
};

// log2(v) + add for a scaled probability of the linear sweep (0 -> -inf)
__device__ __forceinline__ float lin_log2f(double v, double add) {
    double e;
    float f;
    log2_split(v, e, f);
    return (float)(e + add) + f;
}
// Linear-sweep rows: alpha row t came from forward step t-1, beta row t+1 from backward step T-2-t.
// Returns the constant that turns log2(alpha^) + log2(beta^) + log2e*(w + e) into the log2 posterior
// of an arc in frame t:  sum of the other frames' c - shifts - tot.
__device__ __forceinline__ double lin_post_const(const PostArgs& p, const int64_t* U, int u, int t, int T) {
    return p.ctot2[u] - p.cmax[U[WF_U_ROW0] + t] * kLog2eD - p.tot2[u] -
           p.aoff[U[WF_U_OFFS] + lin_shift_index(t - 1, p.lin, p.lag)] - p.boff[U[WF_U_OFFS] + lin_shift_index(T - 2 - t, p.lin, p.lag)];
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
    // row format: scaled probabilities (linear sweep) or natural-log scores (log-semiring sweep / fallback)
    const bool lin = p.lin && !p.flag[u];
    // posterior = exp2(xs * (la + lb) + es * (w + e)): la / lb of linear-sweep rows are already log2
    const float xs = lin ? 1.f : kLog2e, es = kLog2e;
    auto ldA = [&](int64_t i) { return abase[i]; };
    auto ldB = [&](int64_t i) { return bbase[i]; };
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
        const bool masked = p.mA != nullptr;
        const float* mra = masked ? p.mA + U[WF_U_ALPHA] + arow : nullptr;
        const float* mrb = masked ? p.mB + U[WF_U_BETA] + brow : nullptr;
        const float mthr = masked ? p.thr[u] : 0.f;
        // reference point: the best alpha of this frame, so that both LDS vectors hold small numbers
        double mx = (double)-INFINITY;
        double ca = pa, cb = pb;
        if (SMALL) {
            mx = ca;
            if (stage_e && tid < K) le[tid] = pe;
            prefetch(t + 1 < T && f + 1 < f1 ? t + 1 : -1);
        } else {
            if (!lin)
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
        if (lin) {   // log2 of the scaled probabilities, both centred on the rescale target
            a_ref = lin_post_const(p, U, u, t, T) + (double)kLinBias;
            b_ref = -(double)kLinBias;
        }
        auto cvA = [&](double v) { return lin ? lin_log2f(v, a_ref) : (float)(v - a_ref); };
        auto cvB = [&](double v) { return lin ? lin_log2f(v, b_ref) : (float)(v - b_ref); };
        if (SMALL) {
            if (tid < S) { la[tid] = cvA(ca); lb[tid] = cvB(cb); }
        } else {
            for (int s = tid; s < S; s += nt) {
                la[s] = cvA(ldA(arow + s));
                lb[s] = cvB(ldB(brow + s));
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
                float pv = __builtin_amdgcn_exp2f((la[lsrc[a]] + lb[ldst[a]]) * xs + (lw[a] + e) * es);
                if (masked && !(mra[lsrc[a]] + (lw[a] + e) + mrb[ldst[a]] >= mthr)) pv = 0.f;   // pruned lattice arc
                lpost[a] = pv;
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
                    float pv = __builtin_amdgcn_exp2f((la[lsrc[a]] + lb[ldst[a]]) * xs + (lw[a] + e) * es);   // 0 for dead states
                    if (masked && !(mra[lsrc[a]] + (lw[a] + e) + mrb[ldst[a]] >= mthr)) pv = 0.f;   // pruned lattice arc
                    acc += pv;
                }
                grow[direct ? k : labels[k]] = gs * acc;
            }
        }
        __syncthreads();   // la / lb / le / lpost are reused by the next frame
    }
}

// Gradient rows of graphs whose alpha / beta rows do not fit the LDS, or are long enough that staging them costs
// more than gathering: no staging, alpha_t[src] and beta_{t+1}[dst] gathered from memory in fp64 per arc (log-domain
// rows of the big-graph / log-semiring sweeps, or the scaled-probability rows of sweep_linbig_kernel: log2 per arc).
// Columns with up to kPostHeavy arcs: one lane each; heavier ones: summed by a whole wavefront over
// coalesced slices of the column-sorted arc arrays (fixed reduction tree: bit-stable, no atomics).
// grid = (Tmax, num_seqs), 256 threads.
__global__ void __launch_bounds__(256) post_big_kernel(PostArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x, q = blockIdx.y;
    float* grow = p.grad + (int64_t)q * p.strideB + (int64_t)f * p.strideT;
    const int V = p.V;
    const int u = p.seq2utt[q];
    const bool has = u >= 0;
    const int64_t* U = p.utt + (int64_t)(has ? u : 0) * WF_U_WORDS;
    const int start = (int)U[WF_U_START], T = (int)U[WF_U_T];
    const double tot = has ? p.totd[u] : 0.0;
    const int t = f - start;
    if (p.pass == 1 && !(has && p.flag[u])) return;   // after post_hub_kernel: only the utterances it left to this kernel
    if (!has || tot == (double)-INFINITY || t < 0 || t >= T) {
        if (!p.skip_zero) zero_span(grow, V, tid, 256);
        return;
    }
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = I[WF_I_S], K = I[WF_I_K], direct = I[WF_I_DIRECT];
    const double* arow = p.alpha + U[WF_U_ALPHA] + (int64_t)t * S;
    const double* brow = p.beta + U[WF_U_BETA] + (int64_t)(t + 1) * S;
    const float* erow = direct ? p.lp + (int64_t)q * p.strideB + (int64_t)(start + t) * p.strideT
                               : p.E + U[WF_U_E] + (int64_t)t * K;
    const int32_t* loff = p.blob + I[WF_I_LAB_OFF];
    const int32_t* lsrc = p.blob + I[WF_I_LAB_SRC];
    const int32_t* ldst = p.blob + I[WF_I_LAB_DST];
    const float* lw = (const float*)(p.blob + I[WF_I_LAB_W]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const float gs = p.gscale ? p.gscale[u] : 1.f;
    if (!p.skip_zero) {   // columns the graph does not use
        if (!direct) zero_span(grow, V, tid, 256);
        else if (K < V) zero_span(grow + K, V - K, tid, 256);
        if (!direct) __syncthreads();
    }
    // row format: scaled probabilities + shifts (linear-domain sweep) or natural-log scores (log-semiring sweep / fallback)
    const bool lin = p.lin && !p.flag[u];
    const double cst = lin ? lin_post_const(p, U, u, t, T) : 0.0;
    const bool masked = p.mA != nullptr;   // output_beam pruning: the kept lattice arcs only (rule of the masked sweeps)
    const float* mra = masked ? p.mA + U[WF_U_ALPHA] + (int64_t)t * S : nullptr;
    const float* mrb = masked ? p.mB + U[WF_U_BETA] + (int64_t)(t + 1) * S : nullptr;
    const float mthr = masked ? p.thr[u] : 0.f;
    auto post = [&](int a, float e) {
        if (masked && !(mra[lsrc[a]] + (lw[a] + e) + mrb[ldst[a]] >= mthr)) return 0.f;
        if (lin) {
            double ea, eb;
            float fa, fb;
            log2_split(arow[lsrc[a]], ea, fa);
            log2_split(brow[ldst[a]], eb, fb);
            return __builtin_amdgcn_exp2f((float)(ea + eb + cst) + (fa + fb) + (lw[a] + e) * kLog2e);   // exp2(-inf) = 0 for dead states
        }
        const double x = arow[lsrc[a]] + brow[ldst[a]] - tot;   // -inf for dead states
        return __builtin_amdgcn_exp2f(((float)x + (lw[a] + e)) * kLog2e);
    };
    for (int k0 = wv * 64; k0 < K; k0 += 256) {
        const int k = k0 + lane;
        int a0 = 0, a1 = 0;
        if (k < K) { a0 = loff[k]; a1 = loff[k + 1]; }
        const bool heavy = a1 - a0 > kPostHeavyBig;
        if (k < K && !heavy) {
            const float e = erow[k];
            float acc = 0.f;
            for (int a = a0; a < a1; ++a) acc += post(a, e);
            grow[direct ? k : labels[k]] = gs * acc;
        }
        unsigned long long hm = __ballot(heavy);
        while (hm) {
            const int j = __builtin_ctzll(hm);
            hm &= hm - 1;
            const int kk = k0 + j;
            const int b0 = __shfl(a0, j), b1 = __shfl(a1, j);
            const float e = erow[kk];
            float acc = 0.f;
            for (int a = b0 + lane; a < b1; a += 64) acc += post(a, e);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += shfl_xor(acc, o);
            if (lane == 0) grow[direct ? kk : labels[kk]] = gs * acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// post_hub_kernel: gradient rows of a graph in hub form (see sweep_hub_kernel).  One 256-lane workgroup per (frame,
// sequence); a lane takes groups tid, tid + 256, ...: the group's alpha_t / beta_{t+1} values, its NC log-probs and its
// NC gradient entries are all contiguous runs shared with the neighbouring lanes -- a pure streaming kernel (read the
// alpha row, the beta row and the log_probs row, write the gradient row; nothing is gathered through arc lists, where
// post_big_kernel re-reads 12 B per arc from L2 in every workgroup).  All arcs carrying a label sit in one group, so the
// per-label sums are lane-local and in a fixed order (bit-stable).  Every column of the row is written (the hub form
// covers columns 0..K-1; K..V-1 are zeroed), so no zero-fill is needed.
// pass 0: the frame's posteriors must sum to 1, else flag_w[u] is set (the log-semiring fallback + post_big_kernel redo
// the utterance afterwards); pass -1: utterances flagged by check_kernel are left to post_big_kernel.
// ------------------------------------------------------------------------------------------
template <int G, int NC>
__global__ void __launch_bounds__(256) post_hub_kernel(PostArgs p) {
    constexpr int W = G + 1;
    __shared__ float rchk[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x, q = blockIdx.y;
    float* grow = p.grad + (int64_t)q * p.strideB + (int64_t)f * p.strideT;
    const int V = p.V;
    const int u = p.seq2utt[q];
    const bool has = u >= 0;
    const int64_t* U = p.utt + (int64_t)(has ? u : 0) * WF_U_WORDS;
    const int start = (int)U[WF_U_START], T = (int)U[WF_U_T];
    const double tot2 = has ? p.tot2[u] : 0.0;
    const int t = f - start;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const bool last_launch = p.hseg_den == 0 || p.hseg_n >= p.hseg_den;
    if (p.pass == 0 && has && f == 0 && tid == 0 && last_launch && tot2 == (double)-INFINITY && T >= I[WF_I_MINLEN]) p.flag_w[u] = 1;
    if (has && p.pass != 0 && p.flag[u]) return;   // log-domain rows: post_big_kernel's
    if (p.hseg_den > 0) {
        if (has && tot2 != (double)-INFINITY && t >= 0 && t < T) {
            if (!hub_seg_has(T, t, p.hseg_n, p.hseg_den) || hub_seg_has(T, t, p.hseg_prev, p.hseg_den)) return;
        } else if (!last_launch) return;
    }
    if (!has || tot2 == (double)-INFINITY || t < 0 || t >= T) {
        zero_span(grow, V, tid, 256);
        return;
    }
    const int S = I[WF_I_S], K = I[WF_I_K];
    const int NG = I[WF_I_HUB_NG], cs = I[WF_I_HUB_CS], NX = I[WF_I_HUB_NX];
    const int32_t* c0p = p.blob + I[WF_I_HUB_C0];
    const float* cfp = (const float*)(p.blob + I[WF_I_HUB_COEF]);
    const int32_t* xcol = p.blob + I[WF_I_HUB_XCOL];
    float cf[W][W][NC];
    unsigned present = 0;
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                cf[i][j][c] = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, cfp[(i * W + j) * NC + c])));
                if (cf[i][j][c] != 0.f) present |= 1u << (i * W + j);
            }
    present = (unsigned)uni((int)present);
    int c0[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) c0[c] = uni(c0p[c]);
    const float* arow = (const float*)p.alpha + U[WF_U_ALPHA] + (int64_t)t * S;      // (sweep_hub_kernel's fp32 rows)
    const float* brow = (const float*)p.beta + U[WF_U_BETA] + (int64_t)(t + 1) * S;
    const float* erow = p.lp + (int64_t)q * p.strideB + (int64_t)(start + t) * p.strideT;
    const float gs = p.gscale ? p.gscale[u] : 1.f;
    // alpha row t came from forward step t-1, beta row t+1 from backward step T-2-t (one shift per step, c_t = 0)
    const double kk = -tot2 - p.aoff[U[WF_U_OFFS] + lin_shift_index(t - 1, 1, 0)] - p.boff[U[WF_U_OFFS] + lin_shift_index(T - 2 - t, 1, 0)];
    // alpha^ * beta^ * 2^kk with alpha^, beta^ near 1: split the scale over both factors so that neither leaves fp32's range
    const int k1 = (int)floor(kk * 0.5);
    const float sa = (float)exp2(kk - (double)k1 - (double)k1);   // in [1, 4): the fractional part
    const float ha = ldexpf(arow[0] * sa, k1), hb = ldexpf(brow[0], k1);
    float chk = 0.f;
    // one group: posteriors of its NC labels from alpha_t / beta_{t+1} of its G states (and the hub's)
    auto group = [&](const float* ag, const float* bg, const float* eg, float* out) __attribute__((always_inline)) {
        float av[W], bv[W];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            av[i] = ldexpf(ag[i] * sa, k1);
            bv[i] = ldexpf(bg[i], k1);
        }
        av[G] = ha; bv[G] = hb;
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.f;
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (present & (1u << (i * W + j))) {   // arcs j -> i: alpha_t[j] * beta_{t+1}[i]
                    const float w = av[j] * bv[i];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = fmaf(cf[i][j][c], w, acc[c]);
                }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float val = acc[c] * __builtin_amdgcn_exp2f(eg[c] * kLog2e);
            chk += val;
            out[c] = gs * val;
        }
    };
    // the reference's topologies number a phone's NC tokens consecutively: then a group's log-probs and gradient entries
    // are one NC-float run like its states' values, and a lane moves each with one load / store (rows are only 4-byte
    // aligned: packed types), kPostHubUn groups per iteration with every load issued before the first use
    bool inter = cs == NC;
#pragma unroll
    for (int c = 1; c < NC; ++c) inter = inter && c0[c] == c0[0] + c;
    if (inter) {
        constexpr int UN = 4;
        for (int g0 = tid; g0 < NG; g0 += 256 * UN) {
            HubRow<G> A[UN], Bv[UN];
            HubRow<NC> E[UN];
#pragma unroll
            for (int r = 0; r < UN; ++r) {
                const int g = min(g0 + r * 256, NG - 1);
                A[r] = *(const HubRow<G>*)(arow + 1 + g * G);
                Bv[r] = *(const HubRow<G>*)(brow + 1 + g * G);
                E[r] = *(const HubRow<NC>*)(erow + c0[0] + g * NC);
            }
#pragma unroll
            for (int r = 0; r < UN; ++r) {
                const int g = g0 + r * 256;
                if (g < NG) {
                    HubRow<NC> O;
                    group(A[r].v, Bv[r].v, E[r].v, O.v);
                    *(HubRow<NC>*)(grow + c0[0] + g * NC) = O;   // (a non-temporal store here costs 7 %: measured)
                }
            }
        }
    } else {
        for (int g = tid; g < NG; g += 256) {
            float ag[G], bg[G], eg[NC], out[NC];
#pragma unroll
            for (int i = 0; i < G; ++i) { ag[i] = arow[1 + g * G + i]; bg[i] = brow[1 + g * G + i]; }
#pragma unroll
            for (int c = 0; c < NC; ++c) eg[c] = erow[c0[c] + g * cs];
            group(ag, bg, eg, out);
#pragma unroll
            for (int c = 0; c < NC; ++c) grow[c0[c] + g * cs] = out[c];
        }
    }
    for (int e = tid; e < NX; e += 256) {
        const int col = xcol[e];
        const float val = (ha * hb) * __builtin_amdgcn_exp2f(erow[col] * kLog2e);
        chk += val;
        grow[col] = gs * val;
    }
    if (K < V) zero_span(grow + K, V - K, tid, 256);
    if (p.pass == 0) {
        chk = wave_sum(chk);
        if (lane == 0) rchk[wv] = chk;
        __syncthreads();
        if (tid == 0) {
            const float sum = (rchk[0] + rchk[1]) + (rchk[2] + rchk[3]);
            if (p.force || !(fabsf(sum - 1.f) <= kCheckTol)) p.flag_w[u] = 1;
        }
    }
}

// Gradient rows of small compact graphs (numerators: S, K <= 64 * PSS, A <= 128 * PSS), rows pre-zeroed
// (skip_zero).  One wavefront per kPostFR consecutive frames of one utterance: the arc descriptors are
// frame-independent and stay in registers, the next frame's alpha / beta / emission values are loaded
// while the current frame is reduced, and a column shared by many arcs (the blank of a CTC numerator
// sits on ~40% of the arcs) is summed by the whole wave (fixed tree: bit-stable) instead of one lane.
constexpr int kPostFR = 4;
constexpr int kPostHeavy = 48;   // arcs per column from which the wave sums the column cooperatively (a lane sums a shorter column alone: all columns in parallel, ~30 cycles an arc, against ~250 cycles per column of the wave-wide path)

// MASK (output_beam pruning): the tropical rows alpha^_t / beta^_{t+1} and the raw emissions are staged too, and an arc
// whose alpha^_t[src] + fl32(w + lp) + beta^_{t+1}[dst] falls below the utterance's threshold contributes nothing -- the
// expression, operand order and fp32 rows of band_mask_kernel and of the masked sweep_kernel.
template <int PSS, bool MASK = false>
__global__ void __launch_bounds__(64) post_small_kernel(PostArgs p) {
    constexpr int PSA = 2 * PSS;
    extern __shared__ __align__(16) unsigned char smem[];
    float* la = (float*)smem;
    float* lb = la + p.smax;
    float* le = lb + p.smax;
    float* lpost = le + p.ecap;
    float* lma = lpost + p.acap;            // MASK: alpha^_t[s], beta^_{t+1}[s], raw emissions
    float* lmb = lma + p.smax;
    float* ler = lmb + p.smax;
    const int lane = threadIdx.x;
    const int f0 = blockIdx.x * kPostFR, q = blockIdx.y;
    const int u = p.seq2utt[q];
    const int64_t* U = p.utt + (int64_t)max(u, 0) * WF_U_WORDS;
    const int start = (int)U[WF_U_START], T = (int)U[WF_U_T];
    const double tot = u >= 0 ? p.totd[u] : (double)-INFINITY;
    const int t0 = max(f0 - start, 0), t1 = (u >= 0 && tot != (double)-INFINITY) ? min(f0 + kPostFR - start, T) : 0;
    const int K = (int)U[WF_U_K], direct = (int)U[WF_U_DIRECT];
    if (p.pass == 0 && u >= 0 && f0 == 0 && lane == 0 && tot == (double)-INFINITY) {
        // zero total: fine when no path of this length exists, otherwise the linear sweep underflowed
        const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
        if (T >= I[WF_I_MINLEN]) p.flag_w[u] = 1;
    }
    if (!p.skip_zero) {
        // rows not pre-zeroed (graphs whose columns are the labels themselves: no side-stream fill): this wave zeroes what
        // it does not write -- whole rows outside the segment / of infeasible utterances, columns K.. of the others
        for (int f = f0; f < min(f0 + kPostFR, p.Tmax); ++f) {
            const int t = f - start;
            const bool live = t >= t0 && t < t1;
            float* row = p.grad + (int64_t)q * p.strideB + (int64_t)f * p.strideT;
            for (int c = (live ? K : 0) + lane; c < p.V; c += 64) row[c] = 0.f;
        }
    }
    if (t0 >= t1) return;
    const int S = (int)U[WF_U_S], A = (int)U[WF_U_A];
    const double* abase = p.alpha + U[WF_U_ALPHA];
    const double* bbase = p.beta + U[WF_U_BETA];
    const float* ebase = direct ? p.lp + (int64_t)q * p.strideB + (int64_t)start * p.strideT : p.E + U[WF_U_E];
    const int64_t epitch = direct ? p.strideT : K;
    const bool lin = p.pass == 0 ? true : !p.flag[u];   // flagged: the fallback left natural-log fp64 rows
    const int32_t* loff = p.blob + U[WF_U_LAB_OFF];
    const int32_t* lsrc = p.blob + U[WF_U_LAB_SRC];
    const int32_t* ldst = p.blob + U[WF_U_LAB_DST];
    const int32_t* lcol = p.blob + U[WF_U_LAB_COL];
    const float* lw = (const float*)(p.blob + U[WF_U_LAB_W]);
    const int32_t* labels = p.blob + U[WF_U_LABELS];
    const float gs = p.gscale ? p.gscale[u] : 1.f;
    // per-utterance constants of the linear-sweep row format
    const double ctot2 = lin ? p.ctot2[u] : 0.0, tot2 = lin ? p.tot2[u] : 0.0;
    const double* aoffs = p.aoff + U[WF_U_OFFS];
    const double* boffs = p.boff + U[WF_U_OFFS];
    const double* cmax = p.cmax + U[WF_U_ROW0];

    // ---- frame-independent: this lane's arcs and columns (indices clamped so that no load is conditional)
    int asrc[PSA], adst[PSA], acol[PSA];
    float aw[PSA], awr[MASK ? PSA : 1];
#pragma unroll
    for (int i = 0; i < PSA; ++i) {
        const int a = min(lane + i * 64, A - 1);
        asrc[i] = lsrc[a]; adst[i] = ldst[a]; acol[i] = lcol[a]; aw[i] = lw[a] * kLog2e;
        if (MASK) awr[i] = lw[a];
    }
    const float* mabase = MASK ? p.mA + U[WF_U_ALPHA] : nullptr;
    const float* mbbase = MASK ? p.mB + U[WF_U_BETA] : nullptr;
    const float mthr = MASK ? p.thr[u] : 0.f;
    int k0[PSS], k1[PSS], kend[PSS], klab[PSS];
    unsigned long long heavy[PSS];   // wave mask of the columns of slot i that are summed cooperatively
#pragma unroll
    for (int i = 0; i < PSS; ++i) {
        const int k = min(lane + i * 64, K - 1);
        k0[i] = loff[k]; k1[i] = kend[i] = loff[k + 1]; klab[i] = direct ? k : labels[k];
        const bool mine = lane + i * 64 < K;
        heavy[i] = __ballot(mine && k1[i] - k0[i] > kPostHeavy);
        if (!mine || k1[i] - k0[i] > kPostHeavy) k1[i] = k0[i];   // nothing left for the per-lane loop
    }

    // ---- row prefetch registers (frame t): alpha_t, beta_{t+1}, emissions, offsets
    double av[PSS], bv[PSS], c_t, ga, gb;
    float ev[PSS], mav[MASK ? PSS : 1], mbv[MASK ? PSS : 1];
    auto fetch = [&](const int t) __attribute__((always_inline)) {
        const double* arow = abase + (int64_t)t * S;
        const double* brow = bbase + (int64_t)(t + 1) * S;
        const float* erow = ebase + (int64_t)t * epitch;
#pragma unroll
        for (int i = 0; i < PSS; ++i) {
            const int s = min(lane + i * 64, S - 1);
            av[i] = arow[s];
            bv[i] = brow[s];
            ev[i] = erow[min(lane + i * 64, K - 1)];
            if (MASK) { mav[i] = mabase[(int64_t)t * S + s]; mbv[i] = mbbase[(int64_t)(t + 1) * S + s]; }
        }
        c_t = lin ? cmax[t] : 0.0;
        ga = aoffs[lin ? lin_shift_index(t - 1, p.lin, p.lag) : 0];
        gb = boffs[lin ? lin_shift_index(T - 2 - t, p.lin, p.lag) : 0];
    };
    const unsigned long long c_start = __builtin_amdgcn_s_memtime();
    fetch(t0);
    unsigned long long c_f[kPostFR + 1];
    int nfr = 0;
    bool bad = p.pass == 0 && p.force;
    for (int t = t0; t < t1; ++t) {
        if (p.prof) c_f[nfr++] = __builtin_amdgcn_s_memtime();
        // ---- this frame's values -> LDS (log2 domain, both vectors centred)
        double a_ref, b_ref;
        float xs;
        if (lin) {
            a_ref = ctot2 - c_t * kLog2eD - tot2 - ga - gb + (double)kLinBias;
            b_ref = -(double)kLinBias;
            xs = 1.f;
        } else {
            double mx = (double)-INFINITY;
#pragma unroll
            for (int i = 0; i < PSS; ++i) mx = fmax(mx, av[i]);   // clamped duplicates do not change the maximum
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, shfl_xor(mx, o));
            a_ref = mx;
            b_ref = tot - mx;
            xs = kLog2e;
        }
#pragma unroll
        for (int i = 0; i < PSS; ++i) {
            const int s = lane + i * 64;
            if (s < S) {
                la[s] = lin ? lin_log2f(av[i], a_ref) : (float)(av[i] - a_ref) * xs;
                lb[s] = lin ? lin_log2f(bv[i], b_ref) : (float)(bv[i] - b_ref) * xs;
                if (MASK) { lma[s] = mav[i]; lmb[s] = mbv[i]; }
            }
            if (s < K) { le[s] = ev[i] * kLog2e; if (MASK) ler[s] = ev[i]; }
        }
        if (t + 1 < t1) fetch(t + 1);   // next frame's loads fly while this frame is reduced
        // One wavefront: its LDS operations execute in order, so no s_barrier is needed -- and __syncthreads() would
        // first drain vmcnt, i.e. wait for the prefetch just issued and for the previous frame's scattered stores
        // (measured: ~4 us of exposed memory latency per frame).  wave_barrier only pins the instruction order.
        __builtin_amdgcn_wave_barrier();
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < PSA; ++i) {
            const int a = lane + i * 64;
            if (a < A) {
                float pv = __builtin_amdgcn_exp2f(la[asrc[i]] + lb[adst[i]] + (aw[i] + le[acol[i]]));
                if (MASK && !(lma[asrc[i]] + (awr[i] + ler[acol[i]]) + lmb[adst[i]] >= mthr)) pv = 0.f;   // pruned lattice arc
                lpost[a] = pv;
                chk += pv;
            }
        }
        if (p.pass == 0 && !(fabsf(wave_sum(chk) - 1.f) <= kCheckTol)) bad = true;   // wave-uniform
        __builtin_amdgcn_wave_barrier();
        float* grow = p.grad + (int64_t)q * p.strideB + (int64_t)(start + t) * p.strideT;
        // per-column sums in arc order (bit-stable).  The rows were zero-filled: a posterior that is exactly 0 in fp32
        // (half of them on BASELINE's synthetic cfg3 batch, more for a trained model's peaked posteriors) is not stored --
        // the kernel is bound by the number of scattered stores (one 128-B line each), not by their bytes.
        const bool store_zeros = !p.skip_zero;
#pragma unroll
        for (int i = 0; i < PSS; ++i) {
            if (lane + i * 64 < K && !((heavy[i] >> lane) & 1)) {
                float acc = 0.f;
                for (int a = k0[i]; a < k1[i]; ++a) acc += lpost[a];
                if (acc != 0.f || store_zeros) grow[klab[i]] = p.accum ? grow[klab[i]] + gs * acc : gs * acc;
            }
            unsigned long long hm = heavy[i];
            while (hm) {   // wave-uniform
                const int hl = __builtin_ctzll(hm);
                hm &= hm - 1;
                const int a0 = __shfl(k0[i], hl, 64), a1 = __shfl(kend[i], hl, 64);
                float acc = 0.f;
                for (int a = a0 + lane; a < a1; a += 64) acc += lpost[a];
                acc = wave_sum(acc);
                if (lane == hl && (acc != 0.f || store_zeros)) grow[klab[i]] = p.accum ? grow[klab[i]] + gs * acc : gs * acc;
            }
        }
        __builtin_amdgcn_wave_barrier();   // la / lb / le / lpost are rewritten by the next frame
    }
    if (bad && lane == 0) p.flag_w[u] = 1;
    if (p.prof && blockIdx.x == 60 && blockIdx.y == 5 && lane == 0) {
        const unsigned long long c_end = __builtin_amdgcn_s_memtime();
        printf("post_small: setup %llu, frames %llu %llu %llu, last+drain %llu cycles (100 MHz ticks x24)\n", c_f[0] - c_start,
               nfr > 1 ? c_f[1] - c_f[0] : 0ull, nfr > 2 ? c_f[2] - c_f[1] : 0ull, nfr > 3 ? c_f[3] - c_f[2] : 0ull, c_end - c_f[nfr - 1]);
    }
}

// ------------------------------------------------------------------------------------------
// rescue_kernel: the rare path behind post_small_kernel's row check (pass 0).  One workgroup per utterance; exits at
// once unless flag[u] is set.  For a flagged utterance it redoes everything in the log semiring in fp64 (exp / log
// in fp64 as well): forward rows, total, backward rows (left in the arenas as natural-log values, the format of the
// fallback sweep), then the gradient rows -- every label column of every frame is rewritten, zeros included, over
// whatever pass 0 scattered.  Simple and slow (a state's arcs are summed by one lane), ~1 ms for T = 500: it runs only
// when the probability-domain sweep lost mass to underflow.  ONE launch instead of a fallback sweep + a second
// posterior pass: an empty launch costs ~4 us of stream time.
// ------------------------------------------------------------------------------------------
struct RescueArgs {
    const int32_t* blob;
    const int64_t* utt;
    const int32_t* flag;
    const float* gscale;
    const float* lp;
    int64_t strideB, strideT;
    const float* E;
    double* alpha;
    double* beta;
    float* tot;
    double* totd;
    float* grad;
    const float* mA;   // output_beam pruning (null: none): tropical rows and thresholds; only kept lattice arcs count
    const float* mB;
    const float* thr;
};

__device__ __forceinline__ double lse2_d(double a, double b) {   // log(exp(a) + exp(b)), -inf safe
    const double m = fmax(a, b);
    if (m == (double)-INFINITY) return m;
    return m + log(exp(a - m) + exp(b - m));
}

__global__ void __launch_bounds__(256) rescue_kernel(RescueArgs p) {
    const int u = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (!p.flag[u]) return;
    __shared__ double s_tot;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int S = I[WF_I_S], K = I[WF_I_K], direct = I[WF_I_DIRECT], T = (int)U[WF_U_T];
    const int seq = (int)U[WF_U_SEQ], start = (int)U[WF_U_START];
    const float* ebase = direct ? p.lp + (int64_t)seq * p.strideB + (int64_t)start * p.strideT : p.E + U[WF_U_E];
    const int64_t epitch = direct ? p.strideT : K;
    double* A_ = p.alpha + U[WF_U_ALPHA];
    double* B_ = p.beta + U[WF_U_BETA];
    const bool masked = p.mA != nullptr;
    const float* MA_ = masked ? p.mA + U[WF_U_ALPHA] : nullptr;
    const float* MB_ = masked ? p.mB + U[WF_U_BETA] : nullptr;
    const float mthr = masked ? p.thr[u] : 0.f;
    // kept(t, src, x, dst): the lattice arc survives the beam (the expression of the masked sweeps / post kernels)
    auto kept = [&](int t, int s_, float x, int d_) { return !masked || MA_[(int64_t)t * S + s_] + x + MB_[(int64_t)(t + 1) * S + d_] >= mthr; };
    // ---- forward
    {
        const int32_t* off = p.blob + I[WF_I_IN_OFF];
        const int32_t* src = p.blob + I[WF_I_IN_SRC];
        const int32_t* col = p.blob + I[WF_I_IN_COL];
        const float* w = (const float*)(p.blob + I[WF_I_IN_W]);
        for (int s = tid; s < S; s += nt) A_[s] = s == 0 ? 0.0 : (double)-INFINITY;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const double* cur = A_ + (int64_t)t * S;
            double* nxt = A_ + (int64_t)(t + 1) * S;
            const float* erow = ebase + (int64_t)t * epitch;
            for (int s = tid; s < S; s += nt) {
                double m = (double)-INFINITY;
                for (int a = off[s]; a < off[s + 1]; ++a)
                    if (kept(t, src[a], w[a] + erow[col[a]], s)) m = fmax(m, cur[src[a]] + (double)(w[a] + erow[col[a]]));
                double sum = 0.0;
                if (m > (double)-INFINITY)
                    for (int a = off[s]; a < off[s + 1]; ++a)
                        if (kept(t, src[a], w[a] + erow[col[a]], s)) sum += exp(cur[src[a]] + (double)(w[a] + erow[col[a]]) - m);
                nxt[s] = m > (double)-INFINITY ? m + log(sum) : m;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int NF = I[WF_I_NF];
            const int32_t* fsrc = p.blob + I[WF_I_FIN_SRC];
            const float* fw = (const float*)(p.blob + I[WF_I_FIN_W]);
            const double* last = A_ + (int64_t)T * S;
            double tot = (double)-INFINITY;
            for (int i = 0; i < NF; ++i) tot = lse2_d(tot, last[fsrc[i]] + (double)fw[i]);
            s_tot = tot;
            p.tot[u] = (float)tot;
            p.totd[u] = tot;
        }
    }
    // ---- backward
    {
        const int32_t* off = p.blob + I[WF_I_OUT_OFF];
        const int32_t* dst = p.blob + I[WF_I_OUT_DST];
        const int32_t* col = p.blob + I[WF_I_OUT_COL];
        const float* w = (const float*)(p.blob + I[WF_I_OUT_W]);
        const float* fb = (const float*)(p.blob + I[WF_I_FINB_LOG]);
        for (int s = tid; s < S; s += nt) B_[(int64_t)T * S + s] = (double)fb[s];
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            const double* nxt = B_ + (int64_t)(t + 1) * S;
            double* cur = B_ + (int64_t)t * S;
            const float* erow = ebase + (int64_t)t * epitch;
            for (int s = tid; s < S; s += nt) {
                double m = (double)-INFINITY;
                for (int a = off[s]; a < off[s + 1]; ++a)
                    if (kept(t, s, w[a] + erow[col[a]], dst[a])) m = fmax(m, nxt[dst[a]] + (double)(w[a] + erow[col[a]]));
                double sum = 0.0;
                if (m > (double)-INFINITY)
                    for (int a = off[s]; a < off[s + 1]; ++a)
                        if (kept(t, s, w[a] + erow[col[a]], dst[a])) sum += exp(nxt[dst[a]] + (double)(w[a] + erow[col[a]]) - m);
                cur[s] = m > (double)-INFINITY ? m + log(sum) : m;
            }
            __syncthreads();
        }
    }
    // ---- gradient rows: per label column, its arcs in column-sorted (= arc) order
    if (!p.grad) return;
    const double tot = s_tot;
    const int32_t* loff = p.blob + I[WF_I_LAB_OFF];
    const int32_t* lsrc = p.blob + I[WF_I_LAB_SRC];
    const int32_t* ldst = p.blob + I[WF_I_LAB_DST];
    const float* lw = (const float*)(p.blob + I[WF_I_LAB_W]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const float gs = p.gscale ? p.gscale[u] : 1.f;
    for (int64_t j = tid; j < (int64_t)T * K; j += nt) {
        const int t = (int)(j / K), k = (int)(j - (int64_t)t * K);
        float val = 0.f;
        if (tot > (double)-INFINITY) {
            const double* ar = A_ + (int64_t)t * S;
            const double* br = B_ + (int64_t)(t + 1) * S;
            const float ef = ebase[(int64_t)t * epitch + k];
            const double e = (double)ef;
            double acc = 0.0;
            for (int a = loff[k]; a < loff[k + 1]; ++a)
                if (kept(t, lsrc[a], lw[a] + ef, ldst[a])) acc += exp(ar[lsrc[a]] + ((double)lw[a] + e) + br[ldst[a]] - tot);
            val = gs * (float)acc;
        }
        p.grad[(int64_t)seq * p.strideB + (int64_t)(start + t) * p.strideT + (direct ? k : labels[k])] = val;
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
// Beam-pruned frame-synchronous search (k2.get_lattice(search_beam, output_beam, min/max_active_states),
// utils/lattice_analysis.py:125): token passing over the OUTGOING CSR of a decoding graph of any size.
//
// One 1024-lane workgroup per utterance runs all of its frames (utterances are independent: no grid-wide
// synchronisation).  Per frame:
//   expand : every surviving state of frame t pushes score + weight + emission along its outgoing arcs with one
//            64-bit atomic max per arc on best[dst] = (ordered score << 32 | ~arc id): the maximum is the tropical
//            value, ties go to the lowest arc index (the rule of wf_viterbi); the first arc to touch a state appends
//            it to the frame's candidate list.  The arcs, not the states, are dealt to the lanes (see the kernel).
//   (log)  : optionally a second pass over the same arcs adds exp(log-alpha + arc - best[dst] - G) into lsum[dst]
//            (fp64 atomic add; G = max(log-alpha - tropical score) over the survivors, so no term exceeds 1): the
//            log-semiring total of the PRUNED lattice, what k2 reports for lattice.get_tot_scores(True, True).
//   prune  : candidates below (frame best - beam) are dropped; when more than max_active remain (or fewer than
//            min_active), a 1024-bin histogram of the scores moves the cut-off (bin granularity).  Survivors get a
//            record (arc, record of the source state) in an arena that only grows by the active count, and best /
//            lsum are reset for the touched states only.
// The best path is then a pointer chase through T records.  With beam = inf and no max_active the survivors are
// all reachable states and scores / paths equal wf_viterbi's bit for bit (same fp32 association v + (w + e)).
// ------------------------------------------------------------------------------------------
constexpr int kSearchThreads = 1024;
constexpr int kSearchXL = 2048;   // survivors whose arcs / scores / degree prefix are in the LDS at a time
constexpr int kSearchCL = 6144;   // candidates of a frame kept in the LDS (the rest in the workspace)
constexpr int kSearchBins = 1024;
constexpr int kSearchUnroll = 1;
enum { WF_SR_SOFF = 0, WF_SR_XOFF = 1, WF_SR_CAP = 2, WF_SR_AOFF = 3, WF_SR_ARCOFF = 4, WF_SR_WORDS = 6 };

struct SearchArgs {
    const int32_t* blob;
    const int64_t* utt;
    const int64_t* tab;          // [B][WF_SR_WORDS]
    const float* lp;
    int64_t strideB, strideT;
    unsigned long long* best;    // [nstates]
    double* lsum;                // [nstates] or null
    int32_t* slot;               // [nstates] state -> record of the newest frame it survived in
    int32_t* cand;               // [nstates] states touched by the frame (beyond the LDS part)
    unsigned long long* cpk;     // [nstates] packed best per candidate (second compaction run)
    double* cls;                 // [nstates] same for lsum
    int32_t* arcown;             // [narcs] arc id -> index of the survivor that expanded it this frame
    int32_t* xstate;             // survivors of the current frame
    float* xscore;
    int2* xarcs;
    double* xla;
    int2* arena;
    int64_t nstates;
    float beam;
    int max_active, min_active;
    float* best_score;
    int32_t* best_arcs;
    int Tmax;
    void* tot_log;
    int tot_double;
    int32_t* n_active;           // [B][Tmax+1] or null
    // wf_search_lattice (the backward output_beam pass reads them), else null: (state, tropical score bits, first
    // outgoing arc, out-degree) of every record of the arena, and per utterance the first record of every frame:
    // ftab[t] .. ftab[t + 1] are frame t's
    int4* rinfo;
    int64_t* ftab;               // [B][ftab_stride]
    int ftab_stride;
    int prof;                    // measurement aid (WF_SEARCH_PROFILE=1): workgroup 0 prints the cycles of its phases
    // search_mw_kernel: `wgs` workgroups per utterance
    int wgs, B, solo_max;
    struct SearchCtl* ctl;       // [B], zeroed before the launch
    const int32_t* redo;         // search_kernel after search_mw_kernel: [B] stride of SearchCtl, run only the utterances whose flag is set
    int redo_stride;
    int test_drop;               // test aid (WF_SEARCH_TEST_DROP, bit 0: search_mw_kernel, bit 1: search_back_kernel, bit 2: search_words_kernel): the utterance's last workgroup leaves at once, as if it had never
                                 // become resident -- its partners' barrier runs out and the one-workgroup launch redoes the utterance
};

// Per-utterance control block of search_mw_kernel (device memory, zero at launch): the barrier counter of the utterance's
// workgroups, the abort flag, the frame's counters in three generations (frame t uses c[t % 3] and zeroes c[(t + 1) % 3],
// while c[(t - 1) % 3] is still being read) and the histogram of the second compaction.
struct SearchCtl {
    unsigned bar;
    int abort;
    // workgroup 0 -> the utterance's other workgroups, after frames it ran alone: the frame at which they join again (INT_MAX:
    // never) and the loop state they continue with
    int resume_t, r_nx;
    long long r_base, r_base_prev;
    unsigned r_g;
    int pad;
    struct Gen { int ncand; unsigned fmax; int nsurv1, nsurv2; unsigned smin, G1, G2; int pad; } c[3];
    int hist[1024];
    // search_back_kernel (several workgroups per utterance): its barrier counter, its abort flag, the lattice's arc count
    unsigned bbar;
    int babort;
    unsigned long long narcs;
    // search_words_kernel: barrier counter, abort flag, and per frame (three generations, as c[]) the smallest and largest
    // word count with a finite cell
    unsigned wbar;
    int wabort;
    int kband[3][2];
};

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu));
}
// loads of data that other lanes of the workgroup produced in an earlier phase (plain stores or L2 atomics): not from a
// possibly stale line of the vector L1
template <typename T> __device__ __forceinline__ T ld_fresh(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Hand-offs between workgroups inside a launch (search_mw_kernel, search_back_kernel): EVERY wave waits for its own
// stores and atomics before the workgroup's barrier (s_barrier waits for no counter), then one lane releases at agent
// scope, waits again, and signals; the waiting lane polls relaxed, acquires once, waits, and the workgroup's barrier
// lets the other waves load.
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ void __launch_bounds__(kSearchThreads) search_kernel(SearchArgs p) {
    __shared__ int s_ncand, s_nsurv, s_keepbin, s_narcs, s_bin0;
    __shared__ uint32_t s_fmax, s_min, s_G;
    __shared__ unsigned long long s_pkmax;   // the frame's best candidate (score, then lowest arc): kept whatever the cut-off does
    __shared__ int s_pfx[kSearchXL];        // exclusive prefix of the survivors' out-degrees (arc -> owner lookup)
    __shared__ int s_xa0[kSearchXL];        // first outgoing arc and tropical score of the first kSearchXL survivors
    __shared__ float s_xv[kSearchXL];
    __shared__ int s_cand[kSearchCL];       // the first kSearchCL candidates of the frame
    __shared__ int s_hist[kSearchBins];
    __shared__ int s_wsum[16];
    __shared__ unsigned long long s_fin;
    __shared__ double s_lsum;
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.redo && !p.redo[(size_t)u * p.redo_stride]) return;   // (after search_mw_kernel: only what it gave up on)
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int64_t* R = p.tab + (int64_t)u * WF_SR_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int T = (int)U[WF_U_T];
    const int direct = I[WF_I_DIRECT];
    const int32_t* off = p.blob + I[WF_I_OUT_OFF];
    const int4* orec = (const int4*)(p.blob + I[WF_I_OUT_REC]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const int64_t soff = R[WF_SR_SOFF];
    const int cap = (int)R[WF_SR_CAP];
    unsigned long long* best = p.best + soff;
    double* lsum = p.lsum ? p.lsum + soff : nullptr;
    if (p.redo) {   // the run that was given up left values behind
        for (int i = tid; i < I[WF_I_S]; i += kSearchThreads) { best[i] = 0ull; if (lsum) lsum[i] = 0.0; }
        __syncthreads();
    }
    int32_t* cand = p.cand + soff;
    unsigned long long* cpk = p.cpk + soff;    // packed best of every candidate, kept while best[] is being reset
    double* cls = p.lsum ? p.cls + soff : nullptr;
    int32_t* slot = p.slot + soff;             // state -> record of the newest frame it survived in
    int32_t* arcown = p.arcown + R[WF_SR_ARCOFF];   // arc id -> index of the survivor that expanded it this frame
    int32_t* xstate = p.xstate + R[WF_SR_XOFF];
    float* xscore = p.xscore + R[WF_SR_XOFF];
    int2* xarcs = p.xarcs + R[WF_SR_XOFF];     // (first outgoing arc, out-degree) of every survivor
    double* xla = p.xla + R[WF_SR_XOFF];
    int2* arena = p.arena + R[WF_SR_AOFF];
    int32_t* out = p.best_arcs + (int64_t)u * (p.Tmax + 1);
    int32_t* nact = p.n_active ? p.n_active + (int64_t)u * (p.Tmax + 1) : nullptr;
    int4* rinfo = p.rinfo ? p.rinfo + R[WF_SR_AOFF] : nullptr;
    int64_t* ftab = p.ftab ? p.ftab + (int64_t)u * p.ftab_stride : nullptr;
    for (int i = tid; i <= p.Tmax; i += kSearchThreads) { out[i] = -1; if (nact) nact[i] = 0; }
    const float* lp0 = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
    const float ninf = -INFINITY;

    // frame 0: the start state
    if (tid == 0) {
        xstate[0] = 0; xscore[0] = 0.f; xla[0] = 0.0;
        xarcs[0] = make_int2(off[0], off[1] - off[0]);
        s_xa0[0] = off[0]; s_xv[0] = 0.f; s_pfx[0] = off[1] - off[0];   // (s_pfx holds the DEGREES until the scan)
        arena[0] = make_int2(-1, -1);
        if (rinfo) { rinfo[0] = make_int4(0, __float_as_int(0.f), off[0], off[1] - off[0]); ftab[0] = 0; ftab[1] = 1; }
        slot[0] = 0;
        if (nact) nact[0] = 1;
        s_G = f2ord(0.f);
    }
    int nX = 1;
    int64_t base = 1;        // records used so far
    int64_t base_prev = 0;   // first record of the current frame's survivors
    bool dead = false;
    __syncthreads();
    unsigned long long pc[5] = {0, 0, 0, 0, 0}, pt0 = 0;
    long long parcs = 0, pcand = 0, ptight = 0;
#define SR_MARK(i) do { if (p.prof) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pc[i] += now_ - pt0; pt0 = now_; } } while (0)
    if (p.prof) pt0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < T; ++t) {
        const float* erow = lp0 + (int64_t)t * p.strideT;
        if (tid == 0) { s_ncand = 0; s_fmax = 0; s_nsurv = 0; s_min = 0xffffffffu; s_keepbin = 0x7fffffff; s_pkmax = 0ull; }
        const float G = ord2f(s_G);
        const bool xl = nX <= kSearchXL;   // (uniform) the survivors' arcs / scores are in the LDS
        // ---------------- expand: ARC-parallel.  The survivors' out-degrees are prefix-summed into the LDS (chunks of
        // kSearchXL survivors); lane l then owns arcs l, l + 1024, ... of the chunk's arc list, kSearchUnroll of them in
        // flight together, and finds each arc's survivor by a branch-free binary search of the prefix -- the same work per
        // lane whatever the degrees are (a state with 10^5 outgoing arcs is simply spread over all lanes).
        uint32_t lmax = 0;
        for (int log_pass = 0; log_pass < (lsum ? 2 : 1); ++log_pass) {   // (the log pass needs every tropical value of the frame)
        for (int c0 = 0; c0 < nX; c0 += kSearchXL) {
            const int m = min(kSearchXL, nX - c0);
            if (!(log_pass && xl)) {   // (one chunk: the prefix of the tropical pass is still there)
                constexpr int kPer = kSearchXL / kSearchThreads;
                int deg[kPer], run = 0;
                if (!xl) __syncthreads();   // (the previous chunk's searches are over)
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    const int i = tid * kPer + k;
                    deg[k] = i < m ? (xl ? s_pfx[i] : ld_fresh(&xarcs[c0 + i].y)) : 0;
                    if (!xl && i < m) { s_xa0[i] = ld_fresh(&xarcs[c0 + i].x); s_xv[i] = ld_fresh(xscore + c0 + i); }
                }
#pragma unroll
                for (int k = 0; k < kPer; ++k) { const int x = deg[k]; deg[k] = run; run += x; }
                int v = run;
                for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(v, o, 64); if (lane >= o) v += x; }
                if (lane == 63) s_wsum[wave] = v;
                __syncthreads();
                int add = v - run;
                for (int k = 0; k < wave; ++k) add += s_wsum[k];
#pragma unroll
                for (int k = 0; k < kPer; ++k) s_pfx[tid * kPer + k] = add + deg[k];
                if (tid == kSearchThreads - 1) s_narcs = add + run;
            }
            __syncthreads();
            const int NA = s_narcs;
            if (!log_pass) parcs += NA;
            SR_MARK(0);
            for (int k0 = 0; k0 < NA; k0 += kSearchUnroll * kSearchThreads) {
                int own[kSearchUnroll], q[kSearchUnroll];
                bool ok[kSearchUnroll];
#pragma unroll
                for (int j = 0; j < kSearchUnroll; ++j) {
                    const int k = k0 + j * kSearchThreads + tid;
                    ok[j] = k < NA;
                    int lo = 0;
#pragma unroll
                    for (int st = kSearchXL / 2; st > 0; st >>= 1)
                        if (lo + st < m && s_pfx[lo + st] <= k) lo += st;
                    own[j] = lo;
                    q[j] = ok[j] ? k - s_pfx[lo] + s_xa0[lo] : 0;
                }
                int d[kSearchUnroll], ar[kSearchUnroll];
                float we[kSearchUnroll];
#pragma unroll
                for (int j = 0; j < kSearchUnroll; ++j) {
                    const int4 rec = orec[q[j]];   // {dst, column, weight, arc id}: one 16-byte access per arc
                    d[j] = rec.x;
                    ar[j] = rec.w;
                    we[j] = __int_as_float(rec.z) + erow[direct ? rec.y : labels[rec.y]];
                }
                if (!log_pass) {
                    // one 64-bit atomic max per arc: the tropical value of the destination, ties to the lowest arc index
                    unsigned long long old[kSearchUnroll];
#pragma unroll
                    for (int j = 0; j < kSearchUnroll; ++j) {
                        const float sc = s_xv[own[j]] + we[j];
                        ok[j] = ok[j] && sc > ninf;
                        const uint32_t o = f2ord(sc);
                        old[j] = 1ull;
                        if (ok[j]) {
                            lmax = max(lmax, o);
                            arcown[ar[j]] = c0 + own[j];
                            old[j] = atomicMax(&best[d[j]], ((unsigned long long)o << 32) | (0xffffffffu - (uint32_t)ar[j]));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kSearchUnroll; ++j) {   // first touch: append to the candidate list, one LDS atomic per wavefront
                        const bool fresh = old[j] == 0ull;
                        const unsigned long long mk = __ballot(fresh);
                        if (fresh) {
                            const int leader = __ffsll((long long)mk) - 1;
                            int wb = 0;
                            if (lane == leader) wb = atomicAdd(&s_ncand, __popcll(mk));
                            wb = __shfl(wb, leader, 64);
                            const int idx = wb + __popcll(mk & ((1ull << lane) - 1ull));
                            if (idx < kSearchCL) s_cand[idx] = d[j]; else cand[idx] = d[j];
                        }
                    }
                } else {
                    // log semiring: exp(log-alpha + arc - tropical value of the destination - G) <= 1
                    float tb[kSearchUnroll];
                    double la[kSearchUnroll];
#pragma unroll
                    for (int j = 0; j < kSearchUnroll; ++j) {
                        tb[j] = ord2f((uint32_t)(ld_fresh(best + d[j]) >> 32));
                        la[j] = ld_fresh(xla + c0 + own[j]);
                    }
#pragma unroll
                    for (int j = 0; j < kSearchUnroll; ++j)
                        if (ok[j] && we[j] > ninf) atomicAdd(&lsum[d[j]], exp(la[j] + (double)we[j] - (double)tb[j] - (double)G));
                }
            }
        }
        __syncthreads();   // every value of the pass is final
        SR_MARK(1);
        }
        for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, o, 64));
        if (lane == 0 && lmax) atomicMax(&s_fmax, lmax);
        __syncthreads();
        // ---------------- prune
        const int n = s_ncand;
        if (n == 0) { dead = true; break; }   // (uniform)
        const float fmax = ord2f(s_fmax);
        float cutoff = fmax - p.beam;          // -inf for an infinite beam
        float inv_width = 0.f;
        int keepbin = 0x7fffffff;
        const int want_min = min(p.min_active, n);
        // Compaction: a candidate that passes (cutoff, keepbin) gets its record and goes to the survivor arrays.  The first
        // run (from_best) also moves the packed value of every candidate out of best[] / lsum[] (which go back to "empty"
        // for the next frame) into cpk / cls, where a second run with a moved cut-off finds them.
        uint32_t gmax = 0, lmin = 0xffffffffu;
        bool ovf = false;   // second run only: the best bin alone holds more than cap candidates (exact ties)
        unsigned long long lpk = 0ull;
        auto compact = [&](const bool from_best) __attribute__((always_inline)) {
            gmax = 0;
            const unsigned long long pkbest = from_best ? 0ull : s_pkmax;
            const int lim = ovf ? cap - 1 : cap;   // (ovf: slot cap - 1 belongs to the frame's best candidate)
            for (int i0 = 0; i0 < n; i0 += kSearchUnroll * kSearchThreads) {
                int s[kSearchUnroll];
                unsigned long long pk[kSearchUnroll];
                bool in[kSearchUnroll], keep[kSearchUnroll];
#pragma unroll
                for (int j = 0; j < kSearchUnroll; ++j) {
                    const int i = i0 + j * kSearchThreads + tid;
                    in[j] = i < n;
                    s[j] = !in[j] ? 0 : (i < kSearchCL ? s_cand[i] : ld_fresh(cand + i));
                }
#pragma unroll
                for (int j = 0; j < kSearchUnroll; ++j) {
                    const int i = i0 + j * kSearchThreads + tid;
                    pk[j] = !in[j] ? 0ull : (from_best ? ld_fresh(best + s[j]) : ld_fresh(cpk + i));
                }
                double ls[kSearchUnroll];
                int o0[kSearchUnroll], o1[kSearchUnroll], ow[kSearchUnroll];
#pragma unroll
                for (int j = 0; j < kSearchUnroll; ++j) {
                    const int i = i0 + j * kSearchThreads + tid;
                    const float sc = ord2f((uint32_t)(pk[j] >> 32));
                    keep[j] = in[j] && sc >= cutoff &&
                              (keepbin == 0x7fffffff || min(kSearchBins - 1, (int)((fmax - sc) * inv_width)) <= keepbin);
                    if (in[j]) { lmin = min(lmin, (uint32_t)(pk[j] >> 32)); if (from_best) lpk = max(lpk, pk[j]); }
                    ls[j] = 0.0;
                    if (lsum && in[j] && (from_best || keep[j])) ls[j] = from_best ? ld_fresh(lsum + s[j]) : ld_fresh(cls + i);
                    o0[j] = keep[j] ? off[s[j]] : 0;
                    o1[j] = keep[j] ? off[s[j] + 1] : 0;
                    ow[j] = keep[j] ? ld_fresh(arcown + (int)(0xffffffffu - (uint32_t)pk[j])) : 0;
                }
#pragma unroll
                for (int j = 0; j < kSearchUnroll; ++j) {
                    const int i = i0 + j * kSearchThreads + tid;
                    if (from_best && in[j]) {
                        best[s[j]] = 0ull; cpk[i] = pk[j];
                        if (lsum) { lsum[s[j]] = 0.0; cls[i] = ls[j]; }
                    }
                    const bool isbest = ovf && keep[j] && pk[j] == pkbest;
                    const unsigned long long mk = __ballot(keep[j] && !isbest);   // wave-aggregated append (uniform control flow here)
                    int wb = 0;
                    if (lane == 0 && mk) wb = atomicAdd(&s_nsurv, __popcll(mk));
                    wb = __shfl(wb, 0, 64);
                    if (keep[j]) {
                        const int k = isbest ? cap - 1 : wb + __popcll(mk & ((1ull << lane) - 1ull));
                        if (isbest || k < lim) {
                            const float sc = ord2f((uint32_t)(pk[j] >> 32));
                            arena[base + k] = make_int2((int)(0xffffffffu - (uint32_t)pk[j]), (int)(base_prev + ow[j]));
                            if (rinfo) rinfo[base + k] = make_int4(s[j], __float_as_int(sc), o0[j], o1[j] - o0[j]);
                            slot[s[j]] = (int)(base + k);
                            xstate[k] = s[j];
                            xscore[k] = sc;
                            xarcs[k] = make_int2(o0[j], o1[j] - o0[j]);
                            if (k < kSearchXL) { s_xa0[k] = o0[j]; s_xv[k] = sc; s_pfx[k] = o1[j] - o0[j]; }
                            if (lsum) {
                                const double la = (double)sc + (double)G + log(ls[j]);
                                xla[k] = la;
                                gmax = max(gmax, f2ord((float)(la - (double)sc)));
                            }
                        }
                    }
                }
            }
        };
        pcand += n;
        SR_MARK(2);
        compact(true);
        for (int o = 32; o > 0; o >>= 1) lmin = min(lmin, (uint32_t)__shfl_xor((int)lmin, o, 64));
        if (lane == 0) atomicMin(&s_min, lmin);
        for (int o = 32; o > 0; o >>= 1) lpk = max(lpk, (unsigned long long)__shfl_xor((long long)lpk, o, 64));
        if (lane == 0 && lpk) atomicMax(&s_pkmax, lpk);
        __syncthreads();
        const int cnt = s_nsurv;
        const bool tighten = cnt > cap, loosen = !tighten && cnt < want_min;
        SR_MARK(3);
        if (tighten || loosen) {
            ++ptight;   // (uniform) move the cut-off along a histogram of the scores (best first), compact again
            if (loosen) cutoff = ninf;
            const float lo = max(cutoff, ord2f(s_min));
            inv_width = fmax > lo ? (float)kSearchBins / (fmax - lo) : 0.f;
            // When the best bin alone holds more than cap candidates the histogram is taken again over that bin's range (up
            // to three times: 2^-30 of the first range); candidates that are still tied then -- equal scores -- are cut in
            // arrival order, but never the frame's best one (s_pkmax: it takes the last slot).
            for (int refine = 0;; ++refine) {
                s_hist[tid] = 0;
                __syncthreads();   // (every lane has read s_nsurv by now; cpk / cls of the first run are visible)
                if (tid == 0) { s_nsurv = 0; s_keepbin = tighten ? 0 : kSearchBins - 1; }
                for (int i0 = 0; i0 < n; i0 += kSearchUnroll * kSearchThreads) {
                    unsigned long long pk[kSearchUnroll];
#pragma unroll
                    for (int j = 0; j < kSearchUnroll; ++j) {
                        const int i = i0 + j * kSearchThreads + tid;
                        pk[j] = i < n ? ld_fresh(cpk + i) : 0ull;
                    }
#pragma unroll
                    for (int j = 0; j < kSearchUnroll; ++j) {
                        const float sc = ord2f((uint32_t)(pk[j] >> 32));
                        if (pk[j] && sc >= cutoff) atomicAdd(&s_hist[min(kSearchBins - 1, (int)((fmax - sc) * inv_width))], 1);
                    }
                }
                __syncthreads();
                int v = s_hist[tid];   // inclusive scan of the 1024 bins
                for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(v, o, 64); if (lane >= o) v += x; }
                if (lane == 63) s_wsum[wave] = v;
                __syncthreads();
                for (int k = 0; k < wave; ++k) v += s_wsum[k];
                if (tid == 0) s_bin0 = v;
                // tighten: the last bin whose running count still fits; loosen: the first bin that reaches min_active
                if (tighten) { if (v <= cap) atomicMax(&s_keepbin, tid); }
                else { if (v >= want_min) atomicMin(&s_keepbin, tid); }
                __syncthreads();
                keepbin = s_keepbin;
                ovf = tighten && keepbin == 0 && s_bin0 > cap;
                if (!ovf || refine == 3 || !(inv_width > 0.f) || !(inv_width < 1e30f)) break;
                cutoff = max(cutoff, fmax - 1.f / inv_width);   // the best bin's lower edge
                inv_width *= (float)kSearchBins;
                __syncthreads();   // (s_bin0 / s_keepbin are rewritten)
            }
            compact(false);
            __syncthreads();
        }
        if (lsum) {
            for (int o = 32; o > 0; o >>= 1) gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, o, 64));
            if (tid == 0) s_G = 0;
            __syncthreads();
            if (lane == 0 && gmax) atomicMax(&s_G, gmax);
        }
        nX = ovf ? cap : min(s_nsurv, cap);
        if (tid == 0 && nact) nact[t + 1] = nX;
        base_prev = base;
        base += nX;
        if (tid == 0 && ftab) ftab[t + 2] = base;
        __syncthreads();
        SR_MARK(4);
    }
    if (p.prof && u == 0 && tid == 0)
        printf("search_kernel wg 0: T=%d arcs %lld candidates %lld second-compaction frames %lld | cycles: scan %llu expand %llu reduce %llu compact %llu tighten+tail %llu\n",
               T, parcs, pcand, ptight, pc[0], pc[1], pc[2], pc[3], pc[4]);
    // ---------------- final arcs over the survivors of frame T
    const int NF = I[WF_I_NF];
    const int32_t* fsrc = p.blob + I[WF_I_FIN_SRC];
    const float* fw = (const float*)(p.blob + I[WF_I_FIN_W]);
    const int32_t* farc = p.blob + I[WF_I_FIN_ARC];
    if (tid == 0) { s_fin = 0ull; s_fmax = 0; s_lsum = 0.0; }
    __syncthreads();
    const int32_t* slotT = slot;
    const int32_t* asrc = p.blob + I[WF_I_ARC_SRC];
    const int64_t baseT = base - nX;
    auto survivor = [&](int s) -> int {   // index of state s among the survivors of frame T, or -1
        const int64_t r = (int64_t)ld_fresh(slotT + s) - baseT;
        if (r < 0 || r >= nX) return -1;
        return ld_fresh(xstate + r) == s ? (int)r : -1;
    };
    if (!dead) {
        unsigned long long fb = 0ull;
        uint32_t lm = 0;
        for (int i = tid; i < NF; i += kSearchThreads) {
            const int j = survivor(fsrc[i]);
            if (j < 0) continue;
            const float v = ld_fresh(xscore + j) + fw[i];
            if (!(v > ninf)) continue;
            fb = max(fb, ((unsigned long long)f2ord(v) << 32) | (0xffffffffu - (uint32_t)farc[i]));
            if (lsum) lm = max(lm, f2ord((float)(ld_fresh(xla + j) + (double)fw[i])));
        }
        if (fb) atomicMax(&s_fin, fb);
        if (lm) atomicMax(&s_fmax, lm);
        __syncthreads();
        if (lsum && s_fmax) {
            const double ref = (double)ord2f(s_fmax);
            double acc = 0.0;
            for (int i = tid; i < NF; i += kSearchThreads) {
                const int j = survivor(fsrc[i]);
                if (j < 0 || !(fw[i] > ninf)) continue;
                acc += exp(ld_fresh(xla + j) + (double)fw[i] - ref);
            }
            acc = wave_sum_d(acc);
            if (lane == 0 && acc != 0.0) atomicAdd(&s_lsum, acc);
            __syncthreads();
        }
    }
    if (tid == 0) {
        const unsigned long long fb = dead ? 0ull : s_fin;
        double totl = -INFINITY;
        if (fb == 0ull) {
            p.best_score[u] = ninf;
        } else {
            p.best_score[u] = ord2f((uint32_t)(fb >> 32));
            const int fa = (int)(0xffffffffu - (uint32_t)fb);
            const int abase = I[WF_I_ARC_BASE];
            out[T] = abase + fa;
            int64_t r = ld_fresh(slotT + asrc[fa]);
            for (int t = T - 1; t >= 0; --t) {
                const int2 rec = arena[r];
                out[t] = abase + rec.x;
                r = rec.y;
            }
            if (lsum && s_fmax) totl = (double)ord2f(s_fmax) + log(s_lsum);
        }
        if (p.tot_log) {
            if (p.tot_double) ((double*)p.tot_log)[u] = totl;
            else ((float*)p.tot_log)[u] = (float)totl;
        }
    }
}

// ------------------------------------------------------------------------------------------
// search_mw_kernel: the same search with SEVERAL workgroups per utterance (wf_search picks it when the chip has room:
// B x wgs workgroups, all resident).  One workgroup's memory pipeline is what bounds search_kernel at 10^3 - 10^4
// active states (a frame is ~10 scattered cache lines per active state, dealt with by one CU); here the frame's
// survivors are dealt to `wgs` workgroups (contiguous blocks), every workgroup expands its block with the same atomics
// on best[], keeps the candidates IT touched first in a list of its own and compacts that list itself; survivor slots
// come from one global counter.  The utterance's workgroups meet at a barrier in device memory (a monotonic counter)
// twice per frame (after the expansion: every tropical value is final; after the compaction: the next frame's
// survivors are complete), four times when the histogram moves the cut-off.  Frames with fewer than solo_max
// survivors are run by workgroup 0 alone and cost the others one barrier.
//
// Every wait is bounded: a workgroup that has waited ~0.1 s (its partners are not running: the chip is shared with
// other work and they did not fit) raises the utterance's abort flag, everyone leaves, and wf_search's second launch
// (search_kernel with `redo`) runs that utterance with one workgroup.  Workgroup b serves utterance b % B: consecutive
// workgroup ids go round the XCDs, so an utterance's workgroups share an L2 when B is a multiple of 8.
// ------------------------------------------------------------------------------------------
constexpr long kSearchSpinLimit = 200000;   // polls of the barrier counter (>= 0.5 us each) before giving up

__global__ void __launch_bounds__(kSearchThreads) search_mw_kernel(SearchArgs p) {
    __shared__ int s_ncand, s_keepbin, s_narcs, s_abort, s_resume;
    __shared__ uint32_t s_fmax;
    __shared__ SearchCtl::Gen s_gen;        // the frame's counters while workgroup 0 runs the frame alone
    __shared__ int s_pfx[kSearchXL];
    __shared__ int s_xa0[kSearchXL];
    __shared__ float s_xv[kSearchXL];
    __shared__ int s_cand[kSearchCL];
    __shared__ int s_hist[kSearchBins];
    __shared__ int s_wsum[16];
    __shared__ unsigned long long s_fin;
    __shared__ double s_lsum;
    const int NW = p.wgs;
    const int u = blockIdx.x % p.B, g = blockIdx.x / p.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int64_t* R = p.tab + (int64_t)u * WF_SR_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int T = (int)U[WF_U_T];
    const int direct = I[WF_I_DIRECT];
    const int32_t* off = p.blob + I[WF_I_OUT_OFF];
    const int4* orec = (const int4*)(p.blob + I[WF_I_OUT_REC]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const int64_t soff = R[WF_SR_SOFF];
    const int cap = (int)R[WF_SR_CAP];
    unsigned long long* best = p.best + soff;
    double* lsum = p.lsum ? p.lsum + soff : nullptr;
    // this workgroup's own candidate arrays (a state is first touched by exactly one arc, hence listed by one workgroup)
    int32_t* cand = p.cand + (int64_t)g * p.nstates + soff;
    unsigned long long* cpk = p.cpk + (int64_t)g * p.nstates + soff;
    double* cls = p.lsum ? p.cls + (int64_t)g * p.nstates + soff : nullptr;
    int32_t* slot = p.slot + soff;
    int32_t* arcown = p.arcown + R[WF_SR_ARCOFF];
    int32_t* xstate = p.xstate + R[WF_SR_XOFF];
    float* xscore = p.xscore + R[WF_SR_XOFF];
    int2* xarcs = p.xarcs + R[WF_SR_XOFF];
    double* xla = p.xla + R[WF_SR_XOFF];
    int2* arena = p.arena + R[WF_SR_AOFF];
    int32_t* out = p.best_arcs + (int64_t)u * (p.Tmax + 1);
    int32_t* nact = p.n_active ? p.n_active + (int64_t)u * (p.Tmax + 1) : nullptr;
    int4* rinfo = p.rinfo ? p.rinfo + R[WF_SR_AOFF] : nullptr;
    int64_t* ftab = p.ftab ? p.ftab + (int64_t)u * p.ftab_stride : nullptr;
    SearchCtl* ctl = p.ctl + u;
    const float* lp0 = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
    const float ninf = -INFINITY;
    const int solo_max = max(p.solo_max, 2);
    unsigned nbar = 0;
    if ((p.test_drop & 1) && g == NW - 1) return;

    // barrier of the utterance's workgroups; true: the utterance was given up (every lane of every workgroup leaves)
    auto grid_sync = [&]() -> bool {
        drain_vm();        // every wave: its stores and atomics have completed ...
        __syncthreads();   // ... before one lane releases them for the whole workgroup
        if (tid == 0) {
            ++nbar;
            const unsigned target = nbar * (unsigned)NW;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // (the XCD's L2 writes back what this workgroup stored)
            drain_vm();
            atomicAdd(&ctl->bar, 1u);
            long spins = 0;
            bool ab = false;
            while (ld_fresh(&ctl->bar) < target) {
                if (++spins > 64) __builtin_amdgcn_s_sleep(2);   // (the partners are usually a round trip or two away)
                if ((spins & 255) == 0 && (ld_fresh(&ctl->abort) || spins > kSearchSpinLimit)) { ab = true; break; }
            }
            if (ab) atomicExch(&ctl->abort, 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (this compute unit's L1 drops its lines; the wait holds the barrier below for it)
            drain_vm();
            s_abort = ab || ld_fresh(&ctl->abort);
        }
        __syncthreads();
        return s_abort != 0;
    };

    if (g == 0) {
        for (int i = tid; i <= p.Tmax; i += kSearchThreads) { out[i] = -1; if (nact) nact[i] = 0; }
        if (tid == 0) {   // frame 0: the start state (one survivor: workgroup 0 runs the frame alone)
            xstate[0] = 0; xscore[0] = 0.f; xla[0] = 0.0;
            xarcs[0] = make_int2(off[0], off[1] - off[0]);
            arena[0] = make_int2(-1, -1);
            if (rinfo) { rinfo[0] = make_int4(0, __float_as_int(0.f), off[0], off[1] - off[0]); ftab[0] = 0; ftab[1] = 1; }
            slot[0] = 0;
            if (nact) nact[0] = 1;
        }
    }
    unsigned long long mc[6] = {0, 0, 0, 0, 0, 0}, mt0 = 0;
    long long mfr[2] = {0, 0};
#define MW_MARK(i) do { if (p.prof) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); mc[i] += now_ - mt0; mt0 = now_; } } while (0)
    int nX = 1;
    int64_t base = 1, base_prev = 0;
    bool dead = false;
    bool lds_valid = false;   // workgroup 0: its LDS holds the survivors' arcs / scores / degrees (it compacted them alone)
    float G = 0.f;            // max over the survivors of (log-alpha - tropical score): keeps the log pass's terms <= 1
    __syncthreads();
    int t = 0;
    while (t < T) {
        const bool solo = nX < solo_max || NW == 1;   // (uniform over the utterance's workgroups)
        if (solo && g != 0) {
            // Workgroup 0 runs frames with few survivors alone, at the speed of search_kernel (no barrier in device
            // memory); the others sleep until it publishes the frame at which they are needed again, or the end.
            // (Workgroup 0 has the lowest id of the utterance: it is resident before any of its waiters.)
            if (tid == 0) {
                int r;
                long spins = 0;
                while ((r = ld_fresh(&ctl->resume_t)) <= t) {
                    __builtin_amdgcn_s_sleep(16);
                    if ((++spins & 255) == 0 && ld_fresh(&ctl->abort)) { r = 0x7fffffff; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                drain_vm();
                s_resume = r;
            }
            __syncthreads();
            const int r = s_resume;
            if (r >= T) return;   // the utterance ended (or died) in workgroup 0's hands
            nX = ld_fresh(&ctl->r_nx);
            base = ld_fresh(&ctl->r_base);
            base_prev = ld_fresh(&ctl->r_base_prev);
            G = ord2f(ld_fresh(&ctl->r_g));
            t = r;
            __syncthreads();
            continue;
        }
        SearchCtl::Gen* C = solo ? &s_gen : &ctl->c[t % 3];
        if (p.prof) { mt0 = __builtin_amdgcn_s_memtime(); ++mfr[solo ? 0 : 1]; }
        if (g == 0 && tid == 0) {
            SearchCtl::Gen* Cn = &ctl->c[(t + 1) % 3];   // (last used two frames ago; the next frame's counters)
            Cn->ncand = 0; Cn->fmax = 0; Cn->nsurv1 = 0; Cn->nsurv2 = 0; Cn->smin = 0xffffffffu; Cn->G1 = 0; Cn->G2 = 0;
            if (solo) { s_gen.ncand = 0; s_gen.fmax = 0; s_gen.nsurv1 = 0; s_gen.nsurv2 = 0; s_gen.smin = 0xffffffffu; s_gen.G1 = 0; s_gen.G2 = 0; }
        }
        const float* erow = lp0 + (int64_t)t * p.strideT;
        if (tid == 0) { s_ncand = 0; s_fmax = 0; s_keepbin = 0x7fffffff; }
        const int x0 = solo ? 0 : (int)((long long)nX * g / NW), x1 = solo ? nX : (int)((long long)nX * (g + 1) / NW);
        const bool in_lds = solo && lds_valid && nX <= kSearchXL;   // (one chunk, and its data is in the LDS already)
        auto sync_all = [&]() -> bool { if (solo) { __syncthreads(); return false; } return grid_sync(); };
        // ---------------- expand (as search_kernel, over this workgroup's block of the survivors)
        uint32_t lmax = 0;
        for (int log_pass = 0; log_pass < (lsum ? 2 : 1); ++log_pass) {
            for (int c0 = x0; c0 < x1; c0 += kSearchXL) {
                const int m = min(kSearchXL, x1 - c0);
                if (!(log_pass && in_lds)) {   // (one chunk in the LDS: the prefix of the tropical pass is still there)
                    constexpr int kPer = kSearchXL / kSearchThreads;
                    int deg[kPer], run = 0;
                    __syncthreads();   // (the previous chunk's searches are over; frame 0's survivor is written)
#pragma unroll
                    for (int k = 0; k < kPer; ++k) {
                        const int i = tid * kPer + k;
                        deg[k] = i < m ? (in_lds ? s_pfx[i] : ld_fresh(&xarcs[c0 + i].y)) : 0;
                        if (!in_lds && i < m) { s_xa0[i] = ld_fresh(&xarcs[c0 + i].x); s_xv[i] = ld_fresh(xscore + c0 + i); }
                    }
#pragma unroll
                    for (int k = 0; k < kPer; ++k) { const int x = deg[k]; deg[k] = run; run += x; }
                    int v = run;
                    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(v, o, 64); if (lane >= o) v += x; }
                    if (lane == 63) s_wsum[wave] = v;
                    __syncthreads();
                    int add = v - run;
                    for (int k = 0; k < wave; ++k) add += s_wsum[k];
#pragma unroll
                    for (int k = 0; k < kPer; ++k) s_pfx[tid * kPer + k] = add + deg[k];
                    if (tid == kSearchThreads - 1) s_narcs = add + run;
                }
                __syncthreads();
                const int NA = s_narcs;
                for (int k0 = 0; k0 < NA; k0 += kSearchThreads) {
                    const int k = k0 + tid;
                    bool ok = k < NA;
                    int lo = 0;
#pragma unroll
                    for (int st = kSearchXL / 2; st > 0; st >>= 1)
                        if (lo + st < m && s_pfx[lo + st] <= k) lo += st;
                    const int q = ok ? k - s_pfx[lo] + s_xa0[lo] : 0;
                    const int4 rec = orec[q];
                    const int d = rec.x, ar = rec.w;
                    const float we = __int_as_float(rec.z) + erow[direct ? rec.y : labels[rec.y]];
                    if (!log_pass) {
                        const float sc = s_xv[lo] + we;
                        ok = ok && sc > ninf;
                        const uint32_t o = f2ord(sc);
                        unsigned long long old = 1ull;
                        if (ok) {
                            lmax = max(lmax, o);
                            arcown[ar] = c0 + lo;
                            old = atomicMax(&best[d], ((unsigned long long)o << 32) | (0xffffffffu - (uint32_t)ar));
                        }
                        const bool fresh = old == 0ull;
                        const unsigned long long mk = __ballot(fresh);
                        if (fresh) {
                            const int leader = __ffsll((long long)mk) - 1;
                            int wb = 0;
                            if (lane == leader) wb = atomicAdd(&s_ncand, __popcll(mk));
                            wb = __shfl(wb, leader, 64);
                            const int idx = wb + __popcll(mk & ((1ull << lane) - 1ull));
                            if (idx < kSearchCL) s_cand[idx] = d; else cand[idx] = d;
                        }
                    } else {
                        const float tb = ord2f((uint32_t)(ld_fresh(best + d) >> 32));
                        const double la = ld_fresh(xla + c0 + lo);
                        if (ok && we > ninf) atomicAdd(&lsum[d], exp(la + (double)we - (double)tb - (double)G));
                    }
                }
            }
            if (!log_pass) {
                for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, o, 64));
                if (lane == 0 && lmax) atomicMax(&s_fmax, lmax);
                __syncthreads();
                if (tid == 0) {
                    if (s_fmax) atomicMax(&C->fmax, s_fmax);
                    if (s_ncand) atomicAdd(&C->ncand, s_ncand);
                }
            }
            if (!solo) MW_MARK(0);
            if (sync_all()) return;   // every value of the pass is final, for every workgroup
            if (!solo) MW_MARK(1);
        }
        // ---------------- prune
        const int n = s_ncand;                     // this workgroup's candidates
        const int ntot = ld_fresh(&C->ncand);      // the frame's
        if (ntot > 0) {
            const float fmax = ord2f(ld_fresh(&C->fmax));
            float cutoff = fmax - p.beam;
            float inv_width = 0.f;
            int keepbin = 0x7fffffff;
            const int want_min = min(p.min_active, ntot);
            uint32_t gmax = 0, lmin = 0xffffffffu;
            auto compact = [&](const bool from_best) __attribute__((always_inline)) {
                gmax = 0;
                int* counter = from_best ? &C->nsurv1 : &C->nsurv2;
                for (int i0 = 0; i0 < n; i0 += kSearchThreads) {
                    const int i = i0 + tid;
                    const bool in = i < n;
                    const int s = !in ? 0 : (i < kSearchCL ? s_cand[i] : ld_fresh(cand + i));
                    const unsigned long long pk = !in ? 0ull : (from_best ? ld_fresh(best + s) : ld_fresh(cpk + i));
                    const float sc = ord2f((uint32_t)(pk >> 32));
                    const bool keep = in && sc >= cutoff &&
                                      (keepbin == 0x7fffffff || min(kSearchBins - 1, (int)((fmax - sc) * inv_width)) <= keepbin);
                    if (in) lmin = min(lmin, (uint32_t)(pk >> 32));
                    double ls = 0.0;
                    if (lsum && in && (from_best || keep)) ls = from_best ? ld_fresh(lsum + s) : ld_fresh(cls + i);
                    const int o0 = keep ? off[s] : 0, o1 = keep ? off[s + 1] : 0;
                    const int ow = keep ? ld_fresh(arcown + (int)(0xffffffffu - (uint32_t)pk)) : 0;
                    if (from_best && in) {
                        best[s] = 0ull; cpk[i] = pk;
                        if (lsum) { lsum[s] = 0.0; cls[i] = ls; }
                    }
                    const unsigned long long mk = __ballot(keep);
                    int wb = 0;
                    if (lane == 0 && mk) wb = atomicAdd(counter, __popcll(mk));
                    wb = __shfl(wb, 0, 64);
                    if (keep) {
                        const int k = wb + __popcll(mk & ((1ull << lane) - 1ull));
                        if (k < cap) {
                            arena[base + k] = make_int2((int)(0xffffffffu - (uint32_t)pk), (int)(base_prev + ow));
                            if (rinfo) rinfo[base + k] = make_int4(s, __float_as_int(sc), o0, o1 - o0);
                            slot[s] = (int)(base + k);
                            xstate[k] = s;
                            xscore[k] = sc;
                            xarcs[k] = make_int2(o0, o1 - o0);
                            if (solo && k < kSearchXL) { s_xa0[k] = o0; s_xv[k] = sc; s_pfx[k] = o1 - o0; }
                            if (lsum) {
                                const double la = (double)sc + (double)G + log(ls);
                                xla[k] = la;
                                gmax = max(gmax, f2ord((float)(la - (double)sc)));
                            }
                        }
                    }
                }
                if (lsum) {
                    for (int o = 32; o > 0; o >>= 1) gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, o, 64));
                    if (lane == 0 && gmax) atomicMax(from_best ? &C->G1 : &C->G2, gmax);
                }
            };
            compact(true);
            for (int o = 32; o > 0; o >>= 1) lmin = min(lmin, (uint32_t)__shfl_xor((int)lmin, o, 64));
            if (lane == 0 && lmin != 0xffffffffu) atomicMin(&C->smin, lmin);
            if (!solo) MW_MARK(2);
            if (sync_all()) return;
            if (!solo) MW_MARK(3);
            const int cnt = ld_fresh(&C->nsurv1);
            const bool tighten = cnt > cap, loosen = !tighten && cnt < want_min;
            if (tighten || loosen) {   // (uniform over the utterance's workgroups)
                if (loosen) cutoff = ninf;
                const float lo = max(cutoff, ord2f(ld_fresh(&C->smin)));
                inv_width = fmax > lo ? (float)kSearchBins / (fmax - lo) : 0.f;
                s_hist[tid] = 0;
                __syncthreads();
                if (tid == 0) s_keepbin = tighten ? 0 : kSearchBins - 1;
                for (int i0 = 0; i0 < n; i0 += kSearchThreads) {
                    const int i = i0 + tid;
                    const unsigned long long pk = i < n ? ld_fresh(cpk + i) : 0ull;
                    const float sc = ord2f((uint32_t)(pk >> 32));
                    if (pk && sc >= cutoff) atomicAdd(&s_hist[min(kSearchBins - 1, (int)((fmax - sc) * inv_width))], 1);
                }
                __syncthreads();
                if (!solo) {   // the workgroups' histograms meet in device memory
                    if (s_hist[tid]) atomicAdd(&ctl->hist[tid], s_hist[tid]);
                    if (grid_sync()) return;
                }
                int v = solo ? s_hist[tid] : ld_fresh(&ctl->hist[tid]);
                for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(v, o, 64); if (lane >= o) v += x; }
                if (lane == 63) s_wsum[wave] = v;
                __syncthreads();
                for (int k = 0; k < wave; ++k) v += s_wsum[k];
                if (tighten) { if (v <= cap) atomicMax(&s_keepbin, tid); }
                else { if (v >= want_min) atomicMin(&s_keepbin, tid); }
                __syncthreads();
                keepbin = s_keepbin;
                if (tighten && keepbin == 0 && (solo ? s_hist[0] : ld_fresh(&ctl->hist[0])) > cap) {
                    // the best bin alone overflows (exact ties): search_kernel refines the histogram and keeps the frame's
                    // best candidate; the utterance goes to it (every workgroup of the frame sees the same bin)
                    if (tid == 0) atomicExch(&ctl->abort, 1);
                    return;
                }
                compact(false);
                if (!solo) {
                    if (grid_sync()) return;          // every workgroup has read the histogram and placed its survivors
                    if (g == 0) ctl->hist[tid] = 0;   // (its next use is behind the barriers of a later frame)
                } else {
                    __syncthreads();
                }
            }
        }
        // ---------------- the next frame's survivors (every workgroup of the frame reads the same counters)
        {
            const int n2 = ld_fresh(&C->nsurv2), n1 = ld_fresh(&C->nsurv1), nc = ld_fresh(&C->ncand);
            const bool second = nc > 0 && (n1 > cap || n1 < min(p.min_active, nc));   // the frame ran its second compaction
            nX = min(second ? n2 : n1, cap);
            if (lsum && nX > 0) G = ord2f(ld_fresh(second ? &C->G2 : &C->G1));
        }
        __syncthreads();   // (s_gen / s_ncand are rewritten by the next frame)
        MW_MARK(solo ? 5 : 4);
        lds_valid = solo && nX <= kSearchXL;
        if (g == 0 && tid == 0 && nact) nact[t + 1] = nX;
        base_prev = base;
        base += nX;
        if (g == 0 && tid == 0 && ftab) ftab[t + 2] = base;
        ++t;
        if (nX == 0) dead = true;
        if (solo && NW > 1) {
            // workgroup 0 after a frame of its own: call the others back when the next frame is theirs too, or send them home
            const bool over = dead || t >= T;
            if (over || nX >= solo_max) {
                if (tid == 0) {
                    ctl->r_nx = nX; ctl->r_base = base; ctl->r_base_prev = base_prev; ctl->r_g = f2ord(G);
                }
                drain_vm();        // every wave's stores of the solo frames have completed
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    drain_vm();
                    atomicExch(&ctl->resume_t, over ? 0x7fffffff : t);
                }
            }
        }
        if (dead) break;
    }
    if (p.prof && u == 0 && tid == 0 && (g == 0 || g == NW - 1))
        printf("search_mw_kernel utterance 0 workgroup %d of %d: %lld solo + %lld shared frames | cycles in shared frames: expand %llu, wait %llu, compact %llu, wait %llu, second compaction + tail %llu | solo frames %llu\n",
               g, NW, mfr[0], mfr[1], mc[0], mc[1], mc[2], mc[3], mc[4], mc[5]);
    if (g != 0) return;
    // ---------------- final arcs over the survivors of frame T (workgroup 0)
    const int NF = I[WF_I_NF];
    const int32_t* fsrc = p.blob + I[WF_I_FIN_SRC];
    const float* fw = (const float*)(p.blob + I[WF_I_FIN_W]);
    const int32_t* farc = p.blob + I[WF_I_FIN_ARC];
    if (tid == 0) { s_fin = 0ull; s_fmax = 0; s_lsum = 0.0; }
    __syncthreads();
    const int32_t* asrc = p.blob + I[WF_I_ARC_SRC];
    const int64_t baseT = base - nX;
    auto survivor = [&](int s) -> int {
        const int64_t r = (int64_t)ld_fresh(slot + s) - baseT;
        if (r < 0 || r >= nX) return -1;
        return ld_fresh(xstate + r) == s ? (int)r : -1;
    };
    if (!dead) {
        unsigned long long fb = 0ull;
        uint32_t lm = 0;
        for (int i = tid; i < NF; i += kSearchThreads) {
            const int j = survivor(fsrc[i]);
            if (j < 0) continue;
            const float v = ld_fresh(xscore + j) + fw[i];
            if (!(v > ninf)) continue;
            fb = max(fb, ((unsigned long long)f2ord(v) << 32) | (0xffffffffu - (uint32_t)farc[i]));
            if (lsum) lm = max(lm, f2ord((float)(ld_fresh(xla + j) + (double)fw[i])));
        }
        if (fb) atomicMax(&s_fin, fb);
        if (lm) atomicMax(&s_fmax, lm);
        __syncthreads();
        if (lsum && s_fmax) {
            const double ref = (double)ord2f(s_fmax);
            double acc = 0.0;
            for (int i = tid; i < NF; i += kSearchThreads) {
                const int j = survivor(fsrc[i]);
                if (j < 0 || !(fw[i] > ninf)) continue;
                acc += exp(ld_fresh(xla + j) + (double)fw[i] - ref);
            }
            acc = wave_sum_d(acc);
            if (lane == 0 && acc != 0.0) atomicAdd(&s_lsum, acc);
            __syncthreads();
        }
    }
    if (tid == 0) {
        const unsigned long long fb = dead ? 0ull : s_fin;
        double totl = -INFINITY;
        if (fb == 0ull) {
            p.best_score[u] = ninf;
        } else {
            p.best_score[u] = ord2f((uint32_t)(fb >> 32));
            const int fa = (int)(0xffffffffu - (uint32_t)fb);
            const int abase = I[WF_I_ARC_BASE];
            out[T] = abase + fa;
            int64_t r = ld_fresh(slot + asrc[fa]);
            for (int t = T - 1; t >= 0; --t) {
                const int2 rec = arena[r];
                out[t] = abase + rec.x;
                r = rec.y;
            }
            if (lsum && s_fmax) totl = (double)ord2f(s_fmax) + log(s_lsum);
        }
        if (p.tot_log) {
            if (p.tot_double) ((double*)p.tot_log)[u] = totl;
            else ((float*)p.tot_log)[u] = (float)totl;
        }
    }
}

// ------------------------------------------------------------------------------------------
// search_back_kernel: the output_beam of k2.get_lattice (utils/lattice_analysis.py:125) over the lattice the forward
// search left behind.  The lattice is every arc between a survivor of frame t and a survivor of frame t + 1 (and the
// final arcs out of the survivors of frame T); the forward kernels recorded, per record of the arena, the state and
// its tropical forward score (rinfo) and where each frame's records begin (ftab).  One workgroup per utterance walks
// the frames backwards, one lane per record: the tropical backward score bt is the max over the record's arcs of
// (w + e) + bt[dst]; an arc stays in the output lattice when alpha + ((w + e) + bt[dst]) >= best - output_beam (fp32,
// this association); bl, the log-semiring backward score over the arcs that stay (fp64), ends as the pruned lattice's
// total at the start state's record.  A state's record of frame t + 1 is found through slot[] (free after the forward
// pass), rewritten frame by frame and validated against the frame's range and the record's state.
// ------------------------------------------------------------------------------------------
struct SearchBackArgs {
    const int32_t* blob;
    const int64_t* utt;
    const int64_t* tab;
    const float* lp;
    int64_t strideB, strideT;
    int32_t* slot;               // state -> record, frames of the parity of T
    int32_t* slot2;              // the other frames (the forward pass's candidate list, free by now)
    const int4* rinfo;
    float* bt;                   // [records]
    double* bl;                  // [records]
    const int64_t* ftab;
    int ftab_stride;
    const float* best_score;
    float output_beam;
    void* tot_log;
    int tot_double;
    long long* lattice_arcs;     // [B] or null: arcs of the output lattice (final arcs included)
    int wgs, B, solo_max;        // workgroups per utterance (workgroup b serves utterance b % B, as search_mw_kernel)
    SearchCtl* ctl;              // [B] (wgs > 1, or redo)
    int redo;                    // second launch, one workgroup per utterance: only the utterances whose babort is set
    int test_drop;               // test aid, as SearchArgs::test_drop
};

constexpr int kBackRecs = 2;     // records per lane in flight (the frame is a chain of dependent loads: overlap two)
constexpr int kBackArcs = 4;     // arcs of a record in flight

__global__ void __launch_bounds__(kSearchThreads) search_back_kernel(SearchBackArgs p) {
    __shared__ unsigned long long s_arcs;
    __shared__ int s_abort;
    const int u = blockIdx.x % p.B, g = blockIdx.x / p.B, tid = threadIdx.x;
    SearchCtl* ctl = p.ctl ? p.ctl + u : nullptr;
    if (p.redo && !ctl->babort) return;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int64_t* R = p.tab + (int64_t)u * WF_SR_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int T = (int)U[WF_U_T];
    const float ninf = -INFINITY;
    const float best = p.best_score[u];
    auto put = [&](double tot, long long arcs) {
        if (p.tot_double) ((double*)p.tot_log)[u] = tot; else ((float*)p.tot_log)[u] = (float)tot;
        if (p.lattice_arcs) p.lattice_arcs[u] = arcs;
    };
    if (!(best > ninf)) {   // no path survived the forward search
        if (tid == 0 && g == 0) put(-INFINITY, 0);
        return;
    }
    const int64_t* ftab = p.ftab + (int64_t)u * p.ftab_stride;
    // Several workgroups share the frames of an utterance whose lattice is wide (one workgroup's scattered loads are what
    // bounds a frame: ~5 cache lines per arc); a narrow one is walked by workgroup 0 alone, without barriers in device
    // memory.  (Every workgroup of the utterance reads the same record count: the decision is uniform.)
    const bool multi = p.wgs > 1 && ftab[T + 1] >= (int64_t)max(p.solo_max, 2) * (T + 1);   // (records per frame, on average)
    if (!multi && g != 0) return;
    const int NW = multi ? p.wgs : 1;
    unsigned nbar = 0;
    if ((p.test_drop & 2) && multi && g == NW - 1) return;
    // barrier of the utterance's workgroups, bounded as search_mw_kernel's: true = given up (everyone leaves; the second
    // launch walks the utterance with one workgroup)
    auto sync_all = [&]() -> bool {
        if (multi) drain_vm();
        __syncthreads();
        if (!multi) return false;
        if (tid == 0) {
            ++nbar;
            const unsigned target = nbar * (unsigned)NW;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            drain_vm();
            atomicAdd(&ctl->bbar, 1u);
            long spins = 0;
            bool ab = false;
            while (ld_fresh(&ctl->bbar) < target) {
                if (++spins > 64) __builtin_amdgcn_s_sleep(2);
                if ((spins & 255) == 0 && (ld_fresh(&ctl->babort) || spins > kSearchSpinLimit)) { ab = true; break; }
            }
            if (ab) atomicExch(&ctl->babort, 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_vm();
            s_abort = ab || ld_fresh(&ctl->babort);
        }
        __syncthreads();
        return s_abort != 0;
    };
    if (p.redo && tid == 0) ctl->narcs = 0ull;
    const float thresh = best - p.output_beam;   // (-inf with output_beam = inf: every arc on a complete path stays)
    const int direct = I[WF_I_DIRECT];
    const int4* orec = (const int4*)(p.blob + I[WF_I_OUT_REC]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    // slot_of(t): state -> record of frame t.  Two arrays by the frame's parity: frame t's map is written while frame
    // t + 1's is being read.  What other frames (or the forward pass) left in them is told apart by the frame's range of
    // records and the record's state.
    int32_t* slot_a = p.slot + R[WF_SR_SOFF];
    int32_t* slot_b = p.slot2 + R[WF_SR_SOFF];
    auto slot_of = [&](int t) -> int32_t* { return ((T - t) & 1) ? slot_b : slot_a; };
    const int4* rinfo = p.rinfo + R[WF_SR_AOFF];
    float* bt = p.bt + R[WF_SR_AOFF];
    double* bl = p.bl + R[WF_SR_AOFF];
    uint32_t* btu = (uint32_t*)bt;   // frame T: order-preserving bits under atomicMax, 0 = no final arc
    const float* lp0 = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
    if (tid == 0) s_arcs = 0ull;
    long long narcs = 0;

    // ---------------- frame T: the final arcs (workgroup 0)
    if (g == 0) {
        const int64_t b1 = ftab[T];
        const int n1 = (int)(ftab[T + 1] - b1);
        const int NF = I[WF_I_NF];
        const int32_t* fsrc = p.blob + I[WF_I_FIN_SRC];
        const float* fw = (const float*)(p.blob + I[WF_I_FIN_W]);
        int32_t* slot = slot_of(T);
        for (int i = tid; i < n1; i += kSearchThreads) { slot[rinfo[b1 + i].x] = (int)(b1 + i); btu[b1 + i] = 0u; bl[b1 + i] = 0.0; }
        __syncthreads();
        auto record = [&](int s) -> int64_t {
            const int64_t j = ld_fresh(slot + s);
            return (j >= b1 && j < b1 + n1 && rinfo[j].x == s) ? j : -1;
        };
        for (int i = tid; i < NF; i += kSearchThreads) {
            const int64_t j = record(fsrc[i]);
            if (j >= 0 && fw[i] > ninf) atomicMax(&btu[j], f2ord(fw[i]));
        }
        __syncthreads();
        for (int i = tid; i < NF; i += kSearchThreads) {
            const int64_t j = record(fsrc[i]);
            if (j < 0 || !(fw[i] > ninf)) continue;
            if (__int_as_float(rinfo[j].y) + fw[i] >= thresh) {
                atomicAdd(&bl[j], exp((double)fw[i] - (double)ord2f(ld_fresh(btu + j))));
                ++narcs;
            }
        }
        __syncthreads();
        for (int i = tid; i < n1; i += kSearchThreads) {
            const uint32_t o = ld_fresh(btu + b1 + i);
            const double acc = ld_fresh(bl + b1 + i);
            const float m = o ? ord2f(o) : ninf;
            bt[b1 + i] = m;
            bl[b1 + i] = acc > 0.0 ? (double)m + log(acc) : -INFINITY;
        }
    }
    if (sync_all()) return;
    // ---------------- frames T - 1 .. 0 (a contiguous block of the frame's records per workgroup)
    for (int t = T - 1; t >= 0; --t) {
        const int64_t b0 = ftab[t], b1 = ftab[t + 1];
        const int n0 = (int)(b1 - b0), n1 = (int)(ftab[t + 2] - b1);
        const float* erow = lp0 + (int64_t)t * p.strideT;
        const int32_t* slot1 = slot_of(t + 1);
        int32_t* slot0 = slot_of(t);
        const bool shared = NW > 1 && n0 >= 4 * NW;
        const int x0 = shared ? (int)((long long)n0 * g / NW) : 0, x1 = shared ? (int)((long long)n0 * (g + 1) / NW) : (g == 0 ? n0 : 0);
        for (int i0 = x0; i0 < x1; i0 += kBackRecs * kSearchThreads) {
            int4 me[kBackRecs];
            float m[kBackRecs];
            double mx[kBackRecs], acc[kBackRecs];
            int maxdeg = 0;
#pragma unroll
            for (int r = 0; r < kBackRecs; ++r) {
                const int i = i0 + r * kSearchThreads + tid;
                me[r] = i < x1 ? rinfo[b0 + i] : make_int4(0, 0, 0, 0);
                if (i < x1) slot0[me[r].x] = (int)(b0 + i);
                m[r] = ninf; mx[r] = -INFINITY; acc[r] = 0.0;
                maxdeg = max(maxdeg, me[r].w);
            }
            for (int q0 = 0; q0 < maxdeg; q0 += kBackArcs) {
                int d[kBackRecs][kBackArcs];
                float we[kBackRecs][kBackArcs], tb[kBackRecs][kBackArcs];
                int64_t j[kBackRecs][kBackArcs];
                double lb[kBackRecs][kBackArcs];
                int st[kBackRecs][kBackArcs];
#pragma unroll
                for (int r = 0; r < kBackRecs; ++r)
#pragma unroll
                    for (int k = 0; k < kBackArcs; ++k) {
                        const bool in = q0 + k < me[r].w;
                        const int4 rec = orec[me[r].z + (in ? q0 + k : 0)];
                        d[r][k] = in ? rec.x : -1;
                        we[r][k] = __int_as_float(rec.z) + erow[direct ? rec.y : labels[rec.y]];
                    }
#pragma unroll
                for (int r = 0; r < kBackRecs; ++r)
#pragma unroll
                    for (int k = 0; k < kBackArcs; ++k) {
                        const int64_t x = d[r][k] >= 0 ? (int64_t)ld_fresh(slot1 + d[r][k]) : -1;
                        j[r][k] = (x >= b1 && x < b1 + n1) ? x : -1;
                    }
#pragma unroll
                for (int r = 0; r < kBackRecs; ++r)
#pragma unroll
                    for (int k = 0; k < kBackArcs; ++k) {
                        const int64_t x = j[r][k] >= 0 ? j[r][k] : b1;   // (b1: a record that exists; its values are not used)
                        st[r][k] = rinfo[x].x;
                        tb[r][k] = ld_fresh(bt + x);
                        lb[r][k] = ld_fresh(bl + x);
                    }
#pragma unroll
                for (int r = 0; r < kBackRecs; ++r)
#pragma unroll
                    for (int k = 0; k < kBackArcs; ++k) {
                        if (j[r][k] < 0 || st[r][k] != d[r][k]) continue;
                        const float vb = we[r][k] + tb[r][k];
                        if (!(vb > ninf)) continue;
                        m[r] = max(m[r], vb);
                        if (__int_as_float(me[r].y) + vb >= thresh) {
                            ++narcs;
                            const double v = (double)we[r][k] + lb[r][k];
                            if (v > mx[r]) { acc[r] = acc[r] * exp(mx[r] - v) + 1.0; mx[r] = v; }
                            else if (v > -INFINITY) acc[r] += exp(v - mx[r]);
                        }
                    }
            }
#pragma unroll
            for (int r = 0; r < kBackRecs; ++r) {
                const int i = i0 + r * kSearchThreads + tid;
                if (i < x1) {
                    bt[b0 + i] = m[r];
                    bl[b0 + i] = acc[r] > 0.0 ? mx[r] + log(acc[r]) : -INFINITY;
                }
            }
        }
        if (sync_all()) return;   // (frame t's scores and map are complete; every lookup of frame t + 1 is over)
    }
    if (narcs) atomicAdd(&s_arcs, (unsigned long long)narcs);
    __syncthreads();
    if (!ctl) {
        if (tid == 0) put(ld_fresh(bl + 0), (long long)s_arcs);
        return;
    }
    if (tid == 0 && s_arcs) atomicAdd(&ctl->narcs, s_arcs);
    if (sync_all()) return;
    if (tid == 0 && g == 0) put(ld_fresh(bl + 0), (long long)ld_fresh(&ctl->narcs));
}

// ------------------------------------------------------------------------------------------
// search_words_kernel: `latticeW = k2.connect(k2.compose(lattice, W))` with W = k2.linear_fst(words, words) and its
// get_tot_scores (utils/lattice_analysis.py:137-139,161-162), after search_back_kernel: the output lattice restricted
// to the paths whose non-epsilon aux labels spell the utterance's word sequence.  Cell (record, k words consumed):
// L = log-semiring backward score (fp64), M = tropical (fp32); an arc of the lattice (kept by output_beam: the test of
// search_back_kernel, from its bt) with aux 0 goes to (dst, k), with aux = words[k] to (dst, k + 1), any other aux
// nowhere; frame T starts from search_back_kernel's values at k = n.  Two frames of cells per utterance (ring by frame
// parity), one lane per cell, one workgroup per utterance; the start record's cell k = 0 holds the two totals.
// ------------------------------------------------------------------------------------------
struct SearchWordsArgs {
    SearchBackArgs b;            // (bt / bl: what search_back_kernel left)
    const int32_t* aux;          // device, per arc of the FsaVec (0 = epsilon, -1 on final arcs)
    const int64_t* woff;         // [B] first cell of the utterance's ring
    const int32_t* wtab;         // [B][wstride]: n, then the n word ids
    int wstride;
    double* wl;
    float* wt;
    void* tot_w;                 // fp32 / fp64 [B] (b.tot_double)
    float* best_w;               // [B]
};

__global__ void __launch_bounds__(kSearchThreads) search_words_kernel(SearchWordsArgs q) {
    __shared__ int s_band[3][2];   // {smallest, largest} word count with a finite cell, frame t in generation (T - t) % 3
    __shared__ int s_abort;
    const SearchBackArgs& p = q.b;
    const int u = blockIdx.x % p.B, g = blockIdx.x / p.B, tid = threadIdx.x;
    SearchCtl* ctl = p.ctl ? p.ctl + u : nullptr;
    if (p.redo && !ctl->wabort) return;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int64_t* R = p.tab + (int64_t)u * WF_SR_WORDS;
    const int32_t* I = p.blob + WF_HDR_WORDS + (int)U[WF_U_FSA] * WF_INFO_WORDS;
    const int T = (int)U[WF_U_T];
    const float ninf = -INFINITY;
    const float best = p.best_score[u];
    auto put = [&](double tot, float bw) {
        if (p.tot_double) ((double*)q.tot_w)[u] = tot; else ((float*)q.tot_w)[u] = (float)tot;
        q.best_w[u] = bw;
    };
    if (!(best > ninf)) {
        if (tid == 0 && g == 0) put(-INFINITY, ninf);
        return;
    }
    const int64_t* ftab = p.ftab + (int64_t)u * p.ftab_stride;
    // as search_back_kernel: a wide lattice is shared by the utterance's workgroups (contiguous blocks of a frame's cells,
    // one bounded barrier in device memory per frame, a one-workgroup second launch for what it gives up on)
    const bool multi = p.wgs > 1 && ftab[T + 1] >= (int64_t)max(p.solo_max, 2) * (T + 1);
    if (!multi && g != 0) return;
    const int NW = multi ? p.wgs : 1;
    unsigned nbar = 0;
    if ((p.test_drop & 4) && multi && g == NW - 1) return;
    auto sync_all = [&]() -> bool {
        if (multi) drain_vm();
        __syncthreads();
        if (!multi) return false;
        if (tid == 0) {
            ++nbar;
            const unsigned target = nbar * (unsigned)NW;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            drain_vm();
            atomicAdd(&ctl->wbar, 1u);
            long spins = 0;
            bool ab = false;
            while (ld_fresh(&ctl->wbar) < target) {
                if (++spins > 64) __builtin_amdgcn_s_sleep(2);
                if ((spins & 255) == 0 && (ld_fresh(&ctl->wabort) || spins > kSearchSpinLimit)) { ab = true; break; }
            }
            if (ab) atomicExch(&ctl->wabort, 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            drain_vm();
            s_abort = ab || ld_fresh(&ctl->wabort);
        }
        __syncthreads();
        return s_abort != 0;
    };
    int (*band)[2] = multi ? ctl->kband : s_band;   // (device memory when several workgroups collect into it)
    const float thresh = best - p.output_beam;
    const int direct = I[WF_I_DIRECT];
    const int4* orec = (const int4*)(p.blob + I[WF_I_OUT_REC]);
    const int32_t* labels = p.blob + I[WF_I_LABELS];
    const int32_t* aux = q.aux + I[WF_I_ARC_BASE];
    int32_t* slot_a = p.slot + R[WF_SR_SOFF];
    int32_t* slot_b = p.slot2 + R[WF_SR_SOFF];
    auto slot_of = [&](int t) -> int32_t* { return ((T - t) & 1) ? slot_b : slot_a; };
    const int4* rinfo = p.rinfo + R[WF_SR_AOFF];
    const float* bt = p.bt + R[WF_SR_AOFF];
    const double* bl = p.bl + R[WF_SR_AOFF];
    const float* lp0 = p.lp + U[WF_U_SEQ] * p.strideB + U[WF_U_START] * p.strideT;
    const int32_t* W = q.wtab + (int64_t)u * q.wstride;
    const int n = W[0], N1 = n + 1;
    ++W;
    const int64_t cap = R[WF_SR_CAP];
    double* Lr[2] = {q.wl + q.woff[u], q.wl + q.woff[u] + cap * N1};
    float* Mr[2] = {q.wt + q.woff[u], q.wt + q.woff[u] + cap * N1};
    // ---------------- frame T (workgroup 0)
    if (g == 0) {
        const int64_t b1 = ftab[T];
        const int n1 = (int)(ftab[T + 1] - b1);
        int32_t* slot = slot_of(T);
        double* L = Lr[0];
        float* M = Mr[0];
        if (tid == 0) {
            band[0][0] = n; band[0][1] = n;
            band[1][0] = 0x7fffffff; band[1][1] = -1;
            band[2][0] = 0x7fffffff; band[2][1] = -1;
        }
        for (int64_t x = tid; x < (int64_t)n1 * N1; x += kSearchThreads) {
            const int i = (int)(x / N1), k = (int)(x % N1);
            const double l = bl[b1 + i];
            const bool end = k == n && l > -INFINITY;
            L[x] = end ? l : -INFINITY;
            M[x] = end ? bt[b1 + i] : ninf;
            if (k == 0) slot[rinfo[b1 + i].x] = (int)(b1 + i);
        }
    }
    if (sync_all()) return;
    // ---------------- frames T - 1 .. 0
    for (int t = T - 1; t >= 0; --t) {
        const int64_t b0 = ftab[t], b1 = ftab[t + 1];
        const int n0 = (int)(b1 - b0), n1 = (int)(ftab[t + 2] - b1);
        const float* erow = lp0 + (int64_t)t * p.strideT;
        const int32_t* slot1 = slot_of(t + 1);
        int32_t* slot0 = slot_of(t);
        const double* Ln = Lr[(T - t - 1) & 1];
        const float* Mn = Mr[(T - t - 1) & 1];
        double* L = Lr[(T - t) & 1];
        float* M = Mr[(T - t) & 1];
        const int gn = (T - t - 1) % 3, gc = (T - t) % 3, gp = (T - t + 1) % 3;   // frame t + 1's band, this frame's, frame t - 1's
        const int blo = ld_fresh(&band[gn][0]), bhi = ld_fresh(&band[gn][1]);
        if (bhi < 0) {   // (uniform) no cell of frame t + 1 is finite: the lattice holds no path with these words
            if (tid == 0 && g == 0) put(-INFINITY, ninf);
            return;
        }
        // (frame t - 1 collects into generation gp, last read -- as frame t + 2's band -- before the barrier that ended frame t + 1)
        if (tid == 0 && g == 0) { band[gp][0] = 0x7fffffff; band[gp][1] = -1; }
        const int klo = blo - 1, khi = bhi;   // a cell reaches k or k + 1 of the next frame
        // Only the word counts that can be finite are computed, plus one column on either side that frame t - 1 may read
        // (written as -inf): cells [kb0, kb1] of every record.
        const int kb0 = max(klo - 1, 0), kb1 = min(khi + 1, n), KW = kb1 - kb0 + 1;
        const int64_t ncell = (int64_t)n0 * KW;
        const int64_t y0 = NW > 1 ? ncell * g / NW : 0, y1 = NW > 1 ? ncell * (g + 1) / NW : ncell;
        int my_lo = 0x7fffffff, my_hi = -1;
        for (int64_t y = y0 + tid; y < y1; y += kSearchThreads) {
            const int i = (int)(y / KW), k = kb0 + (int)(y % KW);
            const int64_t x = (int64_t)i * N1 + k;
            const int4 me = rinfo[b0 + i];
            if (k == kb0) slot0[me.x] = (int)(b0 + i);
            const float a = __int_as_float(me.y);
            float m = ninf;
            double mx = -INFINITY, acc = 0.0;
            const int narc = (k < klo || k > khi) ? 0 : me.w;
            for (int e = 0; e < narc; ++e) {
                const int4 rec = orec[me.z + e];
                const int ax = aux[rec.w];
                int k2 = k;
                if (ax > 0) {
                    if (k >= n || W[k] != ax) continue;
                    k2 = k + 1;
                }
                const int d = rec.x;
                const int64_t j = ld_fresh(slot1 + d);
                if (j < b1 || j >= b1 + n1 || rinfo[j].x != d) continue;
                const float we = __int_as_float(rec.z) + erow[direct ? rec.y : labels[rec.y]];
                const float vb = we + bt[j];
                if (!(vb > ninf) || !(a + vb >= thresh)) continue;       // not an arc of the output lattice
                const int64_t c = (j - b1) * N1 + k2;
                const double v = (double)we + ld_fresh(Ln + c);
                m = max(m, we + ld_fresh(Mn + c));
                if (v > mx) { acc = acc * exp(mx - v) + 1.0; mx = v; }
                else if (v > -INFINITY) acc += exp(v - mx);
            }
            L[x] = acc > 0.0 ? mx + log(acc) : -INFINITY;
            M[x] = m;
            if (acc > 0.0) { my_lo = min(my_lo, k); my_hi = max(my_hi, k); }
        }
        if (my_hi >= 0) { atomicMin(&band[gc][0], my_lo); atomicMax(&band[gc][1], my_hi); }
        if (sync_all()) return;
    }
    // (the start record's cell k = 0: written only when frame 0's band of finite word counts reaches down to 0)
    if (tid == 0 && g == 0) {
        if (ld_fresh(&band[T % 3][0]) == 0) put(ld_fresh(Lr[T & 1] + 0), ld_fresh(Mr[T & 1] + 0));
        else put(-INFINITY, ninf);
    }
}

// ------------------------------------------------------------------------------------------
// host side: workspace planning and launches
// ------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Plan {
    int B = 0;
    size_t off_hubsave = 0;
    size_t off_utt = 0, off_seq2utt = 0, off_gscale = 0, off_totd = 0, off_tot2 = 0, off_ctot2 = 0, off_flag = 0, off_aoff = 0,
           off_boff = 0, off_E = 0, off_P = 0, off_cmax = 0, off_alpha = 0, off_beta = 0, off_bp = 0, off_finarc = 0, total = 0;
    std::vector<int64_t> utt;  // B * WF_U_WORDS
    int smax = 0, kmax_compact = 0, kmax = 0, amax = 0, hmax = 0, maxlabdeg = 0;
    bool any_compact = false, all_compact = true, any_heavy = false, arcs_lds_ok = true;
    // register-resident sweep eligibility (per direction)
    bool reg_ok_in = true, reg_ok_out = true;
    int over_deg_in = 0, over_deg_out = 0;
    int64_t n_E = 0, n_ab = 0, n_P = 0, n_rows = 0;
    bool lin = false;   // forward-backward by a linear-domain sweep (decided in make_plan)
    bool pruned = false;   // output_beam pruning: tropical rows + thresholds live in the workspace
    bool pruned_fast = false;   // ... on the banded probability-domain path (trop_band_kernel + band_mask_kernel + MASK sweep)
    size_t off_mA = 0, off_mB = 0, off_best = 0, off_thr = 0, off_bmask = 0;
    bool big = false;       // graphs too large for the LDS sweeps: one launch per frame
    bool shared = false;    // ... and the whole batch uses one graph (forward-only calls): kNB utterances per lane
    bool one_fsa = true;
    bool linbig = false;    // lin, by sweep_linbig_kernel (emission probabilities kept as fp32)
    bool hub = false;       // lin, by sweep_hub_kernel / post_hub_kernel: every graph is in hub form with the same (G, NC)
    int hub_g = -1, hub_nc = 0, hub_ng = 0;   // (-1: no graph seen yet; 0: not all in the same hub form)
    int hub_pat = -2;                         // index into kHubPatterns when every graph has that pattern (-2: none seen, -1: other)
    int hub_kpf = 0, hub_kpb = 0;             // split ring of sweep_hub_kernel: buffer pitches that hold every graph's parts (-1: some graph has no split)
    size_t off_big = 0, off_Et = 0;
    int64_t nstates = 0;
    int nheavy_in_max = 0, nheavy_out_max = 0;
    int band_bw = 0, band_sn = 0, band_m = 1;
    bool band_we = true;   // every arc weight of every graph is 0   // banded form: widest band / most positions (band_bw < 0: some graph is not banded)
    int Tlong = 0;
};

int check_blob(const void* graph_host) {
    if (!graph_host) return wf_fail(WF_EINVAL, "null prepared graph");
    const int32_t* H = (const int32_t*)graph_host;
    if (H[WF_H_MAGIC] != WF_BLOB_MAGIC) return wf_fail(WF_EINVAL, "prepared graph: bad magic");
    return WF_OK;
}

// Every segment inside the log_probs box [num_seqs][Tmax][V]: the only guard between a bad supervision_segments row
// and an out-of-bounds device read, so every compute entry point runs it (the Python drop-in asserts the same).
int check_segments(const char* who, const int32_t* seg, int B, int num_seqs, int Tmax) {
    if (!seg || B < 1) return wf_fail(WF_EINVAL, "%s: bad B/seg", who);
    for (int i = 0; i < B; ++i) {
        const int64_t b = seg[3 * i], st = seg[3 * i + 1], T = seg[3 * i + 2];
        if (b < 0 || b >= num_seqs || st < 0 || T < 0 || st + T > Tmax)
            return wf_fail(WF_ERANGE, "%s: segment %d = (%d,%d,%d) outside log_probs [%d][%d][.]", who, i, (int)b, (int)st, (int)T, num_seqs, Tmax);
    }
    return WF_OK;
}

// mode: 0 = forward-backward (alpha+beta arenas), 1 = tot only, 2 = viterbi (bp arena)
struct RegCfg {
    bool ok = false;
    int threads = 0, R = 1, HD = 0, smax = 0;
    size_t lds = 0;
};
RegCfg lin_config(const Plan& P);
size_t sweep_lds_bytes(const Plan& P, size_t acc_bytes);
constexpr int kBigMinStates = 0x7fffffff;   // per-frame launches only where nothing else fits, until measured otherwise
// Test / measurement hooks (environment variables WF_*): read ONCE, at the first call of the process, and again only
// when a test calls wf_debug_reload_env() after changing them -- no getenv on the production path.
struct Hooks {
    int no_band, no_gen, no_lin, no_linbig, no_reg, no_wave, no_fwd_via_fb, no_gather_small, no_post_small, no_overlap,
        zero_memset, zero_blocks, force_fallback, big_min_s, no_shared_big, post_big_min_s, sweep_ablate, no_arc_cache, no_pruned_fast,
        post_profile, sweep_profile, quad_r, no_post_check, fill_side_stream, no_hub, no_hub_pattern, no_tab_kernarg, no_hub_split, hub_force_split, no_dbl_fast, no_hub_overlap;
};
Hooks g_hooks;
bool g_hooks_loaded = false;
int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
void load_hooks() {
    Hooks h{};
    h.no_band = env_int("WF_NO_BAND_SWEEP", 0); h.no_gen = env_int("WF_NO_GEN_SWEEP", 0);
    h.no_lin = env_int("WF_NO_LIN_SWEEP", 0); h.no_linbig = env_int("WF_NO_LINBIG_SWEEP", 0);
    h.no_reg = env_int("WF_NO_REG_SWEEP", 0); h.no_wave = env_int("WF_NO_WAVE_SWEEP", 0);
    h.no_fwd_via_fb = env_int("WF_NO_FWD_VIA_FB", 0); h.no_gather_small = env_int("WF_NO_GATHER_SMALL", 0);
    h.no_post_small = getenv("WF_NO_POST_SMALL") != nullptr; h.no_overlap = getenv("WF_NO_OVERLAP") != nullptr;
    h.zero_memset = getenv("WF_ZERO_MEMSET") != nullptr; h.zero_blocks = env_int("WF_ZERO_BLOCKS", 0);
    h.force_fallback = env_int("WF_FORCE_FALLBACK", 0); h.big_min_s = env_int("WF_BIG_MIN_S", kBigMinStates);
    h.no_shared_big = env_int("WF_NO_SHARED_BIG", 0); h.post_big_min_s = env_int("WF_POST_BIG_MIN_S", 4096);
    h.sweep_ablate = env_int("WF_SWEEP_ABLATE", 0); h.no_arc_cache = env_int("WF_NO_ARC_CACHE", 0);
    h.post_profile = env_int("WF_POST_PROFILE", 0); h.sweep_profile = env_int("WF_SWEEP_PROFILE", 0);
    h.quad_r = env_int("WF_QUAD_R", 0);
    h.no_post_check = env_int("WF_NO_POST_CHECK", 0); h.fill_side_stream = env_int("WF_FILL_SIDE_STREAM", 0);
    h.no_hub = env_int("WF_NO_HUB_SWEEP", 0); h.no_hub_pattern = env_int("WF_NO_HUB_PATTERN", 0);
    h.no_tab_kernarg = env_int("WF_NO_TAB_KERNARG", 0); h.no_hub_split = env_int("WF_NO_HUB_SPLIT", 0);
    h.hub_force_split = env_int("WF_HUB_FORCE_SPLIT", 0);
    h.no_dbl_fast = env_int("WF_NO_DOUBLE_FAST", 0);
    h.no_hub_overlap = env_int("WF_NO_HUB_OVERLAP", 0);
    h.no_pruned_fast = env_int("WF_NO_PRUNED_FAST", 0);
    g_hooks = h;
    g_hooks_loaded = true;
}
inline const Hooks& hk() {
    if (!g_hooks_loaded) load_hooks();
    return g_hooks;
}
struct BandSel { bool ok = false; int R = 0, BW = 0, M = 1, grp = 0; bool we = false; bool gen = false; };
BandSel band_config(const Plan& P);
bool linbig_config(const Plan& P);
bool hub_config(const Plan& P);
bool post_small_fits(const Plan& P);

// dbl_exact: WF_DOUBLE_ACC (k2 use_double_scores=True): exp/log in fp64 too, log-semiring sweep only
int make_plan(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int B, int V,
              int num_seqs, size_t acc_bytes, int mode, Plan* P, bool dbl_exact = false, bool pruned = false) {
    int rc = check_blob(graph_host);
    if (rc) return rc;
    if (B < 1 || !seg || V < 1) return wf_fail(WF_EINVAL, "bad B/seg/V");
    const int32_t* H = (const int32_t*)graph_host;
    const int G = H[WF_H_NUM_FSAS];
    if (!utt2fsa && G != B) return wf_fail(WF_EINVAL, "graph has %d FSAs but %d segments (pass utt2fsa to share graphs)", G, B);
    P->B = B;
    P->utt.assign((size_t)B * WF_U_WORDS, 0);
    int64_t nE = 0, nab = 0, nP = 0, row0 = 0;
    for (int i = 0; i < B; ++i) {
        const int g = utt2fsa ? utt2fsa[i] : i;
        if (g < 0 || g >= G) return wf_fail(WF_ERANGE, "utt2fsa[%d] = %d out of range", i, g);
        const int32_t* I = H + WF_HDR_WORDS + (size_t)g * WF_INFO_WORDS;
        if (I[WF_I_MAX_LABEL] >= V) return wf_fail(WF_ERANGE, "fsa %d uses label %d but log_probs has V=%d columns", g, I[WF_I_MAX_LABEL], V);
        const int b = seg[3 * i], st = seg[3 * i + 1], T = seg[3 * i + 2];
        if (b < 0 || st < 0 || T < 0 || (num_seqs > 0 && b >= num_seqs)) return wf_fail(WF_ERANGE, "segment %d = (%d,%d,%d) out of range", i, b, st, T);
        int64_t* U = &P->utt[(size_t)i * WF_U_WORDS];
        U[WF_U_FSA] = g; U[WF_U_SEQ] = b; U[WF_U_START] = st; U[WF_U_T] = T;
        if (g != (utt2fsa ? utt2fsa[0] : 0)) P->one_fsa = false;
        U[WF_U_ALPHA] = nab; U[WF_U_BETA] = nab; U[WF_U_E] = nE; U[WF_U_ROW0] = row0;
        U[WF_U_OFFS] = row0 + i;   // sum of (T+1) over the utterances before this one
        U[WF_U_SOFF] = P->nstates;
        P->nstates += I[WF_I_S];
        P->nheavy_in_max = std::max(P->nheavy_in_max, (int)I[WF_I_NHEAVY_IN]);
        P->nheavy_out_max = std::max(P->nheavy_out_max, (int)I[WF_I_NHEAVY_OUT]);
        U[WF_U_P] = nP;
        U[WF_U_PP] = (I[WF_I_K] + 1) & ~1;
        nP += (int64_t)T * U[WF_U_PP];
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
        if (I[WF_I_BAND_BW] < 0) P->band_bw = -1;
        else if (P->band_bw >= 0) P->band_bw = std::max(P->band_bw, (int)I[WF_I_BAND_BW]);
        P->band_sn = std::max(P->band_sn, (int)I[WF_I_S] - 1);
        P->band_m = std::max(P->band_m, (int)I[WF_I_BAND_M]);
        if (!I[WF_I_BAND_WE]) P->band_we = false;
        {
            int pat = -1;
            if (I[WF_I_HUB_G] == 2) {   // does the coefficient table equal one of the compile-time patterns?
                const int nc = I[WF_I_HUB_NC];
                const float* cfh = (const float*)(H + I[WF_I_HUB_COEF]);
                for (int q = 0; q < kHubNumPatterns && pat < 0; ++q) {
                    float want[9 * WF_HUB_MAX_NC] = {};
                    bool fits = true;
                    for (int e = 0; e < kHubPatterns[q].n; ++e) {
                        const int i = kHubPatterns[q].e[e] >> 4, j = (kHubPatterns[q].e[e] >> 2) & 3, c = kHubPatterns[q].e[e] & 3;
                        if (c >= nc) { fits = false; break; }
                        want[(i * 3 + j) * nc + c] += 1.f;
                    }
                    if (fits && !memcmp(want, cfh, sizeof(float) * 9 * nc)) pat = q;
                }
            }
            if (I[WF_I_HUB_G] > 0 && P->hub_kpf >= 0) {
                // split ring: tokens of a group consecutive; the largest front part such that three front and two back
                // buffers fit the LDS
                const int nc = I[WF_I_HUB_NC], cs = I[WF_I_HUB_CS], K = I[WF_I_K];
                const int32_t* c0h = H + I[WF_I_HUB_C0];
                bool inter = cs == nc;
                for (int c = 1; c < nc; ++c) inter = inter && c0h[c] == c0h[0] + c;
                int best_f = 0, best_b = 0;
                for (int k = kHubSlots - 1; k >= 1 && inter && !best_f; --k) {
                    const int64_t f = (int64_t)c0h[0] + (int64_t)kHubCompute * k * cs;
                    if (f >= K) continue;
                    const size_t pf = align_up((size_t)f, 64) + 64, pb = align_up((size_t)(K - f), 64) + 64;
                    if (kHubLdsFixed + (3 * pf + 2 * pb) * 4 <= kLdsBudget) { best_f = (int)pf; best_b = (int)pb; }
                }
                if (!best_f) P->hub_kpf = -1;
                else { P->hub_kpf = std::max(P->hub_kpf, best_f); P->hub_kpb = std::max(P->hub_kpb, best_b); }
            }
            if (P->hub_g == -1) { P->hub_g = I[WF_I_HUB_G]; P->hub_nc = I[WF_I_HUB_NC]; P->hub_pat = pat; }
            else if (P->hub_g != I[WF_I_HUB_G] || P->hub_nc != I[WF_I_HUB_NC]) P->hub_g = 0;
            else if (P->hub_pat != pat) P->hub_pat = -1;
        }
        row0 += T;
    }
    // graphs whose sweep state does not fit the LDS (or, test hook, with at least WF_BIG_MIN_S states): per-frame launches
    P->big = sweep_lds_bytes(*P, acc_bytes) > kLdsBudget || P->smax >= hk().big_min_s;
    if (mode == 0) {
        P->pruned = pruned;
        const bool hub_ok = hub_config(*P);
        // output_beam pruning stays on the probability-domain path where the banded wave sweep applies (every numerator,
        // the small denominators); elsewhere: tropical + masked log-semiring sweeps (sweep_kernel / big_step_kernel)
        P->pruned_fast = pruned && acc_bytes == 8 && !dbl_exact && !P->big && band_config(*P).ok && post_small_fits(*P) &&
                         P->kmax <= kBandMaskKMax && !hk().no_pruned_fast;
        // WF_DOUBLE_ACC: the banded wave sweep is fp64 from the emission probabilities to the total (sweep_band_kernel); the
        // other probability-domain kernels keep fp32 pieces, so their graphs take the fp64 log-semiring sweep
        const bool dbl_fast = dbl_exact && !pruned && band_config(*P).ok && !hk().no_dbl_fast;
        P->lin = acc_bytes == 8 && (!dbl_exact || dbl_fast) && !P->big &&
                 (pruned ? P->pruned_fast : (band_config(*P).ok || lin_config(*P).ok || hub_ok || linbig_config(*P)));
        P->hub = P->lin && !band_config(*P).ok && !lin_config(*P).ok && hub_ok;
        P->linbig = P->lin && !band_config(*P).ok && !lin_config(*P).ok && !P->hub;
        if (P->hub) nP = 0;   // no probability arena: the sweep reads log_probs itself
    }
    P->n_E = nE; P->n_ab = nab; P->n_P = nP; P->n_rows = row0;
    size_t o = 0;
    P->off_utt = o; o = align_up(o + sizeof(int64_t) * WF_U_WORDS * B, 256);
    P->off_seq2utt = o; o = align_up(o + sizeof(int32_t) * std::max(num_seqs, 1), 256);
    P->off_gscale = o; o = align_up(o + sizeof(float) * B, 256);
    P->off_finarc = o; o = align_up(o + sizeof(int32_t) * B, 256);
    P->off_totd = o; o = align_up(o + sizeof(double) * B, 256);
    P->off_hubsave = o; o = align_up(o + sizeof(double) * 8 * (size_t)B, 256);
    P->off_tot2 = o; o = align_up(o + sizeof(double) * B, 256);
    P->off_E = o; o = align_up(o + sizeof(float) * (size_t)nE, 256);
    if (mode == 0) {
        P->off_ctot2 = o; o = align_up(o + sizeof(double) * B, 256);
        P->off_flag = o; o = align_up(o + sizeof(int32_t) * B, 256);
        if (P->lin) {
            P->off_P = o; o = align_up(o + (P->linbig ? sizeof(float) * ((size_t)nP + kLBSlot * kMaxThreads) : sizeof(double) * (size_t)nP), 256);
            P->off_cmax = o; o = align_up(o + sizeof(double) * (size_t)std::max<int64_t>(row0, 1), 256);
        }
        P->off_alpha = o; o = align_up(o + acc_bytes * (size_t)nab, 256);
        P->off_beta = o; o = align_up(o + acc_bytes * (size_t)nab, 256);
        if (pruned) {
            P->off_mA = o; o = align_up(o + sizeof(float) * (size_t)nab, 256);
            P->off_mB = o; o = align_up(o + sizeof(float) * (size_t)nab, 256);
            P->off_best = o; o = align_up(o + sizeof(float) * B, 256);
            P->off_thr = o; o = align_up(o + sizeof(float) * B, 256);
            if (P->pruned_fast) { P->off_bmask = o; o = align_up(o + 2 * sizeof(unsigned long long) * 64 * (size_t)std::max<int64_t>(row0, 1), 256); }
        }
        P->off_aoff = o; o = align_up(o + sizeof(double) * (size_t)(row0 + B), 256);
        P->off_boff = o; o = align_up(o + sizeof(double) * (size_t)(row0 + B), 256);
    } else if (mode == 2) {
        P->off_bp = o; o = align_up(o + sizeof(int32_t) * (size_t)nab, 256);
    }
    if (mode != 0 && P->big && P->one_fsa && B >= 2 && !hk().no_shared_big) {
        P->shared = true;
        const size_t groups = (size_t)(B + kNB - 1) / kNB;
        P->off_big = o; o = align_up(o + 2 * acc_bytes * groups * (size_t)P->smax * kNB, 256);
        P->off_Et = o; o = align_up(o + sizeof(float) * groups * (size_t)std::max(P->Tlong, 1) * (size_t)std::max(P->kmax, 1) * kNB, 256);
    } else if (mode != 0 && P->big) {
        P->off_big = o; o = align_up(o + 2 * acc_bytes * (size_t)P->nstates, 256);
    }
    P->total = o;
    return WF_OK;
}

// Test aid: every allocating call the library makes (pinned staging ring, streams, events) bumps this counter, so that a
// test can show that an entry point made none (wf_debug_alloc_count).  The library never allocates device memory.
std::atomic<int64_t> g_alloc_count{0};

// optional per-kernel timing (bench.py's roofline leg): HIP events recorded on the caller's stream
// around each kernel of the last wf_fb_logsum / wf_fb_backward call of this thread.
static __device__ unsigned long long g_tspan_dev[2];   // SweepArgs::tspan (timing runs only)
struct Timing {
    unsigned long long* tspan = nullptr;   // device address of g_tspan_dev on the device of the last timed call
    bool have_span = false;
    bool on = false;
    hipEvent_t ev[8] = {};   // t0 | gather | sweep | (post t0) | post | zero t0 | zero t1 | check + fallback
    bool made = false;
    bool have_fwd = false, have_post = false, have_zero = false, have_check = false;
    int check_from = 2;   // event that opens the "check + fallback" interval (2: after the sweep; 4: after the posterior kernel)
};
thread_local Timing g_tm;
int tm_record(int i, hipStream_t st) {
    if (!g_tm.on) return WF_OK;
    if (!g_tm.made) {
        g_alloc_count.fetch_add(1);
        for (auto& e : g_tm.ev) HIP_TRY(hipEventCreate(&e));
        g_tm.made = true;
    }
    HIP_TRY(hipEventRecord(g_tm.ev[i], st));
    return WF_OK;
}

// ---- scratch of the allocation-free entry point (wf_fb_logsum_static): one file-scope device array per GPU.
// In host code the array's name is only a stand-in; its device address is resolved once per device by wf_init()
// (hipGetSymbolAddress) and kept in g_scratch_ptr[device] -- host code and kernels only ever see that pointer.
constexpr size_t kScratchBytes = (size_t)WF_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 16;
void* g_scratch_ptr[kMaxDevices] = {};
std::once_flag g_scratch_once[kMaxDevices];   // host threads may set up their contexts on one device concurrently

// Per host thread and device: side streams for work that can overlap the latency-bound sweeps, their events, and a
// ring of pinned staging slots for the per-call tables (so that the host-to-device copy of a call is a true
// asynchronous copy from memory that outlives the call; a slot is reused only after the copy that read it finished).
constexpr int kRingSlots = 8;
constexpr size_t kRingSlotInit = 64 * 1024;
struct Side {
    hipStream_t stream = nullptr, stream2 = nullptr;   // zero-fill | backward direction of the big-graph sweep
    hipEvent_t fork = nullptr, join = nullptr, fork2 = nullptr, join2 = nullptr;
    bool fill_attr = false;
    unsigned char* ring = nullptr;   // pinned, kRingSlots x slot_bytes
    size_t slot_bytes = 0;
    hipEvent_t ring_ev[kRingSlots] = {};
    bool ring_used[kRingSlots] = {};
    int ring_next = 0;
    void* scratch_ptr = nullptr;
    std::vector<std::pair<const void*, int>> lds_attr;   // kernels whose dynamic-LDS limit was raised on this device
};
thread_local Side g_side[kMaxDevices];

int ring_alloc(Side& s, size_t slot_bytes) {
    g_alloc_count.fetch_add(1);
    if (s.ring) {
        for (int i = 0; i < kRingSlots; ++i)
            if (s.ring_used[i]) { HIP_TRY(hipEventSynchronize(s.ring_ev[i])); s.ring_used[i] = false; }
        HIP_TRY(hipHostFree(s.ring));
        s.ring = nullptr;
    }
    HIP_TRY(hipHostMalloc((void**)&s.ring, kRingSlots * slot_bytes, hipHostMallocDefault));
    s.slot_bytes = slot_bytes;
    return WF_OK;
}

// create = false: the allocation-free entry point; the context must exist already (wf_init)
int side_get(Side** out, bool create = true) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return wf_fail(WF_EUNSUPPORTED, "device index %d", dev);
    Side& s = g_side[dev];
    if (!s.stream) {
        if (!create) return wf_fail(WF_EINVAL, "wf_init() has not been called by this thread on device %d", dev);
        g_alloc_count.fetch_add(1);
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.fork2, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.join2, hipEventDisableTiming));
        for (auto& e : s.ring_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        int rc = ring_alloc(s, kRingSlotInit);
        if (rc) return rc;
        hipError_t sym_err = hipSuccess;
        std::call_once(g_scratch_once[dev], [&]() {
            void* ptr = nullptr;
            sym_err = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));   // this device's copy of the array
            g_scratch_ptr[dev] = sym_err == hipSuccess ? ptr : nullptr;
        });
        if (sym_err != hipSuccess || !g_scratch_ptr[dev]) return wf_fail(WF_EHIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(sym_err));
        s.scratch_ptr = g_scratch_ptr[dev];
    }
    *out = &s;
    return WF_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, thread) instead of on every launch
int ensure_lds(const void* fn, size_t bytes) {
    if (bytes <= 48 * 1024) return WF_OK;
    Side* s = nullptr;
    int rc = side_get(&s);
    if (rc) return rc;
    for (auto& e : s->lds_attr)
        if (e.first == fn) {
            if (e.second >= (int)bytes) return WF_OK;
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            e.second = (int)bytes;
            return WF_OK;
        }
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    s->lds_attr.emplace_back(fn, (int)bytes);
    return WF_OK;
}

// Makes the caller's stream wait for whatever was forked to a side stream, on EVERY exit path of an entry point
// (an early error return must not leave side-stream work running on memory the caller is about to release).
struct ForkGuard {
    hipStream_t st;
    hipStream_t side = nullptr;
    hipEvent_t join = nullptr;
    bool recorded = false, joined = false;
    explicit ForkGuard(hipStream_t s) : st(s) {}
    void fork(hipStream_t side_stream, hipEvent_t join_event) { side = side_stream; join = join_event; recorded = false; joined = false; }
    int record() { if (side && !recorded) { HIP_TRY(hipEventRecord(join, side)); recorded = true; } return WF_OK; }
    int wait() {
        if (!side || joined) return WF_OK;
        int rc = record();
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(st, join, 0));
        joined = true;
        return WF_OK;
    }
    ~ForkGuard() { (void)wait(); }
};

constexpr int kFillLds = 84 * 1024;   // see the zero-fill launch in fb_logsum_impl
int ensure_band_lds();

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

// the fallback re-run: no LDS staging of arcs or emissions (rare path, any size that the linear sweep takes)
SweepCfg sweep_config_minimal(const Plan& P, size_t acc_bytes) {
    SweepCfg c{};
    c.threads = std::min(kMaxThreads, std::max(64, (int)align_up(P.smax, 64)));
    c.smax = (int)align_up(P.smax, 4);
    c.hcap = std::max(1, std::min(kMaxHeavy, P.hmax));
    c.kpad = std::max(P.kmax, 1);
    c.estage = false; c.arcs_lds = false; c.chunk = 0; c.arcs_words = 0;
    c.lds = 2 * (size_t)c.smax * acc_bytes + (size_t)c.hcap * 16 * 2 * acc_bytes + (size_t)c.hcap * 16 * 4;
    return c;
}

size_t sweep_lds_bytes(const Plan& P, size_t acc_bytes) { return sweep_config(P, acc_bytes).lds; }

// big-graph sweep: init, one step launch per frame (and, with a.beta set, per direction: the backward
// launches go to the side stream and run beside the forward ones), final reduction
template <typename Acc, typename Sum, bool TROP, bool MASK = false>
int launch_big(const Plan& P, BigArgs a, int B, hipStream_t st) {
    a.nstates = P.nstates;
    a.nlb = (P.smax + 255) / 256;
    hipLaunchKernelGGL((big_init_kernel<Acc, TROP>), dim3(std::min(a.nlb, 1024), B), dim3(256), 0, st, a);
    Side* side = nullptr;
    ForkGuard guard(st);
    if (a.beta) {
        int rc = side_get(&side);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(side->fork2, st));
        HIP_TRY(hipStreamWaitEvent(side->stream2, side->fork2, 0));
        guard.fork(side->stream2, side->join2);
        const dim3 gridb(a.nlb + P.nheavy_out_max, B);
        BigArgs b = a;
        for (int t = P.Tlong - 1; t >= 0; --t) {
            b.t = t;
            hipLaunchKernelGGL((big_step_kernel<Acc, Sum, TROP, 1, MASK>), gridb, dim3(256), 0, side->stream2, b);
        }
        if ((rc = guard.record())) return rc;
    }
    const dim3 grid(a.nlb + P.nheavy_in_max, B);
    for (int t = 0; t < P.Tlong; ++t) {
        a.t = t;
        hipLaunchKernelGGL((big_step_kernel<Acc, Sum, TROP, 0, MASK>), grid, dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL((big_final_kernel<Acc, Sum, TROP>), dim3(B), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return guard.wait();
}

// big-graph forward sweep of a batch that shares one graph (Plan::shared)
template <typename Acc, typename Sum, bool TROP>
int launch_big_shared(const Plan& P, unsigned char* ws, const BigArgs& a, int B, hipStream_t st) {
    BigSharedArgs q{};
    q.blob = a.blob; q.utt = a.utt; q.lp = a.lp; q.strideB = a.strideB; q.strideT = a.strideT; q.E = a.E;
    q.Et = (float*)(ws + P.off_Et); q.buf = ws + P.off_big; q.bp = a.bp;
    q.B = B; q.Tlong = P.Tlong; q.S = P.smax; q.K = P.kmax; q.nlb = (P.smax + 255) / 256;
    const int groups = (B + kNB - 1) / kNB;
    if (P.Tlong > 0) hipLaunchKernelGGL(big_tr_kernel, dim3(P.Tlong, groups), dim3(256), 0, st, q);
    hipLaunchKernelGGL(big_init_shared_kernel<Acc>, dim3(std::min((P.smax * kNB + 255) / 256, 2048), groups), dim3(256), 0, st, q);
    const dim3 grid(q.nlb + P.nheavy_in_max, groups);
    for (int t = 0; t < P.Tlong; ++t) {
        q.t = t;
        hipLaunchKernelGGL((big_step_shared_kernel<Acc, Sum, TROP>), grid, dim3(256), 0, st, q);
    }
    BigArgs f = a;
    f.buf = q.buf; f.alpha = nullptr; f.nstates = 0;
    f.shared_S = P.smax; f.shared_par = P.Tlong & 1; f.shared_groups = groups;
    hipLaunchKernelGGL((big_final_kernel<Acc, Sum, TROP>), dim3(B), dim3(256), 0, st, f);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

template <typename Acc, typename Sum, int MODE>
int launch_sweep(const SweepCfg& c, const SweepArgs& a_, int B, hipStream_t st) {
    SweepArgs a = a_;
    a.ablate |= hk().sweep_ablate;
    if (hk().no_arc_cache) a.ablate |= 16;   // test hook: streamed arcs re-read in every frame
    const int ndir = (MODE == M_FB || MODE == M_FB_MAX) ? 2 : 1;
    dim3 grid(B * ndir), block(c.threads);
    if (c.lds > kLdsBudget)
        return wf_fail(WF_EUNSUPPORTED, "graph too large for the LDS-resident sweep: %d states need %zu B of LDS (> %zu)",
                       c.smax, c.lds, kLdsBudget);
#define WF_LAUNCH(AL, ES)                                                                            \
    do {                                                                                             \
        auto k = sweep_kernel<Acc, Sum, MODE, AL, ES>;                                                    \
        { int rc_ = ensure_lds((const void*)k, c.lds); if (rc_) return rc_; }                         \
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

// output_beam pruning, second pass: the log-semiring forward-backward over the kept lattice arcs only
int launch_sweep_masked(const SweepCfg& c, const SweepArgs& a, int B, hipStream_t st) {
    if (c.lds > kLdsBudget)
        return wf_fail(WF_EUNSUPPORTED, "graph too large for the LDS-resident sweep: %d states need %zu B of LDS (> %zu)",
                       c.smax, c.lds, kLdsBudget);
    auto k = sweep_kernel<double, float, M_FB, false, false, true>;
    { int rc_ = ensure_lds((const void*)k, c.lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(k, dim3(B * 2), dim3(c.threads), c.lds, st, a);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// ---- register-resident quad sweep: eligibility and launch

RegCfg reg_config(const Plan& P, bool both_dirs) {
    RegCfg c;
    const int no_reg = hk().no_reg;   // test hook
    if (no_reg) return c;
    if (!P.arcs_lds_ok || P.smax > kMaxThreads || !P.reg_ok_in || (both_dirs && !P.reg_ok_out)) return c;
    // four lanes per state; R states per quad once 4*S exceeds the workgroup
    c.R = P.smax * 4 <= kMaxThreads ? 1 : (P.smax * 2 <= kMaxThreads ? 2 : 4);
    const int force_r = hk().quad_r;
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

// linear-domain forward-backward sweep: lane = state, so S <= 1024; every state but (at most) one per
// direction has <= WF_REG_DEG arcs; the exceptional state's arcs fit HD <= 2 per lane
RegCfg lin_config(const Plan& P) {
    RegCfg c;
    const int no_lin = hk().no_lin;   // test hook
    if (no_lin) return c;
    if (P.smax > kMaxThreads || !P.reg_ok_in || !P.reg_ok_out) return c;
    c.threads = std::max(64, (int)align_up(P.smax, 64));
    const int od = std::max(P.over_deg_in, P.over_deg_out);
    c.HD = (od + c.threads - 1) / c.threads;
    if (c.HD > 2) return c;
    c.smax = (int)align_up(P.smax, 4);
    c.lds = 2 * (size_t)c.smax * 8 + 16 * 8 + 16 * 4;
    c.ok = true;
    return c;
}

// frames per rescale group = ring depth of the kernel variant launch_sweep_lin() picks
int lin_group(const RegCfg& c) { return c.threads <= 256 ? 8 : 4; }

// linear-domain sweep for graphs whose arcs do not fit the LDS (sweep_linbig_kernel): weights-elided graphs of up to
// kLBSlot * 1024 states with at most kLBHeavy heavy states, when neither the banded nor the lane-per-state sweep applies
size_t linbig_lds(const Plan& P) { return (size_t)kLBHeavy * 16 * 8 + 32 * 4 + 2 * ((size_t)align_up(P.smax, 4) + 4) * 4 + align_up(P.kmax, 4) * 4; }
bool linbig_config(const Plan& P) {
    if (hk().no_lin == 1 || hk().no_linbig) return false;   // test hooks (WF_NO_LIN_SWEEP=2: only the lane-per-state kernel is off)
    if (band_config(P).ok || lin_config(P).ok) return false;
    return P.band_we && P.smax <= kLBSlot * kMaxThreads && P.kmax <= kLBSlot * kMaxThreads && P.hmax <= kLBHeavy &&
           linbig_lds(P) <= kLdsBudget;
}

// hub form (sweep_hub_kernel): every graph of the call, the same (G, NC) instantiation, groups within kHubSlots per
// compute lane, a probability row within the loader lanes' reach and two of them within the LDS
size_t hub_kp(const Plan& P) { return align_up((size_t)P.kmax, 64) + 64; }   // a row buffer: whole 64-float DMA pieces
int hub_nb(const Plan& P) { return kHubLdsFixed + 3 * hub_kp(P) * 4 <= kLdsBudget && !hk().hub_force_split ? 3 : 2; }   // (test hook: rows as if too long for three)
// rows too long for three buffers: the split ring (three buffers of the front part, two of the back part), when every
// graph of the call has a split (Plan::hub_kpf / hub_kpb, see hub_split) and the parts fit
bool hub_is_split(const Plan& P) {
    return hub_nb(P) == 2 && P.hub_kpf > 0 && !hk().no_hub_split && kHubLdsFixed + (3 * (size_t)P.hub_kpf + 2 * (size_t)P.hub_kpb) * 4 <= kLdsBudget;
}
size_t hub_lds(const Plan& P) {
    if (hub_is_split(P)) return kHubLdsFixed + (3 * (size_t)P.hub_kpf + 2 * (size_t)P.hub_kpb) * 4;
    return kHubLdsFixed + (size_t)hub_nb(P) * hub_kp(P) * 4;
}
bool hub_config(const Plan& P) {
    if (hk().no_hub || hk().no_lin == 1) return false;
    if (P.hub_g < 2 || P.hub_g > WF_HUB_MAX_G || P.hub_nc < 2 || P.hub_nc > 3 || (P.hub_g == 3 && P.hub_nc != 3)) return false;
    return P.hub_ng <= kHubSlots * kHubCompute && hub_lds(P) <= kLdsBudget;
}

template <int G, int NC, int PAT>
int launch_sweep_hub_one(const Plan& P, SweepArgs a, int B, hipStream_t st) {
    const size_t lds = hub_lds(P);
    a.kpad = (int)hub_kp(P);
    a.chunk = hub_nb(P);
    a.hub_kpf = a.hub_kpb = 0;
    if (hub_is_split(P)) { a.hub_kpf = P.hub_kpf; a.hub_kpb = P.hub_kpb; }
    a.ablate = hk().sweep_ablate;
    auto k = sweep_hub_kernel<G, NC, PAT>;
    { int rc_ = ensure_lds((const void*)k, lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(k, dim3(B * 2), dim3(kMaxThreads), lds, st, a);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

int launch_sweep_hub(const Plan& P, const SweepArgs& a, int B, hipStream_t st) {
    const int pat = hk().no_hub_pattern ? -1 : P.hub_pat;
    if (P.hub_g == 2 && P.hub_nc == 2) {
        if (pat == 0) return launch_sweep_hub_one<2, 2, 0>(P, a, B, st);
        if (pat == 1) return launch_sweep_hub_one<2, 2, 1>(P, a, B, st);
        if (pat == 2) return launch_sweep_hub_one<2, 2, 2>(P, a, B, st);
        if (pat == 3) return launch_sweep_hub_one<2, 2, 3>(P, a, B, st);
        return launch_sweep_hub_one<2, 2, -1>(P, a, B, st);
    }
    if (P.hub_g == 2 && P.hub_nc == 3) {
        if (pat == 4) return launch_sweep_hub_one<2, 3, 4>(P, a, B, st);
        return launch_sweep_hub_one<2, 3, -1>(P, a, B, st);
    }
    if (P.hub_g == 3 && P.hub_nc == 3) return launch_sweep_hub_one<3, 3, -1>(P, a, B, st);
    return wf_fail(WF_EUNSUPPORTED, "internal: no hub sweep for G=%d NC=%d", P.hub_g, P.hub_nc);
}

// rescale bookkeeping of the linear-domain sweep that a call uses: frames per group and lag (lin_shift_index)
void lin_grp_lag(const Plan& P, int* grp, int* lag) {
    const BandSel bs = band_config(P);
    if (bs.ok) { *grp = bs.grp; *lag = 0; }
    else if (lin_config(P).ok) { *grp = lin_group(lin_config(P)); *lag = 1; }
    else { *grp = 1; *lag = 0; }   // sweep_linbig_kernel
}

int launch_sweep_linbig(const Plan& P, SweepArgs a, int B, hipStream_t st) {
    a.smax = (int)align_up(P.smax, 4);
    const size_t lds = linbig_lds(P);
    { int rc_ = ensure_lds((const void*)sweep_linbig_kernel, lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(sweep_linbig_kernel, dim3(B * 2), dim3(kMaxThreads), lds, st, a);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// banded wave sweep: every graph of the call is banded and fits one wavefront; picks the instantiation
BandSel band_config(const Plan& P) {
    BandSel b;
    const int no_band = hk().no_band;   // test hook
    if (no_band || P.band_sn < 1 || P.kmax >= kBandKP) return b;
    b.grp = kBandC;
    if (P.band_bw < 0) {
        // no banded order (a cycle besides self-loops: denominators): the general-graph mode of the wave kernel, for
        // graphs of at most 64 non-final states with <= WF_REG_DEG arcs per state but for one state of up to 64
        if (hk().no_gen || P.band_sn > 64 || !P.reg_ok_in || !P.reg_ok_out || P.over_deg_in > 64 || P.over_deg_out > 64)
            return b;
        b.gen = true;
        b.R = b.BW = b.M = 1;
        b.we = P.band_we;
        b.ok = true;
        return b;
    }
    b.BW = P.band_bw <= 2 ? 2 : 4;
    b.R = std::max(b.BW, (P.band_sn + 63) / 64);
    if (b.R > WF_BAND_MAX_R) return b;
    b.M = P.band_m <= 1 ? 1 : 2;
    b.we = b.M == 1 && P.band_we;
    b.ok = true;
    return b;
}

template <int R, int BW, int MM, bool WE, bool GEN = false, bool MASK = false>
int launch_band_one(SweepArgs a, int B, hipStream_t st, int nfill) {
    auto k = sweep_band_kernel<R, BW, MM, WE, GEN, MASK>;
    { int rc_ = ensure_lds((const void*)k, kBandLds); if (rc_) return rc_; }
    a.nsweep = B * 2;
    hipLaunchKernelGGL(k, dim3(B * 2 + nfill), dim3(128), kBandLds, st, a);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// nfill > 0: that many extra workgroups zero-fill a.zp / a.ztail (see SweepArgs)
// masked: output_beam pruning, a.bmask holds the keep-mask words (launch_trop_band ran before)
int launch_sweep_band(const BandSel& b, SweepArgs a, int B, hipStream_t st, int nfill = 0, bool masked = false) {
    const int prof = hk().sweep_profile;
    a.ablate = prof;
    if (masked) {
        if (b.gen) return b.we ? launch_band_one<1, 1, 1, true, true, true>(a, B, st, nfill) : launch_band_one<1, 1, 1, false, true, true>(a, B, st, nfill);
#define WF_BAND_CASE(RR, WW)                                                                     \
    if (b.BW == WW && b.R == RR)                                                                  \
        return b.we ? launch_band_one<RR, WW, 1, true, false, true>(a, B, st, nfill)                                   \
                    : (b.M == 1 ? launch_band_one<RR, WW, 1, false, false, true>(a, B, st, nfill) : launch_band_one<RR, WW, 2, false, false, true>(a, B, st, nfill))
        WF_BAND_CASE(2, 2);
        WF_BAND_CASE(3, 2);
        WF_BAND_CASE(4, 2);
        WF_BAND_CASE(4, 4);
#undef WF_BAND_CASE
        return wf_fail(WF_EUNSUPPORTED, "internal: no banded sweep for R=%d BW=%d", b.R, b.BW);
    }
    if (b.gen) return b.we ? launch_band_one<1, 1, 1, true, true>(a, B, st, nfill) : launch_band_one<1, 1, 1, false, true>(a, B, st, nfill);
#define WF_BAND_CASE(RR, WW)                                                                     \
    if (b.BW == WW && b.R == RR)                                                                  \
        return b.we ? launch_band_one<RR, WW, 1, true>(a, B, st, nfill)                                   \
                    : (b.M == 1 ? launch_band_one<RR, WW, 1, false>(a, B, st, nfill) : launch_band_one<RR, WW, 2, false>(a, B, st, nfill))
    WF_BAND_CASE(2, 2);
    WF_BAND_CASE(3, 2);
    WF_BAND_CASE(4, 2);
    WF_BAND_CASE(4, 4);
#undef WF_BAND_CASE
    return wf_fail(WF_EUNSUPPORTED, "internal: no banded sweep for R=%d BW=%d", b.R, b.BW);
}

// output_beam pruning on the banded path, first two launches: the tropical forward-backward rows (a.alpha / a.beta / a.tot
// = the fp32 arenas and best[]), then the keep-mask words
template <int R, int BW, int MM, bool WE, bool GEN = false>
int launch_trop_one(const SweepArgs& a, const BandMaskArgs& m, int B, int Tlong, hipStream_t st) {
    auto k = trop_band_kernel<R, BW, MM, WE, GEN>;
    { int rc_ = ensure_lds((const void*)k, kBandLds); if (rc_) return rc_; }
    hipLaunchKernelGGL(k, dim3(B * 2), dim3(128), kBandLds, st, a);
    hipLaunchKernelGGL((band_mask_kernel<R, BW, MM, GEN>), dim3((std::max(Tlong, 1) + kMaskFR - 1) / kMaskFR, B), dim3(64),
                       (size_t)kMaskFR * (2 * m.sp + m.kp) * sizeof(float), st, m);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

int launch_trop_band(const BandSel& b, const SweepArgs& a, const BandMaskArgs& m, int B, int Tlong, hipStream_t st) {
    if (b.gen) return b.we ? launch_trop_one<1, 1, 1, true, true>(a, m, B, Tlong, st) : launch_trop_one<1, 1, 1, false, true>(a, m, B, Tlong, st);
#define WF_TROP_CASE(RR, WW) \
    if (b.BW == WW && b.R == RR) return b.we ? launch_trop_one<RR, WW, 1, true>(a, m, B, Tlong, st) : b.M == 1 ? launch_trop_one<RR, WW, 1, false>(a, m, B, Tlong, st) : launch_trop_one<RR, WW, 2, false>(a, m, B, Tlong, st)
    WF_TROP_CASE(2, 2);
    WF_TROP_CASE(3, 2);
    WF_TROP_CASE(4, 2);
    WF_TROP_CASE(4, 4);
#undef WF_TROP_CASE
    return wf_fail(WF_EUNSUPPORTED, "internal: no banded tropical sweep for R=%d BW=%d", b.R, b.BW);
}

// wf_init: the dynamic-LDS attribute of every banded instantiation, so that a later launch finds it set
int ensure_band_lds() {
    int rc = 0;
#define WF_TROP_ATTR(...) if ((rc = ensure_lds((const void*)trop_band_kernel<__VA_ARGS__>, kBandLds))) return rc
    WF_TROP_ATTR(1, 1, 1, true, true); WF_TROP_ATTR(1, 1, 1, false, true);
    WF_TROP_ATTR(2, 2, 1, true); WF_TROP_ATTR(2, 2, 1, false); WF_TROP_ATTR(2, 2, 2, false);
    WF_TROP_ATTR(3, 2, 1, true); WF_TROP_ATTR(3, 2, 1, false); WF_TROP_ATTR(3, 2, 2, false);
    WF_TROP_ATTR(4, 2, 1, true); WF_TROP_ATTR(4, 2, 1, false); WF_TROP_ATTR(4, 2, 2, false);
    WF_TROP_ATTR(4, 4, 1, true); WF_TROP_ATTR(4, 4, 1, false); WF_TROP_ATTR(4, 4, 2, false);
#undef WF_TROP_ATTR
#define WF_BAND_ATTR(...) if ((rc = ensure_lds((const void*)sweep_band_kernel<__VA_ARGS__>, kBandLds))) return rc
    WF_BAND_ATTR(1, 1, 1, true, true); WF_BAND_ATTR(1, 1, 1, false, true);
    WF_BAND_ATTR(2, 2, 1, true); WF_BAND_ATTR(2, 2, 1, false); WF_BAND_ATTR(2, 2, 2, false);
    WF_BAND_ATTR(3, 2, 1, true); WF_BAND_ATTR(3, 2, 1, false); WF_BAND_ATTR(3, 2, 2, false);
    WF_BAND_ATTR(4, 2, 1, true); WF_BAND_ATTR(4, 2, 1, false); WF_BAND_ATTR(4, 2, 2, false);
    WF_BAND_ATTR(4, 4, 1, true); WF_BAND_ATTR(4, 4, 1, false); WF_BAND_ATTR(4, 4, 2, false);
    WF_BAND_ATTR(1, 1, 1, true, true, true); WF_BAND_ATTR(1, 1, 1, false, true, true);
    WF_BAND_ATTR(2, 2, 1, true, false, true); WF_BAND_ATTR(2, 2, 1, false, false, true); WF_BAND_ATTR(2, 2, 2, false, false, true);
    WF_BAND_ATTR(3, 2, 1, true, false, true); WF_BAND_ATTR(3, 2, 1, false, false, true); WF_BAND_ATTR(3, 2, 2, false, false, true);
    WF_BAND_ATTR(4, 2, 1, true, false, true); WF_BAND_ATTR(4, 2, 1, false, false, true); WF_BAND_ATTR(4, 2, 2, false, false, true);
    WF_BAND_ATTR(4, 4, 1, true, false, true); WF_BAND_ATTR(4, 4, 1, false, false, true); WF_BAND_ATTR(4, 4, 2, false, false, true);
#undef WF_BAND_ATTR
    return WF_OK;
}

int launch_sweep_lin(const RegCfg& c, SweepArgs a, int B, hipStream_t st) {
    dim3 grid(B * 2), block(c.threads);
    a.smax = c.smax;
    const int prof = hk().sweep_profile;
    a.ablate = prof;   // bit 8: print cycles per step; bits 1/2/4: ablations (row stores / ring refills / rescale) for timing only
    // two register budgets: workgroups of <= 256 lanes may use up to 512 VGPRs per lane
#define WF_LAUNCH_LIN(HH)                                                                             \
    do {                                                                                              \
        if (c.threads == 64 && !hk().no_wave) hipLaunchKernelGGL((sweep_lin_kernel<HH, 256, true>), grid, block, c.lds, st, a); \
        else if (c.threads <= 256) hipLaunchKernelGGL((sweep_lin_kernel<HH, 256>), grid, block, c.lds, st, a); \
        else hipLaunchKernelGGL((sweep_lin_kernel<HH, kMaxThreads>), grid, block, c.lds, st, a);        \
    } while (0)
    if (c.HD == 0) WF_LAUNCH_LIN(0);
    else if (c.HD == 1) WF_LAUNCH_LIN(1);
    else WF_LAUNCH_LIN(2);
#undef WF_LAUNCH_LIN
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

template <int MODE>
int launch_sweep_quad(const RegCfg& c, SweepArgs a, int B, hipStream_t st) {
    dim3 grid(B), block(c.threads);
    a.smax = c.smax;
    const int prof = hk().sweep_profile;
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

// Plan of a wf_fb_logsum call.  rows: the caller wants alpha / beta kept (gradient, WF_KEEP_AB) or pruning needs them.
// A forward-only call (torch.no_grad() denominators, conformer.py:413) on a graph that sweep_linbig_kernel handles
// is planned as a forward-backward call without the posterior kernel (*fwd_via_fb): both linear-domain sweeps + the
// row check (which needs beta) take 3.8 ms on cfg4's denominators, the log-semiring forward sweep alone 7.5 ms.
int plan_fb(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int B, int V, int num_seqs, bool rows,
            bool dbl, bool pruned, Plan* P, bool* fwd_via_fb) {
    *fwd_via_fb = false;
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, rows ? 0 : 1, P, dbl, pruned);
    if (rc || rows || dbl || P->smax < 4096 || P->big || hk().no_fwd_via_fb) return rc;
    Plan P0;
    if (make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, 0, &P0, false, false) == WF_OK && (P0.linbig || P0.hub)) {
        *P = P0;
        *fwd_via_fb = true;
    }
    return WF_OK;
}

extern "C" int wf_fb_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg,
                                     int32_t B, int32_t num_seqs, int32_t V, uint32_t flags, int32_t need_grad, size_t* bytes) {
    if (!bytes) return wf_fail(WF_EINVAL, "null bytes");
    Plan P;
    // the seq -> utterance map is sized by num_seqs, exactly as the compute calls plan it (a log_probs tensor may hold
    // many more sequences than the segments name)
    bool fwd_via_fb = false;
    int rc = plan_fb(graph_host, utt2fsa, seg, B, V, num_seqs, need_grad || (flags & (WF_KEEP_AB | WF_PRUNED)),
                     (flags & WF_DOUBLE_ACC) != 0, (flags & WF_PRUNED) != 0, &P, &fwd_via_fb);
    if (rc) return rc;
    // (asked with WF_ACCUMULATE: also says whether wf_fb_logsum will take the flag for these graphs)
    if ((flags & WF_ACCUMULATE) && (!P.lin || P.hub || !post_small_fits(P)))
        return wf_fail(WF_EUNSUPPORTED, "wf_fb_workspace_bytes: WF_ACCUMULATE is for graphs of the probability-domain path with at most 512 states / 1024 arcs (numerators)");
    *bytes = P.total + 256;
    if ((flags & WF_DOUBLE_ACC) && !(flags & (WF_TROPICAL | WF_PRUNED | WF_KEEP_AB)) && !need_grad) {
        // wf_tot_scores(WF_DOUBLE_ACC) runs the fp64 banded sweeps where they apply: room for their rows too
        Plan Q;
        if (plan_fb(graph_host, utt2fsa, seg, B, V, num_seqs, true, true, false, &Q, &fwd_via_fb) == WF_OK && Q.lin)
            *bytes = std::max(*bytes, Q.total + 256);
    }
    return WF_OK;
}

// Does wf_fb_logsum(flags | WF_PRUNED) keep these graphs on the fast (probability-domain, banded) kernels?  The Python
// drop-in's "auto" beam mode asks before it decides whether output_beam is honoured for a graph vector.
extern "C" int wf_fb_pruned_is_fast(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                                    int32_t num_seqs, int32_t V, uint32_t flags, int32_t* fast) {
    if (!fast) return wf_fail(WF_EINVAL, "wf_fb_pruned_is_fast: null pointer");
    Plan P;
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, 0, &P, (flags & WF_DOUBLE_ACC) != 0, true);
    if (rc) return rc;
    *fast = P.pruned_fast ? 1 : 0;
    return WF_OK;
}

// Test aid: how many utterances of the last wf_fb_logsum call on this workspace were redone by the log-semiring fallback
// (their probability-domain rows failed the row check, or WF_FORCE_FALLBACK).  Same (graph, seg, V, num_seqs, flags,
// need_grad) as that call.  Synchronises the stream.
extern "C" int wf_debug_fallback_count(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                                       int32_t num_seqs, int32_t V, uint32_t flags, int32_t need_grad,
                                       const void* workspace, size_t workspace_bytes, void* stream, int32_t* count) {
    if (!workspace || !count) return wf_fail(WF_EINVAL, "wf_debug_fallback_count: null pointer");
    Plan P;
    bool fwd_via_fb = false;
    int rc = plan_fb(graph_host, utt2fsa, seg, B, V, num_seqs, need_grad || (flags & (WF_KEEP_AB | WF_PRUNED)),
                     (flags & WF_DOUBLE_ACC) != 0, (flags & WF_PRUNED) != 0, &P, &fwd_via_fb);
    if (rc) return rc;
    const unsigned char* ws = (const unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (const unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small");
    *count = 0;
    if (!P.lin) return WF_OK;   // (no probability-domain sweep, nothing to fall back from)
    std::vector<int32_t> f(B);
    HIP_TRY(hipMemcpyAsync(f.data(), ws + P.off_flag, sizeof(int32_t) * B, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < B; ++i) *count += f[i] != 0;
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
// Copies the per-call tables (utterance table, then the seq -> utterance map: adjacent workspace regions, one copy)
// into the workspace: staged in a pinned ring slot owned by this thread's context, so the copy is asynchronous
// and its source outlives the call.  may_grow = false (allocation-free entry point): a table larger than a slot is an
// error instead of a reallocation.
int upload_tables(const Plan& P, unsigned char* ws, const int32_t* seq2utt, int num_seqs, hipStream_t st, bool may_grow = true) {
    const size_t utt_bytes = sizeof(int64_t) * P.utt.size();
    const size_t span = seq2utt ? (P.off_seq2utt - P.off_utt) + sizeof(int32_t) * (size_t)num_seqs : utt_bytes;
    Side* s = nullptr;
    int rc = side_get(&s, may_grow);
    if (rc) return rc;
    if (span > s->slot_bytes) {
        if (!may_grow) return wf_fail(WF_ENOSPC, "per-call tables need %zu B but a staging slot holds %zu B (B or num_seqs too large for the allocation-free entry point)", span, s->slot_bytes);
        if ((rc = ring_alloc(*s, align_up(span * 2, 4096)))) return rc;
    }
    const int slot = s->ring_next;
    s->ring_next = (slot + 1) % kRingSlots;
    if (s->ring_used[slot]) HIP_TRY(hipEventSynchronize(s->ring_ev[slot]));   // the copy issued kRingSlots calls ago
    unsigned char* stage = s->ring + (size_t)slot * s->slot_bytes;
    memcpy(stage, P.utt.data(), utt_bytes);
    if (seq2utt) {
        memset(stage + utt_bytes, 0, (P.off_seq2utt - P.off_utt) - utt_bytes);
        memcpy(stage + (P.off_seq2utt - P.off_utt), seq2utt, sizeof(int32_t) * (size_t)num_seqs);
    }
    HIP_TRY(hipMemcpyAsync(ws + P.off_utt, stage, span, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(s->ring_ev[slot], st));
    s->ring_used[slot] = true;
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

// The per-call tables can ride in the arguments of gather_lin_small_tab_kernel (see there) when that kernel opens the
// call, the batch is small, every table word fits 32 bits and every sequence of log_probs is named by a segment.
bool tab_in_kernarg(const Plan& P, int num_seqs, bool want_seq2utt) {
    if (hk().no_tab_kernarg || !P.lin || P.hub || P.linbig || P.kmax > 256 || hk().no_gather_small) return false;
    if (P.B > kTabMaxB || (want_seq2utt && num_seqs != P.B)) return false;
    for (int64_t v : P.utt) if (v < 0 || v > 0x7fffffff) return false;
    return true;
}

int launch_gather(const Plan& P, const void* graph_dev, unsigned char* ws, const float* lp,
                  int64_t strideB, int64_t strideT, hipStream_t st, bool tab = false, bool tab_seq2utt = false) {
    if (P.hub) return WF_OK;   // sweep_hub_kernel reads log_probs itself (and clears the fallback flags)
    if (P.lin) {
        GatherLinArgs g{(const int32_t*)graph_dev, (const int64_t*)(ws + P.off_utt), lp, strideB, strideT,
                        (float*)(ws + P.off_E), (double*)(ws + P.off_P), (double*)(ws + P.off_cmax),
                        (int32_t*)(ws + P.off_flag), P.linbig ? (float*)(ws + P.off_P) : nullptr};
        if (tab) {
            GatherLinTabArgs a;
            a.g = g; a.utt_out = (int64_t*)(ws + P.off_utt); a.seq2utt_out = tab_seq2utt ? (int32_t*)(ws + P.off_seq2utt) : nullptr;
            for (size_t i = 0; i < P.utt.size(); ++i) a.tab.w[i] = (int32_t)P.utt[i];
            const dim3 grid((std::max(P.Tlong, 1) + kGFR - 1) / kGFR, P.B);
            if (P.kmax <= 128) hipLaunchKernelGGL(gather_lin_small_tab_kernel, grid, dim3(64), 0, st, a);
            else hipLaunchKernelGGL(gather_lin_small4_tab_kernel, grid, dim3(64), 0, st, a);
        } else if (!P.linbig && P.kmax <= 128 && !hk().no_gather_small)
            hipLaunchKernelGGL(gather_lin_small_kernel, dim3((std::max(P.Tlong, 1) + kGFR - 1) / kGFR, P.B), dim3(64), 0, st, g);
        else if (!P.linbig && P.kmax <= 256 && !hk().no_gather_small)
            hipLaunchKernelGGL(gather_lin_small4_kernel, dim3((std::max(P.Tlong, 1) + kGFR - 1) / kGFR, P.B), dim3(64), 0, st, g);
        else
            hipLaunchKernelGGL(gather_lin_kernel, dim3(std::max(P.Tlong, 1), P.B), dim3(64), 0, st, g);
        HIP_TRY(hipGetLastError());
        return WF_OK;
    }
    if (!P.any_compact) return WF_OK;
    GatherArgs g{(const int32_t*)graph_dev, (const int64_t*)(ws + P.off_utt), lp, strideB, strideT,
                 (float*)(ws + P.off_E), P.B};
    int64_t per = (int64_t)P.Tlong * P.kmax_compact;
    int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (per + 255) / 256), 4096);
    hipLaunchKernelGGL(gather_kernel, dim3(gx, P.B), dim3(256), 0, st, g);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// post_small_kernel's size limits (one wavefront per 4 frames, <= 8 arc / 8 state / 8 column slots per lane)
bool post_small_fits(const Plan& P) {
    return !hk().no_post_small && P.amax >= 1 && P.amax <= 128 * 8 && P.smax <= 64 * 8 && P.kmax <= 64 * 8;
}

// zero-fill of the B x Tmax x V gradient box on `st`; blocks_hint: workgroups of the flat kernel (0: as many as it takes)
int launch_zero(float* grad, int64_t strideB, int64_t strideT, int num_seqs, int Tmax, int V, int blocks_hint, Side* side,
                hipStream_t st) {
    if (strideT == V && strideB == (int64_t)Tmax * V) {
        const size_t n = (size_t)num_seqs * Tmax * V;
        if (((uintptr_t)grad & 15) == 0 && !hk().zero_memset) {
            const size_t n4 = n / 4;
            int want = blocks_hint > 0 ? blocks_hint : 2048;
            if (hk().zero_blocks > 0) want = hk().zero_blocks;
            const int blocks = (int)std::min<size_t>((n4 + 255) / 256, (size_t)want);
            // The fill blocks ask for (and never touch) more than half of a CU's LDS: the latency-bound sweep
            // workgroups do the same, so no fill block lands on a sweep workgroup's CU, where its four waves
            // would compete with the chain wave for issue slots (measured: the co-resident sweep workgroups
            // ran 1.6-3.8x longer and set the kernel's duration).
            if (!side->fill_attr) {   // once per device (the attribute belongs to the device's copy of the kernel)
                HIP_TRY(hipFuncSetAttribute((const void*)zero_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFillLds));
                side->fill_attr = true;
            }
            hipLaunchKernelGGL(zero_flat_kernel, dim3(std::max(blocks, 1)), dim3(256), kFillLds, st,
                               (float4*)grad, n4, grad + n4 * 4, (int)(n - n4 * 4));
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(hipMemsetAsync(grad, 0, sizeof(float) * n, st));
        }
    } else {
        ZeroArgs z{grad, strideB, strideT, V};
        hipLaunchKernelGGL(zero_kernel, dim3(Tmax, num_seqs), dim3(256), 0, st, z);
        HIP_TRY(hipGetLastError());
    }
    return WF_OK;
}

// pass: PostArgs::pass (-1: check_kernel ran before; 0: post_small_kernel checks the rows itself)
int launch_post(const Plan& P, const void* graph_dev, unsigned char* ws, const float* lp,
                int64_t strideB, int64_t strideT, int num_seqs, int Tmax, int V, const float* gscale_dev,
                float* grad, bool skip_zero, hipStream_t st, int pass = -1, bool accum = false, int hseg_n = 0,
                int hseg_prev = 0, int hseg_den = 0) {
    PostArgs q{};
    q.hseg_n = hseg_n; q.hseg_prev = hseg_prev; q.hseg_den = hseg_den;
    q.accum = accum ? 1 : 0;
    q.pass = pass; q.flag_w = (int32_t*)(ws + P.off_flag); q.force = hk().force_fallback;
    {
        int grp = 0, lag = 0;
        lin_grp_lag(P, &grp, &lag);
        q.lin = P.lin ? grp : 0;
        q.lag = lag;
    }
    q.prof = hk().post_profile;
    q.ctot2 = (const double*)(ws + P.off_ctot2); q.cmax = (const double*)(ws + P.off_cmax);
    q.flag = (const int32_t*)(ws + P.off_flag);
    if (P.pruned) { q.mA = (const float*)(ws + P.off_mA); q.mB = (const float*)(ws + P.off_mB); q.thr = (const float*)(ws + P.off_thr); }
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
    // pre-zeroed rows (some graph compact: the side-stream fill covered the whole box), or only graphs whose columns are the
    // labels (the kernel zeroes the few columns it does not write)
    if (P.hub && pass != 1) {
        const dim3 grid(Tmax, num_seqs);
        // beside a cut sweep (hseg_den > 0, not the last launch): fewer posterior workgroups per CU, so that the sweeps'
        // LDS-DMA loads are not starved (an LDS request that is never used is the throttle)
        // (measured on cfg4's mmictc denominator, ms per step by the request: 0 -> 1.52, 32 KB -> 1.50, 64 KB (two workgroups
        // per CU) -> 1.445, 80 KB and more (one per CU: the posterior kernel falls behind) -> 1.58 = no overlap at all)
        const size_t plds = (hseg_den > 0 && hseg_n < hseg_den) ? (size_t)env_int("WF_HUB_POST_LDS", 65536) : 0;
#define WF_PHUB_CASE(GG, CC) if (P.hub_g == GG && P.hub_nc == CC) { \
            if (plds > 0) { int rc_ = ensure_lds((const void*)post_hub_kernel<GG, CC>, plds); if (rc_) return rc_; } \
            hipLaunchKernelGGL((post_hub_kernel<GG, CC>), grid, dim3(256), plds, st, q); }
        WF_PHUB_CASE(2, 2) else WF_PHUB_CASE(2, 3) else WF_PHUB_CASE(3, 3)
        else return wf_fail(WF_EUNSUPPORTED, "internal: no hub posterior kernel for G=%d NC=%d", P.hub_g, P.hub_nc);
#undef WF_PHUB_CASE
        HIP_TRY(hipGetLastError());
        if (pass == 0) return WF_OK;   // the caller runs the fallback sweep, then this function again with pass = 1
        q.pass = 1;                    // pass -1: what check_kernel flagged (log-domain rows) is post_big_kernel's
        hipLaunchKernelGGL(post_big_kernel, grid, dim3(256), 0, st, q);
        HIP_TRY(hipGetLastError());
        return WF_OK;
    }
    if (P.hub) {   // pass 1
        hipLaunchKernelGGL(post_big_kernel, dim3(Tmax, num_seqs), dim3(256), 0, st, q);
        HIP_TRY(hipGetLastError());
        return WF_OK;
    }
    if (P.lin && (skip_zero || !P.any_compact) && post_small_fits(P)) {
        q.ecap = (int)align_up(P.kmax, 4);
        q.acap = (int)align_up(P.amax, 4);
        size_t lds2 = (2 * (size_t)q.smax + q.ecap + q.acap) * 4;
        const int need = std::max(std::max((P.smax + 63) / 64, (P.kmax + 63) / 64), (P.amax + 127) / 128);
        const dim3 grid((Tmax + kPostFR - 1) / kPostFR, num_seqs);
        if (P.pruned) {   // masked posteriors: the tropical rows and raw emissions are staged as well
            lds2 += (2 * (size_t)q.smax + q.ecap) * 4;
            if (need <= 2) hipLaunchKernelGGL((post_small_kernel<2, true>), grid, dim3(64), lds2, st, q);
            else if (need <= 3) hipLaunchKernelGGL((post_small_kernel<3, true>), grid, dim3(64), lds2, st, q);
            else if (need <= 4) hipLaunchKernelGGL((post_small_kernel<4, true>), grid, dim3(64), lds2, st, q);
            else if (need <= 5) hipLaunchKernelGGL((post_small_kernel<5, true>), grid, dim3(64), lds2, st, q);
            else if (need <= 6) hipLaunchKernelGGL((post_small_kernel<6, true>), grid, dim3(64), lds2, st, q);
            else hipLaunchKernelGGL((post_small_kernel<8, true>), grid, dim3(64), lds2, st, q);
            HIP_TRY(hipGetLastError());
            return WF_OK;
        }
        if (need <= 2) hipLaunchKernelGGL(post_small_kernel<2>, grid, dim3(64), lds2, st, q);
        else if (need <= 3) hipLaunchKernelGGL(post_small_kernel<3>, grid, dim3(64), lds2, st, q);   // cfg3: S ~ 130, A ~ 323 (the kernel is instruction-bound)
        else if (need <= 4) hipLaunchKernelGGL(post_small_kernel<4>, grid, dim3(64), lds2, st, q);
        else if (need <= 5) hipLaunchKernelGGL(post_small_kernel<5>, grid, dim3(64), lds2, st, q);   // cfg2 numerators: A ~ 562
        else if (need <= 6) hipLaunchKernelGGL(post_small_kernel<6>, grid, dim3(64), lds2, st, q);
        else hipLaunchKernelGGL(post_small_kernel<8>, grid, dim3(64), lds2, st, q);
        HIP_TRY(hipGetLastError());
        return WF_OK;
    }
    // rows beyond the LDS, or long enough that staging them costs more than gathering per arc (measured on cfg4's
    // denominators, S = 10 002: 2.4 ms against 3.9 ms): gather from memory.  Log-domain, unpruned rows only.
    if (lds > kLdsBudget || ((!P.lin || P.linbig) && !P.pruned && P.smax >= hk().post_big_min_s)) {
        hipLaunchKernelGGL(post_big_kernel, dim3(Tmax, num_seqs), dim3(256), 0, st, q);
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
        { int rc_ = ensure_lds((const void*)post_kernel<false>, lds); if (rc_) return rc_; }
        hipLaunchKernelGGL(post_kernel<false>, grid, block, lds, st, q);
    }
    HIP_TRY(hipGetLastError());
    return WF_OK;
}
}  // namespace

namespace {
// use_scratch: the allocation-free entry point -- the workspace is this device's static scratch array, nothing may
// be allocated (the thread's context must exist: wf_init)
int fb_logsum_impl(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                   const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                   int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, float output_beam,
                   uint32_t flags, const float* grad_scale, float* tot, float* grad,
                   void* workspace, size_t workspace_bytes, void* stream, bool use_scratch) {
    if (!graph_dev || !log_probs || !tot || (!workspace && !use_scratch) || !seg) return wf_fail(WF_EINVAL, "wf_fb_logsum: null pointer");
    const bool pruned = output_beam > 0 && !std::isinf(output_beam);
    if (pruned != ((flags & WF_PRUNED) != 0))
        return wf_fail(WF_EINVAL, "wf_fb_logsum: pass WF_PRUNED (also to wf_fb_workspace_bytes / wf_fb_backward) exactly when 0 < output_beam < inf");
    hipStream_t st = (hipStream_t)stream;
    const bool dbl = flags & WF_DOUBLE_ACC;
    if ((flags & WF_TOT_F64) && !dbl) return wf_fail(WF_EINVAL, "wf_fb_logsum: WF_TOT_F64 goes with WF_DOUBLE_ACC");
    const bool need_ab = grad || (flags & WF_KEEP_AB);
    const size_t ab = 8;
    Plan P;
    bool fwd_via_fb = false;
    int rc = plan_fb(graph_host, utt2fsa, seg, B, V, num_seqs, need_ab || pruned, dbl, pruned, &P, &fwd_via_fb);
    if (rc) return rc;
    if ((rc = check_segments("wf_fb_logsum", seg, B, num_seqs, Tmax))) return rc;
    Side* side = nullptr;
    if (use_scratch) {
        if ((rc = side_get(&side, false))) return rc;
        workspace = side->scratch_ptr;
        workspace_bytes = kScratchBytes;
        if (P.total + 256 > kScratchBytes)
            return wf_fail(WF_ENOSPC, "wf_fb_logsum_static: this call needs %zu B of scratch, the static array holds %zu B (use wf_fb_logsum with a caller workspace)", P.total + 256, kScratchBytes);
        if (P.big) return wf_fail(WF_EUNSUPPORTED, "wf_fb_logsum_static: graphs beyond the LDS-resident sweeps are not supported");
    }
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    std::vector<int32_t> seq2utt;
    if (need_ab || pruned) { rc = build_seq2utt(seg, B, num_seqs, &seq2utt); if (rc) return rc; }
    const bool tab = tab_in_kernarg(P, num_seqs, need_ab || pruned) && (!pruned || P.pruned_fast);
    if (!tab) {
        rc = upload_tables(P, ws, (need_ab || pruned) ? seq2utt.data() : nullptr, num_seqs, st, !use_scratch);
        if (rc) return rc;
    }
    g_tm.have_fwd = g_tm.have_post = g_tm.have_zero = g_tm.have_check = false;
    // Fork: zero-fill the gradient box on the side stream while the sweeps (which occupy only 2B
    // workgroups) run; post_kernel then only scatters the K posteriors per row.
    // WF_ACCUMULATE: the box holds another call's gradient -- no fill, and the posteriors are added once the rows are
    // final (check_kernel and the fallback first: a row check folded into the scatter could not take its additions back)
    const bool accum = flags & WF_ACCUMULATE;
    if (accum) {
        if (!grad || pruned || (flags & (WF_PREZERO | WF_KEEP_AB)))
            return wf_fail(WF_EINVAL, "wf_fb_logsum: WF_ACCUMULATE goes with grad != NULL and none of WF_PRUNED / WF_PREZERO / WF_KEEP_AB");
        if (!P.lin || P.hub || !post_small_fits(P))
            return wf_fail(WF_EUNSUPPORTED, "wf_fb_logsum: WF_ACCUMULATE is for graphs of the probability-domain path with at most 512 states / 1024 arcs (numerators)");
    }
    const bool overlap = grad && P.any_compact && !hk().no_overlap && !accum;
    // the row check folded into post_small_kernel (no check_kernel launch): when the gradient is computed by this call
    const bool postcheck = P.lin && grad && !(flags & WF_PREZERO) && !hk().no_post_check && !accum &&
                           (P.hub || ((overlap || !P.any_compact) && post_small_fits(P)));
    ForkGuard guard(st);   // every return below first makes `st` wait for the side stream
    // the last step of a successful call.  WF_TOT_F64: every kernel left the total in the workspace in fp64 as well (the
    // fp32 values they wrote into the front of `tot` are overwritten here)
    auto finish = [&]() -> int {
        int frc = guard.wait();
        if (frc) return frc;
        if (flags & WF_TOT_F64) HIP_TRY(hipMemcpyAsync(tot, ws + P.off_totd, sizeof(double) * B, hipMemcpyDeviceToDevice, st));
        return WF_OK;
    };
    g_tm.have_span = false;
    if (g_tm.on && P.lin && !P.hub && band_config(P).ok) {   // (reset in front of the timed intervals)
        void* sp = nullptr;
        HIP_TRY(hipGetSymbolAddress(&sp, HIP_SYMBOL(g_tspan_dev)));
        g_tm.tspan = (unsigned long long*)sp;
        HIP_TRY(hipMemsetAsync(sp, 0xff, 8, st));
        HIP_TRY(hipMemsetAsync((char*)sp + 8, 0, 8, st));
        g_tm.have_span = true;
    }
    if ((rc = tm_record(0, st))) return rc;
    rc = launch_gather(P, graph_dev, ws, log_probs, strideB, strideT, st, tab, need_ab || pruned);
    if (rc) return rc;
    if ((rc = tm_record(1, st))) return rc;
    // The fill rides in the banded sweep's own launch (extra workgroups) when the box is one flat, 16-byte aligned run and
    // sweep + fill workgroups all fit the chip at one per CU; otherwise it goes to the side stream.
    int fill_want = (int)((double)num_seqs * Tmax * V * 4.0 / (std::max(P.Tlong, 1) * 0.15e-6) / 70e9) + 1;
    fill_want = std::max(32, std::min(fill_want, 2048));
    if (hk().zero_blocks > 0) fill_want = hk().zero_blocks;
    const bool flat = strideT == V && strideB == (int64_t)Tmax * V && ((uintptr_t)grad & 15) == 0;
    // 128-lane fill workgroups in the sweep's launch: ~80 GB/s each, sized to end with the sweep (~0.095 us per frame step).
    // More of them slow the chain waves down (measured on cfg3, ms per step: 32 -> 0.177, 44 -> 0.144, 68 -> 0.112,
    // 80 -> 0.1037, 100 -> 0.1044, 124 -> 0.110, 180 -> 0.119)
    // (the chain's time per frame step beside the fill, by instantiation: <3,2> cfg3 0.095 us; four rows per lane <4,2>
    // 0.16 us, distance-4 bands <4,4> 0.19 us, the general-graph mode 0.17 us -- a fill sized for a faster chain than
    // the launch has crowds the memory pipeline and slows the chain further: cfg4's mmictc numerators 555 cycles per
    // step with 169 fill workgroups against 378 alone)
    // Measured on cfg4's numerators (V = 10 001 / 15 001, ms per step by fill workgroups): mmictc 84 -> 0.175, 96 -> 0.164,
    // 112 -> 0.155, 128 -> 0.168, 169 -> 0.191; 3state-skip 112 -> 0.235, 128 -> 0.227, 160 -> 0.231, side stream 0.258.
    // Beside these slower chains a fill workgroup reaches ~62 GB/s, ~120 of them saturate the HBM, and the launch does
    // best with a whole number of workgroups per XCD.
    const BandSel bsf = band_config(P);
    const bool slow_chain = bsf.ok && (bsf.gen || bsf.BW == 4 || bsf.R == 4);
    const double step_s = !slow_chain ? 0.095e-6 : bsf.gen ? 0.17e-6 : bsf.BW == 4 ? 0.19e-6 : 0.16e-6;
    int nfill_sweep = (int)((double)num_seqs * Tmax * V * 4.0 / (std::max(P.Tlong, 1) * step_s) / (slow_chain ? 62e9 : 80e9)) + 1;
    if (slow_chain) {
        nfill_sweep = std::min(nfill_sweep, 128);
        nfill_sweep += (8 - (2 * B + nfill_sweep) % 8) % 8;
    }
    nfill_sweep = std::max(16, nfill_sweep);
    if (hk().zero_blocks > 0) nfill_sweep = 2 * hk().zero_blocks;
    const bool fill_in_sweep = overlap && P.lin && band_config(P).ok && flat && !hk().fill_side_stream && !hk().zero_memset &&
                               2 * B + nfill_sweep <= 256;
    if (fill_in_sweep) g_tm.have_zero = false;
    if (overlap && !fill_in_sweep) {
        // forked after the gather: both are HBM-bound and would only slow each other down
        if ((rc = side_get(&side, !use_scratch))) return rc;
        HIP_TRY(hipEventRecord(side->fork, st));
        HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
        guard.fork(side->stream, side->join);
        if ((rc = tm_record(5, side->stream))) return rc;
        // Size the grid so that the fill ends about when the sweeps do (~0.15 us per frame step,
        // ~70 GB/s of streaming stores per workgroup): more workgroups than that only steal
        // issue slots and memory-queue space from the latency-bound sweep workgroups
        // (measured on cfg3: 64 blocks -> fill 73 us, sweep 80 us; 256 blocks -> 62 / 98; 1024 -> 74 / 114).
        if ((rc = launch_zero(grad, strideB, strideT, num_seqs, Tmax, V, fill_want, side, side->stream))) return rc;
        if ((rc = tm_record(6, side->stream))) return rc;
        g_tm.have_zero = g_tm.on;
        if ((rc = guard.record())) return rc;
    }
    SweepCfg c = sweep_config(P, ab);
    SweepArgs a{};
    a.blob = (const int32_t*)graph_dev; a.utt = (const int64_t*)(ws + P.off_utt);
    a.lp = log_probs; a.strideB = strideB; a.strideT = strideT;
    a.E = (const float*)(ws + P.off_E);
    a.flag_clear = (int32_t*)(ws + P.off_flag); a.cmax_w = (double*)(ws + P.off_cmax);
    a.alpha = ws + P.off_alpha; a.beta = ws + P.off_beta; a.tot = tot; a.bp = nullptr; a.fin_arc = nullptr;
    a.totd = (double*)(ws + P.off_totd);
    a.chunk = c.chunk; a.kpad = c.kpad; a.smax = c.smax; a.arcs_lds_words = c.arcs_words; a.hcap = c.hcap; a.tot_double = 0;
    a.aoff = (double*)(ws + P.off_aoff); a.boff = (double*)(ws + P.off_boff);
    a.tot2 = (double*)(ws + P.off_tot2);
    a.P = (const double*)(ws + P.off_P); a.cmax = (const double*)(ws + P.off_cmax); a.ctot2 = (double*)(ws + P.off_ctot2);
    if (P.lin) {
        // linear-domain sweep, then the row identity check and the log-semiring re-run of whatever failed it
        const BandSel bs = band_config(P);
        int lgrp = 0, llag = 0;
        lin_grp_lag(P, &lgrp, &llag);
        int nfill = 0;
        if (fill_in_sweep) {
            const size_t n = (size_t)num_seqs * Tmax * V;
            a.zp = (float4*)grad; a.zn4 = n / 4; a.ztail = grad + (n / 4) * 4; a.zntail = (int)(n - (n / 4) * 4);
            nfill = nfill_sweep;
        }
        if (P.pruned_fast) {
            // output_beam: tropical rows + best, then the keep-mask words the MASK sweep stages (the banded form only)
            SweepArgs ta = a;
            ta.alpha = ws + P.off_mA; ta.beta = ws + P.off_mB; ta.tot = ws + P.off_best; ta.totd = nullptr;
            BandMaskArgs bm{a.blob, a.utt, log_probs, strideB, strideT, a.E, (const float*)(ws + P.off_mA), (const float*)(ws + P.off_mB),
                            (const float*)(ws + P.off_best), output_beam, (float*)(ws + P.off_thr),
                            (unsigned long long*)(ws + P.off_bmask), P.n_rows, (int)align_up(P.smax, 4), (int)align_up(std::max(P.kmax, 1), 4)};
            if ((rc = launch_trop_band(bs, ta, bm, B, P.Tlong, st))) return rc;
            a.bmask = (unsigned long long*)(ws + P.off_bmask); a.bmask_rows = P.n_rows;
        }
        if (g_tm.have_span) a.tspan = g_tm.tspan;
        if (P.hub && postcheck && !hk().no_hub_overlap && P.Tlong >= kHubSegMinT) {
            // Hub-form denominators: 2B sweep workgroups leave three quarters of the chip idle, and the posterior kernel
            // (bandwidth bound, as long as half a sweep) used to wait for both sweeps to finish.  A frame's posteriors need
            // alpha_t and beta_{t+1} only: once both sweeps are s steps in, frames T - s - 1 .. s are ready.  The sweeps
            // run in four launches (5/8 of the steps, then three times 1/8: the state that is not in the rows travels
            // through hubsave); after each, the side stream computes the posterior rows that have become ready -- a
            // quarter of them each time, beside the sweeps' next eighth.  The total they normalise with comes from the
            // middle frame as soon as the sweeps have met (hub_tot_kernel).
            int kSegN[17] = {0, 5, 6, 7, 8}, nseg = 4, den = 8;
            if (const char* e = getenv("WF_HUB_SEGS")) {   // (tuning aid: "den:n1,n2,...,den")
                den = atoi(e);
                nseg = 0;
                for (const char* q = strchr(e, ':'); q && nseg < 16; q = strchr(q + 1, ',')) kSegN[++nseg] = atoi(q + 1);
            }
            if ((rc = side_get(&side, !use_scratch))) return rc;
            a.hubsave = (double*)(ws + P.off_hubsave);
            a.seg_den = den;
            for (int k = 1; k <= nseg; ++k) {
                a.seg_n0 = kSegN[k - 1]; a.seg_n1 = kSegN[k];
                if ((rc = launch_sweep_hub(P, a, B, st))) return rc;
                if (k == 1) {
                    HubTotArgs ht{a.blob, a.utt, a.alpha, a.beta, a.aoff, a.boff, a.tot2, a.ctot2, kSegN[1], den};
                    hipLaunchKernelGGL(hub_tot_kernel, dim3(B), dim3(256), 0, st, ht);
                    HIP_TRY(hipGetLastError());
                }
                if (k == nseg && (rc = tm_record(2, st))) return rc;
                HIP_TRY(hipEventRecord(side->fork, st));
                HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
                if (k == 1) guard.fork(side->stream, side->join);
                rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, overlap,
                                 side->stream, 0, false, kSegN[k], k > 1 ? kSegN[k - 1] : 0, den);
                if (rc) return rc;
            }
            if ((rc = tm_record(3, st))) return rc;
            if ((rc = guard.wait())) return rc;
            if ((rc = tm_record(4, st))) return rc;
            g_tm.have_post = g_tm.on;
            // what the row check flagged: the workgroup-wide fallback sweep and post_big_kernel (both exit at once otherwise)
            SweepCfg hc = sweep_config_minimal(P, ab);
            SweepArgs ha = a;
            ha.seg_den = 0;
            ha.flag = (const int32_t*)(ws + P.off_flag);
            ha.chunk = hc.chunk; ha.kpad = hc.kpad; ha.smax = hc.smax; ha.arcs_lds_words = 0; ha.hcap = hc.hcap;
            if ((rc = launch_sweep<double, float, M_FB>(hc, ha, B, st))) return rc;
            rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, overlap, st, 1);
            if (rc) return rc;
            if ((rc = tm_record(7, st))) return rc;
            g_tm.have_fwd = g_tm.on;
            g_tm.have_check = g_tm.on;
            g_tm.check_from = 4;
            return finish();
        }
        rc = P.hub ? launch_sweep_hub(P, a, B, st) : bs.ok ? launch_sweep_band(bs, a, B, st, nfill, P.pruned_fast) : (lin_config(P).ok ? launch_sweep_lin(lin_config(P), a, B, st) : launch_sweep_linbig(P, a, B, st));
        if (rc) return rc;
        if ((rc = tm_record(2, st))) return rc;
        SweepCfg fc = sweep_config_minimal(P, ab);
        SweepArgs fa = a;
        fa.flag = (const int32_t*)(ws + P.off_flag);
        fa.chunk = fc.chunk; fa.kpad = fc.kpad; fa.smax = fc.smax; fa.arcs_lds_words = 0; fa.hcap = fc.hcap;
        if (P.pruned_fast) {   // the fallback of a pruned call is the masked log-semiring sweep
            fa.mA = (const float*)(ws + P.off_mA); fa.mB = (const float*)(ws + P.off_mB);
            fa.best = (const float*)(ws + P.off_best); fa.thr = (float*)(ws + P.off_thr); fa.beam = output_beam;
        }
        if (postcheck) {
            // posterior rows first (pass 0 checks every frame and flags what fails), then ONE launch that redoes the
            // flagged utterances in the log semiring (exits at once when nothing is flagged)
            if ((rc = guard.wait())) return rc;
            if ((rc = tm_record(3, st))) return rc;
            rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, overlap, st, 0);
            if (rc) return rc;
            if ((rc = tm_record(4, st))) return rc;
            g_tm.have_post = g_tm.on;
            if (P.hub) {
                // a hub state's thousands of arcs are no job for rescue_kernel's one lane per state: the workgroup-wide
                // fallback sweep and post_big_kernel redo the flagged utterances (both exit at once otherwise)
                rc = launch_sweep<double, float, M_FB>(fc, fa, B, st);
                if (rc) return rc;
                rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, overlap, st, 1);
                if (rc) return rc;
            } else {
                RescueArgs r{a.blob, a.utt, (const int32_t*)(ws + P.off_flag), grad_scale, log_probs, strideB, strideT, a.E,
                             (double*)a.alpha, (double*)a.beta, tot, a.totd, grad, nullptr, nullptr, nullptr};
                if (P.pruned_fast) { r.mA = fa.mA; r.mB = fa.mB; r.thr = fa.thr; }
                hipLaunchKernelGGL(rescue_kernel, dim3(B), dim3(256), 0, st, r);
                HIP_TRY(hipGetLastError());
            }
            if ((rc = tm_record(7, st))) return rc;
            g_tm.have_fwd = g_tm.on;
            g_tm.have_check = g_tm.on;
            g_tm.check_from = 4;
            return finish();
        }
        CheckArgs k{a.blob, a.utt, (const double*)a.alpha, (const double*)a.beta, a.aoff, a.boff, a.tot2, a.ctot2,
                    (int32_t*)(ws + P.off_flag), lgrp, llag,
                    hk().force_fallback, P.hub ? 1 : 0};
        hipLaunchKernelGGL(check_kernel, dim3((P.Tlong + 1 + kCheckFR - 1) / kCheckFR, B), dim3(64), 0, st, k);
        HIP_TRY(hipGetLastError());
        rc = P.pruned_fast ? launch_sweep_masked(fc, fa, B, st) : launch_sweep<double, float, M_FB>(fc, fa, B, st);
        if (rc) return rc;
        if ((rc = tm_record(7, st))) return rc;
        g_tm.check_from = 2;
    } else if (pruned && P.big) {
        // the same two passes with the per-frame launches of graphs beyond the LDS
        BigArgs g{};
        g.blob = a.blob; g.utt = a.utt; g.lp = log_probs; g.strideB = strideB; g.strideT = strideT; g.E = a.E;
        g.alpha = ws + P.off_mA; g.beta = ws + P.off_mB; g.tot = ws + P.off_best; g.tot_double = 0;
        rc = launch_big<float, float, true>(P, g, B, st);
        if (rc) return rc;
        g.alpha = a.alpha; g.beta = a.beta; g.tot = tot; g.totd = a.totd;
        g.mA = (const float*)(ws + P.off_mA); g.mB = (const float*)(ws + P.off_mB);
        g.best = (const float*)(ws + P.off_best); g.thr = (float*)(ws + P.off_thr); g.beam = output_beam;
        rc = dbl ? launch_big<double, double, false, true>(P, g, B, st) : launch_big<double, float, false, true>(P, g, B, st);
        if (rc) return rc;
        if ((rc = tm_record(2, st))) return rc;
    } else if (pruned) {
        // pass 1: tropical forward-backward (fp32, every row kept); pass 2: log semiring over the kept arcs
        SweepCfg tc = sweep_config(P, 4);
        SweepArgs ta = a;
        ta.alpha = ws + P.off_mA; ta.beta = ws + P.off_mB; ta.tot = ws + P.off_best; ta.totd = nullptr;
        ta.chunk = tc.chunk; ta.kpad = tc.kpad; ta.smax = tc.smax; ta.arcs_lds_words = tc.arcs_words; ta.hcap = tc.hcap;
        rc = launch_sweep<float, float, M_FB_MAX>(tc, ta, B, st);
        if (rc) return rc;
        SweepCfg mc = sweep_config_minimal(P, ab);
        SweepArgs ma = a;
        ma.mA = (const float*)(ws + P.off_mA); ma.mB = (const float*)(ws + P.off_mB);
        ma.best = (const float*)(ws + P.off_best); ma.thr = (float*)(ws + P.off_thr); ma.beam = output_beam;
        ma.chunk = mc.chunk; ma.kpad = mc.kpad; ma.smax = mc.smax; ma.arcs_lds_words = 0; ma.hcap = mc.hcap;
        rc = launch_sweep_masked(mc, ma, B, st);
        if (rc) return rc;
        if ((rc = tm_record(2, st))) return rc;
    } else {
        RegCfg rc_ = (dbl || need_ab || P.big) ? RegCfg() : reg_config(P, false);
        if (P.big) {
            // graph beyond the LDS sweeps: per-frame launches (both directions when alpha/beta rows are wanted)
            BigArgs g{};
            g.blob = a.blob; g.utt = a.utt; g.lp = log_probs; g.strideB = strideB; g.strideT = strideT; g.E = a.E;
            g.tot = tot; g.tot_double = 0;
            g.totd = a.totd;
            if (need_ab) { g.alpha = a.alpha; g.beta = a.beta; }
            else g.buf = ws + P.off_big;
            if (!need_ab && P.shared) rc = dbl ? launch_big_shared<double, double, false>(P, ws, g, B, st) : launch_big_shared<double, float, false>(P, ws, g, B, st);
            else rc = dbl ? launch_big<double, double, false>(P, g, B, st) : launch_big<double, float, false>(P, g, B, st);
        } else if (rc_.ok) rc = launch_sweep_quad<M_TOT_LOG>(rc_, a, B, st);
        else if (need_ab) rc = dbl ? launch_sweep<double, double, M_FB>(c, a, B, st) : launch_sweep<double, float, M_FB>(c, a, B, st);
        else rc = dbl ? launch_sweep<double, double, M_TOT_LOG>(c, a, B, st) : launch_sweep<double, float, M_TOT_LOG>(c, a, B, st);
        if (rc) return rc;
        if ((rc = tm_record(2, st))) return rc;
    }
    g_tm.have_fwd = g_tm.on;
    g_tm.have_check = g_tm.on && P.lin;
    if (grad && (flags & WF_PREZERO)) {
        // loss now, gradient later (wf_fb_backward with WF_PREZERO): the fill is done, ordered before whatever follows
        if ((rc = guard.wait())) return rc;
    } else if (grad) {
        if ((rc = guard.wait())) return rc;
        if ((rc = tm_record(3, st))) return rc;
        rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, overlap || accum, st, -1, accum);
        if (rc) return rc;
        if ((rc = tm_record(4, st))) return rc;
        g_tm.have_post = g_tm.on;
    }
    return finish();
}
}  // namespace

extern "C" int wf_fb_logsum(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                            const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                            int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, float output_beam,
                            uint32_t flags, const float* grad_scale, float* tot, float* grad,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return fb_logsum_impl(graph_host, graph_dev, utt2fsa, log_probs, strideB, strideT, num_seqs, Tmax, V, seg, B, output_beam,
                          flags, grad_scale, tot, grad, workspace, workspace_bytes, stream, false);
}

// The allocation-free variant of wf_fb_logsum: same kernels, same results, the workspace is the library's static
// per-device scratch array (WF_STATIC_SCRATCH_BYTES).  Calls nothing that allocates: the thread's context (side
// streams, pinned staging ring, scratch address) must have been set up by wf_init() on the current device.
extern "C" int wf_fb_logsum_static(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                                   const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                                   int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, float output_beam,
                                   uint32_t flags, const float* grad_scale, float* tot, float* grad, void* stream) {
    if (flags & (WF_KEEP_AB | WF_PREZERO))
        return wf_fail(WF_EUNSUPPORTED, "wf_fb_logsum_static: WF_KEEP_AB / WF_PREZERO (a later wf_fb_backward) need a caller-owned workspace");
    return fb_logsum_impl(graph_host, graph_dev, utt2fsa, log_probs, strideB, strideT, num_seqs, Tmax, V, seg, B, output_beam,
                          flags, grad_scale, tot, grad, nullptr, 0, stream, true);
}

extern "C" int wf_debug_alloc_count(int64_t* count) {
    if (!count) return wf_fail(WF_EINVAL, "null count");
    *count = g_alloc_count.load();
    return WF_OK;
}

extern "C" int wf_static_scratch_bytes(size_t* bytes) {
    if (!bytes) return wf_fail(WF_EINVAL, "null bytes");
    *bytes = kScratchBytes;
    return WF_OK;
}

// Set-up for the calling host thread on the current device (idempotent): side streams and their events, the pinned
// staging ring, the device address of the static scratch array.  Other entry points do this lazily on first use;
// wf_fb_logsum_static requires it to have happened.
extern "C" int wf_init(void) {
    Side* s = nullptr;
    int rc = side_get(&s, true);
    if (rc) return rc;
    (void)hk();
    if (!s->fill_attr) {
        HIP_TRY(hipFuncSetAttribute((const void*)zero_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFillLds));
        s->fill_attr = true;
    }
    // raise the dynamic-LDS limit of the kernels the allocation-free path may launch, so that it never has to
    return ensure_band_lds();
}

// Releases what wf_init / the lazy set-up created for the calling thread (all devices).  Waits for the side streams.
extern "C" int wf_shutdown(void) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    for (int d = 0; d < kMaxDevices; ++d) {
        Side& s = g_side[d];
        if (!s.stream) continue;
        HIP_TRY(hipSetDevice(d));
        HIP_TRY(hipStreamSynchronize(s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream2));
        for (int i = 0; i < kRingSlots; ++i) {
            if (s.ring_used[i]) HIP_TRY(hipEventSynchronize(s.ring_ev[i]));
            HIP_TRY(hipEventDestroy(s.ring_ev[i]));
        }
        HIP_TRY(hipHostFree(s.ring));
        HIP_TRY(hipStreamDestroy(s.stream));
        HIP_TRY(hipStreamDestroy(s.stream2));
        HIP_TRY(hipEventDestroy(s.fork)); HIP_TRY(hipEventDestroy(s.join));
        HIP_TRY(hipEventDestroy(s.fork2)); HIP_TRY(hipEventDestroy(s.join2));
        s = Side();
    }
    if (g_tm.made) {
        for (auto& e : g_tm.ev) HIP_TRY(hipEventDestroy(e));
        g_tm = Timing();
    }
    HIP_TRY(hipSetDevice(cur));
    return WF_OK;
}

extern "C" int wf_debug_reload_env(void) {
    load_hooks();
    return WF_OK;
}

// [-sum of the finite tot, n_frames, number of infeasible utterances] in one launch (the buffer of the one fused
// all-reduce per step); B <= any, one workgroup
__global__ void __launch_bounds__(256) loss_stats_kernel(const float* tot, int B, float n_frames, float* out) {
    __shared__ float rs[4], rn[4];
    float s = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float v = tot[i];
        if (v > -INFINITY && v < INFINITY) s -= v; else n += 1.f;
    }
    s = wave_sum(s);
    n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rn[threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
        out[1] = n_frames;
        out[2] = (rn[0] + rn[1]) + (rn[2] + rn[3]);
    }
}

extern "C" int wf_loss_stats(const float* tot, int32_t B, float n_frames, float* out, void* stream) {
    if (!tot || !out || B < 0) return wf_fail(WF_EINVAL, "wf_loss_stats: bad argument");
    hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tot, B, n_frames, out);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// Per-label column sums of a gradient box grad[n, t, :] (n < N, t < T): the tail of the fused all-reduce buffer
// ([loss_sum, n_frames, n_infeasible, sum_{b,t} grad[b,t,label]]).  A streaming read of the box: a
// workgroup owns 1024 columns (4 per lane, 256 apart: coalesced) over a run of kLabelRows rows and adds its partial sums
// into out[] with one atomic per column.  Rows outside the segments hold zeros by wf_fb_logsum's contract.
constexpr int kLabelRows = 40;
__global__ void __launch_bounds__(256) label_sums_kernel(const float* grad, int64_t strideB, int64_t strideT, int N, int T,
                                                          int V, float* out) {
    const int c0 = blockIdx.x * 1024 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * kLabelRows, r1 = min(r0 + kLabelRows, (int64_t)N * T);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // four rows per round: sixteen independent loads in flight per lane (one row at a time left the kernel at 3.2 TB/s)
    for (int64_t r = r0; r < r1; r += 4) {
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t rr = min(r + i, r1 - 1);
            const float* row = grad + (rr / T) * strideB + (rr % T) * strideT;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int c = c0 + j * 256; v[i][j] = (c < V && r + i < r1) ? row[c] : 0.f; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[i][j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = c0 + j * 256; if (c < V && acc[j] != 0.f) atomicAdd(out + c, acc[j]); }
}

extern "C" int wf_label_grad_sums(const float* grad, int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t Tmax,
                                  int32_t V, float* out, void* stream) {
    if (!grad || !out || num_seqs < 0 || Tmax < 0 || V < 1) return wf_fail(WF_EINVAL, "wf_label_grad_sums: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)V, st));
    const int64_t rows = (int64_t)num_seqs * Tmax;
    if (rows == 0) return WF_OK;
    const dim3 grid((V + 1023) / 1024, (unsigned)((rows + kLabelRows - 1) / kLabelRows));
    hipLaunchKernelGGL(label_sums_kernel, grid, dim3(256), 0, st, grad, strideB, strideT, num_seqs, Tmax, V, out);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// ---- the drop-in's eager gradient (waterfall_amd/graph.py): loss reduction and upstream-gradient scaling ----------
// loss_reduce_kernel: out[0] = sum_b w[b] * (-tot[b]) (w == NULL: 1), one workgroup.  graph.py:66-76's reductions
// ('sum'; 'mean' with w[b] = 1 / (B * target_lengths[b])) without three elementwise torch launches.
__global__ void __launch_bounds__(256) loss_reduce_kernel(const float* tot, const float* w, int B, float* out) {
    __shared__ float rs[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s -= tot[i] * (w ? w[i] : 1.f);   // -(-inf) = +inf: an infeasible utterance
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) rs[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
}

extern "C" int wf_loss_reduce(const float* tot, const float* w, int32_t B, float* out, void* stream) {
    if (!tot || !out || B < 0) return wf_fail(WF_EINVAL, "wf_loss_reduce: bad argument");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tot, w, B, out);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// grad_scale_kernel: grad[seq, t, label_k] *= g for the label columns of the utterance's (compact) graph, g = gs[u]
// (per_utt) or gs[0]; a wavefront whose factor is exactly 1 exits at once -- the usual loss.backward().
struct GradScaleArgs {
    const int32_t* blob;
    const int64_t* utt;
    const int32_t* seq2utt;
    const float* gs;
    int per_utt;
    float* grad;
    int64_t strideB, strideT;
    int Tmax;
};

__global__ void __launch_bounds__(64) grad_scale_kernel(GradScaleArgs p) {
    const int lane = threadIdx.x;
    const int f0 = blockIdx.x * kPostFR, q = blockIdx.y;
    const int u = p.seq2utt[q];
    if (u < 0) return;
    const float g = p.gs[p.per_utt ? u : 0];
    if (g == 1.f) return;
    const int64_t* U = p.utt + (int64_t)u * WF_U_WORDS;
    const int start = (int)U[WF_U_START], T = (int)U[WF_U_T], K = (int)U[WF_U_K], direct = (int)U[WF_U_DIRECT];
    const int32_t* labels = p.blob + U[WF_U_LABELS];
    const int t0 = max(f0 - start, 0), t1 = min(f0 + kPostFR - start, T);
    for (int k = lane; k < K; k += 64) {
        const int c = direct ? k : labels[k];
        for (int t = t0; t < t1; ++t) {
            float* e = p.grad + (int64_t)q * p.strideB + (int64_t)(start + t) * p.strideT + c;
            const float v = *e;
            if (v != 0.f) *e = v * g;
        }
    }
}

// Must follow a wf_fb_logsum(grad != NULL) call with the same graph / seg / workspace on the same stream (its tables
// are still in the workspace).  gs: DEVICE fp32, [B] (per_utt != 0) or [1].
extern "C" int wf_grad_scale(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const int32_t* seg,
                             int32_t B, int32_t num_seqs, int32_t Tmax, int32_t V, int64_t strideB, int64_t strideT,
                             const float* gs, int32_t per_utt, float* grad, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!graph_dev || !seg || !gs || !grad || !workspace) return wf_fail(WF_EINVAL, "wf_grad_scale: null pointer");
    Plan P;
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, 0, &P);
    if (rc) return rc;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    // only the per-call tables at the head of the workspace are read (the same offsets under every plan: the call before
    // may have been planned with WF_PRUNED / WF_DOUBLE_ACC, whose workspaces differ further down)
    if (ws + P.off_gscale > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.off_gscale + 256);
    GradScaleArgs a{(const int32_t*)graph_dev, (const int64_t*)(ws + P.off_utt), (const int32_t*)(ws + P.off_seq2utt), gs, per_utt ? 1 : 0,
                    grad, strideB, strideT, Tmax};
    hipLaunchKernelGGL(grad_scale_kernel, dim3((Tmax + kPostFR - 1) / kPostFR, num_seqs), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_timing_enable(int on) { g_tm.on = on != 0; g_tm.have_fwd = g_tm.have_post = g_tm.have_zero = false; return WF_OK; }

// ms[0..4] = gather, sweep, post, zero-fill, check + fallback kernel time of the last timed call(s); ms[5] = the banded
// sweep launch's own duration on the device (in-kernel s_memrealtime: latest workgroup end - earliest start);
// -1 = not measured.  Synchronises on the recorded events (a measurement aid, not part of the hot path).
extern "C" int wf_timing_read(float* ms) {
    if (!ms) return wf_fail(WF_EINVAL, "wf_timing_read: null");
    ms[0] = ms[1] = ms[2] = ms[3] = ms[4] = ms[5] = -1.f;
    if (g_tm.have_check) {
        HIP_TRY(hipEventSynchronize(g_tm.ev[7]));
        HIP_TRY(hipEventElapsedTime(&ms[4], g_tm.ev[g_tm.check_from], g_tm.ev[7]));
    }
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
    if (g_tm.have_span && g_tm.have_fwd) {
        unsigned long long sp[2] = {0, 0};
        HIP_TRY(hipMemcpy(sp, g_tm.tspan, sizeof(sp), hipMemcpyDeviceToHost));   // (the events above were waited for)
        if (sp[1] > sp[0]) ms[5] = (float)((double)(sp[1] - sp[0]) * 1e-5);       // 100 MHz ticks -> ms
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
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, 0, &P, dbl, (flags & WF_PRUNED) != 0);
    if (rc) return rc;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    (void)tot;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = tm_record(3, st))) return rc;
    // P.lin repeats the decision the forward call made; the fallback flags are still in the workspace
    // WF_PREZERO: the forward call zero-filled the box under exactly the condition below (its `overlap`)
    const bool prezeroed = (flags & WF_PREZERO) && P.any_compact && !hk().no_overlap;
    // WF_ACCUMULATE: the box holds another call's gradient; this call's posteriors are added (see wf_fb_logsum)
    const bool accum = flags & WF_ACCUMULATE;
    if (accum && ((flags & (WF_PRUNED | WF_PREZERO)) || !P.lin || P.hub || !post_small_fits(P)))
        return wf_fail(WF_EUNSUPPORTED, "wf_fb_backward: WF_ACCUMULATE is for graphs of the probability-domain path with at most 512 states / 1024 arcs (numerators), without WF_PRUNED / WF_PREZERO");
    rc = launch_post(P, graph_dev, ws, log_probs, strideB, strideT, num_seqs, Tmax, V, grad_scale, grad, prezeroed || accum, st, -1, accum);
    if (rc) return rc;
    if ((rc = tm_record(4, st))) return rc;
    g_tm.have_post = g_tm.on;
    return WF_OK;
}

extern "C" int wf_state_posteriors(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                                   const int32_t* seg, int32_t B, int32_t V, int32_t num_seqs, uint32_t flags,
                                   float* out, int64_t out_strideB, int64_t out_strideT, int32_t Smax, int32_t Tmax,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !out || !workspace || !seg) return wf_fail(WF_EINVAL, "wf_state_posteriors: null pointer");
    if (flags & WF_PRUNED) return wf_fail(WF_EUNSUPPORTED, "wf_state_posteriors: not available with output_beam pruning");
    Plan P;
    int rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, 8, 0, &P, (flags & WF_DOUBLE_ACC) != 0);
    if (rc) return rc;
    if (P.smax > Smax) return wf_fail(WF_ERANGE, "wf_state_posteriors: a graph has %d states but out has room for %d", P.smax, Smax);
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + P.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, P.total + 256);
    const BandSel bs = band_config(P);
    StatePostArgs a{};
    a.utt = (const int64_t*)(ws + P.off_utt);
    a.alpha = (const double*)(ws + P.off_alpha); a.beta = (const double*)(ws + P.off_beta);
    a.aoff = (const double*)(ws + P.off_aoff); a.boff = (const double*)(ws + P.off_boff);
    a.tot2 = (const double*)(ws + P.off_tot2); a.ctot2 = (const double*)(ws + P.off_ctot2);
    a.totd = (const double*)(ws + P.off_totd); a.flag = (const int32_t*)(ws + P.off_flag);
    a.lin = P.lin ? 1 : 0;
    lin_grp_lag(P, &a.grp, &a.lag);
    if (!P.lin) a.grp = 1;
    a.out = out; a.strideB = out_strideB; a.strideT = out_strideT; a.Smax = Smax; a.Tmax = Tmax;
    a.rows_f32 = P.hub ? 1 : 0;
    hipLaunchKernelGGL(state_post_kernel, dim3(std::max(Tmax, 1), B), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_tot_scores(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                             const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t Tmax,
                             int32_t V, const int32_t* seg, int32_t B, uint32_t flags, void* tot,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !log_probs || !tot || !workspace) return wf_fail(WF_EINVAL, "wf_tot_scores: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool dbl = flags & WF_DOUBLE_ACC;
    const size_t ab = (dbl || !(flags & WF_TROPICAL)) ? 8 : 4;
    Plan P;
    int rc = check_segments("wf_tot_scores", seg, B, num_seqs, Tmax);
    if (rc) return rc;
    if (dbl && !(flags & WF_TROPICAL)) {
        // get_tot_scores(log_semiring=True, use_double_scores=True) (lattice_analysis.py:128,148): the fp64 banded sweeps
        // where they apply (and the workspace is the one wf_fb_workspace_bytes(WF_DOUBLE_ACC, need_grad = 0) promised)
        Plan Q;
        bool via = false;
        if (plan_fb(graph_host, utt2fsa, seg, B, V, num_seqs, true, true, false, &Q, &via) == WF_OK && Q.lin &&
            Q.total + 256 <= workspace_bytes)
            return fb_logsum_impl(graph_host, graph_dev, utt2fsa, log_probs, strideB, strideT, num_seqs, Tmax, V, seg, B, 0.0f,
                                  WF_DOUBLE_ACC | WF_KEEP_AB | WF_TOT_F64, nullptr, (float*)tot, nullptr, workspace,
                                  workspace_bytes, stream, false);
    }
    rc = make_plan(graph_host, utt2fsa, seg, B, V, num_seqs, ab, 1, &P);
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
    if (P.big) {
        BigArgs g{};
        g.blob = a.blob; g.utt = a.utt; g.lp = log_probs; g.strideB = strideB; g.strideT = strideT; g.E = a.E;
        g.buf = ws + P.off_big; g.tot = tot; g.tot_double = a.tot_double;
        if (P.shared) {
            if (flags & WF_TROPICAL) return dbl ? launch_big_shared<double, double, true>(P, ws, g, B, st) : launch_big_shared<float, float, true>(P, ws, g, B, st);
            return dbl ? launch_big_shared<double, double, false>(P, ws, g, B, st) : launch_big_shared<double, float, false>(P, ws, g, B, st);
        }
        if (flags & WF_TROPICAL) return dbl ? launch_big<double, double, true>(P, g, B, st) : launch_big<float, float, true>(P, g, B, st);
        return dbl ? launch_big<double, double, false>(P, g, B, st) : launch_big<double, float, false>(P, g, B, st);
    }
    RegCfg rc_ = dbl ? RegCfg() : reg_config(P, false);
    if (rc_.ok) return (flags & WF_TROPICAL) ? launch_sweep_quad<M_TOT_MAX>(rc_, a, B, st)
                                             : launch_sweep_quad<M_TOT_LOG>(rc_, a, B, st);
    if (flags & WF_TROPICAL) return dbl ? launch_sweep<double, double, M_TOT_MAX>(c, a, B, st) : launch_sweep<float, float, M_TOT_MAX>(c, a, B, st);
    return dbl ? launch_sweep<double, double, M_TOT_LOG>(c, a, B, st) : launch_sweep<double, float, M_TOT_LOG>(c, a, B, st);
}

extern "C" int wf_viterbi(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                          const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V,
                          const int32_t* seg, int32_t B, int32_t Tmax, uint32_t flags, float* best_score,
                          int32_t* best_arcs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!graph_dev || !log_probs || !best_score || !best_arcs || !workspace || !seg) return wf_fail(WF_EINVAL, "wf_viterbi: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool dbl = flags & WF_DOUBLE_ACC;
    const size_t ab = dbl ? 8 : 4;
    Plan P;
    int rc = check_segments("wf_viterbi", seg, B, num_seqs, Tmax);
    if (rc) return rc;
    int maxseq = 0;   // (sizes the plan exactly as wf_viterbi_workspace_bytes does)
    for (int i = 0; i < B; ++i) maxseq = std::max(maxseq, seg[3 * i] + 1);
    rc = make_plan(graph_host, utt2fsa, seg, B, V, maxseq, ab, 2, &P);
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
    RegCfg rc_ = (dbl || P.big) ? RegCfg() : reg_config(P, false);
    if (P.big) {
        BigArgs g{};
        g.blob = a.blob; g.utt = a.utt; g.lp = log_probs; g.strideB = strideB; g.strideT = strideT; g.E = a.E;
        g.buf = ws + P.off_big; g.tot = best_score; g.tot_double = 0; g.bp = a.bp; g.fin_arc = a.fin_arc;
        if (P.shared) rc = dbl ? launch_big_shared<double, double, true>(P, ws, g, B, st) : launch_big_shared<float, float, true>(P, ws, g, B, st);
        else rc = dbl ? launch_big<double, double, true>(P, g, B, st) : launch_big<float, float, true>(P, g, B, st);
    } else if (rc_.ok) rc = launch_sweep_quad<M_VITERBI>(rc_, a, B, st);
    else rc = dbl ? launch_sweep<double, double, M_VITERBI>(c, a, B, st) : launch_sweep<float, float, M_VITERBI>(c, a, B, st);
    if (rc) return rc;
    TraceArgs t{a.blob, a.utt, a.bp, a.fin_arc, nullptr, best_arcs, B, Tmax};
    hipLaunchKernelGGL(backtrace_kernel, dim3((B + 63) / 64), dim3(64), 0, st, t);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// ---- beam-pruned search -------------------------------------------------------------------------
namespace {
struct SearchPlan {
    Plan P;   // only P.utt (utterance table followed by the search table) and P.off_utt are used
    size_t off_best = 0, off_lsum = 0, off_slot = 0, off_cand = 0, off_cpk = 0, off_cls = 0, off_arcown = 0, off_xs = 0, off_xv = 0, off_xl = 0, off_xa = 0, off_arena = 0, off_ctl = 0, off_rinfo = 0, off_bt = 0, off_bl = 0, off_ftab = 0, off_wl = 0, off_wt = 0, total = 0;
    int64_t nstates = 0;
    int wgs = 1;
    int ftab_stride = 0;   // (lattice: longest segment + 2)
    int wstride = 0;       // (words: longest word sequence + 1)
    size_t tab_words = 0;  // int64 words of the utterance + search tables (the word tables follow them in P.utt)
};
// Workgroups per utterance (search_mw_kernel when > 1): up to 8 while the launch fills at most half of the chip, and only
// where the active set may grow past what one workgroup handles well.  WF_SEARCH_WGS forces a number (1: search_kernel).
int search_wgs(int B, int max_active) {
    const int forced = env_int("WF_SEARCH_WGS", 0);
    if (forced > 0) return std::min(forced, 16);
    if (max_active > 0 && max_active < 1024) return 1;
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? n : 0;
        (void)hipGetLastError();
    }
    int g = 8;
    while (g > 1 && (long long)B * g > cus / 2) g >>= 1;
    return g;
}
int plan_search(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int B, int V, int max_active,
                bool logsum, bool lattice, SearchPlan* S, const int32_t* words = nullptr, const int32_t* word_off = nullptr) {
    int rc = check_blob(graph_host);
    if (rc) return rc;
    if (B < 1 || !seg || V < 1) return wf_fail(WF_EINVAL, "bad B/seg/V");
    if (max_active < 0) return wf_fail(WF_EINVAL, "max_active_states < 0");
    const int32_t* H = (const int32_t*)graph_host;
    const int G = H[WF_H_NUM_FSAS];
    if (!utt2fsa && G != B) return wf_fail(WF_EINVAL, "graph has %d FSAs but %d segments (pass utt2fsa to share graphs)", G, B);
    S->P.B = B;
    S->P.utt.assign((size_t)B * (WF_U_WORDS + WF_SR_WORDS), 0);
    int64_t ns = 0, nx = 0, nr = 0, na = 0;
    for (int i = 0; i < B; ++i) {
        const int g = utt2fsa ? utt2fsa[i] : i;
        if (g < 0 || g >= G) return wf_fail(WF_ERANGE, "utt2fsa[%d] = %d out of range", i, g);
        const int32_t* I = H + WF_HDR_WORDS + (size_t)g * WF_INFO_WORDS;
        if (I[WF_I_MAX_LABEL] >= V) return wf_fail(WF_ERANGE, "fsa %d uses label %d but log_probs has V=%d columns", g, I[WF_I_MAX_LABEL], V);
        if (I[WF_I_A] > 0 && !I[WF_I_OUT_REC]) return wf_fail(WF_EUNSUPPORTED, "prepared graph has no outgoing-arc index (prepared by an older library)");
        const int b = seg[3 * i], st = seg[3 * i + 1], T = seg[3 * i + 2];
        if (b < 0 || st < 0 || T < 0) return wf_fail(WF_ERANGE, "segment %d = (%d,%d,%d) out of range", i, b, st, T);
        int64_t* U = &S->P.utt[(size_t)i * WF_U_WORDS];
        U[WF_U_FSA] = g; U[WF_U_SEQ] = b; U[WF_U_START] = st; U[WF_U_T] = T;
        int64_t* R = &S->P.utt[(size_t)B * WF_U_WORDS + (size_t)i * WF_SR_WORDS];
        const int cap = max_active > 0 ? std::min(max_active, (int)I[WF_I_S]) : (int)I[WF_I_S];
        R[WF_SR_SOFF] = ns; R[WF_SR_XOFF] = nx; R[WF_SR_CAP] = cap; R[WF_SR_AOFF] = nr; R[WF_SR_ARCOFF] = na;
        ns += I[WF_I_S];
        na += (int64_t)I[WF_I_A] + I[WF_I_NF];
        nx += cap;
        nr += (int64_t)(T + 1) * cap + 1;
        S->ftab_stride = std::max(S->ftab_stride, T + 2);
    }
    S->nstates = ns;
    S->tab_words = S->P.utt.size();
    int64_t ncell = 0;
    if (word_off) {   // search_words_kernel: per utterance the ring's first cell, then [n, words...] rows (int32, packed)
        if (!words && word_off[B] > 0) return wf_fail(WF_EINVAL, "null words");
        for (int i = 0; i < B; ++i) {
            const int n = word_off[i + 1] - word_off[i];
            if (n < 0) return wf_fail(WF_EINVAL, "word_off must not decrease");
            S->wstride = std::max(S->wstride, n + 1);
        }
        std::vector<int32_t> rows((size_t)B * S->wstride + 1, 0);
        for (int i = 0; i < B; ++i) {
            const int n = word_off[i + 1] - word_off[i];
            const int64_t cap = S->P.utt[(size_t)B * WF_U_WORDS + (size_t)i * WF_SR_WORDS + WF_SR_CAP];
            S->P.utt.push_back(ncell);
            ncell += 2 * cap * (n + 1);
            rows[(size_t)i * S->wstride] = n;
            for (int k = 0; k < n; ++k) {
                const int w = words[word_off[i] + k];
                if (w <= 0) return wf_fail(WF_EINVAL, "word ids must be > 0 (0 is epsilon)");
                rows[(size_t)i * S->wstride + 1 + k] = w;
            }
        }
        const size_t at = S->P.utt.size();
        S->P.utt.resize(at + (rows.size() + 1) / 2, 0);
        memcpy(&S->P.utt[at], rows.data(), rows.size() * sizeof(int32_t));
    }
    const size_t W = (size_t)(S->wgs = search_wgs(B, max_active));   // (every workgroup of an utterance lists its own candidates)
    size_t o = 0;
    S->P.off_utt = o; o = align_up(o + sizeof(int64_t) * S->P.utt.size(), 256);
    S->off_ctl = o; o = align_up(o + sizeof(SearchCtl) * (size_t)B, 256);
    S->off_best = o; o = align_up(o + 8 * (size_t)ns, 256);
    S->off_lsum = o; if (logsum) o = align_up(o + 8 * (size_t)ns, 256);
    S->off_slot = o; o = align_up(o + 4 * (size_t)ns, 256);
    S->off_cand = o; o = align_up(o + 4 * (size_t)ns * W, 256);
    S->off_cpk = o; o = align_up(o + 8 * (size_t)ns * W, 256);
    S->off_cls = o; if (logsum) o = align_up(o + 8 * (size_t)ns * W, 256);
    S->off_arcown = o; o = align_up(o + 4 * (size_t)na, 256);
    S->off_xs = o; o = align_up(o + 4 * (size_t)nx, 256);
    S->off_xv = o; o = align_up(o + 4 * (size_t)nx, 256);
    S->off_xl = o; o = align_up(o + 8 * (size_t)nx, 256);
    S->off_xa = o; o = align_up(o + 8 * (size_t)nx, 256);
    S->off_arena = o; o = align_up(o + 8 * (size_t)nr, 256);
    if (lattice) {   // the backward output_beam pass: per record (state, score), tropical / log backward scores; frame table
        S->off_rinfo = o; o = align_up(o + 16 * (size_t)nr, 256);
        S->off_bt = o; o = align_up(o + 4 * (size_t)nr, 256);
        S->off_bl = o; o = align_up(o + 8 * (size_t)nr, 256);
        S->off_ftab = o; o = align_up(o + 8 * (size_t)B * S->ftab_stride, 256);
    }
    if (word_off) {
        S->off_wl = o; o = align_up(o + 8 * (size_t)ncell, 256);
        S->off_wt = o; o = align_up(o + 4 * (size_t)ncell, 256);
    }
    S->total = o;
    return WF_OK;
}
}  // namespace

extern "C" int wf_search_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                                         int32_t V, int32_t max_active_states, int32_t want_log_tot, size_t* bytes) {
    if (!bytes) return wf_fail(WF_EINVAL, "null bytes");
    SearchPlan S;
    int rc = plan_search(graph_host, utt2fsa, seg, B, V, max_active_states, want_log_tot != 0, false, &S);
    if (rc) return rc;
    *bytes = S.total + 256;
    return WF_OK;
}

extern "C" int wf_search_lattice_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                                                 int32_t V, int32_t max_active_states, size_t* bytes) {
    if (!bytes) return wf_fail(WF_EINVAL, "null bytes");
    SearchPlan S;
    int rc = plan_search(graph_host, utt2fsa, seg, B, V, max_active_states, false, true, &S);
    if (rc) return rc;
    *bytes = S.total + 256;
    return WF_OK;
}

namespace {
int search_impl(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
                int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B, int32_t Tmax,
                float search_beam, int32_t min_active_states, int32_t max_active_states, uint32_t flags,
                float* best_score, int32_t* best_arcs, void* tot_log, int32_t* n_active, void* workspace,
                size_t workspace_bytes, void* stream, bool lattice, float output_beam, int64_t* lattice_arcs,
                const int32_t* aux_dev = nullptr, const int32_t* words = nullptr, const int32_t* word_off = nullptr,
                void* tot_w = nullptr, float* best_w = nullptr) {
    if (!graph_dev || !log_probs || !best_score || !best_arcs || !workspace || !seg) return wf_fail(WF_EINVAL, "wf_search: null pointer");
    if (!(search_beam > 0.f)) return wf_fail(WF_EINVAL, "wf_search: search_beam must be > 0 (inf: no beam)");
    if (min_active_states < 0) return wf_fail(WF_EINVAL, "wf_search: min_active_states < 0");
    hipStream_t st = (hipStream_t)stream;
    int rc = check_segments("wf_search", seg, B, num_seqs, Tmax);
    if (rc) return rc;
    SearchPlan S;
    // (with the lattice pass the total comes from search_back_kernel: the forward kernels skip their log-semiring pass)
    const bool fwd_log = tot_log != nullptr && !lattice;
    rc = plan_search(graph_host, utt2fsa, seg, B, V, max_active_states, fwd_log, lattice, &S, words, word_off);
    if (rc) return rc;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (ws + S.total > (unsigned char*)workspace + workspace_bytes) return wf_fail(WF_ENOSPC, "workspace too small: %zu < %zu", workspace_bytes, S.total + 256);
    rc = upload_tables(S.P, ws, nullptr, 0, st);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ws + S.off_best, 0, 8 * (size_t)S.nstates, st));
    if (fwd_log) HIP_TRY(hipMemsetAsync(ws + S.off_lsum, 0, 8 * (size_t)S.nstates, st));
    SearchArgs a{};
    a.blob = (const int32_t*)graph_dev;
    a.utt = (const int64_t*)(ws + S.P.off_utt);
    a.tab = a.utt + (size_t)B * WF_U_WORDS;
    a.lp = log_probs; a.strideB = strideB; a.strideT = strideT;
    a.best = (unsigned long long*)(ws + S.off_best);
    a.lsum = fwd_log ? (double*)(ws + S.off_lsum) : nullptr;
    a.slot = (int32_t*)(ws + S.off_slot);
    a.cand = (int32_t*)(ws + S.off_cand);
    a.cpk = (unsigned long long*)(ws + S.off_cpk);
    a.cls = fwd_log ? (double*)(ws + S.off_cls) : nullptr;
    a.arcown = (int32_t*)(ws + S.off_arcown);
    a.xstate = (int32_t*)(ws + S.off_xs);
    a.xscore = (float*)(ws + S.off_xv);
    a.xla = (double*)(ws + S.off_xl);
    a.xarcs = (int2*)(ws + S.off_xa);
    a.arena = (int2*)(ws + S.off_arena);
    a.nstates = S.nstates;
    a.beam = search_beam; a.max_active = max_active_states; a.min_active = min_active_states;
    a.best_score = best_score; a.best_arcs = best_arcs; a.Tmax = Tmax;
    a.tot_log = fwd_log ? tot_log : nullptr; a.tot_double = (flags & WF_DOUBLE_ACC) ? 1 : 0;
    a.n_active = n_active;
    if (lattice) {
        a.rinfo = (int4*)(ws + S.off_rinfo);
        a.ftab = (int64_t*)(ws + S.off_ftab);
        a.ftab_stride = S.ftab_stride;
    }
    static const int search_prof = getenv("WF_SEARCH_PROFILE") ? atoi(getenv("WF_SEARCH_PROFILE")) : 0;
    a.prof = search_prof;
    if (S.wgs > 1) {
        // several workgroups per utterance; what they give up on (their partners never arrived: a chip shared with other
        // work) the one-workgroup kernel redoes in the second launch, which otherwise exits at once
        HIP_TRY(hipMemsetAsync(ws + S.off_ctl, 0, sizeof(SearchCtl) * (size_t)B, st));
        a.wgs = S.wgs; a.B = B; a.solo_max = env_int("WF_SEARCH_SOLO_MAX", 256);
        a.test_drop = env_int("WF_SEARCH_TEST_DROP", 0);
        a.ctl = (SearchCtl*)(ws + S.off_ctl);
        hipLaunchKernelGGL(search_mw_kernel, dim3(B * S.wgs), dim3(kSearchThreads), 0, st, a);
        a.redo = &a.ctl[0].abort;
        a.redo_stride = (int)(sizeof(SearchCtl) / sizeof(int32_t));
    }
    hipLaunchKernelGGL(search_kernel, dim3(B), dim3(kSearchThreads), 0, st, a);
    if (lattice) {   // the forward total is replaced by the total of the lattice that output_beam leaves
        SearchBackArgs b{};
        b.blob = a.blob; b.utt = a.utt; b.tab = a.tab;
        b.lp = log_probs; b.strideB = strideB; b.strideT = strideT;
        b.slot = a.slot; b.slot2 = a.cand; b.rinfo = a.rinfo;
        b.bt = (float*)(ws + S.off_bt);
        b.bl = (double*)(ws + S.off_bl);
        b.ftab = a.ftab; b.ftab_stride = a.ftab_stride;
        b.best_score = best_score;
        b.output_beam = output_beam;
        b.tot_log = tot_log; b.tot_double = a.tot_double;
        b.lattice_arcs = (long long*)lattice_arcs;
        b.wgs = 1; b.B = B;
        if (S.wgs > 1) {   // (the control blocks were zeroed before the forward launch)
            b.wgs = S.wgs; b.ctl = a.ctl; b.solo_max = a.solo_max; b.test_drop = a.test_drop;
            hipLaunchKernelGGL(search_back_kernel, dim3(B * S.wgs), dim3(kSearchThreads), 0, st, b);
            b.wgs = 1; b.redo = 1; b.test_drop = 0;   // what the barrier gave up on: one workgroup, from the start
        }
        hipLaunchKernelGGL(search_back_kernel, dim3(B), dim3(kSearchThreads), 0, st, b);
        if (word_off) {   // the output lattice restricted to each utterance's word sequence
            SearchWordsArgs w{};
            w.b = b; w.b.redo = 0; w.b.ctl = nullptr; w.b.wgs = 1;
            w.aux = aux_dev;
            w.woff = a.utt + S.tab_words;
            w.wtab = (const int32_t*)(w.woff + B);
            w.wstride = S.wstride;
            w.wl = (double*)(ws + S.off_wl);
            w.wt = (float*)(ws + S.off_wt);
            w.tot_w = tot_w; w.best_w = best_w;
            if (S.wgs > 1) {
                w.b.wgs = S.wgs; w.b.ctl = a.ctl; w.b.solo_max = a.solo_max; w.b.test_drop = a.test_drop;
                hipLaunchKernelGGL(search_words_kernel, dim3(B * S.wgs), dim3(kSearchThreads), 0, st, w);
                w.b.wgs = 1; w.b.redo = 1; w.b.test_drop = 0;
            }
            hipLaunchKernelGGL(search_words_kernel, dim3(B), dim3(kSearchThreads), 0, st, w);
        }
    }
    HIP_TRY(hipGetLastError());
    return WF_OK;
}
}  // namespace

extern "C" int wf_search(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
                         int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B, int32_t Tmax,
                         float search_beam, int32_t min_active_states, int32_t max_active_states, uint32_t flags,
                         float* best_score, int32_t* best_arcs, void* tot_log, int32_t* n_active, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return search_impl(graph_host, graph_dev, utt2fsa, log_probs, strideB, strideT, num_seqs, V, seg, B, Tmax, search_beam,
                       min_active_states, max_active_states, flags, best_score, best_arcs, tot_log, n_active, workspace,
                       workspace_bytes, stream, false, 0.f, nullptr);
}

extern "C" int wf_search_lattice_words_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg,
                                                       int32_t B, int32_t V, int32_t max_active_states,
                                                       const int32_t* words, const int32_t* word_off, size_t* bytes) {
    if (!bytes || !word_off) return wf_fail(WF_EINVAL, "null bytes / word_off");
    SearchPlan S;
    int rc = plan_search(graph_host, utt2fsa, seg, B, V, max_active_states, false, true, &S, words, word_off);
    if (rc) return rc;
    *bytes = S.total + 256;
    return WF_OK;
}

extern "C" int wf_search_lattice_words(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
                                       int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B,
                                       int32_t Tmax, float search_beam, float output_beam, int32_t min_active_states,
                                       int32_t max_active_states, uint32_t flags, const int32_t* aux_dev, const int32_t* words,
                                       const int32_t* word_off, float* best_score, int32_t* best_arcs, void* tot_log,
                                       void* tot_words, float* best_words, void* workspace, size_t workspace_bytes, void* stream) {
    if (!tot_log || !aux_dev || !word_off || !tot_words || !best_words) return wf_fail(WF_EINVAL, "wf_search_lattice_words: null pointer");
    if (!(output_beam > 0.f)) return wf_fail(WF_EINVAL, "wf_search_lattice_words: output_beam must be > 0 (inf: no pruning)");
    return search_impl(graph_host, graph_dev, utt2fsa, log_probs, strideB, strideT, num_seqs, V, seg, B, Tmax, search_beam,
                       min_active_states, max_active_states, flags, best_score, best_arcs, tot_log, nullptr, workspace,
                       workspace_bytes, stream, true, output_beam, nullptr, aux_dev, words, word_off, tot_words, best_words);
}

extern "C" int wf_search_lattice(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
                                 int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B,
                                 int32_t Tmax, float search_beam, float output_beam, int32_t min_active_states,
                                 int32_t max_active_states, uint32_t flags, float* best_score, int32_t* best_arcs, void* tot_log,
                                 int64_t* lattice_arcs, int32_t* n_active, void* workspace, size_t workspace_bytes, void* stream) {
    if (!tot_log) return wf_fail(WF_EINVAL, "wf_search_lattice: null tot_log");
    if (!(output_beam > 0.f)) return wf_fail(WF_EINVAL, "wf_search_lattice: output_beam must be > 0 (inf: no pruning)");
    return search_impl(graph_host, graph_dev, utt2fsa, log_probs, strideB, strideT, num_seqs, V, seg, B, Tmax, search_beam,
                       min_active_states, max_active_states, flags, best_score, best_arcs, tot_log, n_active, workspace,
                       workspace_bytes, stream, true, output_beam, lattice_arcs);
}
