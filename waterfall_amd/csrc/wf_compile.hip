// Device numerator-graph compiler: the reference rebuilds the numerator graphs on every training
// step (waterfall/models.py:263-270: linear_fsa -> remove_epsilon_and_add_self_loops -> arc_sort -> compose(topo, .,
// treat_epsilons_specially=False); waterfall/utils/datapipe.py:233-256).  numerator_compile_kernel builds, from the flat
// transcripts and the topology tables of wf_topo_device_build, the SAME graphs and the SAME prepared tables as the host's
// wf_numerator_compile (wf_graph.cpp: compose_linear_one + prepare_one, word for word) -- one wavefront per utterance,
// nothing but the transcripts crosses PCIe, and the host does no graph work at all.
//
// What makes it parallel: product states (q, i) are numbered position-major, so
//   1. every transcript position lists for itself the topology states that can be alive there (the states its phone's
//      arcs enter, closed under the arcs that stay at the position) and their arcs as two 8 x 8 bit matrices;
//   2. one pass over the positions carries the distance from the start through those matrices (8 x 8 lanes, min-plus):
//      reachable states, and the fewest frames an accepted path needs, fall out of it;
//   3. state ids are a prefix sum over the positions; a state's arcs are enumerated by the lane that owns it, in the
//      topology's order; arc offsets are prefix sums again;
//   4. the incoming-arc CSR, the column-sorted arcs and the distinct labels are bitonic sorts in the LDS; the banded order
//      is "by position, inside a position by depth, then by state" (prepare_one's pos_hint order).
// Limits (WF_NC_*): <= 256 positions, <= 8 live topology states per position, <= 257 states, <= 1024 arcs; anything else
// sets the utterance's status and the caller compiles that batch on the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/wf_dwfst.h"
#include "wf_internal.h"

int wf_fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return wf_fail(WF_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

constexpr int kNP = WF_NC_MAX_POS, kNC = WF_NC_MAX_CAND, kSMax = WF_NC_MAX_S, kAMax = WF_NC_MAX_A, kNFMax = WF_NC_MAX_NF,
              kKMax = WF_NC_MAX_K;
constexpr int kInf = 0x3fffffff;
constexpr unsigned kNinfBits = 0xff800000u;

enum { NC_OK = 0, NC_LONG = 1, NC_PHONE = 2, NC_CAND = 3, NC_STATES = 4, NC_ARCS = 5, NC_FINALS = 6, NC_LABELS = 7, NC_SPACE = 8 };

struct NcTopo {
    const int32_t* arcs; const int32_t* aux; const int32_t* index;
    const int32_t* soff; const int32_t* perm; const int32_t* key; const int32_t* f1s; const int32_t* z1s; const int32_t* dense;
    const int32_t* pstate; const int32_t* ptoken; const int32_t* pdoff; const int32_t* pdst;
    int nA, nS, cross, nph, maxph, numph, maxd;
};

__device__ __forceinline__ NcTopo nc_topo(const int32_t* t) {
    NcTopo o;
    o.nA = t[WF_T_NA]; o.nS = t[WF_T_NS]; o.cross = t[WF_T_CROSS]; o.nph = t[WF_T_NPH]; o.maxph = t[WF_T_MAXPH];
    o.numph = t[WF_T_NUMPH]; o.maxd = t[WF_T_MAXD];
    o.arcs = t + t[WF_T_OARCS]; o.aux = t + t[WF_T_OAUX]; o.index = t + t[WF_T_OINDEX];
    o.soff = o.index; o.perm = o.index + o.nS + 1; o.key = o.perm + o.nA; o.f1s = o.key + o.nA; o.z1s = o.f1s + o.nS;
    o.dense = o.z1s + o.nS;
    o.pstate = t + t[WF_T_OPSTATE]; o.ptoken = t + t[WF_T_OPTOKEN]; o.pdoff = t + t[WF_T_OPDOFF]; o.pdst = t + t[WF_T_OPDST];
    return o;
}

// The emitting arcs of product state (q, i) in the order compose_linear_one (wf_graph.cpp) emits them: f(q2, adv, tok,
// weight bits, phone) with adv = 1 for an arc that consumes transcript position i.  Returns their number.
template <typename F>
__device__ __forceinline__ int nc_enum(const NcTopo& t, const int q, const int i, const int L, const int32_t* seq, F&& f) {
    const int hi = t.soff[q + 1];
    const int f1 = t.f1s[q], z1 = t.z1s[q];
    int p0 = hi, p1 = hi, nph = 0;
    if (i < L) {
        nph = seq[i];
        if (t.dense[q]) {
            if (nph <= t.maxph) { const int32_t* tb = t.index + t.dense[q]; p0 = tb[nph]; p1 = tb[nph + 1]; }
        } else {
            p0 = z1;
            while (p0 < hi && t.key[p0] < nph) ++p0;
            p1 = p0;
            while (p1 < hi && t.key[p1] == nph) ++p1;
        }
    }
    int cq = 0, ctok = 0;
    const bool cross = t.cross && q != 0 && i < L && t.pstate[nph] > 0 && t.pstate[nph] != q;
    if (cross) { cq = t.pstate[nph]; ctok = t.ptoken[nph]; }
    int x = f1, y = p0, n = 0;
    if (!cross) {
        while (x < z1 || y < p1) {
            const int ax = x < z1 ? t.perm[x] : 0x7fffffff, ay = y < p1 ? t.perm[y] : 0x7fffffff;
            if (ax < ay) { const int32_t* r = t.arcs + 4 * (size_t)ax; f(r[1], 0, r[2], r[3], 0); ++x; }
            else { const int32_t* r = t.arcs + 4 * (size_t)ay; f(r[1], 1, r[2], r[3], nph); ++y; }
            ++n;
        }
    } else {
        for (; x < z1; ++x, ++n) { const int32_t* r = t.arcs + 4 * (size_t)t.perm[x]; f(r[1], 0, r[2], r[3], 0); }
        bool done = false;
        for (; y < p1; ++y, ++n) {
            const int32_t* r = t.arcs + 4 * (size_t)t.perm[y];
            if (!done && r[2] > ctok) { f(cq, 1, ctok, 0, nph); done = true; ++n; }
            f(r[1], 1, r[2], r[3], nph);
        }
        if (!done) { f(cq, 1, ctok, 0, nph); ++n; }
    }
    return n;
}

struct NcArgs {
    const int32_t* topo;        // device copy of wf_topo_device_build's table
    const int32_t* phones;      // device, flat transcripts
    const int32_t* phones_off;  // device, [B + 1]
    int B;
    int32_t* blob;              // device: the prepared graph vector
    int64_t blob_words;
    int32_t* stage;             // device scratch [arc capacity][5]: {src, dst, label, weight, aux} per utterance at its capacity offset
    int32_t* counts;            // device scratch [4 B]: arcs | states | status | max label
    int32_t* arcs;              // outputs of nc_finish_kernel: [A][4]
    int32_t* aux;               // [A]
    int32_t* meta;              // [B + 1] arc offsets | [B] states | [B] status
};

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// exclusive prefix sum over the lanes of the wave; *total = the sum
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int* total) {
    int s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(s, o, 64);
        if (lane >= o) s += u;
    }
    *total = __shfl(s, 63, 64);
    return s - v;
}

// ascending bitonic sort of k[0 .. N), N a power of two >= 2, by one wavefront (blockDim.x == 64)
__device__ void nc_sort(unsigned* k, const int N, const int lane) {
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = lane; t < N / 2; t += 64) {
                const int i = (t / stride) * stride * 2 + (t % stride), j = i + stride;
                const bool up = (i & size) == 0;
                const unsigned a = k[i], b = k[j];
                if ((a > b) == up) { k[i] = b; k[j] = a; }
            }
        }
    __syncthreads();
}
__device__ __forceinline__ int pow2_at_least(int n) { int N = 2; while (N < n) N <<= 1; return N; }
// first index in sorted k[0 .. n) whose value is >= v
__device__ __forceinline__ int nc_lower_bound(const unsigned* k, int n, unsigned v) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (k[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(64) numerator_compile_kernel(NcArgs p) {
    // ---- LDS
    __shared__ int32_t cand[kNP * kNC];                  // topology states that can be alive at position i, ascending
    __shared__ unsigned char ncand[kNP], reach[kNP];
    __shared__ unsigned long long m0bits[kNP], tbits[kNP];   // bit c*8 + c2: an arc candidate c -> candidate c2 (same position | next position)
    __shared__ unsigned char rounds[kNP];
    __shared__ int32_t distv[kNP * kNC];
    __shared__ int32_t sbase[kNP + 1];
    __shared__ int32_t st_q[kSMax];
    __shared__ unsigned short st_i[kSMax], st_c[kSMax], spos[kSMax], sord[kSMax];
    __shared__ int32_t ebase[kSMax + 1], abase[kSMax + 1], fbase[kSMax + 1];
    __shared__ unsigned short e_src[kAMax], e_dst[kAMax], e_col[kAMax], e_orig[kAMax];
    __shared__ int32_t e_lab[kAMax], e_w[kAMax];
    __shared__ unsigned short fin_src[kNFMax], fin_orig[kNFMax];
    __shared__ int32_t fin_w[kNFMax];
    __shared__ unsigned keys[kAMax];
    __shared__ int32_t labs[kKMax];
    __shared__ __align__(4) unsigned short cellmin[(kSMax - 1) * (WF_BAND_MAX_BW + 1)];
    __shared__ __align__(4) unsigned char cellcnt[(kSMax - 1) * (WF_BAND_MAX_BW + 1)];
    __shared__ int32_t info[WF_INFO_WORDS];
    __shared__ int32_t s_status;

    const int b = blockIdx.x, lane = threadIdx.x;
    const NcTopo t = nc_topo(p.topo);
    const int off0 = p.phones_off[b], L = p.phones_off[b + 1] - off0;
    const int32_t* seq = p.phones + off0;
    // this utterance's blob region and arc staging offset: sums of the capacities of the utterances before it
    size_t start = 0;
    long long acap0 = 0;
    {
        long long w = 0, a = 0;
        for (int u = lane; u < b; u += 64) {
            const int Lu = p.phones_off[u + 1] - p.phones_off[u];
            const int Lc = min(max(Lu, 0), kNP - 1);
            w += (long long)wf_fsa_words_bound((size_t)wf_nc_state_cap(Lc), (size_t)wf_nc_arc_cap(Lc, t.maxd));
            a += wf_nc_arc_cap(Lc, t.maxd);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { w += __shfl_xor(w, o, 64); a += __shfl_xor(a, o, 64); }
        start = wf_pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * p.B) + (size_t)w;
        acap0 = a;
    }
    const int Lc = min(max(L, 0), kNP - 1);
    const size_t limit = start + wf_fsa_words_bound((size_t)wf_nc_state_cap(Lc), (size_t)wf_nc_arc_cap(Lc, t.maxd));
    const int acap = wf_nc_arc_cap(Lc, t.maxd);
    int32_t* B_ = p.blob;
    int32_t* I = B_ + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * b;

    if (lane == 0) s_status = (L < 0 || L + 1 > kNP || limit > (size_t)p.blob_words) ? (L + 1 > kNP ? NC_LONG : NC_SPACE) : NC_OK;
    __syncthreads();
    auto fail = [&](int code) { atomicCAS(&s_status, NC_OK, code); };
    auto ok = [&]() { __syncthreads(); return s_status == NC_OK; };   // (uniform: every lane reads the same word after the barrier)

    if (ok()) {
        for (int i = lane; i < L; i += 64)
            if (seq[i] < 1 || (t.cross && seq[i] >= t.numph)) fail(NC_PHONE);
    }
    // ---- 1. candidates per position
    if (ok()) {
        for (int i = lane; i <= L; i += 64) {
            int32_t* c = cand + i * kNC;
            int n = 0;
            bool ovf = false;
            auto add = [&](int q) {
                for (int j = 0; j < n; ++j) if (c[j] == q) return;
                if (n < kNC) c[n++] = q; else ovf = true;
            };
            if (i == 0) add(0);
            else {
                const int ph = seq[i - 1];
                if (ph <= t.nph) for (int x = t.pdoff[ph]; x < t.pdoff[ph + 1]; ++x) add(t.pdst[x]);
            }
            for (int j = 0; j < n; ++j) {
                const int q = c[j];
                for (int x = t.f1s[q]; x < t.z1s[q]; ++x) add(t.arcs[4 * (size_t)t.perm[x] + 1]);
            }
            for (int j = 1; j < n; ++j) {   // ascending
                const int v = c[j];
                int k = j - 1;
                while (k >= 0 && c[k] > v) { c[k + 1] = c[k]; --k; }
                c[k + 1] = v;
            }
            ncand[i] = (unsigned char)n;
            if (ovf) fail(NC_CAND);
        }
    }
    // ---- 2. arcs between candidates as bit matrices; an upper bound of the depth inside a position
    if (ok()) {
        for (int i = lane; i <= L; i += 64) {
            const int n = ncand[i], n2 = i < L ? ncand[i + 1] : 0;
            unsigned long long m0 = 0, tb = 0;
            for (int c = 0; c < n; ++c) {
                nc_enum(t, cand[i * kNC + c], i, L, seq, [&](int q2, int adv, int, int, int) {
                    const int32_t* cc = cand + (i + adv) * kNC;
                    const int nn = adv ? n2 : n;
                    for (int c2 = 0; c2 < nn; ++c2)
                        if (cc[c2] == q2) { (adv ? tb : m0) |= 1ull << (c * 8 + c2); break; }
                });
            }
            m0bits[i] = m0;
            tbits[i] = tb;
            // longest chain of same-position arcs (self-loops aside), at most n - 1; n: a cycle (then this many rounds still bound every shortest path)
            int depth[kNC];
#pragma unroll
            for (int c = 0; c < kNC; ++c) depth[c] = 0;
            int mx = 0;
            for (int r = 0; r < n; ++r) {
                bool ch = false;
#pragma unroll
                for (int c = 0; c < kNC; ++c)
#pragma unroll
                    for (int c2 = 0; c2 < kNC; ++c2)
                        if (c != c2 && c < n && c2 < n && ((m0 >> (c * 8 + c2)) & 1) && depth[c2] < depth[c] + 1) { depth[c2] = depth[c] + 1; ch = true; }
                if (!ch) break;
            }
#pragma unroll
            for (int c = 0; c < kNC; ++c) mx = max(mx, depth[c]);
            rounds[i] = (unsigned char)min(mx, n);
        }
    }
    // ---- 3. distance from the start, position by position: lane = (candidate c2) * 8 + (candidate c)
    if (ok()) {
        const int c = lane & 7, c2 = lane >> 3;
        int dist = (c2 == 0) ? 0 : kInf;   // state 0 is the smallest topology state: candidate 0 of position 0
        auto relax = [&](unsigned long long bits, bool same_pos) {
            const int dsrc = __shfl(dist, c * 8 + c2, 64);                  // the distance of candidate c
            int v = (((bits >> (c * 8 + c2)) & 1) && !(same_pos && c == c2) && dsrc < kInf) ? dsrc + 1 : kInf;
            v = min(v, __shfl_xor(v, 1, 64));
            v = min(v, __shfl_xor(v, 2, 64));
            v = min(v, __shfl_xor(v, 4, 64));
            return v;                                                       // min over c, in every lane of candidate c2
        };
        for (int i = 0; i <= L; ++i) {
            if (i > 0) dist = relax(tbits[i - 1], false);
            const unsigned long long m0 = m0bits[i];
            const int nr = rounds[i];
            for (int r = 0; r < nr; ++r) dist = min(dist, relax(m0, true));
            if (c == 0) distv[i * kNC + c2] = (c2 < ncand[i]) ? dist : kInf;
        }
    }
    // ---- 4. reachable states, state ids (position-major), the banded order
    int S = 0;
    if (ok()) {
        for (int i = lane; i <= L; i += 64) {
            unsigned r = 0;
            for (int c = 0; c < ncand[i]; ++c) if (distv[i * kNC + c] < kInf) r |= 1u << c;
            reach[i] = (unsigned char)r;
        }
        __syncthreads();
        int carry = 0;
        for (int i0 = 0; i0 <= L; i0 += 64) {
            const int i = i0 + lane;
            const int n = i <= L ? __popc((unsigned)reach[i]) : 0;
            int tot;
            const int ex = wave_excl_scan(n, lane, &tot);
            if (i <= L) sbase[i] = carry + ex;
            carry += tot;
        }
        if (lane == 0) sbase[L + 1] = carry;
        S = carry + 1;
        if (S > kSMax) fail(NC_STATES);
    }
    bool band_ok = true;
    if (ok()) {
        bool cyc = false;
        for (int i = lane; i <= L; i += 64) {
            const unsigned r = reach[i];
            const unsigned long long m0 = m0bits[i];
            const int n = ncand[i];
            int depth[kNC];
#pragma unroll
            for (int c = 0; c < kNC; ++c) depth[c] = 0;
            bool ch = true;
            for (int rd = 0; rd <= n && ch; ++rd) {
                ch = false;
#pragma unroll
                for (int c = 0; c < kNC; ++c)
#pragma unroll
                    for (int c2 = 0; c2 < kNC; ++c2)
                        if (c != c2 && ((r >> c) & 1) && ((r >> c2) & 1) && ((m0 >> (c * 8 + c2)) & 1) && depth[c2] < depth[c] + 1) { depth[c2] = depth[c] + 1; ch = true; }
            }
            if (ch) cyc = true;
            const int base = sbase[i];
#pragma unroll
            for (int c = 0; c < kNC; ++c) {
                if (!((r >> c) & 1)) continue;
                const int id = base + __popc(r & ((1u << c) - 1));
                int rank = 0;
#pragma unroll
                for (int c2 = 0; c2 < kNC; ++c2)
                    if (((r >> c2) & 1) && (depth[c2] < depth[c] || (depth[c2] == depth[c] && c2 < c))) ++rank;
                st_q[id] = cand[i * kNC + c];
                st_i[id] = (unsigned short)i;
                st_c[id] = (unsigned short)c;
                spos[id] = (unsigned short)(base + rank);
                sord[base + rank] = (unsigned short)id;
            }
        }
        band_ok = !__any(cyc);
    }
    // ---- 5. arcs per state
    int A = 0, NF = 0, NA = 0;
    if (ok()) {
        int ce = 0, ca = 0, cf = 0;
        for (int s0 = 0; s0 < S - 1; s0 += 64) {
            const int id = s0 + lane;
            int ne = 0, nf = 0;
            if (id < S - 1) {
                const int q = st_q[id], i = st_i[id];
                ne = nc_enum(t, q, i, L, seq, [&](int, int, int, int, int) {});
                nf = (i == L) ? t.f1s[q] - t.soff[q] : 0;
                if (nf > 1) fail(NC_FINALS);   // (beta_T of a state with several final arcs is a log-sum: the host's job)
            }
            int te, ta, tf;
            const int xe = wave_excl_scan(ne, lane, &te), xa = wave_excl_scan(ne + nf, lane, &ta), xf = wave_excl_scan(nf, lane, &tf);
            if (id < S - 1) { ebase[id] = ce + xe; abase[id] = ca + xa; fbase[id] = cf + xf; }
            ce += te; ca += ta; cf += tf;
        }
        A = ce; NA = ca; NF = cf;
        if (lane == 0) { ebase[S - 1] = A; ebase[S] = A; abase[S - 1] = NA; fbase[S - 1] = NF; }
        if (A > kAMax || NA > acap) fail(NC_ARCS);
        if (NF > kNFMax) fail(NC_FINALS);
    }
    int max_label = -1, we = 1;
    if (ok()) {
        int32_t* st = p.stage + 5 * (size_t)acap0;
        int ml = -1, w0 = 1;
        for (int id = lane; id < S - 1; id += 64) {
            const int q = st_q[id], i = st_i[id];
            int k = 0;
            const int eb = ebase[id], ab = abase[id];
            nc_enum(t, q, i, L, seq, [&](int q2, int adv, int tok, int wbits, int phv) {
                const int i2 = i + adv;
                const int32_t* cc = cand + i2 * kNC;
                int c2 = 0;
                while (c2 < kNC - 1 && cc[c2] != q2) ++c2;
                const int dst = sbase[i2] + __popc((unsigned)reach[i2] & ((1u << c2) - 1));
                const int j = eb + k, o = ab + k;
                e_src[j] = (unsigned short)id; e_dst[j] = (unsigned short)dst; e_lab[j] = tok; e_w[j] = wbits; e_orig[j] = (unsigned short)o;
                int32_t* r = st + 5 * (size_t)o;
                r[0] = id; r[1] = dst; r[2] = tok; r[3] = wbits; r[4] = phv;
                ml = max(ml, tok);
                if (wbits != 0 && (unsigned)wbits != 0x80000000u) w0 = 0;
                ++k;
            });
            if (i == L) {
                const int nf = t.f1s[q] - t.soff[q];
                for (int f = 0; f < nf; ++f) {
                    const int o = ab + k + f, fi = fbase[id] + f;
                    const int wb = t.arcs[4 * (size_t)t.perm[t.soff[q] + f] + 3];
                    fin_src[fi] = (unsigned short)id; fin_w[fi] = wb; fin_orig[fi] = (unsigned short)o;
                    int32_t* r = st + 5 * (size_t)o;
                    r[0] = id; r[1] = S - 1; r[2] = -1; r[3] = wb; r[4] = -1;
                }
            }
        }
        max_label = wave_max_i32(ml);
        we = wave_min_i32(w0);
    }
    // ---- 6. distinct labels and columns
    int K = 0, direct = 0;
    if (ok()) {
        const int N = pow2_at_least(max(A, 2));
        for (int j = lane; j < N; j += 64) keys[j] = j < A ? (unsigned)e_lab[j] : 0xffffffffu;
        nc_sort(keys, N, lane);
        int k0 = 0;
        for (int j0 = 0; j0 < A; j0 += 64) {
            const int j = j0 + lane;
            const bool first = j < A && (j == 0 || keys[j] != keys[j - 1]);
            const unsigned long long m = __ballot(first);
            if (first) {
                const int at = k0 + __popcll(m & ((1ull << lane) - 1));
                if (at < kKMax) labs[at] = (int32_t)keys[j];
            }
            k0 += __popcll(m);
        }
        direct = (max_label >= 0 && (long long)k0 * 2 >= (long long)max_label + 1) ? 1 : 0;
        K = direct ? max_label + 1 : k0;
        if (K > kKMax || k0 > kKMax) fail(NC_LABELS);
        __syncthreads();
        if (s_status == NC_OK)
            for (int j = lane; j < A; j += 64) {
                const int l = e_lab[j];
                int col = l;
                if (!direct) { int lo = 0, hi = k0; while (lo < hi) { const int mid = (lo + hi) >> 1; if (labs[mid] < l) lo = mid + 1; else hi = mid; } col = lo; }
                e_col[j] = (unsigned short)col;
            }
    }
    // ---- 7. the tables, in prepare_one's order
    if (ok()) {
        size_t cur = start;
        auto take = [&](size_t n) {
            const size_t o = cur;
            cur += wf_pad4(n);
            if (o + n + lane < cur && lane < 3) B_[o + n + lane] = 0;   // the padding words of a table are zeroed
            return (int32_t)o;
        };
        for (int w = lane; w < WF_INFO_WORDS; w += 64) info[w] = 0;
        __syncthreads();
        const float ninf = -INFINITY;
        // incoming CSR: stable sort of the emitting arcs by destination
        {
            const int o_off = take(S + 1), o_src = take(A), o_col = take(A), o_w = take(A), o_arc = take(A);
            const int N = pow2_at_least(max(A, 2));
            for (int j = lane; j < N; j += 64) keys[j] = j < A ? ((unsigned)e_dst[j] << 16 | (unsigned)j) : 0xffffffffu;
            nc_sort(keys, N, lane);
            for (int q = lane; q < A; q += 64) {
                const int j = keys[q] & 0xffff;
                B_[o_src + q] = e_src[j]; B_[o_col + q] = e_col[j]; B_[o_w + q] = e_w[j]; B_[o_arc + q] = e_orig[j];
            }
            const int o_h = take(S);
            int nh = 0, maxin = 0, nover = 0, over = 0;
            for (int s0 = 0; s0 < S + 1; s0 += 64) {
                const int s2 = s0 + lane;
                int lo = 0, deg = 0;
                if (s2 <= S) lo = nc_lower_bound(keys, A, (unsigned)s2 << 16);
                if (s2 <= S) B_[o_off + s2] = lo;
                if (s2 < S) deg = nc_lower_bound(keys, A, (unsigned)(s2 + 1) << 16) - lo;
                const bool hv = s2 < S && deg > WF_HEAVY_DEG;
                const unsigned long long m = __ballot(hv);
                if (hv) B_[o_h + nh + __popcll(m & ((1ull << lane) - 1))] = s2;
                nh += __popcll(m);
                maxin = max(maxin, wave_max_i32(deg));
                const bool ov = s2 < S && deg > WF_REG_DEG;
                nover += __popcll(__ballot(ov));
                over = max(over, wave_max_i32(ov ? s2 : 0));
            }
            for (int x = nh + lane; x < S; x += 64) B_[o_h + x] = 0;
            if (lane == 0) {
                info[WF_I_IN_OFF] = o_off; info[WF_I_IN_SRC] = o_src; info[WF_I_IN_COL] = o_col; info[WF_I_IN_W] = o_w; info[WF_I_IN_ARC] = o_arc;
                info[WF_I_HEAVY_IN] = o_h; info[WF_I_NHEAVY_IN] = nh; info[WF_I_MAXIN] = maxin; info[WF_I_NOVER_IN] = nover; info[WF_I_OVER_IN] = over;
            }
        }
        // outgoing CSR: the emitting arcs are grouped by source already
        {
            const int o_off = take(S + 1), o_dst = take(A), o_col = take(A), o_w = take(A), o_arc = take(A), o_rec = take(4 * (size_t)A);
            for (int s2 = lane; s2 <= S; s2 += 64) B_[o_off + s2] = ebase[s2];
            for (int j = lane; j < A; j += 64) {
                B_[o_dst + j] = e_dst[j]; B_[o_col + j] = e_col[j]; B_[o_w + j] = e_w[j]; B_[o_arc + j] = e_orig[j];
                int32_t* rec = B_ + o_rec + 4 * (size_t)j;
                rec[0] = e_dst[j]; rec[1] = e_col[j]; rec[2] = e_w[j]; rec[3] = e_orig[j];
            }
            const int o_h = take(S);
            int nh = 0, maxout = 0, nover = 0, over = 0;
            for (int s0 = 0; s0 < S; s0 += 64) {
                const int s2 = s0 + lane;
                const int deg = s2 < S ? ebase[s2 + 1] - ebase[s2] : 0;
                const bool hv = s2 < S && deg > WF_HEAVY_DEG;
                const unsigned long long m = __ballot(hv);
                if (hv) B_[o_h + nh + __popcll(m & ((1ull << lane) - 1))] = s2;
                nh += __popcll(m);
                maxout = max(maxout, wave_max_i32(deg));
                const bool ov = s2 < S && deg > WF_REG_DEG;
                nover += __popcll(__ballot(ov));
                over = max(over, wave_max_i32(ov ? s2 : 0));
            }
            for (int x = nh + lane; x < S; x += 64) B_[o_h + x] = 0;
            if (lane == 0) {
                info[WF_I_OUT_OFF] = o_off; info[WF_I_OUT_DST] = o_dst; info[WF_I_OUT_COL] = o_col; info[WF_I_OUT_W] = o_w; info[WF_I_OUT_ARC] = o_arc;
                info[WF_I_OUT_REC] = o_rec;
                info[WF_I_HEAVY_OUT] = o_h; info[WF_I_NHEAVY_OUT] = nh; info[WF_I_MAXOUT] = maxout; info[WF_I_NOVER_OUT] = nover; info[WF_I_OVER_OUT] = over;
            }
        }
        // column-sorted arcs
        {
            const int o_off = take(K + 1), o_src = take(A), o_dst = take(A), o_w = take(A), o_lab = take(K), o_lcol = take(A);
            const int N = pow2_at_least(max(A, 2));
            __syncthreads();
            for (int j = lane; j < N; j += 64) keys[j] = j < A ? ((unsigned)e_col[j] << 16 | (unsigned)j) : 0xffffffffu;
            nc_sort(keys, N, lane);
            for (int q = lane; q < A; q += 64) {
                const int j = keys[q] & 0xffff;
                B_[o_src + q] = e_src[j]; B_[o_dst + q] = e_dst[j]; B_[o_w + q] = e_w[j]; B_[o_lcol + q] = e_col[j];
            }
            int maxdeg = 0;
            for (int k0 = 0; k0 < K + 1; k0 += 64) {
                const int k = k0 + lane;
                int lo = 0, deg = 0;
                if (k <= K) { lo = nc_lower_bound(keys, A, (unsigned)k << 16); B_[o_off + k] = lo; }
                if (k < K) deg = nc_lower_bound(keys, A, (unsigned)(k + 1) << 16) - lo;
                maxdeg = max(maxdeg, wave_max_i32(deg));
            }
            for (int k = lane; k < K; k += 64) B_[o_lab + k] = direct ? k : labs[k];
            if (lane == 0) {
                info[WF_I_LAB_COL] = o_lcol; info[WF_I_LAB_OFF] = o_off; info[WF_I_LAB_SRC] = o_src; info[WF_I_LAB_DST] = o_dst;
                info[WF_I_LAB_W] = o_w; info[WF_I_LABELS] = o_lab; info[WF_I_MAXLAB_DEG] = maxdeg;
            }
        }
        // final arcs, beta_T, the source state of every arc
        {
            const int o_src = take(NF), o_w = take(NF), o_arc = take(NF);
            for (int f = lane; f < NF; f += 64) { B_[o_src + f] = fin_src[f]; B_[o_w + f] = fin_w[f]; B_[o_arc + f] = fin_orig[f]; }
            const int o_fl = take(S), o_fm = take(S), o_as = take((size_t)A + NF);
            for (int s2 = lane; s2 < S; s2 += 64) { B_[o_fl + s2] = (int32_t)kNinfBits; B_[o_fm + s2] = (int32_t)kNinfBits; }
            __syncthreads();
            for (int f = lane; f < NF; f += 64) { B_[o_fm + fin_src[f]] = fin_w[f]; B_[o_fl + fin_src[f]] = fin_w[f]; B_[o_as + fin_orig[f]] = fin_src[f]; }
            for (int j = lane; j < A; j += 64) B_[o_as + e_orig[j]] = e_src[j];
            if (lane == 0) {
                info[WF_I_FIN_SRC] = o_src; info[WF_I_FIN_W] = o_w; info[WF_I_FIN_ARC] = o_arc;
                info[WF_I_FINB_LOG] = o_fl; info[WF_I_FINB_MAX] = o_fm; info[WF_I_ARC_SRC] = o_as;
            }
        }
        // fewest frames of an accepted path: the distance of the nearest state with a final arc
        {
            int ml = 0x7fffffff;
            for (int f = lane; f < NF; f += 64) ml = min(ml, distv[(int)st_i[fin_src[f]] * kNC + st_c[fin_src[f]]]);
            ml = wave_min_i32(ml);
            if (lane == 0) info[WF_I_MINLEN] = ml;
        }
        // banded form
        {
            const int Sn = S - 1;
            int bw = -1, m = 0;
            if (band_ok) {
                int d = 0;
                bool back = false;
                for (int j = lane; j < A; j += 64) {
                    const int dd = (int)spos[e_dst[j]] - (int)spos[e_src[j]];
                    d = max(d, dd);
                    if (dd < 0) back = true;
                }
                bw = wave_max_i32(d);
                if (__any(back)) bw = -1;
            }
            if (bw >= 0 && bw <= WF_BAND_MAX_BW) {
                for (int x = lane; x < Sn * (bw + 1); x += 64) { cellcnt[x] = 0; cellmin[x] = 0xffff; }
                __syncthreads();
                // (byte and halfword LDS atomics do not exist: a cell is counted through the 32-bit word that holds it)
                for (int j = lane; j < A; j += 64) {
                    const int pd = spos[e_dst[j]], d = pd - (int)spos[e_src[j]];
                    const int cell = pd * (bw + 1) + d;
                    atomicAdd((unsigned*)cellcnt + (cell >> 2), 1u << (8 * (cell & 3)));
                    const unsigned sh = 16 * (cell & 1);
                    unsigned* wp = (unsigned*)cellmin + (cell >> 1);
                    unsigned old = *wp;
                    for (;;) {
                        const unsigned curv = (old >> sh) & 0xffff;
                        if (curv <= (unsigned)j) break;
                        const unsigned neww = (old & ~(0xffffu << sh)) | ((unsigned)j << sh);
                        const unsigned seen = atomicCAS(wp, old, neww);
                        if (seen == old) break;
                        old = seen;
                    }
                }
                __syncthreads();
                int mm = 0;
                for (int x = lane; x < Sn * (bw + 1); x += 64) mm = max(mm, (int)cellcnt[x]);
                m = wave_max_i32(mm);
            }
            if (bw >= 0 && bw <= WF_BAND_MAX_BW && m <= WF_BAND_MAX_M) {
                m = max(m, 1);
                const size_t cells = (size_t)Sn * (bw + 1) * m;
                const int o_ord = take(Sn), o_f = take(cells * 2), o_b = take(cells * 2);
                for (int x = lane; x < Sn; x += 64) B_[o_ord + x] = sord[x];
                for (size_t x = lane; x < cells; x += 64) { B_[o_f + 2 * x] = 0; B_[o_f + 2 * x + 1] = (int32_t)kNinfBits; B_[o_b + 2 * x] = 0; B_[o_b + 2 * x + 1] = (int32_t)kNinfBits; }
                __syncthreads();
                for (int j = lane; j < A; j += 64) {
                    const int ps = spos[e_src[j]], pd = spos[e_dst[j]], d = pd - ps;
                    const int cell = pd * (bw + 1) + d;
                    const int slot = (cellmin[cell] == (unsigned short)j) ? 0 : 1;
                    const size_t ef = (size_t)cell * m + slot;
                    B_[o_f + 2 * ef] = e_col[j]; B_[o_f + 2 * ef + 1] = e_w[j];
                    const size_t eb = ((size_t)(Sn - 1 - ps) * (bw + 1) + d) * m + slot;
                    B_[o_b + 2 * eb] = e_col[j]; B_[o_b + 2 * eb + 1] = e_w[j];
                }
                if (lane == 0) { info[WF_I_BAND_BW] = bw; info[WF_I_BAND_M] = m; info[WF_I_BAND_ORDER] = o_ord; info[WF_I_BANDF] = o_f; info[WF_I_BANDB] = o_b; }
            } else if (lane == 0) {
                info[WF_I_BAND_BW] = -1; info[WF_I_BAND_M] = 0;
            }
            if (lane == 0) info[WF_I_BAND_WE] = we;
        }
        if (lane == 0) {
            info[WF_I_S] = S; info[WF_I_A] = A; info[WF_I_NF] = NF; info[WF_I_K] = K; info[WF_I_DIRECT] = direct;
            info[WF_I_MAX_LABEL] = max_label; info[WF_I_HUB_G] = 0;
        }
        if (cur > limit) fail(NC_SPACE);
        __syncthreads();
        for (int w = lane; w < WF_INFO_WORDS; w += 64) I[w] = info[w];   // (WF_I_ARC_BASE: nc_finish_kernel)
    }
    __syncthreads();
    const int status = s_status;
    if (status != NC_OK) {
        for (int w = lane; w < WF_INFO_WORDS; w += 64) I[w] = 0;
        S = 0; NA = 0; max_label = -1;
    }
    if (lane == 0) {
        p.counts[b] = NA; p.counts[p.B + b] = S; p.counts[2 * p.B + b] = status; p.counts[3 * p.B + b] = max_label;
    }
}

// Second launch: arc offsets (a prefix sum over the utterances), the arcs from their staging place to the FsaVec arrays,
// WF_I_ARC_BASE, the blob header.  grid = B.
__global__ void __launch_bounds__(64) nc_finish_kernel(NcArgs p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const NcTopo t = nc_topo(p.topo);
    long long off = 0, acap0 = 0, words = 0;
    int bad = 0, gmaxS = 0, gmaxA = 0, gmaxlab = -1;
    for (int u = lane; u < p.B; u += 64) {
        const int Lu = p.phones_off[u + 1] - p.phones_off[u];
        const int Lc = min(max(Lu, 0), kNP - 1);
        if (u < b) { off += p.counts[u]; acap0 += wf_nc_arc_cap(Lc, t.maxd); }
        words += (long long)wf_fsa_words_bound((size_t)wf_nc_state_cap(Lc), (size_t)wf_nc_arc_cap(Lc, t.maxd));
        bad |= p.counts[2 * p.B + u] != 0;
        gmaxS = max(gmaxS, p.counts[p.B + u]);
        gmaxlab = max(gmaxlab, p.counts[3 * p.B + u]);
        gmaxA = max(gmaxA, p.blob[WF_HDR_WORDS + (size_t)WF_INFO_WORDS * u + WF_I_A]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); acap0 += __shfl_xor(acap0, o, 64); words += __shfl_xor(words, o, 64); }
    bad = __any(bad);
    gmaxS = wave_max_i32(gmaxS); gmaxA = wave_max_i32(gmaxA); gmaxlab = wave_max_i32(gmaxlab);
    const int n = p.counts[b];
    if (lane == 0) {
        p.blob[WF_HDR_WORDS + (size_t)WF_INFO_WORDS * b + WF_I_ARC_BASE] = (int32_t)off;
        p.meta[b] = (int32_t)off;
        if (b == p.B - 1) p.meta[p.B] = (int32_t)(off + n);
        p.meta[p.B + 1 + b] = p.counts[p.B + b];
        p.meta[2 * p.B + 1 + b] = p.counts[2 * p.B + b];
    }
    if (b == 0 && lane < WF_HDR_WORDS) {
        int32_t v = 0;
        if (lane == WF_H_MAGIC) v = bad ? 0 : WF_BLOB_MAGIC;   // a batch with a failed utterance is not a usable graph vector
        if (lane == WF_H_NUM_FSAS) v = p.B;
        if (lane == WF_H_WORDS) v = (int32_t)(wf_pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * p.B) + words);
        if (lane == WF_H_MAX_LABEL) v = gmaxlab;
        if (lane == WF_H_MAX_STATES) v = gmaxS;
        if (lane == WF_H_MAX_ARCS) v = gmaxA;
        p.blob[lane] = v;
    }
    const int32_t* st = p.stage + 5 * (size_t)acap0;
    for (int a = lane; a < n; a += 64) {
        const int32_t* r = st + 5 * (size_t)a;
        int32_t* o = p.arcs + 4 * (size_t)(off + a);
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
        p.aux[off + a] = r[4];
    }
}

int nc_sizes(const int32_t* topo, const int32_t* phones_off, int B, size_t* blob_words, long long* arc_cap) {
    if (!topo || topo[WF_T_MAGIC] != WF_TOPO_MAGIC) return wf_fail(WF_EINVAL, "device compiler: bad topology table");
    if (!phones_off || B < 1) return wf_fail(WF_EINVAL, "device compiler: bad B / phones_off");
    size_t w = wf_pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * B);
    long long a = 0;
    for (int b = 0; b < B; ++b) {
        const int L = phones_off[b + 1] - phones_off[b];
        if (L < 0) return wf_fail(WF_EINVAL, "device compiler: transcript %d has a negative length", b);
        const int Lc = L < kNP - 1 ? L : kNP - 1;
        w += wf_fsa_words_bound((size_t)wf_nc_state_cap(Lc), (size_t)wf_nc_arc_cap(Lc, topo[WF_T_MAXD]));
        a += wf_nc_arc_cap(Lc, topo[WF_T_MAXD]);
    }
    if (w >= (size_t)1 << 31) return wf_fail(WF_EUNSUPPORTED, "device compiler: prepared graph exceeds 2^31 words");
    *blob_words = w;
    *arc_cap = a;
    return WF_OK;
}

}  // namespace

extern "C" int wf_numerator_compile_device_bytes(const int32_t* topo_tab, const int32_t* phones_off, int32_t B,
                                                 size_t* blob_bytes, int64_t* arc_capacity, size_t* work_bytes) {
    if (!blob_bytes || !arc_capacity || !work_bytes) return wf_fail(WF_EINVAL, "wf_numerator_compile_device_bytes: null pointer");
    size_t w = 0;
    long long a = 0;
    int rc = nc_sizes(topo_tab, phones_off, B, &w, &a);
    if (rc) return rc;
    *blob_bytes = w * 4;
    *arc_capacity = a;
    *work_bytes = sizeof(int32_t) * (5 * (size_t)a + 4 * (size_t)B) + 256;
    return WF_OK;
}

extern "C" int wf_numerator_compile_device(const int32_t* topo_tab, const void* topo_tab_dev, const int32_t* phones_dev,
                                           const int32_t* phones_off_dev, const int32_t* phones_off, int32_t B,
                                           void* blob_dev, size_t blob_bytes, int32_t* arcs_dev, int32_t* aux_dev,
                                           int64_t arc_capacity, int32_t* meta_dev, void* work_dev, size_t work_bytes,
                                           void* stream) {
    if (!topo_tab_dev || !phones_off_dev || !blob_dev || !arcs_dev || !aux_dev || !meta_dev || !work_dev)
        return wf_fail(WF_EINVAL, "wf_numerator_compile_device: null pointer");
    size_t w = 0;
    long long a = 0;
    int rc = nc_sizes(topo_tab, phones_off, B, &w, &a);
    if (rc) return rc;
    if (!phones_dev && phones_off[B] > 0) return wf_fail(WF_EINVAL, "wf_numerator_compile_device: null transcripts");
    if (blob_bytes < w * 4 || arc_capacity < a || work_bytes < sizeof(int32_t) * (5 * (size_t)a + 4 * (size_t)B) + 256)
        return wf_fail(WF_ENOSPC, "wf_numerator_compile_device: needs a %zu B blob, %lld arcs, %zu B of work space", w * 4, a,
                       sizeof(int32_t) * (5 * (size_t)a + 4 * (size_t)B) + 256);
    NcArgs p{};
    p.topo = (const int32_t*)topo_tab_dev; p.phones = phones_dev; p.phones_off = phones_off_dev; p.B = B;
    p.blob = (int32_t*)blob_dev; p.blob_words = (int64_t)(blob_bytes / 4);
    int32_t* work = (int32_t*)(((uintptr_t)work_dev + 255) & ~(uintptr_t)255);
    p.stage = work; p.counts = work + 5 * (size_t)a;
    p.arcs = arcs_dev; p.aux = aux_dev; p.meta = meta_dev;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(numerator_compile_kernel, dim3(B), dim3(64), 0, st, p);
    hipLaunchKernelGGL(nc_finish_kernel, dim3(B), dim3(64), 0, st, p);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}
