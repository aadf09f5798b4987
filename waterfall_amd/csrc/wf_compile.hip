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
//   4. the distinct labels are a bitmap and its prefix popcounts; the incoming-arc CSR is gathered by the destination
//      state (its sources sit at its own position and the one before); the column-sorted arcs are a stable counting sort
//      (lanes with equal columns find each other with ballots); the banded order is "by position, inside a position by
//      depth, then by state" (prepare_one's pos_hint order).
// Limits (WF_NC_*): <= 256 positions, <= 8 live topology states per position, <= 257 states, <= 1024 arcs; anything else
// sets the utterance's status and the caller compiles that batch on the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

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
constexpr int kMaxDevices = 64;
constexpr int kBmWords = 2048;     // labels below 65536 take the bitmap route, anything above the sort
constexpr int kNcDynWords = (kSMax + 1) + 2 * (kKMax + 1) + 2 * kBmWords + kNP * kNC / 4;
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
    int prof;                   // measurement aid (WF_NC_PROFILE=1): workgroup 0 prints the cycles of its phases
    // wf_numerator_compile_job: one launch.  The transcripts are read from (and the words the host plans with are written to)
    // pinned host memory; the workgroup that finishes last does nc_finish_kernel's job for the batch, arc compaction aside.
    int32_t* out_host;          // [meta_words] arc offsets [B + 1] | states [B] | status [B] | staging offsets [B] | graphs with direct columns, then the blob's header + info words
    int meta_words;
    unsigned* done;             // device counter (zero between jobs): workgroups that have finished
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

template <int CTRL>
__device__ __forceinline__ int nc_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
// min over each aligned group of 8 lanes, in every lane of the group (every lane of the wave active)
__device__ __forceinline__ int nc_min8(int v) {
    v = min(v, nc_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = min(v, nc_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = min(v, nc_dpp<0x141>(v));   // row_half_mirror: lane k of 8 <- lane 7 - k (the other quad)
    return v;
}
// Longest same-position chains (self-loops aside) among the candidates in `allow`, by at most `max_rounds` sweeps in
// (c, c2) ascending order: depth of candidate c in nibble c of the result; *changed = the last sweep still moved something.
__device__ __forceinline__ unsigned nc_depths(unsigned long long m0, unsigned allow, int max_rounds, bool* changed) {
    unsigned D = 0;
    bool ch = true;
    for (int rd = 0; rd < max_rounds && ch; ++rd) {
        ch = false;
        unsigned rows = allow;
        while (rows) {
            const int c = __ffs(rows) - 1;
            rows &= rows - 1;
            unsigned row = (unsigned)(m0 >> (c * 8)) & 0xffu & allow & ~(1u << c);
            const unsigned dc = (D >> (4 * c)) & 15u;
            while (row) {
                const int c2 = __ffs(row) - 1;
                row &= row - 1;
                if (((D >> (4 * c2)) & 15u) < dc + 1) { D = (D & ~(15u << (4 * c2))) | ((dc + 1) << (4 * c2)); ch = true; }
            }
        }
    }
    *changed = ch;
    return D;
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
    __shared__ int32_t s_seq[kNP];                       // the transcript (which may sit in host memory: read once)
    extern __shared__ int32_t nc_dyn[];                  // kNcDynWords
    int32_t* indeg = nc_dyn;                             // [kSMax + 1] in-degree, then the fill cursor of the incoming CSR
    int32_t* colcnt = indeg + kSMax + 1;                 // [kKMax + 1] arcs per column
    int32_t* colcur = colcnt + kKMax + 1;                // [kKMax + 1] fill cursor of the column-sorted arcs
    unsigned* bm = (unsigned*)(colcur + kKMax + 1);      // [kBmWords] the labels in use, one bit each
    int32_t* pre = (int32_t*)(bm + kBmWords);            // [kBmWords] labels in use below word w
    unsigned char* necnt = (unsigned char*)(pre + kBmWords);   // [kNP * kNC] emitting arcs of candidate c at position i

    const int b = blockIdx.x, lane = threadIdx.x;
    unsigned long long tp[12];
    int ntp = 0;
#define NC_MARK() do { if (p.prof) tp[ntp++] = __builtin_amdgcn_s_memtime(); } while (0)
    NC_MARK();
    const NcTopo t = nc_topo(p.topo);
    const int off0 = p.phones_off[b], L = p.phones_off[b + 1] - off0;
    for (int i = lane; i < min(L, kNP); i += 64) s_seq[i] = p.phones[off0 + i];
    const int32_t* seq = s_seq;
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
    NC_MARK();
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
    NC_MARK();
    if (ok()) {
        for (int i = lane; i <= L; i += 64) {
            const int n = ncand[i], n2 = i < L ? ncand[i + 1] : 0;
            unsigned long long m0 = 0, tb = 0;
            for (int c = 0; c < n; ++c) {
                const int ne = nc_enum(t, cand[i * kNC + c], i, L, seq, [&](int q2, int adv, int, int, int) {
                    const int32_t* cc = cand + (i + adv) * kNC;
                    const int nn = adv ? n2 : n;
                    for (int c2 = 0; c2 < nn; ++c2)
                        if (cc[c2] == q2) { (adv ? tb : m0) |= 1ull << (c * 8 + c2); break; }
                });
                necnt[i * kNC + c] = (unsigned char)min(ne, 255);
            }
            m0bits[i] = m0;
            tbits[i] = tb;
            // longest chain of same-position arcs (self-loops aside), at most n - 1; n: a cycle (then this many rounds still bound every shortest path)
            bool ch;
            const unsigned D = nc_depths(m0, (1u << n) - 1, n, &ch);
            int mx = 0;
            for (int c = 0; c < n; ++c) mx = max(mx, (int)((D >> (4 * c)) & 15u));
            rounds[i] = (unsigned char)min(mx, n);
        }
    }
    // ---- 3. distance from the start, position by position: lane = (candidate c2) * 8 + (candidate c)
    NC_MARK();
    if (ok()) {
        const int c = lane & 7, c2 = lane >> 3, tl = c * 8 + c2;
        int dist = (c2 == 0) ? 0 : kInf;   // state 0 is the smallest topology state: candidate 0 of position 0
        auto relax = [&](unsigned long long bits, bool same_pos) {
            const int dsrc = __builtin_amdgcn_ds_bpermute(tl << 2, dist);   // the distance of candidate c
            const int v = (((bits >> tl) & 1) && !(same_pos && c == c2) && dsrc < kInf) ? dsrc + 1 : kInf;
            return nc_min8(v);                                              // min over c, in every lane of candidate c2
        };
        unsigned long long m0 = m0bits[0];
        int nr = rounds[0], nc = ncand[0];
        for (int i = 0; i <= L; ++i) {
            // (the next position's words are on their way while this one's rounds run)
            const int i1 = min(i + 1, L);
            const unsigned long long tb = tbits[i], m0n = m0bits[i1];
            const int nrn = rounds[i1], ncn = ncand[i1];
            for (int r = 0; r < nr; ++r) dist = min(dist, relax(m0, true));
            if (c == 0) distv[i * kNC + c2] = (c2 < nc) ? dist : kInf;
            if (i < L) dist = relax(tb, false);
            m0 = m0n; nr = nrn; nc = ncn;
        }
    }
    // ---- 4. reachable states, state ids (position-major), the banded order
    NC_MARK();
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
            bool ch;
            const unsigned D = nc_depths(m0, r, n + 1, &ch);
            if (ch) cyc = true;
            const int base = sbase[i];
            for (unsigned rr = r; rr; rr &= rr - 1) {
                const int c = __ffs(rr) - 1;
                const int id = base + __popc(r & ((1u << c) - 1));
                const unsigned dc = (D >> (4 * c)) & 15u;
                int rank = 0;
                for (unsigned r2 = r; r2; r2 &= r2 - 1) {
                    const int c2 = __ffs(r2) - 1;
                    const unsigned d2 = (D >> (4 * c2)) & 15u;
                    if (d2 < dc || (d2 == dc && c2 < c)) ++rank;
                }
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
    NC_MARK();
    int A = 0, NF = 0, NA = 0;
    if (ok()) {
        int ce = 0, ca = 0, cf = 0;
        for (int s0 = 0; s0 < S - 1; s0 += 64) {
            const int id = s0 + lane;
            int ne = 0, nf = 0;
            if (id < S - 1) {
                const int q = st_q[id], i = st_i[id];
                ne = necnt[i * kNC + st_c[id]];
                if (ne == 255) fail(NC_ARCS);
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
        for (int s2 = lane; s2 <= S; s2 += 64) indeg[s2] = 0;
    }
    int max_label = -1, we = 1;
    NC_MARK();
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
                atomicAdd(&indeg[dst], 1);
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
    NC_MARK();
    int K = 0, direct = 0;
    if (ok()) {
        const bool bitmap = max_label < 32 * kBmWords;
        const int W = (max_label >> 5) + 1;   // (max_label = -1, no arcs: no words)
        int k0 = 0;
        if (bitmap) {
            for (int w = lane; w < W; w += 64) bm[w] = 0;
            __syncthreads();
            for (int j = lane; j < A; j += 64) atomicOr(&bm[e_lab[j] >> 5], 1u << (e_lab[j] & 31));
            __syncthreads();
            for (int w0 = 0; w0 < W; w0 += 64) {
                const int w = w0 + lane;
                int tot;
                const int ex = wave_excl_scan(w < W ? __popc(bm[w]) : 0, lane, &tot);
                if (w < W) pre[w] = k0 + ex;
                k0 += tot;
            }
        } else {
            const int N = pow2_at_least(max(A, 2));
            for (int j = lane; j < N; j += 64) keys[j] = j < A ? (unsigned)e_lab[j] : 0xffffffffu;
            nc_sort(keys, N, lane);
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
        }
        direct = (max_label >= 0 && (long long)k0 * 2 >= (long long)max_label + 1) ? 1 : 0;
        K = direct ? max_label + 1 : k0;
        if (K > kKMax || k0 > kKMax) fail(NC_LABELS);
        __syncthreads();
        if (s_status == NC_OK) {
            for (int k = lane; k <= K; k += 64) colcnt[k] = 0;
            if (bitmap && !direct)
                for (int w = lane; w < W; w += 64) {
                    int at = pre[w];
                    for (unsigned bits = bm[w]; bits; bits &= bits - 1) labs[at++] = w * 32 + __ffs(bits) - 1;
                }
            __syncthreads();
            for (int j = lane; j < A; j += 64) {
                const int l = e_lab[j];
                int col = l;
                if (!direct) {
                    if (bitmap) col = pre[l >> 5] + __popc(bm[l >> 5] & ((1u << (l & 31)) - 1));
                    else { int lo = 0, hi = k0; while (lo < hi) { const int mid = (lo + hi) >> 1; if (labs[mid] < l) lo = mid + 1; else hi = mid; } col = lo; }
                }
                e_col[j] = (unsigned short)col;
                atomicAdd(&colcnt[col], 1);
            }
        }
    }
    // ---- 7. the tables, in prepare_one's order
    NC_MARK();
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
            const int o_h = take(S);
            int nh = 0, maxin = 0, nover = 0, over = 0, carry = 0;
            for (int s0 = 0; s0 < S + 1; s0 += 64) {
                const int s2 = s0 + lane;
                const int deg = s2 < S ? indeg[s2] : 0;
                int tot;
                const int lo = carry + wave_excl_scan(deg, lane, &tot);
                carry += tot;
                if (s2 <= S) { B_[o_off + s2] = lo; indeg[s2] = lo; }
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
            __syncthreads();
            // the arcs into state d leave the states of d's position and of the one before it; walking those in state
            // order visits the arcs in arc order: the stable sort by destination, without the sort
            for (int d = lane; d < S - 1; d += 64) {
                const int i2 = st_i[d];
                const int lo = sbase[max(i2 - 1, 0)], hi = sbase[i2 + 1];
                int q = indeg[d];
                for (int j = ebase[lo]; j < ebase[hi]; ++j)
                    if (e_dst[j] == d) {
                        B_[o_src + q] = e_src[j]; B_[o_col + q] = e_col[j]; B_[o_w + q] = e_w[j]; B_[o_arc + q] = e_orig[j];
                        ++q;
                    }
            }
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
            __syncthreads();
            int maxdeg = 0, carry = 0;
            for (int k0 = 0; k0 < K + 1; k0 += 64) {
                const int k = k0 + lane;
                const int deg = k < K ? colcnt[k] : 0;
                int tot;
                const int lo = carry + wave_excl_scan(deg, lane, &tot);
                carry += tot;
                if (k <= K) { B_[o_off + k] = lo; colcur[k] = lo; }
                maxdeg = max(maxdeg, wave_max_i32(deg));
            }
            __syncthreads();
            // stable counting sort by column, 64 arcs at a time: the lanes that hold one column find each other with a
            // ballot per bit of the column, take consecutive places in lane order, and the last of them moves the cursor
            const int nbits = 32 - __clz(max(K - 1, 1));
            for (int j0 = 0; j0 < A; j0 += 64) {
                const int j = j0 + lane;
                const bool valid = j < A;
                const int col = valid ? e_col[j] : 0;
                unsigned long long peers = __ballot(valid);
                for (int bit = 0; bit < nbits; ++bit) {
                    const bool one = (col >> bit) & 1;
                    const unsigned long long bb = __ballot(valid && one);
                    peers &= one ? bb : ~bb;
                }
                if (valid) {
                    const int rank = __popcll(peers & ((1ull << lane) - 1)), cnt = __popcll(peers);
                    const int base = colcur[col], q = base + rank;
                    B_[o_src + q] = e_src[j]; B_[o_dst + q] = e_dst[j]; B_[o_w + q] = e_w[j]; B_[o_lcol + q] = col;
                    if (rank == cnt - 1) colcur[col] = base + cnt;
                }
                __syncthreads();
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
    int32_t* IH = p.out_host ? p.out_host + p.meta_words + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * b : nullptr;
    if (IH && s_status == NC_OK)
        for (int w = lane; w < WF_INFO_WORDS; w += 64) IH[w] = info[w];
    __syncthreads();
    NC_MARK();
    if (p.prof && b == 0 && lane == 0) {
        printf("numerator_compile_kernel wg 0 (L=%d S=%d A=%d): cycles setup %llu | candidates %llu | bit matrices %llu | distances %llu | ids+order %llu | arc counts %llu | arcs %llu | labels %llu | tables %llu\n",
               L, S, A, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], tp[6] - tp[5], tp[7] - tp[6], tp[8] - tp[7], tp[9] - tp[8]);
    }
    const int status = s_status;
    if (status != NC_OK) {
        for (int w = lane; w < WF_INFO_WORDS; w += 64) { I[w] = 0; if (IH) IH[w] = 0; }
        S = 0; NA = 0; max_label = -1;
    }
    if (lane == 0) {
        p.counts[b] = NA; p.counts[p.B + b] = S; p.counts[2 * p.B + b] = status; p.counts[3 * p.B + b] = max_label;
    }
    if (!p.done) return;
    // ---- the last workgroup to get here finishes the batch (every workgroup arrives: no one waits for another)
    // (the hand-off between workgroups inside a launch: this workgroup is one wave; it releases what it stored at agent scope
    // and waits for that before it takes its ticket; the last arriver acquires before it loads the others' results)
    __shared__ unsigned s_ticket;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) s_ticket = atomicAdd(p.done, 1u);
    __syncthreads();
    if (s_ticket != (unsigned)p.B - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) *p.done = 0;
    const int Bn = p.B, M = p.meta_words;
    int carry = 0, acarry = 0, bad = 0, gmaxS = 0, gmaxA = 0, gmaxlab = -1, ndirect = 0;
    long long words = 0;
    for (int u0 = 0; u0 < Bn; u0 += 64) {
        const int u = u0 + lane;
        const bool valid = u < Bn;
        int n = 0, cap = 0;
        if (valid) {
            const int Lu = p.phones_off[u + 1] - p.phones_off[u];
            const int Lcu = min(max(Lu, 0), kNP - 1);
            n = p.counts[u];
            cap = wf_nc_arc_cap(Lcu, t.maxd);
            words += (long long)wf_fsa_words_bound((size_t)wf_nc_state_cap(Lcu), (size_t)cap);
            bad |= p.counts[2 * Bn + u] != 0;
            gmaxS = max(gmaxS, p.counts[Bn + u]);
            gmaxlab = max(gmaxlab, p.counts[3 * Bn + u]);
            gmaxA = max(gmaxA, p.blob[WF_HDR_WORDS + (size_t)WF_INFO_WORDS * u + WF_I_A]);
            ndirect += p.blob[WF_HDR_WORDS + (size_t)WF_INFO_WORDS * u + WF_I_DIRECT] != 0;
        }
        int tot, atot;
        const int ex = carry + wave_excl_scan(n, lane, &tot), aex = acarry + wave_excl_scan(cap, lane, &atot);
        carry += tot; acarry += atot;
        if (valid) {
            p.blob[WF_HDR_WORDS + (size_t)WF_INFO_WORDS * u + WF_I_ARC_BASE] = ex;
            p.out_host[M + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * u + WF_I_ARC_BASE] = ex;
            p.out_host[u] = ex;
            p.out_host[Bn + 1 + u] = p.counts[Bn + u];
            p.out_host[2 * Bn + 1 + u] = p.counts[2 * Bn + u];
            p.out_host[3 * Bn + 1 + u] = aex;
        }
    }
    ndirect = wave_sum_i32(ndirect);
    if (lane == 0) { p.out_host[Bn] = carry; p.out_host[4 * Bn + 1] = ndirect; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) words += __shfl_xor(words, o, 64);
    bad = __any(bad);
    gmaxS = wave_max_i32(gmaxS); gmaxA = wave_max_i32(gmaxA); gmaxlab = wave_max_i32(gmaxlab);
    if (lane < WF_HDR_WORDS) {
        int32_t v = 0;
        if (lane == WF_H_MAGIC) v = bad ? 0 : WF_BLOB_MAGIC;
        if (lane == WF_H_NUM_FSAS) v = Bn;
        if (lane == WF_H_WORDS) v = (int32_t)(wf_pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * Bn) + words);
        if (lane == WF_H_MAX_LABEL) v = gmaxlab;
        if (lane == WF_H_MAX_STATES) v = gmaxS;
        if (lane == WF_H_MAX_ARCS) v = gmaxA;
        p.blob[lane] = v;
        p.out_host[M + lane] = v;
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

static __device__ unsigned g_nc_done[64];   // per job slot: workgroups done (zero between jobs)
unsigned* g_nc_done_ptr[kMaxDevices];

// p: everything but prof, stage, counts, done; a: the batch's arc capacity; done_slot < 0: two launches (nc_finish_kernel
// compacts the arcs), else one launch whose last workgroup closes the batch through g_nc_done[done_slot]
int nc_launch(NcArgs p, long long a, void* work_dev, int done_slot, hipStream_t st) {
    int32_t* work = (int32_t*)(((uintptr_t)work_dev + 255) & ~(uintptr_t)255);
    p.stage = work; p.counts = work + 5 * (size_t)a;
    static const int prof = getenv("WF_NC_PROFILE") ? atoi(getenv("WF_NC_PROFILE")) : 0;
    p.prof = prof;
    static std::once_flag lds_once[kMaxDevices];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return wf_fail(WF_EUNSUPPORTED, "device compiler: device %d", dev);
    hipError_t lds_rc = hipSuccess;
    std::call_once(lds_once[dev], [&] {
        lds_rc = hipFuncSetAttribute((const void*)numerator_compile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(kNcDynWords * sizeof(int32_t)));
        if (lds_rc == hipSuccess) lds_rc = hipGetSymbolAddress((void**)&g_nc_done_ptr[dev], HIP_SYMBOL(g_nc_done));
    });
    HIP_TRY(lds_rc);
    if (!g_nc_done_ptr[dev]) return wf_fail(WF_EHIP, "device compiler: set-up failed earlier on device %d", dev);
    p.done = done_slot >= 0 ? g_nc_done_ptr[dev] + done_slot : nullptr;
    hipLaunchKernelGGL(numerator_compile_kernel, dim3(p.B), dim3(64), kNcDynWords * sizeof(int32_t), st, p);
    if (done_slot < 0) hipLaunchKernelGGL(nc_finish_kernel, dim3(p.B), dim3(64), 0, st, p);
    HIP_TRY(hipGetLastError());
    return WF_OK;
}

// One event per job in flight: a ring per device; a slot is reused only once its previous job has completed.
constexpr int kJobRing = 64;
struct JobRing {
    std::mutex mu;
    hipEvent_t ev[kJobRing] = {};
    hipEvent_t main_ev = nullptr;
    int64_t seq_of[kJobRing] = {};
    hipStream_t cs_of[kJobRing] = {};
    hipEvent_t rel_ev[kJobRing] = {};      // wf_compile_job_release: the streams that read the job's graphs are past them
    int64_t rel_seq[kJobRing] = {};
    hipEvent_t tmp_ev = nullptr;
    int64_t next = 1;
};
JobRing g_jobs[kMaxDevices];
inline size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

struct JobLayout { size_t meta_words, blob_bytes, work_off, total, hdr_words; long long arc_cap; };
int job_layout(const int32_t* topo_tab, const int32_t* phones_off, int B, JobLayout* L) {
    size_t w = 0;
    long long a = 0;
    if (phones_off) {
        int rc = nc_sizes(topo_tab, phones_off, B, &w, &a);
        if (rc) return rc;
        if (phones_off[0] != 0) return wf_fail(WF_EINVAL, "device compiler: phones_off[0] must be 0");
    } else {   // the most any batch of B transcripts can take (the capacities are clamped at the kernel's limits)
        if (!topo_tab || topo_tab[WF_T_MAGIC] != WF_TOPO_MAGIC) return wf_fail(WF_EINVAL, "device compiler: bad topology table");
        if (B < 1) return wf_fail(WF_EINVAL, "device compiler: bad B");
        const int Lc = kNP - 1;
        a = (long long)B * wf_nc_arc_cap(Lc, topo_tab[WF_T_MAXD]);
        w = wf_pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * B) +
            (size_t)B * wf_fsa_words_bound((size_t)wf_nc_state_cap(Lc), (size_t)wf_nc_arc_cap(Lc, topo_tab[WF_T_MAXD]));
        if (w >= (size_t)1 << 31) return wf_fail(WF_EUNSUPPORTED, "device compiler: prepared graph exceeds 2^31 words");
    }
    L->meta_words = (4 * (size_t)B + 2 + 63) / 64 * 64;
    L->blob_bytes = w * 4;
    L->work_off = up256(L->blob_bytes);
    L->total = L->work_off + sizeof(int32_t) * (5 * (size_t)a + 4 * (size_t)B) + 256;
    L->hdr_words = wf_pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * B);
    L->arc_cap = a;
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
    p.arcs = arcs_dev; p.aux = aux_dev; p.meta = meta_dev;
    return nc_launch(p, a, work_dev, -1, (hipStream_t)stream);
}

extern "C" int wf_numerator_compile_job_bytes(const int32_t* topo_tab, const int32_t* phones_off, int32_t B, size_t* dev_bytes,
                                              size_t* out_words, int64_t* layout) {
    if (!dev_bytes || !out_words || !layout) return wf_fail(WF_EINVAL, "wf_numerator_compile_job_bytes: null pointer");
    JobLayout L;
    int rc = job_layout(topo_tab, phones_off, B, &L);
    if (rc) return rc;
    *dev_bytes = L.total;
    *out_words = L.meta_words + L.hdr_words;
    layout[0] = (int64_t)L.meta_words; layout[1] = (int64_t)L.hdr_words; layout[2] = (int64_t)L.blob_bytes;
    layout[3] = (int64_t)L.work_off; layout[4] = (int64_t)L.arc_cap;
    return WF_OK;
}

extern "C" int wf_numerator_compile_job(const int32_t* topo_tab, const void* topo_tab_dev, const int32_t* in_host, int32_t B,
                                        void* dev_buf, size_t dev_bytes, int32_t* out_host, size_t out_words, int32_t device,
                                        void* main_stream, void* compile_stream, int64_t reuse_of, int64_t* job,
                                        int64_t* layout) {
    if (!in_host || !topo_tab_dev || !dev_buf || !out_host || !job || !compile_stream)
        return wf_fail(WF_EINVAL, "wf_numerator_compile_job: null pointer");
    if (device < 0 || device >= kMaxDevices) return wf_fail(WF_EUNSUPPORTED, "wf_numerator_compile_job: device %d", device);
    JobLayout L;
    int rc = job_layout(topo_tab, in_host, B, &L);
    if (rc) return rc;
    if (dev_bytes < L.total || out_words < L.meta_words + L.hdr_words)
        return wf_fail(WF_ENOSPC, "wf_numerator_compile_job: needs %zu B on the device and %zu words on the host", L.total,
                       L.meta_words + L.hdr_words);
    for (int32_t i = 0, n = in_host[B]; i < n; ++i)
        if (in_host[B + 1 + i] < 1) return wf_fail(WF_EINVAL, "wf_numerator_compile_job: labels must be >= 1 (0 is epsilon)");
    if (layout) {
        layout[0] = (int64_t)L.meta_words; layout[1] = (int64_t)L.hdr_words; layout[2] = (int64_t)L.blob_bytes;
        layout[3] = (int64_t)L.work_off; layout[4] = (int64_t)L.arc_cap;
    }
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device) HIP_TRY(hipSetDevice(device));
    struct Restore { int cur, dev; ~Restore() { if (cur != dev) (void)hipSetDevice(cur); } } restore{cur, device};
    JobRing& R = g_jobs[device];
    std::lock_guard<std::mutex> lock(R.mu);
    const int64_t seq = R.next;
    const int slot = (int)(seq % kJobRing);
    if (!R.ev[slot]) HIP_TRY(hipEventCreateWithFlags(&R.ev[slot], hipEventDisableTiming));
    else if (R.seq_of[slot]) HIP_TRY(hipEventSynchronize(R.ev[slot]));   // (its counter g_nc_done[slot] is back at zero)
    hipStream_t cs = (hipStream_t)compile_stream, ms = (hipStream_t)main_stream;
    if (ms && ms != cs) {
        // dev_buf comes from an allocator that orders reuse on the caller's stream: the compile stream starts behind what that
        // stream has queued so far (never behind what it queues later: the loss of this step overlaps the compiler)
        if (!R.main_ev) HIP_TRY(hipEventCreateWithFlags(&R.main_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(R.main_ev, ms));
        HIP_TRY(hipStreamWaitEvent(cs, R.main_ev, 0));
    }
    if (reuse_of > 0) {
        // dev_buf served job `reuse_of` before: behind that job's kernel (if it ran on another stream) and behind the streams
        // that read its graphs (wf_compile_job_release).  A slot that has moved on: those events had completed by then.
        const int rs = (int)(reuse_of % kJobRing);
        if (R.seq_of[rs] == reuse_of && R.cs_of[rs] != cs) HIP_TRY(hipStreamWaitEvent(cs, R.ev[rs], 0));
        if (R.rel_seq[rs] == reuse_of) HIP_TRY(hipStreamWaitEvent(cs, R.rel_ev[rs], 0));
    }
    // One launch: the kernel reads the transcripts from the pinned host buffer (once, into the LDS) and writes the words the
    // host plans with straight into out_host; no copy is enqueued on either side.
    NcArgs p{};
    p.topo = (const int32_t*)topo_tab_dev; p.phones_off = in_host; p.phones = in_host + B + 1; p.B = B;
    p.blob = (int32_t*)dev_buf; p.blob_words = (int64_t)(L.blob_bytes / 4);
    p.out_host = out_host; p.meta_words = (int)L.meta_words;
    rc = nc_launch(p, L.arc_cap, (char*)dev_buf + L.work_off, slot, cs);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(R.ev[slot], cs));
    R.seq_of[slot] = seq;
    R.cs_of[slot] = cs;
    R.next = seq + 1;
    *job = seq;
    return WF_OK;
}

extern "C" int wf_compile_job_wait(int32_t device, int64_t job) {
    if (device < 0 || device >= kMaxDevices || job < 1) return wf_fail(WF_EINVAL, "wf_compile_job_wait: bad job");
    JobRing& R = g_jobs[device];
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lock(R.mu);
        const int slot = (int)(job % kJobRing);
        if (R.seq_of[slot] == job) ev = R.ev[slot];   // (a recycled slot: that job completed before the slot was taken again)
    }
    if (ev) HIP_TRY(hipEventSynchronize(ev));
    return WF_OK;
}

extern "C" int wf_compile_job_join(int32_t device, int64_t job, void* stream) {
    if (device < 0 || device >= kMaxDevices || job < 1) return wf_fail(WF_EINVAL, "wf_compile_job_join: bad job");
    JobRing& R = g_jobs[device];
    std::lock_guard<std::mutex> lock(R.mu);
    const int slot = (int)(job % kJobRing);
    if (R.seq_of[slot] == job) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, R.ev[slot], 0));
    return WF_OK;
}

extern "C" int wf_compile_job_release(int32_t device, int64_t job, void* const* streams, int32_t num_streams) {
    if (device < 0 || device >= kMaxDevices || job < 1 || num_streams < 0 || (num_streams && !streams))
        return wf_fail(WF_EINVAL, "wf_compile_job_release: bad arguments");
    if (num_streams == 0) return WF_OK;
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device) HIP_TRY(hipSetDevice(device));
    struct Restore { int cur, dev; ~Restore() { if (cur != dev) (void)hipSetDevice(cur); } } restore{cur, device};
    JobRing& R = g_jobs[device];
    std::lock_guard<std::mutex> lock(R.mu);
    const int slot = (int)(job % kJobRing);
    if (!R.rel_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&R.rel_ev[slot], hipEventDisableTiming));
    else if (R.rel_seq[slot] && R.rel_seq[slot] != job) HIP_TRY(hipEventSynchronize(R.rel_ev[slot]));
    hipStream_t first = (hipStream_t)streams[0];
    for (int i = 1; i < num_streams; ++i) {      // the first stream gathers the others
        if (!R.tmp_ev) HIP_TRY(hipEventCreateWithFlags(&R.tmp_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(R.tmp_ev, (hipStream_t)streams[i]));
        HIP_TRY(hipStreamWaitEvent(first, R.tmp_ev, 0));
    }
    HIP_TRY(hipEventRecord(R.rel_ev[slot], first));
    R.rel_seq[slot] = job;
    return WF_OK;
}
