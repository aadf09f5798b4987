// Host-side graph preparation and numerator-graph compilation for libwfdwfst.so.
//
// Replaces, on the host, what the reference gets from k2 graph algebra on the way into the loss:
//   waterfall/utils/datapipe.py:180      k2.Fsa.from_openfst -> arc arrays (done in Python, fsa.py)
//   waterfall/models.py:263-270          linear_fsa / remove_epsilon_and_add_self_loops / arc_sort /
//                                        compose(topo, ., treat_epsilons_specially=False)
// and builds the sweep-friendly arc orders that the HIP kernels read (see wf_internal.h).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <unistd.h>

#include "../../include/wf_dwfst.h"
#include "wf_internal.h"

static thread_local char g_err[512] = "";

extern "C" const char* wf_last_error(void) { return g_err; }
extern "C" int wf_version(void) { return 2; }

int wf_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

inline size_t pad4(size_t w) { return wf_pad4(w); }
inline size_t fsa_words_bound(size_t S, size_t A) { return wf_fsa_words_bound(S, A); }   // (wf_internal.h: shared with the device compiler)

// Scratch reused across the FSAs of a call and across calls of a host thread: no heap traffic per graph.
struct PrepScratch {
    std::vector<int32_t> em, fin, col, labs, cnt, pos, colmap, stamp, indeg, order, qbuf, dist, fillf, fillb;
    std::vector<double> fm, fs;
    int32_t epoch = 0;
};
thread_local PrepScratch g_prep;

// Builds FSA g's tables at blob word `cur` (advanced past them).  I: its info words.
// pos_hint (nullable): per non-final state its layer in a layered graph whose states are numbered layer by layer -- the
// transcript position of a topo o transcript product (wf_numerator_compile).  The banded order is then "by layer, inside a
// layer by longest path over the layer's own arcs, then by state id": an order every layer can compute for itself (the
// device compiler does), where the general Kahn order below needs a traversal of the whole graph.
int prepare_one(const wf_graph_host* gh, int g, int32_t* B, int32_t* I, size_t& cur, size_t limit, int* out_maxlab,
                const int32_t* pos_hint = nullptr) {
    PrepScratch& X = g_prep;
    const int a0 = gh->fsa_arc_offsets[g], a1 = gh->fsa_arc_offsets[g + 1];
    const int S = gh->fsa_num_states[g];
    if (a1 < a0) return wf_fail(WF_EINVAL, "fsa %d: arc offsets not monotone", g);
    if (S < 2) return wf_fail(WF_EINVAL, "fsa %d: needs at least a start and a final state (S=%d)", g, S);
    const int32_t* arcs = gh->arcs + 4 * (size_t)a0;
    const int NA = a1 - a0;
    // ---- validation, emitting / final split, distinct labels
    X.em.clear(); X.fin.clear(); X.labs.clear();
    int max_label = -1;
    for (int a = 0; a < NA; ++a) {
        const int32_t* r = arcs + 4 * (size_t)a;
        if (r[0] < 0 || r[0] >= S || r[1] < 0 || r[1] >= S)
            return wf_fail(WF_ERANGE, "fsa %d arc %d: state out of range (%d -> %d, S=%d)", g, a, r[0], r[1], S);
        if (r[2] < -1) return wf_fail(WF_ERANGE, "fsa %d arc %d: label %d < -1", g, a, r[2]);
        if (r[2] == -1) {
            if (r[1] != S - 1) return wf_fail(WF_EINVAL, "fsa %d arc %d: label -1 arc must enter the final state", g, a);
            X.fin.push_back(a);
        } else {
            if (r[1] == S - 1) return wf_fail(WF_EINVAL, "fsa %d arc %d: only label -1 arcs may enter the final state", g, a);
            if (r[0] == S - 1) return wf_fail(WF_EINVAL, "fsa %d arc %d: arc leaves the final state", g, a);
            X.em.push_back(a);
            max_label = std::max(max_label, (int)r[2]);
        }
    }
    const int A = (int)X.em.size(), NF = (int)X.fin.size();
    // label -> column through a stamped table (no search per arc)
    if ((int)X.stamp.size() < max_label + 1) { X.stamp.resize(max_label + 1, 0); X.colmap.resize(max_label + 1, 0); }
    if (++X.epoch == INT32_MAX) { std::fill(X.stamp.begin(), X.stamp.end(), 0); X.epoch = 1; }
    for (int a : X.em) {
        const int l = arcs[4 * (size_t)a + 2];
        if (X.stamp[l] != X.epoch) { X.stamp[l] = X.epoch; X.labs.push_back(l); }
    }
    std::sort(X.labs.begin(), X.labs.end());
    // direct columns when the graph uses at least half of the label range (topologies used as
    // denominators use every token): emissions are then read straight from log_probs.
    const int direct = (max_label >= 0 && X.labs.size() * 2 >= (size_t)max_label + 1) ? 1 : 0;
    const int K = direct ? max_label + 1 : (int)X.labs.size();
    if (!direct) for (int k = 0; k < K; ++k) X.colmap[X.labs[k]] = k;
    X.col.resize(A);
    for (int j = 0; j < A; ++j) { const int l = arcs[4 * (size_t)X.em[j] + 2]; X.col[j] = direct ? l : X.colmap[l]; }
    *out_maxlab = max_label;

    // (the padding words of a table are zeroed: the blob is a pure function of the graph, never a copy of stale memory)
    auto take = [&](size_t n) {
        size_t o = cur;
        cur += pad4(n);
        if (cur <= limit) for (size_t i = o + n; i < cur; ++i) B[i] = 0;
        return (int32_t)o;
    };
    memset(I, 0, sizeof(int32_t) * WF_INFO_WORDS);
    I[WF_I_S] = S; I[WF_I_A] = A; I[WF_I_NF] = NF; I[WF_I_K] = K; I[WF_I_DIRECT] = direct;
    I[WF_I_ARC_BASE] = a0;
    I[WF_I_MAX_LABEL] = max_label;
    const size_t need = 2 * pad4(S + 1) + 12 * pad4(A) + pad4(4 * (size_t)A) + pad4(K + 1) + pad4(K) + 3 * pad4(NF) + 4 * pad4(S) + pad4(A + NF);
    if (cur + need > limit) return wf_fail(WF_ENOSPC, "wf_graph_prepare: blob too small at fsa %d", g);
    X.cnt.assign(std::max(S, K) + 1, 0);
    X.pos.resize(std::max(S, K) + 1);
    int we = 1;
    // ---- incoming CSR: stable sort of emitting arcs by dst
    {
        int o_off = take(S + 1), o_src = take(A), o_col = take(A), o_w = take(A), o_arc = take(A);
        I[WF_I_IN_OFF] = o_off; I[WF_I_IN_SRC] = o_src; I[WF_I_IN_COL] = o_col; I[WF_I_IN_W] = o_w; I[WF_I_IN_ARC] = o_arc;
        int32_t* cnt = X.cnt.data();
        std::fill(cnt, cnt + S + 1, 0);
        for (int a : X.em) cnt[arcs[4 * (size_t)a + 1] + 1]++;
        for (int s2 = 0; s2 < S; ++s2) cnt[s2 + 1] += cnt[s2];
        memcpy(B + o_off, cnt, sizeof(int32_t) * (S + 1));
        int32_t* pos = X.pos.data();
        memcpy(pos, cnt, sizeof(int32_t) * S);
        for (int j = 0; j < A; ++j) {
            const int32_t* r = arcs + 4 * (size_t)X.em[j];
            const int q = pos[r[1]]++;
            B[o_src + q] = r[0];
            B[o_col + q] = X.col[j];
            B[o_w + q] = r[3];
            B[o_arc + q] = X.em[j];
            if (r[3] != 0 && r[3] != (int32_t)0x80000000) we = 0;   // weight != +-0.0
        }
        int o_h = take(S), nh = 0, maxin = 0, nover = 0, over = 0;
        for (int s2 = 0; s2 < S; ++s2) {
            const int deg = cnt[s2 + 1] - cnt[s2];
            maxin = std::max(maxin, deg);
            if (deg > WF_HEAVY_DEG) B[o_h + nh++] = s2;
            if (deg > WF_REG_DEG) { ++nover; over = s2; }
        }
        std::fill(B + o_h + nh, B + o_h + S, 0);
        I[WF_I_HEAVY_IN] = o_h; I[WF_I_NHEAVY_IN] = nh; I[WF_I_MAXIN] = maxin;
        I[WF_I_NOVER_IN] = nover; I[WF_I_OVER_IN] = over;
    }
    // ---- outgoing CSR: emitting arcs in original order grouped by src (stable)
    {
        int o_off = take(S + 1), o_dst = take(A), o_col = take(A), o_w = take(A), o_arc = take(A), o_rec = take(4 * (size_t)A);
        I[WF_I_OUT_OFF] = o_off; I[WF_I_OUT_DST] = o_dst; I[WF_I_OUT_COL] = o_col; I[WF_I_OUT_W] = o_w; I[WF_I_OUT_ARC] = o_arc;
        I[WF_I_OUT_REC] = o_rec;
        int32_t* cnt = X.cnt.data();
        std::fill(cnt, cnt + S + 1, 0);
        for (int a : X.em) cnt[arcs[4 * (size_t)a] + 1]++;
        for (int s2 = 0; s2 < S; ++s2) cnt[s2 + 1] += cnt[s2];
        memcpy(B + o_off, cnt, sizeof(int32_t) * (S + 1));
        int32_t* pos = X.pos.data();
        memcpy(pos, cnt, sizeof(int32_t) * S);
        for (int j = 0; j < A; ++j) {
            const int32_t* r = arcs + 4 * (size_t)X.em[j];
            const int q = pos[r[0]]++;
            B[o_dst + q] = r[1];
            B[o_col + q] = X.col[j];
            B[o_w + q] = r[3];
            B[o_arc + q] = X.em[j];
            int32_t* rec = B + o_rec + 4 * (size_t)q;
            rec[0] = r[1]; rec[1] = X.col[j]; rec[2] = r[3]; rec[3] = X.em[j];
        }
        int o_h = take(S), nh = 0, maxout = 0, nover = 0, over = 0;
        for (int s2 = 0; s2 < S; ++s2) {
            const int deg = cnt[s2 + 1] - cnt[s2];
            maxout = std::max(maxout, deg);
            if (deg > WF_HEAVY_DEG) B[o_h + nh++] = s2;
            if (deg > WF_REG_DEG) { ++nover; over = s2; }
        }
        std::fill(B + o_h + nh, B + o_h + S, 0);
        I[WF_I_HEAVY_OUT] = o_h; I[WF_I_NHEAVY_OUT] = nh; I[WF_I_MAXOUT] = maxout;
        I[WF_I_NOVER_OUT] = nover; I[WF_I_OVER_OUT] = over;
    }
    // ---- label(column)-sorted arcs
    {
        int o_off = take(K + 1), o_src = take(A), o_dst = take(A), o_w = take(A), o_lab = take(K), o_lcol = take(A);
        I[WF_I_LAB_COL] = o_lcol;
        I[WF_I_LAB_OFF] = o_off; I[WF_I_LAB_SRC] = o_src; I[WF_I_LAB_DST] = o_dst; I[WF_I_LAB_W] = o_w; I[WF_I_LABELS] = o_lab;
        int32_t* cnt = X.cnt.data();
        std::fill(cnt, cnt + K + 1, 0);
        for (int j = 0; j < A; ++j) cnt[X.col[j] + 1]++;
        int maxdeg = 0;
        for (int k = 0; k < K; ++k) { maxdeg = std::max(maxdeg, cnt[k + 1]); cnt[k + 1] += cnt[k]; }
        memcpy(B + o_off, cnt, sizeof(int32_t) * (K + 1));
        int32_t* pos = X.pos.data();
        memcpy(pos, cnt, sizeof(int32_t) * K);
        for (int j = 0; j < A; ++j) {
            const int32_t* r = arcs + 4 * (size_t)X.em[j];
            const int q = pos[X.col[j]]++;
            B[o_src + q] = r[0];
            B[o_dst + q] = r[1];
            B[o_w + q] = r[3];
            B[o_lcol + q] = X.col[j];
        }
        for (int k = 0; k < K; ++k) B[o_lab + k] = direct ? k : X.labs[k];
        I[WF_I_MAXLAB_DEG] = maxdeg;
    }
    // ---- final arcs
    {
        int o_src = take(NF), o_w = take(NF), o_arc = take(NF);
        I[WF_I_FIN_SRC] = o_src; I[WF_I_FIN_W] = o_w; I[WF_I_FIN_ARC] = o_arc;
        for (int i = 0; i < NF; ++i) {
            const int a = X.fin[i];
            B[o_src + i] = arcs[4 * (size_t)a];
            B[o_w + i] = arcs[4 * (size_t)a + 3];
            B[o_arc + i] = a;
        }
    }
    // ---- beta_T per state and the source state of every original arc
    {
        int o_fl = take(S), o_fm = take(S), o_as = take(A + NF);
        I[WF_I_FINB_LOG] = o_fl; I[WF_I_FINB_MAX] = o_fm; I[WF_I_ARC_SRC] = o_as;
        const float ninf = -INFINITY;
        int32_t nbits; memcpy(&nbits, &ninf, 4);
        for (int s2 = 0; s2 < S; ++s2) { B[o_fl + s2] = nbits; B[o_fm + s2] = nbits; }
        bool single = true;   // at most one final arc per state (every topology here): beta_T = its weight
        for (int a : X.fin) {
            const int s2 = arcs[4 * (size_t)a];
            if (B[o_fm + s2] != nbits) { single = false; break; }
            B[o_fm + s2] = arcs[4 * (size_t)a + 3];
            B[o_fl + s2] = arcs[4 * (size_t)a + 3];
        }
        if (!single) {
            X.fm.assign(S, -INFINITY); X.fs.assign(S, 0.0);
            for (int a : X.fin) { float w; memcpy(&w, &arcs[4 * (size_t)a + 3], 4); X.fm[arcs[4 * (size_t)a]] = std::max(X.fm[arcs[4 * (size_t)a]], (double)w); }
            for (int a : X.fin) { float w; memcpy(&w, &arcs[4 * (size_t)a + 3], 4); X.fs[arcs[4 * (size_t)a]] += std::exp((double)w - X.fm[arcs[4 * (size_t)a]]); }
            for (int s2 = 0; s2 < S; ++s2) {
                const float fl = (X.fm[s2] == -INFINITY) ? -INFINITY : (float)(X.fm[s2] + std::log(X.fs[s2]));
                const float fm = (float)X.fm[s2];
                memcpy(&B[o_fl + s2], &fl, 4);
                memcpy(&B[o_fm + s2], &fm, 4);
            }
        }
        for (int a = 0; a < A + NF; ++a) B[o_as + a] = arcs[4 * (size_t)a];
    }
    const int32_t* ooff = B + I[WF_I_OUT_OFF];
    const int32_t* odst = B + I[WF_I_OUT_DST];
    // ---- fewest frames any accepted path needs (BFS over emitting arcs from the start state)
    {
        X.dist.assign(S, -1);
        X.qbuf.resize(S);
        int32_t* dist = X.dist.data();
        for (int a : X.fin) if (dist[arcs[4 * (size_t)a]] == -1) dist[arcs[4 * (size_t)a]] = -2;   // -2: unvisited, has a final arc
        int qh = 0, qt = 0, minlen = INT32_MAX;
        const bool start_final = dist[0] == -2;
        dist[0] = 0; X.qbuf[qt++] = 0;
        if (start_final) minlen = 0;
        while (qh < qt && minlen == INT32_MAX) {
            const int s2 = X.qbuf[qh++];
            for (int e = ooff[s2]; e < ooff[s2 + 1]; ++e) {
                const int d = odst[e];
                if (dist[d] >= 0) continue;
                const bool fin_here = dist[d] == -2;
                dist[d] = dist[s2] + 1;
                if (fin_here) { minlen = std::min(minlen, dist[d]); }
                X.qbuf[qt++] = d;
            }
        }
        I[WF_I_MINLEN] = minlen;
    }
    // ---- banded form: Kahn topological order (FIFO, self-loops ignored) of the non-final states
    {
        const int Sn = S - 1;   // the super-final state has no emitting arcs
        I[WF_I_BAND_BW] = -1; I[WF_I_BAND_M] = 0;
        I[WF_I_BAND_WE] = we;
        if (Sn >= 1 && Sn <= 64 * WF_BAND_MAX_R) {
            X.indeg.assign(Sn, 0); X.order.clear(); X.pos.resize(std::max(std::max(S, K) + 1, Sn));
            int32_t* pos = X.pos.data();
            if (pos_hint) {
                // layer by layer; X.indeg holds a state's longest-path depth inside its layer
                bool ok = true;
                for (int s0 = 0; s0 < Sn && ok;) {
                    int s1 = s0 + 1;
                    while (s1 < Sn && pos_hint[s1] == pos_hint[s0]) ++s1;
                    const int n = s1 - s0;
                    bool changed = true;
                    for (int round = 0; round <= n && changed; ++round) {
                        changed = false;
                        for (int s2 = s0; s2 < s1; ++s2)
                            for (int e = ooff[s2]; e < ooff[s2 + 1]; ++e) {
                                const int d = odst[e];
                                if (d == s2 || d < s0 || d >= s1) continue;
                                if (X.indeg[d] < X.indeg[s2] + 1) { X.indeg[d] = X.indeg[s2] + 1; changed = true; }
                            }
                    }
                    if (changed) { ok = false; break; }   // a cycle inside the layer besides self-loops: no banded order
                    for (int dep = 0; dep < n; ++dep)
                        for (int s2 = s0; s2 < s1; ++s2)
                            if (X.indeg[s2] == dep) { pos[s2] = (int)X.order.size(); X.order.push_back(s2); }
                    s0 = s1;
                }
                if (ok)   // arcs must not run backwards across layers (they do not in a product with a linear transcript)
                    for (int s2 = 0; s2 < Sn && ok; ++s2)
                        for (int e = ooff[s2]; e < ooff[s2 + 1]; ++e)
                            if (pos_hint[odst[e]] < pos_hint[s2]) { ok = false; break; }
                if (!ok) X.order.clear();
            } else {
            for (int s2 = 0; s2 < Sn; ++s2)
                for (int e = ooff[s2]; e < ooff[s2 + 1]; ++e)
                    if (odst[e] != s2) X.indeg[odst[e]]++;
            X.qbuf.resize(S);
            int qh = 0, qt = 0;
            for (int s2 = 0; s2 < Sn; ++s2) if (X.indeg[s2] == 0) X.qbuf[qt++] = s2;
            while (qh < qt) {
                const int s2 = X.qbuf[qh++];
                pos[s2] = (int)X.order.size(); X.order.push_back(s2);
                for (int e = ooff[s2]; e < ooff[s2 + 1]; ++e)
                    if (odst[e] != s2 && --X.indeg[odst[e]] == 0) X.qbuf[qt++] = odst[e];
            }
            }
            if ((int)X.order.size() == Sn) {
                int bw = 0;
                for (int a : X.em) bw = std::max(bw, pos[arcs[4 * (size_t)a + 1]] - pos[arcs[4 * (size_t)a]]);
                // arcs per (destination position, distance) cell
                int m = 0;
                if (bw <= WF_BAND_MAX_BW) {
                    X.fillf.assign((size_t)Sn * (bw + 1), 0);
                    for (int a : X.em) {
                        const int pd = pos[arcs[4 * (size_t)a + 1]], d = pd - pos[arcs[4 * (size_t)a]];
                        m = std::max(m, ++X.fillf[(size_t)pd * (bw + 1) + d]);
                    }
                }
                if (bw <= WF_BAND_MAX_BW && m <= WF_BAND_MAX_M) {
                    m = std::max(m, 1);
                    const size_t cells = (size_t)Sn * (bw + 1) * m;
                    if (cur + pad4(Sn) + 2 * pad4(cells * 2) > limit) return wf_fail(WF_ENOSPC, "wf_graph_prepare: blob too small at fsa %d (band tables)", g);
                    int o_ord = take(Sn), o_f = take(cells * 2), o_b = take(cells * 2);
                    I[WF_I_BAND_BW] = bw; I[WF_I_BAND_M] = m; I[WF_I_BAND_ORDER] = o_ord; I[WF_I_BANDF] = o_f; I[WF_I_BANDB] = o_b;
                    memcpy(B + o_ord, X.order.data(), sizeof(int32_t) * Sn);
                    // empty cells: column 0, weight -inf (probability 0)
                    const float ninf = -INFINITY; int32_t nbits; memcpy(&nbits, &ninf, 4);
                    for (size_t c = 0; c < cells; ++c) { B[o_f + 2 * c] = 0; B[o_f + 2 * c + 1] = nbits; B[o_b + 2 * c] = 0; B[o_b + 2 * c + 1] = nbits; }
                    X.fillf.assign((size_t)Sn * (bw + 1), 0); X.fillb.assign((size_t)Sn * (bw + 1), 0);
                    for (int j = 0; j < A; ++j) {
                        const int32_t* r = arcs + 4 * (size_t)X.em[j];
                        const int ps = pos[r[0]], pd = pos[r[1]], d = pd - ps;
                        const int col = X.col[j];
                        const size_t cf = (size_t)pd * (bw + 1) + d;
                        const size_t ef = cf * m + X.fillf[cf]++;
                        B[o_f + 2 * ef] = col; B[o_f + 2 * ef + 1] = r[3];
                        const size_t cb = (size_t)(Sn - 1 - ps) * (bw + 1) + d;
                        const size_t eb = cb * m + X.fillb[cb]++;
                        B[o_b + 2 * eb] = col; B[o_b + 2 * eb + 1] = r[3];
                    }
                }
            }
        }
    }
    // ---- hub form (wf_internal.h WF_I_HUB_*): a topology used as denominator over a large table
    I[WF_I_HUB_G] = 0;
    if (we && direct && S - 2 >= 256 && NF >= 1) {
        bool fin_ok = true;
        for (int a : X.fin) if (arcs[4 * (size_t)a] != 0) fin_ok = false;
        if (fin_ok && NF == 1)
            for (int G = 1; G <= WF_HUB_MAX_G && !I[WF_I_HUB_G]; ++G) {
                if ((S - 2) % G) continue;
                const int NG = (S - 2) / G, W = G + 1;
                auto loc = [&](int st, int* g) { if (st == 0) { *g = -1; return G; } *g = (st - 1) / G; return (st - 1) % G; };
                // column classes from group 0 (sorted), stride from group 1
                int c0[WF_HUB_MAX_NC + 1], nc = 0;
                bool ok = true;
                for (int j = 0; j < A && ok; ++j) {
                    const int32_t* r = arcs + 4 * (size_t)X.em[j];
                    int gs, gd;
                    const int ls = loc(r[0], &gs), ld = loc(r[1], &gd);
                    (void)ls; (void)ld;
                    if (gs >= 0 && gd >= 0 && gs != gd) { ok = false; break; }   // an arc between two groups
                    const int g = gs >= 0 ? gs : gd;
                    if (g != 0) continue;
                    bool have = false;
                    for (int c = 0; c < nc; ++c) if (c0[c] == r[2]) have = true;
                    if (!have) { if (nc == WF_HUB_MAX_NC) { ok = false; break; } c0[nc++] = r[2]; }
                }
                if (!ok || nc == 0) continue;
                std::sort(c0, c0 + nc);
                int cs = 0;
                if (NG >= 2) {   // the smallest column used by group 1 belongs to class 0
                    int m1 = INT32_MAX;
                    for (int j = 0; j < A; ++j) {
                        const int32_t* r = arcs + 4 * (size_t)X.em[j];
                        int gs, gd;
                        loc(r[0], &gs); loc(r[1], &gd);
                        if ((gs >= 0 ? gs : gd) == 1) m1 = std::min(m1, (int)r[2]);
                    }
                    if (m1 == INT32_MAX) continue;
                    cs = m1 - c0[0];
                    if (cs <= 0) continue;
                }
                // count tables: [group][(dst*W + src)*nc + c]; every group must equal group 0; hub self-loops are given to the
                // group that owns their column, the rest are extras
                const int TS = W * W * nc;
                X.fillf.assign((size_t)TS * 2, 0);          // [0..TS): group 0's table, [TS..2TS): the current group's
                X.fillb.assign(NG, 0);                        // arcs seen per group
                std::vector<int32_t>& cntg = X.indeg;         // per group per table entry: built lazily, compared at the end
                cntg.assign((size_t)NG * TS, 0);
                X.order.clear();                              // extra hub self-loop columns
                auto cls = [&](int col, int g) { for (int c = 0; c < nc; ++c) if (c0[c] + (int64_t)g * cs == col) return c; return -1; };
                for (int j = 0; j < A && ok; ++j) {
                    const int32_t* r = arcs + 4 * (size_t)X.em[j];
                    int gs, gd;
                    const int ls = loc(r[0], &gs), ld = loc(r[1], &gd);
                    int g = gs >= 0 ? gs : gd;
                    if (g < 0) {   // hub self-loop: owned by a group when its column is one of that group's
                        g = -1;
                        for (int c = 0; c < nc && g < 0; ++c) {
                            const int64_t d = (int64_t)r[2] - c0[c];
                            if (NG == 1 ? d == 0 : (d >= 0 && d % cs == 0 && d / cs < NG)) g = NG == 1 ? 0 : (int)(d / cs);
                        }
                        if (g < 0) { X.order.push_back(r[2]); continue; }
                    }
                    const int c = cls(r[2], g);
                    if (c < 0) { ok = false; break; }
                    cntg[(size_t)g * TS + (ld * W + ls) * nc + c]++;
                }
                for (int g = 1; g < NG && ok; ++g)
                    if (memcmp(&cntg[(size_t)g * TS], &cntg[0], sizeof(int32_t) * TS)) ok = false;
                // the posterior kernel writes exactly the columns of the hub form: they must be all of 0..K-1, each once
                {
                    std::vector<int32_t>& seen = X.dist;
                    seen.assign(K, 0);
                    for (int g = 0; g < NG && ok; ++g)
                        for (int c = 0; c < nc; ++c) {
                            const int64_t col = c0[c] + (int64_t)g * cs;
                            if (col >= K || seen[col]++) { ok = false; break; }
                        }
                    for (size_t e = 0; e < X.order.size() && ok; ++e)
                        if (X.order[e] >= K || seen[X.order[e]]++) ok = false;
                    if (ok && (int64_t)NG * nc + (int64_t)X.order.size() != K) ok = false;
                }
                if (!ok || cur + pad4(nc) + pad4(TS) + pad4(X.order.size()) > limit) continue;
                const int o_c0 = take(nc), o_cf = take(TS), o_x = take(X.order.size());
                for (int c = 0; c < nc; ++c) B[o_c0 + c] = c0[c];
                for (int e = 0; e < TS; ++e) { const float f = (float)cntg[e]; memcpy(&B[o_cf + e], &f, 4); }
                for (size_t e = 0; e < X.order.size(); ++e) B[o_x + e] = X.order[e];
                I[WF_I_HUB_G] = G; I[WF_I_HUB_NG] = NG; I[WF_I_HUB_NC] = nc; I[WF_I_HUB_C0] = o_c0; I[WF_I_HUB_CS] = cs;
                I[WF_I_HUB_COEF] = o_cf; I[WF_I_HUB_NX] = (int32_t)X.order.size(); I[WF_I_HUB_XCOL] = o_x;
                I[WF_I_HUB_FINW] = arcs[4 * (size_t)X.fin[0] + 3];
            }
    }
    return WF_OK;
}

}  // namespace

// An upper bound that needs no pass over the arcs (wf_graph_prepare fills B[WF_H_WORDS] with the words actually used:
// only those need to travel to the device).
extern "C" int wf_graph_prepare_bytes(const wf_graph_host* g, size_t* bytes) {
    if (!g || !bytes || g->num_fsas < 1 || !g->arcs || !g->fsa_arc_offsets || !g->fsa_num_states)
        return wf_fail(WF_EINVAL, "wf_graph_prepare_bytes: null or empty graph");
    size_t w = pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * g->num_fsas);
    for (int i = 0; i < g->num_fsas; ++i) {
        const int na = g->fsa_arc_offsets[i + 1] - g->fsa_arc_offsets[i];
        if (na < 0 || g->fsa_num_states[i] < 2) return wf_fail(WF_EINVAL, "fsa %d: bad arc offsets or fewer than two states", i);
        w += fsa_words_bound((size_t)g->fsa_num_states[i], (size_t)na);
    }
    if (w >= (size_t)1 << 31) return wf_fail(WF_EUNSUPPORTED, "prepared graph exceeds 2^31 words");
    *bytes = w * 4;
    return WF_OK;
}

extern "C" int wf_graph_prepare(const wf_graph_host* gh, void* blob_host, size_t bytes) {
    if (!gh || gh->num_fsas < 1 || !gh->arcs || !gh->fsa_arc_offsets || !gh->fsa_num_states)
        return wf_fail(WF_EINVAL, "wf_graph_prepare: null or empty graph");
    const int G = gh->num_fsas;
    size_t cur = pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * G);
    if (!blob_host || bytes < cur * 4) return wf_fail(WF_ENOSPC, "wf_graph_prepare: blob too small (%zu B)", bytes);
    int32_t* B = (int32_t*)blob_host;
    const size_t limit = std::min(bytes / 4, (size_t)INT32_MAX);
    memset(B, 0, sizeof(int32_t) * WF_HDR_WORDS);
    B[WF_H_MAGIC] = WF_BLOB_MAGIC;
    B[WF_H_NUM_FSAS] = G;
    int gmaxlab = -1, gmaxS = 0, gmaxA = 0;
    for (int g = 0; g < G; ++g) {
        int32_t* I = B + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * g;
        int maxlab = -1;
        int rc = prepare_one(gh, g, B, I, cur, limit, &maxlab);
        if (rc) return rc;
        gmaxlab = std::max(gmaxlab, maxlab); gmaxS = std::max(gmaxS, (int)I[WF_I_S]); gmaxA = std::max(gmaxA, (int)I[WF_I_A]);
    }
    B[WF_H_WORDS] = (int32_t)cur;
    B[WF_H_MAX_LABEL] = gmaxlab; B[WF_H_MAX_STATES] = gmaxS; B[WF_H_MAX_ARCS] = gmaxA;
    return WF_OK;
}

// ------------------------------------------------------------------------------------------
// Batched numerator-graph compiler: the product of wf_compose_linear for a whole
// batch in one call, against a topology indexed once by (source state, phone).  A product state (q, i) then
// touches only the topology arcs that can match -- phone 0 (stay at position i) and phone b[i] (advance) --
// instead of every arc leaving q (5 001 for the hub state of a BPE-5k topology).  Results are identical to
// wf_compose_linear on the same topology: same BFS state numbering, same arc order.
//
// Index layout (int32 words): soff[nS+1] | perm[nA] (arc ids, per state stably sorted by phone; final arcs
// carry key -1) | key[nA] (the phone of perm[j]).
//
// ctc_cross != 0: the CTC topology of utils/get_token_fst_ctc.py:25-53 in factored form.  Its O(P^2) arcs
// "other phone's state -> this phone's state, p : p" (:37-40) are not stored; phone_state[p] / phone_token[p]
// (0 = phone has no state) describe them, and a product state (q != 0, i) gets the arc to phone_state[b[i]]
// at the place the explicit arc would have in the (phone, token)-sorted arc list of q.
// ------------------------------------------------------------------------------------------
// Index layout (int32 words): soff[nS+1] | perm[nA] | key[nA] | f1[nS] | z1[nS] | dense[nS] | maxph | tables...
//   f1[q] / z1[q]: end of state q's final-arc run / of its phone-0 run inside perm (both start where the previous ends)
//   dense[q]: 0, or the word offset (from the start of the index) of a table t[0 .. maxph + 1] for a state with more than
//   kDenseMin phone-labelled arcs: the arcs with phone p are perm[t[p] .. t[p + 1]) -- one load instead of a binary
//   search over the 5 001 arcs of a BPE-5k hub state in every product state (q, i)
static const int kDenseMin = 8;

static int topo_max_phone(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA) {
    int m = 0;
    for (int a = 0; a < nA; ++a)
        if (topo_arcs[4 * (size_t)a + 2] >= 0) m = std::max(m, (int)topo_aux[a]);
    return m;
}

extern "C" int wf_topo_index_words(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS, int64_t* words) {
    if (!topo_arcs || !topo_aux || nA < 1 || nS < 2 || !words) return wf_fail(WF_EINVAL, "wf_topo_index_words: bad argument");
    std::vector<int> deg(nS, 0);
    for (int a = 0; a < nA; ++a) {
        const int s = topo_arcs[4 * (size_t)a];
        if (s < 0 || s >= nS) return wf_fail(WF_ERANGE, "wf_topo_index_words: topo arc %d state out of range", a);
        if (topo_arcs[4 * (size_t)a + 2] >= 0 && topo_aux[a] > 0) deg[s]++;
    }
    const int64_t maxph = topo_max_phone(topo_arcs, topo_aux, nA);
    int64_t w = (int64_t)nS + 1 + 2 * (int64_t)nA + 3 * (int64_t)nS + 1;
    for (int s = 0; s < nS; ++s) if (deg[s] > kDenseMin) w += maxph + 2;
    *words = w;
    return WF_OK;
}

extern "C" int wf_topo_index_build(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS,
                                   int32_t* index) {
    if (!topo_arcs || !topo_aux || nA < 1 || nS < 2 || !index) return wf_fail(WF_EINVAL, "wf_topo_index_build: bad argument");
    int32_t* soff = index;
    int32_t* perm = index + nS + 1;
    int32_t* key = perm + nA;
    int32_t* f1 = key + nA;
    int32_t* z1 = f1 + nS;
    int32_t* dense = z1 + nS;
    std::fill(soff, soff + nS + 1, 0);
    for (int a = 0; a < nA; ++a) {
        const int s = topo_arcs[4 * (size_t)a], d = topo_arcs[4 * (size_t)a + 1];
        if (s < 0 || s >= nS || d < 0 || d >= nS) return wf_fail(WF_ERANGE, "wf_topo_index_build: topo arc %d state out of range", a);
        if (a && s < topo_arcs[4 * (size_t)(a - 1)]) return wf_fail(WF_EINVAL, "wf_topo_index_build: topo arcs must be sorted by source state");
        if (topo_arcs[4 * (size_t)a + 2] >= 0 && topo_aux[a] < 0) return wf_fail(WF_ERANGE, "wf_topo_index_build: topo arc %d has a negative phone", a);
        soff[s + 1]++;
    }
    for (int s = 0; s < nS; ++s) soff[s + 1] += soff[s];
    for (int a = 0; a < nA; ++a) perm[a] = a;
    auto ph = [&](int a) { return topo_arcs[4 * (size_t)a + 2] < 0 ? -1 : topo_aux[a]; };
    for (int s = 0; s < nS; ++s)
        std::stable_sort(perm + soff[s], perm + soff[s + 1], [&](int x, int y) { return ph(x) < ph(y); });
    for (int j = 0; j < nA; ++j) key[j] = ph(perm[j]);
    const int maxph = topo_max_phone(topo_arcs, topo_aux, nA);
    dense[nS] = maxph;
    int64_t w = (int64_t)nS + 1 + 2 * (int64_t)nA + 3 * (int64_t)nS + 1;
    for (int s = 0; s < nS; ++s) {
        const int32_t* ke = key + soff[s + 1];
        f1[s] = (int32_t)(std::upper_bound((const int32_t*)key + soff[s], ke, -1) - key);
        z1[s] = (int32_t)(std::upper_bound((const int32_t*)key + f1[s], ke, 0) - key);
        dense[s] = 0;
        if (soff[s + 1] - z1[s] > kDenseMin) {
            dense[s] = (int32_t)w;
            int32_t* t = index + w;
            w += maxph + 2;
            int j = z1[s];
            for (int p = 0; p <= maxph + 1; ++p) {   // t[p] = first position with key >= p (t[0] = t[1] = z1: phone 0 is not in this run)
                while (j < soff[s + 1] && key[j] < p) ++j;
                t[p] = j;
            }
        }
    }
    return WF_OK;
}

// Topology tables of the DEVICE numerator compiler (wf_compile.hip numerator_compile_kernel; layout: wf_internal.h
// WF_T_*): the arc arrays, the (state, phone) index above, the factored-CTC tables, and per phone the distinct states its
// arcs enter -- what a transcript position needs to list the topology states that can be alive there without knowing
// the positions before it.  Built once per topology on the host, uploaded once.  out == NULL: only *words_needed.
extern "C" int wf_topo_device_build(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS,
                                    const int32_t* index, int64_t index_words, int32_t ctc_cross, const int32_t* phone_state,
                                    const int32_t* phone_token, int32_t num_phones, int32_t* out, int64_t out_words,
                                    int64_t* words_needed) {
    if (!topo_arcs || !topo_aux || !index || nA < 1 || nS < 2 || index_words < 1 || !words_needed)
        return wf_fail(WF_EINVAL, "wf_topo_device_build: bad argument");
    if (ctc_cross && (!phone_state || !phone_token || num_phones < 1))
        return wf_fail(WF_EINVAL, "wf_topo_device_build: ctc_cross needs the phone -> state / token tables");
    const int maxph = topo_max_phone(topo_arcs, topo_aux, nA);
    const int nph = std::max(maxph, ctc_cross ? num_phones - 1 : 0);
    std::vector<std::vector<int32_t>> dsts((size_t)nph + 1);
    for (int a = 0; a < nA; ++a) {
        const int32_t* r = topo_arcs + 4 * (size_t)a;
        if (r[2] >= 0 && topo_aux[a] > 0) dsts[topo_aux[a]].push_back(r[1]);
    }
    if (ctc_cross)
        for (int p = 1; p < num_phones; ++p) if (phone_state[p] > 0) dsts[p].push_back(phone_state[p]);
    size_t npd = 0;
    for (auto& v : dsts) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); npd += v.size(); }
    const int32_t* soff = index;
    const int32_t* key = index + nS + 1 + nA;
    const int32_t* f1s = key + nA;
    const int32_t* z1s = f1s + nS;
    int maxd = 1;
    for (int q = 0; q < nS; ++q) {
        int run = 0, best = 0;
        for (int x = z1s[q]; x < soff[q + 1]; ++x) { run = (x > z1s[q] && key[x] == key[x - 1]) ? run + 1 : 1; best = std::max(best, run); }
        maxd = std::max(maxd, z1s[q] - f1s[q] + best + (ctc_cross ? 1 : 0));
    }
    const int npt = ctc_cross ? num_phones : 0;
    size_t o = WF_T_HDR;
    const size_t o_arcs = o; o += 4 * (size_t)nA;
    const size_t o_aux = o; o += (size_t)nA;
    const size_t o_index = o; o += (size_t)index_words;
    const size_t o_ps = o; o += (size_t)npt;
    const size_t o_pt = o; o += (size_t)npt;
    const size_t o_pdoff = o; o += (size_t)nph + 2;
    const size_t o_pdst = o; o += npd;
    *words_needed = (int64_t)o;
    if (o >= (size_t)1 << 31) return wf_fail(WF_EUNSUPPORTED, "wf_topo_device_build: table exceeds 2^31 words");
    if (!out) return WF_OK;
    if (out_words < (int64_t)o) return wf_fail(WF_ENOSPC, "wf_topo_device_build: %lld words given, %zu needed", (long long)out_words, o);
    memset(out, 0, sizeof(int32_t) * WF_T_HDR);
    out[WF_T_MAGIC] = WF_TOPO_MAGIC; out[WF_T_NA] = nA; out[WF_T_NS] = nS; out[WF_T_CROSS] = ctc_cross ? 1 : 0;
    out[WF_T_NPH] = nph; out[WF_T_MAXPH] = maxph; out[WF_T_OARCS] = (int32_t)o_arcs; out[WF_T_OAUX] = (int32_t)o_aux;
    out[WF_T_OINDEX] = (int32_t)o_index; out[WF_T_OPSTATE] = (int32_t)o_ps; out[WF_T_OPTOKEN] = (int32_t)o_pt;
    out[WF_T_OPDOFF] = (int32_t)o_pdoff; out[WF_T_OPDST] = (int32_t)o_pdst; out[WF_T_MAXD] = maxd; out[WF_T_WORDS] = (int32_t)o;
    out[WF_T_NUMPH] = npt;
    memcpy(out + o_arcs, topo_arcs, sizeof(int32_t) * 4 * (size_t)nA);
    memcpy(out + o_aux, topo_aux, sizeof(int32_t) * (size_t)nA);
    memcpy(out + o_index, index, sizeof(int32_t) * (size_t)index_words);
    if (npt) { memcpy(out + o_ps, phone_state, sizeof(int32_t) * npt); memcpy(out + o_pt, phone_token, sizeof(int32_t) * npt); }
    int32_t* pdoff = out + o_pdoff;
    int32_t* pdst = out + o_pdst;
    size_t w = 0;
    pdoff[0] = 0;
    for (int p = 0; p <= nph; ++p) {
        for (int32_t d : dsts[p]) pdst[w++] = d;
        pdoff[p + 1] = (int32_t)w;
    }
    return WF_OK;
}

namespace {
// open-addressing map (q, i) -> product state id, cleared per utterance
struct PairMap {
    std::vector<int64_t> k;
    std::vector<int32_t> v;
    size_t mask = 0;
    void reset(size_t want) {
        size_t n = 64;
        while (n < 2 * want) n <<= 1;
        if (k.size() != n) { k.assign(n, -1); v.assign(n, 0); }
        else std::fill(k.begin(), k.end(), (int64_t)-1);
        mask = n - 1;
    }
    void grow() {
        std::vector<int64_t> ok;
        std::vector<int32_t> ov;
        ok.swap(k); ov.swap(v);
        k.assign(ok.size() * 2, -1); v.assign(ok.size() * 2, 0);
        mask = k.size() - 1;
        for (size_t j = 0; j < ok.size(); ++j)
            if (ok[j] >= 0) { size_t h = hash(ok[j]); while (k[h] >= 0) h = (h + 1) & mask; k[h] = ok[j]; v[h] = ov[j]; }
    }
    size_t hash(int64_t x) const { return (size_t)((uint64_t)x * 0x9E3779B97F4A7C15ull >> 20) & mask; }
    // returns the id of key x, inserting next_id if absent (inserted = true)
    int32_t get(int64_t x, int32_t next_id, bool* inserted) {
        size_t h = hash(x);
        while (k[h] >= 0) {
            if (k[h] == x) { *inserted = false; return v[h]; }
            h = (h + 1) & mask;
        }
        k[h] = x; v[h] = next_id; *inserted = true;
        return next_id;
    }
};
}  // namespace

namespace {
struct TopoView {
    const int32_t* arcs; const int32_t* aux; int nA, nS;
    const int32_t* index; int ctc_cross; const int32_t* phone_state; const int32_t* phone_token; int num_phones;
};
struct ComposeScratch {
    PairMap id;
    std::vector<int64_t> order;
    std::vector<int32_t> arcs, aux;   // the composed graph of the last utterance: arcs [n][4] (final arcs: dst = -1), aux [n]
    std::vector<int32_t> head_start, arcs2, aux2, newid, oldof;
    std::vector<int32_t> pos;         // per product state (new numbering): its right-hand state (transcript position)
    int num_states = 0;
};

// Product states are DISCOVERED breadth first from (0, 0), but NUMBERED by (right-hand state r, topology state q): for a
// linear transcript that is position-major order -- a numbering every position can compute for itself from a prefix sum,
// which is what lets the device compiler (numerator_compile_kernel) emit the same graph without a sequential traversal.
// (k2's own numbering in compose is parity unpinned: SURVEY 8(c).)  order[old id] = q * R + r; arcs [n][4] = {head, dst
// or -1, label, weight} grouped by head in discovery order, head_start[h] = index of head h's first arc (n at the end).
// Arcs keep their order within a state; X.pos receives r per new id.
void renumber_position_major(ComposeScratch& X, int64_t R) {
    const int n = (int)X.order.size();
    X.oldof.resize(n);
    for (int i = 0; i < n; ++i) X.oldof[i] = i;
    const int64_t* key = X.order.data();
    std::sort(X.oldof.begin(), X.oldof.end(), [&](int a, int b) {
        const int64_t ra = key[a] % R, rb = key[b] % R;
        return ra != rb ? ra < rb : key[a] < key[b];
    });
    X.newid.resize(n);
    X.pos.resize(n);
    for (int i = 0; i < n; ++i) { X.newid[X.oldof[i]] = i; X.pos[i] = (int32_t)(key[X.oldof[i]] % R); }
    X.arcs2.resize(X.arcs.size());
    X.aux2.resize(X.aux.size());
    size_t w = 0;
    for (int s2 = 0; s2 < n; ++s2) {
        const int old = X.oldof[s2];
        for (int a = X.head_start[old]; a < X.head_start[old + 1]; ++a, ++w) {
            const int32_t* r = &X.arcs[4 * (size_t)a];
            int32_t* o = &X.arcs2[4 * w];
            o[0] = s2; o[1] = r[1] < 0 ? -1 : X.newid[r[1]]; o[2] = r[2]; o[3] = r[3];
            X.aux2[w] = X.aux[a];
        }
    }
    X.arcs.swap(X.arcs2);
    X.aux.swap(X.aux2);
}

// compose(topo, linear transcript seq[0..L)) into X.arcs / X.aux (state ids in BFS order, super-final = X.num_states - 1)
int compose_linear_one(const TopoView& tv, const int32_t* seq, int L, int b, ComposeScratch& X) {
    const int nA = tv.nA, nS = tv.nS;
    const int32_t* topo_arcs = tv.arcs;
    const int32_t* index = tv.index;
    const int32_t* soff = index;
    const int32_t* perm = index + nS + 1;
    const int32_t* key = perm + nA;
    const int32_t* f1s = key + nA;
    const int32_t* z1s = f1s + nS;
    const int32_t* dense = z1s + nS;
    const int maxph = dense[nS];
    const int ctc_cross = tv.ctc_cross;
    const int32_t* phone_state = tv.phone_state;
    const int32_t* phone_token = tv.phone_token;
    PairMap& id = X.id;
    std::vector<int64_t>& order = X.order;
    std::vector<int32_t>& arcs = X.arcs;
    std::vector<int32_t>& aux = X.aux;
    for (int i = 0; i < L; ++i)
        if (seq[i] < 1 || (ctc_cross && seq[i] >= tv.num_phones))
            return wf_fail(WF_ERANGE, "wf_compose_linear_batch: transcript %d position %d: phone %d out of range", b, i, seq[i]);
    id.reset(8 * (size_t)(L + 2));
    order.clear(); arcs.clear(); aux.clear(); X.head_start.clear();
    auto pkey = [&](int q, int i) { return (int64_t)q * (L + 1) + i; };
    bool ins;
    id.get(pkey(0, 0), 0, &ins);
    order.push_back(pkey(0, 0));
    auto emit = [&](int head, int q2, int ni, int tok, int32_t wbits, int phv) {
        if (order.size() * 2 > id.k.size()) id.grow();
        const int32_t nid = id.get(pkey(q2, ni), (int32_t)order.size(), &ins);
        if (ins) order.push_back(pkey(q2, ni));
        arcs.push_back(head); arcs.push_back(nid); arcs.push_back(tok); arcs.push_back(wbits);
        aux.push_back(phv);
    };
    for (size_t head = 0; head < order.size(); ++head) {
        X.head_start.push_back((int32_t)aux.size());
        const int q = (int)(order[head] / (L + 1)), i = (int)(order[head] % (L + 1));
        const int lo = soff[q], hi = soff[q + 1];
        // runs of this state's arcs: final (key -1), phone 0, phone b[i]
        const int f1 = f1s[q], z1 = z1s[q];
        int p0 = hi, p1 = hi, nph = 0;
        if (i < L) {
            nph = seq[i];
            if (dense[q]) {
                if (nph <= maxph) { const int32_t* t = index + dense[q]; p0 = t[nph]; p1 = t[nph + 1]; }
            } else {
                p0 = z1;
                while (p0 < hi && key[p0] < nph) ++p0;
                p1 = p0;
                while (p1 < hi && key[p1] == nph) ++p1;
            }
        }
        // implicit CTC arc (q -> phone_state[b[i]], b[i] : b[i])
        int cq = 0, ctok = 0;
        const bool cross = ctc_cross && q != 0 && i < L && phone_state[nph] > 0 && phone_state[nph] != q;
        if (cross) { cq = phone_state[nph]; ctok = phone_token[nph]; }
        int x = f1, y = p0;
        if (!cross) {
            // emitting arcs in topology order: merge the phone-0 run and the phone-b[i] run by arc id
            while (x < z1 || y < p1) {
                const int ax = x < z1 ? perm[x] : INT32_MAX, ay = y < p1 ? perm[y] : INT32_MAX;
                if (ax < ay) {
                    const int32_t* r = topo_arcs + 4 * (size_t)ax;
                    emit((int)head, r[1], i, r[2], r[3], 0);
                    ++x;
                } else {
                    const int32_t* r = topo_arcs + 4 * (size_t)ay;
                    emit((int)head, r[1], i + 1, r[2], r[3], nph);
                    ++y;
                }
            }
        } else {
            // the factored CTC topology is sorted by (phone, token) per state (prepare_ctc.sh: fstarcsort olabel):
            // phone-0 arcs first, then the phone-b[i] arcs by token, the implicit one among them
            for (; x < z1; ++x) {
                const int32_t* r = topo_arcs + 4 * (size_t)perm[x];
                emit((int)head, r[1], i, r[2], r[3], 0);
            }
            bool done = false;
            for (; y < p1; ++y) {
                const int32_t* r = topo_arcs + 4 * (size_t)perm[y];
                if (!done && r[2] > ctok) { emit((int)head, cq, i + 1, ctok, 0, nph); done = true; }
                emit((int)head, r[1], i + 1, r[2], r[3], nph);
            }
            if (!done) emit((int)head, cq, i + 1, ctok, 0, nph);
        }
        // final arcs of q at the last position: after the emitting arcs of this product state
        if (i == L)
            for (int j = lo; j < f1; ++j) {
                const int32_t* r = topo_arcs + 4 * (size_t)perm[j];
                arcs.push_back((int32_t)head); arcs.push_back(-1); arcs.push_back(-1); arcs.push_back(r[3]);
                aux.push_back(-1);
            }
    }
    X.head_start.push_back((int32_t)aux.size());
    X.num_states = (int)order.size() + 1;
    renumber_position_major(X, (int64_t)L + 1);
    return WF_OK;
}

// copies X's graph to its place in the FsaVec arrays (final arcs get their destination)
void compose_store(const ComposeScratch& X, int32_t* out_arcs, int32_t* out_aux) {
    const int64_t n = (int64_t)X.aux.size();
    const int sf = X.num_states - 1;
    for (int64_t a = 0; a < n; ++a) {
        int32_t* o = out_arcs + 4 * (size_t)a;
        o[0] = X.arcs[4 * a]; o[1] = X.arcs[4 * a + 1] < 0 ? sf : X.arcs[4 * a + 1]; o[2] = X.arcs[4 * a + 2]; o[3] = X.arcs[4 * a + 3];
        out_aux[a] = X.aux[a];
    }
}
}  // namespace

extern "C" int wf_compose_linear_batch(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS,
                                       const int32_t* index, int32_t ctc_cross, const int32_t* phone_state,
                                       const int32_t* phone_token, int32_t num_phones,
                                       const int32_t* phones, const int32_t* phones_off, int32_t B,
                                       int32_t* out_arcs, int32_t* out_aux, int64_t cap, int32_t* out_arc_off,
                                       int32_t* out_num_states) {
    if (!topo_arcs || !topo_aux || !index || nA < 1 || nS < 2 || B < 0 || !phones_off || !out_arc_off || !out_num_states)
        return wf_fail(WF_EINVAL, "wf_compose_linear_batch: bad argument");
    if (ctc_cross && (!phone_state || !phone_token || num_phones < 1))
        return wf_fail(WF_EINVAL, "wf_compose_linear_batch: ctc_cross needs the phone -> state / token tables");
    const TopoView tv{topo_arcs, topo_aux, nA, nS, index, ctc_cross, phone_state, phone_token, num_phones};
    static thread_local ComposeScratch X;
    int64_t w = 0;
    bool fits = out_arcs && out_aux;
    out_arc_off[0] = 0;
    for (int b = 0; b < B; ++b) {
        const int L = phones_off[b + 1] - phones_off[b];
        if (L < 0 || (L > 0 && !phones)) return wf_fail(WF_EINVAL, "wf_compose_linear_batch: transcript %d", b);
        int rc = compose_linear_one(tv, phones + phones_off[b], L, b, X);
        if (rc) return rc;
        const int64_t n = (int64_t)X.aux.size();
        if (w + n > INT32_MAX) return wf_fail(WF_EUNSUPPORTED, "wf_compose_linear_batch: more than 2^31 arcs");
        if (fits && w + n <= cap) compose_store(X, out_arcs + 4 * (size_t)w, out_aux + w);
        else fits = false;
        w += n;
        out_arc_off[b + 1] = (int32_t)w;
        out_num_states[b] = X.num_states;
    }
    if (!fits) return wf_fail(WF_ENOSPC, "wf_compose_linear_batch: arc capacity %lld < %lld", (long long)cap, (long long)w);
    return WF_OK;
}

// compose(topo, linear(phones) with epsilon self-loops), treat_epsilons_specially=False, for ONE transcript against a
// topology given as plain arc arrays (the index of wf_topo_index_build is built for the call): the per-utterance form of
// wf_compose_linear_batch, same graph.  Product states (q, i) numbered position-major (renumber_position_major); arcs
// grouped by source state in the topology's arc order; the super-final state is last.
extern "C" int wf_compose_linear(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA,
                                 int32_t nS, const int32_t* phones, int32_t L, int32_t* out_arcs,
                                 int32_t* out_aux, int32_t* out_num_arcs, int32_t* out_num_states,
                                 int32_t cap) {
    if (!topo_arcs || !topo_aux || nA < 1 || nS < 2 || L < 0 || (!phones && L > 0) || !out_num_arcs || !out_num_states)
        return wf_fail(WF_EINVAL, "wf_compose_linear: bad argument");
    int64_t words = 0;
    int rc = wf_topo_index_words(topo_arcs, topo_aux, nA, nS, &words);
    if (rc) return rc;
    std::vector<int32_t> index((size_t)words);
    if ((rc = wf_topo_index_build(topo_arcs, topo_aux, nA, nS, index.data()))) return rc;
    const TopoView tv{topo_arcs, topo_aux, nA, nS, index.data(), 0, nullptr, nullptr, 0};
    ComposeScratch X;
    if ((rc = compose_linear_one(tv, phones, L, 0, X))) return rc;
    const int n = (int)X.aux.size();
    *out_num_arcs = n;
    *out_num_states = X.num_states;
    if (!out_arcs) return WF_OK;
    if (cap < n || !out_aux) return wf_fail(WF_ENOSPC, "wf_compose_linear: arc capacity %d < %d", cap, n);
    compose_store(X, out_arcs, out_aux);
    return WF_OK;
}

// ------------------------------------------------------------------------------------------
// Host worker pool: the numerator compiler's utterances are independent, so a batch is split over a few persistent
// threads (created on first use, parked on a condition variable in between).  One parallel region at a time; a caller
// that finds the pool busy runs its region alone on its own thread.
// ------------------------------------------------------------------------------------------
namespace {
class WorkerPool {
  public:
    static WorkerPool& get() { static WorkerPool p; return p; }
    // runs fn(worker, nworkers) on `want` workers (the caller is worker 0) and waits for all of them
    void run(int want, const std::function<void(int, int)>& fn) {
        want = std::max(1, std::min(want, kMax));
        if (pid_ != getpid()) {
            // A child forked after the parent used the pool (a DataLoader worker) inherits the pool's memory but none of
            // its threads, and condition variables that still count the parent's sleeping workers as waiters (a notify
            // would wait for them forever): leave that state behind untouched and start this process's own.
            st_ = new State();
            pid_ = getpid();
        }
        State& S = *st_;
        std::unique_lock<std::mutex> own(S.region, std::try_to_lock);
        if (want == 1 || !own.owns_lock()) { fn(0, 1); return; }
        {
            std::lock_guard<std::mutex> lk(S.m);
            while ((int)S.threads.size() < want - 1) {
                const int id = (int)S.threads.size() + 1;
                State* sp = st_;
                S.threads.emplace_back([sp, id] { loop(*sp, id); });
            }
            S.fn = &fn; S.n = want; S.pending = want - 1; ++S.gen;
        }
        S.cv.notify_all();
        fn(0, want);
        std::unique_lock<std::mutex> lk(S.m);
        S.done.wait(lk, [&S] { return S.pending == 0; });
        S.fn = nullptr;
    }
    ~WorkerPool() {
        if (pid_ != getpid()) return;   // (a forked child that never used the pool: nothing of its own to stop)
        State& S = *st_;
        { std::lock_guard<std::mutex> lk(S.m); S.stop = true; }
        S.cv.notify_all();
        for (auto& t : S.threads) t.join();
    }
  private:
    static constexpr int kMax = 16;
    struct State {
        std::mutex m, region;
        std::condition_variable cv, done;
        std::vector<std::thread> threads;
        const std::function<void(int, int)>* fn = nullptr;
        int n = 0, pending = 0;
        uint64_t gen = 0;
        bool stop = false;
    };
    static void loop(State& S, int id) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)>* fn = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> lk(S.m);
                S.cv.wait(lk, [&] { return S.stop || S.gen != seen; });
                if (S.stop) return;
                seen = S.gen;
                if (id < S.n) { fn = S.fn; n = S.n; }
            }
            if (fn) {
                (*fn)(id, n);
                std::lock_guard<std::mutex> lk(S.m);
                if (--S.pending == 0) S.done.notify_one();
            }
        }
    }
    State* st_ = new State();
    pid_t pid_ = getpid();
};
}  // namespace

// ------------------------------------------------------------------------------------------
// wf_numerator_compile: wf_compose_linear_batch + wf_graph_prepare in one call, the utterances spread over
// `nthreads` host threads.  What a training step needs between the data loader and the loss call
// (models.py:263-270 / datapipe.py:233-256 run on every step).  Results are identical to the two calls in sequence,
// except that the blob keeps every graph at the start of its upper-bound region (gaps between graphs; B[WF_H_WORDS]
// = the whole extent), so that the threads never wait for each other's sizes.
// Two-call protocol as for wf_compose_linear_batch: too small an arc capacity or blob returns WF_ENOSPC with the sizes
// needed in out_arc_off[B] and *blob_bytes_needed.
// ------------------------------------------------------------------------------------------
extern "C" int wf_numerator_compile(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS,
                                    const int32_t* index, int32_t ctc_cross, const int32_t* phone_state,
                                    const int32_t* phone_token, int32_t num_phones,
                                    const int32_t* phones, const int32_t* phones_off, int32_t B,
                                    int32_t* out_arcs, int32_t* out_aux, int64_t cap, int32_t* out_arc_off,
                                    int32_t* out_num_states, void* blob, size_t blob_bytes, size_t* blob_bytes_needed,
                                    int32_t nthreads) {
    if (!topo_arcs || !topo_aux || !index || nA < 1 || nS < 2 || B < 1 || !phones_off || !out_arc_off || !out_num_states ||
        !out_arcs || !out_aux || !blob || !blob_bytes_needed)
        return wf_fail(WF_EINVAL, "wf_numerator_compile: bad argument");
    if (ctc_cross && (!phone_state || !phone_token || num_phones < 1))
        return wf_fail(WF_EINVAL, "wf_numerator_compile: ctc_cross needs the phone -> state / token tables");
    for (int b = 0; b < B; ++b)
        if (phones_off[b + 1] < phones_off[b] || (phones_off[b + 1] > phones_off[b] && !phones))
            return wf_fail(WF_EINVAL, "wf_numerator_compile: transcript %d", b);
    const TopoView tv{topo_arcs, topo_aux, nA, nS, index, ctc_cross, phone_state, phone_token, num_phones};
    const int nw = std::max(1, std::min<int>(std::min<int>(nthreads, B), 16));
    // phase 1: every worker composes its (contiguous) share into its own buffers
    struct Share { std::vector<int32_t> arcs, aux, narcs, nstates, pos; int rc = 0; char err[512] = ""; };
    std::vector<Share> shares(nw);
    auto range = [&](int w, int n, int* lo, int* hi) { *lo = (int)((int64_t)B * w / n); *hi = (int)((int64_t)B * (w + 1) / n); };
    WorkerPool::get().run(nw, [&](int w, int n) {
        // (a region that found the pool busy runs with n = 1: worker 0 then takes every share)
        for (int sh = w; sh < nw; sh += n) {
            Share& S = shares[sh];
            int lo, hi;
            range(sh, nw, &lo, &hi);
            static thread_local ComposeScratch X;
            for (int b = lo; b < hi && !S.rc; ++b) {
                S.rc = compose_linear_one(tv, phones + phones_off[b], phones_off[b + 1] - phones_off[b], b, X);
                if (S.rc) { snprintf(S.err, sizeof(S.err), "%s", wf_last_error()); break; }
                const size_t at = S.aux.size();
                S.arcs.resize(4 * (at + X.aux.size()));
                S.aux.resize(at + X.aux.size());
                compose_store(X, S.arcs.data() + 4 * at, S.aux.data() + at);
                S.narcs.push_back((int32_t)X.aux.size());
                S.nstates.push_back(X.num_states);
                S.pos.insert(S.pos.end(), X.pos.begin(), X.pos.end());   // num_states - 1 entries: the transcript positions
            }
        }
    });
    for (auto& S : shares) if (S.rc) return wf_fail(S.rc, "%s", S.err);
    // sizes: arc offsets, blob regions (upper bound per graph)
    int64_t w = 0;
    out_arc_off[0] = 0;
    std::vector<size_t> start(B + 1);
    start[0] = pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * B);
    for (int sh = 0, b = 0; sh < nw; ++sh)
        for (size_t j = 0; j < shares[sh].narcs.size(); ++j, ++b) {
            w += shares[sh].narcs[j];
            if (w > INT32_MAX) return wf_fail(WF_EUNSUPPORTED, "wf_numerator_compile: more than 2^31 arcs");
            out_arc_off[b + 1] = (int32_t)w;
            out_num_states[b] = shares[sh].nstates[j];
            start[b + 1] = start[b] + fsa_words_bound((size_t)shares[sh].nstates[j], (size_t)shares[sh].narcs[j]);
        }
    *blob_bytes_needed = start[B] * 4;
    if (start[B] >= (size_t)1 << 31) return wf_fail(WF_EUNSUPPORTED, "prepared graph exceeds 2^31 words");
    if (w > cap || start[B] * 4 > blob_bytes)
        return wf_fail(WF_ENOSPC, "wf_numerator_compile: needs %lld arcs (capacity %lld) and a %zu B blob (%zu B given)", (long long)w,
                       (long long)cap, start[B] * 4, blob_bytes);
    // phase 2: arcs to their place, then each graph's tables
    int32_t* Bw = (int32_t*)blob;
    wf_graph_host gh{B, out_arcs, out_arc_off, out_num_states};
    std::vector<int> maxlab(B, -1), rcs(nw, 0);
    std::vector<std::string> errs(nw);
    WorkerPool::get().run(nw, [&](int wk, int n) {
        for (int sh = wk; sh < nw; sh += n) {
            int lo, hi;
            range(sh, nw, &lo, &hi);
            const Share& S = shares[sh];
            if (hi > lo) {
                memcpy(out_arcs + 4 * (size_t)out_arc_off[lo], S.arcs.data(), sizeof(int32_t) * S.arcs.size());
                memcpy(out_aux + out_arc_off[lo], S.aux.data(), sizeof(int32_t) * S.aux.size());
            }
            size_t pat = 0;
            for (int b = lo; b < hi && !rcs[sh]; ++b) {
                size_t cur = start[b];
                rcs[sh] = prepare_one(&gh, b, Bw, Bw + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * b, cur, start[b + 1], &maxlab[b],
                                      S.pos.data() + pat);
                pat += (size_t)out_num_states[b] - 1;
                if (rcs[sh]) errs[sh] = wf_last_error();
            }
        }
    });
    for (int sh = 0; sh < nw; ++sh) if (rcs[sh]) return wf_fail(rcs[sh], "%s", errs[sh].c_str());
    memset(Bw, 0, sizeof(int32_t) * WF_HDR_WORDS);
    Bw[WF_H_MAGIC] = WF_BLOB_MAGIC;
    Bw[WF_H_NUM_FSAS] = B;
    Bw[WF_H_WORDS] = (int32_t)start[B];
    int gmaxlab = -1, gmaxS = 0, gmaxA = 0;
    for (int b = 0; b < B; ++b) {
        const int32_t* I = Bw + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * b;
        gmaxlab = std::max(gmaxlab, maxlab[b]); gmaxS = std::max(gmaxS, (int)I[WF_I_S]); gmaxA = std::max(gmaxA, (int)I[WF_I_A]);
    }
    Bw[WF_H_MAX_LABEL] = gmaxlab; Bw[WF_H_MAX_STATES] = gmaxS; Bw[WF_H_MAX_ARCS] = gmaxA;
    return WF_OK;
}

// ------------------------------------------------------------------------------------------
// compose(topo, b, treat_epsilons_specially=False) for ARBITRARY epsilon-free right-hand graphs b (one per utterance):
// the phone-level transcript graphs that a lexicon with alternative pronunciations / optional silence gives
// (datapipe.py:244-254: L_inv o words, inverted, epsilons removed, epsilon self-loops added).  b: label = phone >= 1,
// aux = word id (or NULL: the phone), label -1 arcs into its last state, arcs sorted by source state.  Product states
// (q, r) in BFS order from (0, 0); per product state the topology's arcs in order and, per topology arc with phone p,
// the arcs of r labelled p in b's order (phone 0 pairs with the implicit self-loop of r).  Output label = token,
// aux = b's aux, score = the fp32 sum; a product final arc needs a final arc on both sides.
// For a linear b the result is wf_compose_linear_batch's.
// ------------------------------------------------------------------------------------------
extern "C" int wf_compose_fsa_batch(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS,
                                    const int32_t* index, const int32_t* b_arcs, const int32_t* b_aux,
                                    const int32_t* b_arc_off, const int32_t* b_num_states, int32_t B,
                                    int32_t* out_arcs, int32_t* out_aux, int64_t cap, int32_t* out_arc_off,
                                    int32_t* out_num_states) {
    if (!topo_arcs || !topo_aux || !index || nA < 1 || nS < 2 || B < 0 || !b_arc_off || !b_num_states || !out_arc_off || !out_num_states ||
        (B > 0 && b_arc_off[B] > 0 && !b_arcs))
        return wf_fail(WF_EINVAL, "wf_compose_fsa_batch: bad argument");
    const int32_t* soff = index;
    const int32_t* perm = index + nS + 1;
    const int32_t* key = perm + nA;
    ComposeScratch X;
    PairMap& id = X.id;
    std::vector<int64_t>& order = X.order;
    std::vector<int32_t>& arcs = X.arcs;
    std::vector<int32_t>& aux = X.aux;
    std::vector<int32_t> boff, cand, seen;
    int64_t w = 0;
    bool fits = out_arcs && out_aux;
    out_arc_off[0] = 0;
    auto fadd = [](int32_t xb, int32_t yb) { float x, y; memcpy(&x, &xb, 4); memcpy(&y, &yb, 4); x += y; int32_t r; memcpy(&r, &x, 4); return r; };
    for (int b = 0; b < B; ++b) {
        const int32_t* ba = b_arcs + 4 * (size_t)b_arc_off[b];
        const int32_t* bx = b_aux ? b_aux + b_arc_off[b] : nullptr;
        const int nb = b_arc_off[b + 1] - b_arc_off[b], R = b_num_states[b];
        if (nb < 0 || R < 2) return wf_fail(WF_EINVAL, "wf_compose_fsa_batch: graph %d: needs a start and a final state", b);
        boff.assign(R + 1, 0);
        for (int e = 0; e < nb; ++e) {
            const int s = ba[4 * e], d = ba[4 * e + 1], l = ba[4 * e + 2];
            if (s < 0 || s >= R || d < 0 || d >= R) return wf_fail(WF_ERANGE, "wf_compose_fsa_batch: graph %d arc %d: state out of range", b, e);
            if (e && s < ba[4 * (e - 1)]) return wf_fail(WF_EINVAL, "wf_compose_fsa_batch: graph %d: arcs must be sorted by source state", b);
            if (l == 0 || l < -1) return wf_fail(WF_ERANGE, "wf_compose_fsa_batch: graph %d arc %d: label %d (epsilon-free graphs only)", b, e, l);
            boff[s + 1]++;
        }
        for (int s = 0; s < R; ++s) boff[s + 1] += boff[s];
        id.reset(8 * (size_t)(R + 2));
        order.clear(); arcs.clear(); aux.clear(); X.head_start.clear();
        auto pkey = [&](int q, int r) { return (int64_t)q * R + r; };
        bool ins;
        id.get(pkey(0, 0), 0, &ins);
        order.push_back(pkey(0, 0));
        auto emit = [&](int head, int q2, int r2, int tok, int32_t wbits, int ax) {
            if (order.size() * 2 > id.k.size()) id.grow();
            const int32_t nid = id.get(pkey(q2, r2), (int32_t)order.size(), &ins);
            if (ins) order.push_back(pkey(q2, r2));
            arcs.push_back(head); arcs.push_back(nid); arcs.push_back(tok); arcs.push_back(wbits);
            aux.push_back(ax);
        };
        for (size_t head = 0; head < order.size(); ++head) {
            X.head_start.push_back((int32_t)aux.size());
            const int q = (int)(order[head] / R), r = (int)(order[head] % R);
            const int lo = soff[q], hi = soff[q + 1];
            const int32_t* ke = key + hi;
            const int f1 = (int)(std::upper_bound(key + lo, ke, -1) - key);
            const int z1 = (int)(std::upper_bound(key + f1, ke, 0) - key);
            // candidate topology arcs: phone 0, and the phones on r's arcs; visited in topology arc order
            cand.clear(); seen.clear();
            for (int x = f1; x < z1; ++x) cand.push_back(perm[x]);
            for (int e = boff[r]; e < boff[r + 1]; ++e) {
                const int l = ba[4 * e + 2];
                if (l < 0 || std::find(seen.begin(), seen.end(), l) != seen.end()) continue;
                seen.push_back(l);
                const int p0 = (int)(std::lower_bound(key + z1, ke, l) - key), p1 = (int)(std::upper_bound(key + p0, ke, l) - key);
                for (int y = p0; y < p1; ++y) cand.push_back(perm[y]);
            }
            std::sort(cand.begin(), cand.end());
            for (int a : cand) {
                const int32_t* t = topo_arcs + 4 * (size_t)a;
                const int ph = topo_aux[a];
                if (ph == 0) { emit((int)head, t[1], r, t[2], t[3], 0); continue; }
                for (int e = boff[r]; e < boff[r + 1]; ++e)
                    if (ba[4 * e + 2] == ph) emit((int)head, t[1], ba[4 * e + 1], t[2], fadd(t[3], ba[4 * e + 3]), bx ? bx[e] : ph);
            }
            for (int j = lo; j < f1; ++j)   // final arcs: both sides must be final here
                for (int e = boff[r]; e < boff[r + 1]; ++e)
                    if (ba[4 * e + 2] < 0) {
                        arcs.push_back((int32_t)head); arcs.push_back(-1); arcs.push_back(-1);
                        arcs.push_back(fadd(topo_arcs[4 * (size_t)perm[j] + 3], ba[4 * e + 3]));
                        aux.push_back(-1);
                    }
        }
        X.head_start.push_back((int32_t)aux.size());
        renumber_position_major(X, (int64_t)R);      // (right-hand state, topology state) order, as for linear transcripts
        const int sf = (int)order.size();
        const int64_t n = (int64_t)aux.size();
        if (w + n > INT32_MAX) return wf_fail(WF_EUNSUPPORTED, "wf_compose_fsa_batch: more than 2^31 arcs");
        if (fits && w + n <= cap) {
            for (int64_t a = 0; a < n; ++a) {
                int32_t* o = out_arcs + 4 * (size_t)(w + a);
                o[0] = arcs[4 * a]; o[1] = arcs[4 * a + 1] < 0 ? sf : arcs[4 * a + 1]; o[2] = arcs[4 * a + 2]; o[3] = arcs[4 * a + 3];
                out_aux[w + a] = aux[a];
            }
        } else {
            fits = false;
        }
        w += n;
        out_arc_off[b + 1] = (int32_t)w;
        out_num_states[b] = sf + 1;
    }
    if (!fits) return wf_fail(WF_ENOSPC, "wf_compose_fsa_batch: arc capacity %lld < %lld", (long long)cap, (long long)w);
    return WF_OK;
}
