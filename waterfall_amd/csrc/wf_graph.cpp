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
#include <deque>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "../../include/wf_dwfst.h"
#include "wf_internal.h"

static thread_local char g_err[512] = "";

extern "C" const char* wf_last_error(void) { return g_err; }
extern "C" int wf_version(void) { return 1; }

int wf_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

struct FsaPlan {
    int S = 0, A = 0, NF = 0, K = 0, direct = 0, max_label = -1;
    std::vector<int> em;   // emitting arc ids (local)
    std::vector<int> fin;  // final arc ids (local)
    size_t words = 0;
};

inline size_t pad4(size_t w) { return (w + 3) & ~size_t(3); }

// Validates FSA g and computes its size.
int plan_fsa(const wf_graph_host* gh, int g, FsaPlan* p, std::vector<int>* labels_sorted) {
    const int a0 = gh->fsa_arc_offsets[g], a1 = gh->fsa_arc_offsets[g + 1];
    const int S = gh->fsa_num_states[g];
    if (a1 < a0) return wf_fail(WF_EINVAL, "fsa %d: arc offsets not monotone", g);
    if (S < 2) return wf_fail(WF_EINVAL, "fsa %d: needs at least a start and a final state (S=%d)", g, S);
    p->S = S;
    std::vector<int> labs;
    for (int a = a0; a < a1; ++a) {
        const int32_t* r = gh->arcs + 4 * (size_t)a;
        if (r[0] < 0 || r[0] >= S || r[1] < 0 || r[1] >= S)
            return wf_fail(WF_ERANGE, "fsa %d arc %d: state out of range (%d -> %d, S=%d)", g, a - a0, r[0], r[1], S);
        if (r[2] < -1) return wf_fail(WF_ERANGE, "fsa %d arc %d: label %d < -1", g, a - a0, r[2]);
        if (r[2] == -1) {
            if (r[1] != S - 1) return wf_fail(WF_EINVAL, "fsa %d arc %d: label -1 arc must enter the final state", g, a - a0);
            p->fin.push_back(a - a0);
        } else {
            if (r[1] == S - 1) return wf_fail(WF_EINVAL, "fsa %d arc %d: only label -1 arcs may enter the final state", g, a - a0);
            if (r[0] == S - 1) return wf_fail(WF_EINVAL, "fsa %d arc %d: arc leaves the final state", g, a - a0);
            p->em.push_back(a - a0);
            labs.push_back(r[2]);
            p->max_label = std::max(p->max_label, (int)r[2]);
        }
    }
    p->A = (int)p->em.size();
    p->NF = (int)p->fin.size();
    std::sort(labs.begin(), labs.end());
    labs.erase(std::unique(labs.begin(), labs.end()), labs.end());
    // direct columns when the graph uses at least half of the label range (topologies used as
    // denominators use every token): emissions are then read straight from log_probs.
    p->direct = (p->max_label >= 0 && (size_t)labs.size() * 2 >= (size_t)p->max_label + 1) ? 1 : 0;
    p->K = p->direct ? p->max_label + 1 : (int)labs.size();
    if (labels_sorted) *labels_sorted = labs;
    size_t w = 0;
    w += pad4(S + 1) + 4 * pad4(p->A);          // in_off, in_src, in_col, in_w, in_arc
    w += pad4(S + 1) + 3 * pad4(p->A);          // out_off, out_dst, out_col, out_w
    w += pad4(p->K + 1) + 4 * pad4(p->A);       // lab_off, lab_src, lab_dst, lab_w, lab_col
    w += pad4(p->K);                            // labels
    w += 3 * pad4(p->NF);                       // fin_src, fin_w, fin_arc
    w += 2 * pad4(S);                           // heavy_in, heavy_out (upper bound)
    w += 2 * pad4(S) + pad4(p->A + p->NF);      // finb_log, finb_max, arc_src
    if (S - 1 <= 64 * WF_BAND_MAX_R)   // banded form is only built for graphs one wavefront can hold
        w += pad4(S) + 2 * pad4((size_t)S * (WF_BAND_MAX_BW + 1) * WF_BAND_MAX_M * 2);   // order + two tables (upper bound)
    p->words = w;
    return WF_OK;
}

}  // namespace

extern "C" int wf_graph_prepare_bytes(const wf_graph_host* g, size_t* bytes) {
    if (!g || !bytes || g->num_fsas < 1 || !g->arcs || !g->fsa_arc_offsets || !g->fsa_num_states)
        return wf_fail(WF_EINVAL, "wf_graph_prepare_bytes: null or empty graph");
    size_t w = WF_HDR_WORDS + (size_t)WF_INFO_WORDS * g->num_fsas;
    w = pad4(w);
    for (int i = 0; i < g->num_fsas; ++i) {
        FsaPlan p;
        int rc = plan_fsa(g, i, &p, nullptr);
        if (rc) return rc;
        w += p.words;
    }
    if (w >= (size_t)1 << 31) return wf_fail(WF_EUNSUPPORTED, "prepared graph exceeds 2^31 words");
    *bytes = w * 4;
    return WF_OK;
}

extern "C" int wf_graph_prepare(const wf_graph_host* gh, void* blob_host, size_t bytes) {
    size_t need = 0;
    int rc = wf_graph_prepare_bytes(gh, &need);
    if (rc) return rc;
    if (!blob_host || bytes < need) return wf_fail(WF_ENOSPC, "wf_graph_prepare: blob too small (%zu < %zu)", bytes, need);
    int32_t* B = (int32_t*)blob_host;
    memset(B, 0, need);
    B[WF_H_MAGIC] = WF_BLOB_MAGIC;
    B[WF_H_NUM_FSAS] = gh->num_fsas;
    B[WF_H_WORDS] = (int32_t)(need / 4);
    int gmaxlab = -1, gmaxS = 0, gmaxA = 0;
    // plans (validation, sizes) first: every graph's region of the blob is known before its tables are built
    const int G = gh->num_fsas;
    std::vector<FsaPlan> plans(G);
    std::vector<std::vector<int>> all_labs(G);
    std::vector<size_t> start(G + 1);
    start[0] = pad4(WF_HDR_WORDS + (size_t)WF_INFO_WORDS * G);
    for (int g = 0; g < G; ++g) {
        rc = plan_fsa(gh, g, &plans[g], &all_labs[g]);
        if (rc) return rc;
        start[g + 1] = start[g] + plans[g].words;
        gmaxlab = std::max(gmaxlab, plans[g].max_label); gmaxS = std::max(gmaxS, plans[g].S); gmaxA = std::max(gmaxA, plans[g].A);
    }
    if (start[G] * 4 > need) return wf_fail(WF_ENOSPC, "internal: blob overflow");
    for (int g = 0; g < G; ++g) {
        const FsaPlan& p = plans[g];
        const std::vector<int>& labs = all_labs[g];
        size_t cur = start[g];
        int32_t* I = B + WF_HDR_WORDS + (size_t)WF_INFO_WORDS * g;
        const int32_t* arcs = gh->arcs + 4 * (size_t)gh->fsa_arc_offsets[g];
        const int S = p.S, A = p.A;
        auto take = [&](size_t n) { size_t o = cur; cur += pad4(n); return (int32_t)o; };
        I[WF_I_S] = S; I[WF_I_A] = A; I[WF_I_NF] = p.NF; I[WF_I_K] = p.K; I[WF_I_DIRECT] = p.direct;
        I[WF_I_ARC_BASE] = gh->fsa_arc_offsets[g];
        I[WF_I_MAX_LABEL] = p.max_label;
        auto col_of = [&](int label) {
            if (p.direct) return label;
            return (int)(std::lower_bound(labs.begin(), labs.end(), label) - labs.begin());
        };
        // ---- incoming CSR: stable sort of emitting arcs by dst
        {
            int o_off = take(S + 1), o_src = take(A), o_col = take(A), o_w = take(A), o_arc = take(A);
            I[WF_I_IN_OFF] = o_off; I[WF_I_IN_SRC] = o_src; I[WF_I_IN_COL] = o_col; I[WF_I_IN_W] = o_w; I[WF_I_IN_ARC] = o_arc;
            std::vector<int> cnt(S + 1, 0);
            for (int a : p.em) cnt[arcs[4 * a + 1] + 1]++;
            for (int s = 0; s < S; ++s) cnt[s + 1] += cnt[s];
            memcpy(B + o_off, cnt.data(), sizeof(int) * (S + 1));
            std::vector<int> pos(cnt.begin(), cnt.end() - 1);
            int maxin = 0;
            for (int a : p.em) {
                int d = arcs[4 * a + 1];
                int q = pos[d]++;
                B[o_src + q] = arcs[4 * a + 0];
                B[o_col + q] = col_of(arcs[4 * a + 2]);
                B[o_w + q] = arcs[4 * a + 3];
                B[o_arc + q] = a;
            }
            int o_h = take(S), nh = 0;
            for (int s = 0; s < S; ++s) {
                int deg = cnt[s + 1] - cnt[s];
                maxin = std::max(maxin, deg);
                if (deg > WF_HEAVY_DEG) B[o_h + nh++] = s;
            }
            I[WF_I_HEAVY_IN] = o_h; I[WF_I_NHEAVY_IN] = nh; I[WF_I_MAXIN] = maxin;
            int nover = 0, over = 0;
            for (int s = 0; s < S; ++s) if (cnt[s + 1] - cnt[s] > WF_REG_DEG) { ++nover; over = s; }
            I[WF_I_NOVER_IN] = nover; I[WF_I_OVER_IN] = over;
        }
        // ---- outgoing CSR: emitting arcs in original order grouped by src (stable)
        {
            int o_off = take(S + 1), o_dst = take(A), o_col = take(A), o_w = take(A);
            I[WF_I_OUT_OFF] = o_off; I[WF_I_OUT_DST] = o_dst; I[WF_I_OUT_COL] = o_col; I[WF_I_OUT_W] = o_w;
            std::vector<int> cnt(S + 1, 0);
            for (int a : p.em) cnt[arcs[4 * a + 0] + 1]++;
            for (int s = 0; s < S; ++s) cnt[s + 1] += cnt[s];
            memcpy(B + o_off, cnt.data(), sizeof(int) * (S + 1));
            std::vector<int> pos(cnt.begin(), cnt.end() - 1);
            int maxout = 0;
            for (int a : p.em) {
                int s = arcs[4 * a + 0];
                int q = pos[s]++;
                B[o_dst + q] = arcs[4 * a + 1];
                B[o_col + q] = col_of(arcs[4 * a + 2]);
                B[o_w + q] = arcs[4 * a + 3];
            }
            int o_h = take(S), nh = 0;
            for (int s = 0; s < S; ++s) {
                int deg = cnt[s + 1] - cnt[s];
                maxout = std::max(maxout, deg);
                if (deg > WF_HEAVY_DEG) B[o_h + nh++] = s;
            }
            I[WF_I_HEAVY_OUT] = o_h; I[WF_I_NHEAVY_OUT] = nh; I[WF_I_MAXOUT] = maxout;
            int nover = 0, over = 0;
            for (int s = 0; s < S; ++s) if (cnt[s + 1] - cnt[s] > WF_REG_DEG) { ++nover; over = s; }
            I[WF_I_NOVER_OUT] = nover; I[WF_I_OVER_OUT] = over;
        }
        // ---- label(column)-sorted arcs
        {
            const int K = p.K;
            int o_off = take(K + 1), o_src = take(A), o_dst = take(A), o_w = take(A), o_lab = take(K), o_lcol = take(A);
            I[WF_I_LAB_COL] = o_lcol;
            I[WF_I_LAB_OFF] = o_off; I[WF_I_LAB_SRC] = o_src; I[WF_I_LAB_DST] = o_dst; I[WF_I_LAB_W] = o_w; I[WF_I_LABELS] = o_lab;
            std::vector<int> cnt(K + 1, 0);
            for (int a : p.em) cnt[col_of(arcs[4 * a + 2]) + 1]++;
            int maxdeg = 0;
            for (int k = 0; k < K; ++k) { maxdeg = std::max(maxdeg, cnt[k + 1]); cnt[k + 1] += cnt[k]; }
            memcpy(B + o_off, cnt.data(), sizeof(int) * (K + 1));
            std::vector<int> pos(cnt.begin(), cnt.end() - 1);
            for (int a : p.em) {
                int q = pos[col_of(arcs[4 * a + 2])]++;
                B[o_src + q] = arcs[4 * a + 0];
                B[o_dst + q] = arcs[4 * a + 1];
                B[o_w + q] = arcs[4 * a + 3];
                B[o_lcol + q] = col_of(arcs[4 * a + 2]);
            }
            for (int k = 0; k < K; ++k) B[o_lab + k] = p.direct ? k : labs[k];
            I[WF_I_MAXLAB_DEG] = maxdeg;
        }
        // ---- final arcs
        {
            int o_src = take(p.NF), o_w = take(p.NF), o_arc = take(p.NF);
            I[WF_I_FIN_SRC] = o_src; I[WF_I_FIN_W] = o_w; I[WF_I_FIN_ARC] = o_arc;
            for (int i = 0; i < p.NF; ++i) {
                int a = p.fin[i];
                B[o_src + i] = arcs[4 * a + 0];
                B[o_w + i] = arcs[4 * a + 3];
                B[o_arc + i] = a;
            }
        }
        // ---- beta_T per state and the source state of every original arc
        {
            int o_fl = take(S), o_fm = take(S), o_as = take(A + p.NF);
            I[WF_I_FINB_LOG] = o_fl; I[WF_I_FINB_MAX] = o_fm; I[WF_I_ARC_SRC] = o_as;
            std::vector<double> m(S, -INFINITY), sum(S, 0.0);
            for (int a : p.fin) {
                float w; memcpy(&w, &arcs[4 * a + 3], 4);
                m[arcs[4 * a]] = std::max(m[arcs[4 * a]], (double)w);
            }
            for (int a : p.fin) {
                float w; memcpy(&w, &arcs[4 * a + 3], 4);
                sum[arcs[4 * a]] += std::exp((double)w - m[arcs[4 * a]]);
            }
            for (int s = 0; s < S; ++s) {
                float fl = (m[s] == -INFINITY) ? -INFINITY : (float)(m[s] + std::log(sum[s]));
                float fm = (float)m[s];
                memcpy(&B[o_fl + s], &fl, 4);
                memcpy(&B[o_fm + s], &fm, 4);
            }
            for (int a = 0; a < A + p.NF; ++a) B[o_as + a] = arcs[4 * a];
        }
        // ---- fewest frames any accepted path needs (BFS over emitting arcs from the start state)
        {
            std::vector<int> dist(S, -1);
            std::vector<char> fin_src(S, 0);
            for (int a : p.fin) fin_src[arcs[4 * a]] = 1;
            const int32_t* ooff = B + I[WF_I_OUT_OFF];
            const int32_t* odst = B + I[WF_I_OUT_DST];
            std::deque<int> q;
            dist[0] = 0; q.push_back(0);
            int minlen = INT32_MAX;
            while (!q.empty()) {
                int s = q.front(); q.pop_front();
                if (fin_src[s]) { minlen = dist[s]; break; }   // BFS: the first final-capable state is the nearest
                for (int e = ooff[s]; e < ooff[s + 1]; ++e)
                    if (dist[odst[e]] < 0) { dist[odst[e]] = dist[s] + 1; q.push_back(odst[e]); }
            }
            I[WF_I_MINLEN] = minlen;
        }
        // ---- banded form: Kahn topological order (FIFO, self-loops ignored) of the non-final states
        {
            const int Sn = S - 1;   // the super-final state has no emitting arcs
            I[WF_I_BAND_BW] = -1; I[WF_I_BAND_M] = 0;
            {
                int we = 1;
                for (int a : p.em) { float w; memcpy(&w, &arcs[4 * a + 3], 4); if (w != 0.f) we = 0; }
                I[WF_I_BAND_WE] = we;
            }
            std::vector<int> indeg(Sn, 0), order, pos(Sn, -1);
            const int32_t* ooff = B + I[WF_I_OUT_OFF];
            const int32_t* odst = B + I[WF_I_OUT_DST];
            for (int s = 0; s < Sn; ++s)
                for (int e = ooff[s]; e < ooff[s + 1]; ++e)
                    if (odst[e] != s) indeg[odst[e]]++;
            std::deque<int> q;
            if (Sn <= 64 * WF_BAND_MAX_R)
                for (int s = 0; s < Sn; ++s) if (indeg[s] == 0) q.push_back(s);
            while (!q.empty()) {
                int s = q.front(); q.pop_front();
                pos[s] = (int)order.size(); order.push_back(s);
                for (int e = ooff[s]; e < ooff[s + 1]; ++e)
                    if (odst[e] != s && --indeg[odst[e]] == 0) q.push_back(odst[e]);
            }
            if ((int)order.size() == Sn && Sn >= 1 && Sn <= 64 * WF_BAND_MAX_R) {
                int bw = 0;
                for (int a : p.em) bw = std::max(bw, pos[arcs[4 * a + 1]] - pos[arcs[4 * a]]);
                // arcs per (destination position, distance) cell
                int m = 0;
                if (bw <= WF_BAND_MAX_BW) {
                    std::vector<int> cnt((size_t)Sn * (bw + 1), 0);
                    for (int a : p.em) m = std::max(m, ++cnt[(size_t)pos[arcs[4 * a + 1]] * (bw + 1) + (pos[arcs[4 * a + 1]] - pos[arcs[4 * a]])]);
                }
                if (bw <= WF_BAND_MAX_BW && m <= WF_BAND_MAX_M) {
                    m = std::max(m, 1);
                    const size_t cells = (size_t)Sn * (bw + 1) * m;
                    int o_ord = take(Sn), o_f = take(cells * 2), o_b = take(cells * 2);
                    I[WF_I_BAND_BW] = bw; I[WF_I_BAND_M] = m; I[WF_I_BAND_ORDER] = o_ord; I[WF_I_BANDF] = o_f; I[WF_I_BANDB] = o_b;
                    memcpy(B + o_ord, order.data(), sizeof(int) * Sn);
                    // empty cells: column 0, weight -inf (probability 0)
                    float ninf = -INFINITY; int32_t nbits; memcpy(&nbits, &ninf, 4);
                    for (size_t c = 0; c < cells; ++c) { B[o_f + 2 * c] = 0; B[o_f + 2 * c + 1] = nbits; B[o_b + 2 * c] = 0; B[o_b + 2 * c + 1] = nbits; }
                    std::vector<int> fillf((size_t)Sn * (bw + 1), 0), fillb((size_t)Sn * (bw + 1), 0);
                    for (int a : p.em) {
                        const int ps = pos[arcs[4 * a]], pd = pos[arcs[4 * a + 1]], d = pd - ps;
                        const int col = col_of(arcs[4 * a + 2]);
                        size_t cf = (size_t)pd * (bw + 1) + d;
                        size_t ef = cf * m + fillf[cf]++;
                        B[o_f + 2 * ef] = col; B[o_f + 2 * ef + 1] = arcs[4 * a + 3];
                        size_t cb = (size_t)(Sn - 1 - ps) * (bw + 1) + d;
                        size_t eb = cb * m + fillb[cb]++;
                        B[o_b + 2 * eb] = col; B[o_b + 2 * eb + 1] = arcs[4 * a + 3];
                    }
                }
            }
        }
        if (cur > start[g + 1]) return wf_fail(WF_ENOSPC, "internal: blob overflow in fsa %d", g);
    }
    B[WF_H_MAX_LABEL] = gmaxlab; B[WF_H_MAX_STATES] = gmaxS; B[WF_H_MAX_ARCS] = gmaxA;
    return WF_OK;
}

// compose(topo, linear(phones) with epsilon self-loops), treat_epsilons_specially=False.
// Product states (q, i) are numbered in BFS discovery order from (0, 0); arcs are emitted grouped
// by source state in the topology's arc order; the super-final state is last.
extern "C" int wf_compose_linear(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA,
                                 int32_t nS, const int32_t* phones, int32_t L, int32_t* out_arcs,
                                 int32_t* out_aux, int32_t* out_num_arcs, int32_t* out_num_states,
                                 int32_t cap) {
    if (!topo_arcs || !topo_aux || nA < 1 || nS < 2 || L < 0 || (!phones && L > 0) || !out_num_arcs || !out_num_states)
        return wf_fail(WF_EINVAL, "wf_compose_linear: bad argument");
    std::vector<int> off(nS + 1, 0);
    for (int a = 0; a < nA; ++a) {
        int s = topo_arcs[4 * a];
        if (s < 0 || s >= nS || topo_arcs[4 * a + 1] < 0 || topo_arcs[4 * a + 1] >= nS)
            return wf_fail(WF_ERANGE, "wf_compose_linear: topo arc %d state out of range", a);
        if (a && s < topo_arcs[4 * (a - 1)]) return wf_fail(WF_EINVAL, "wf_compose_linear: topo arcs must be sorted by source state");
        off[s + 1]++;
    }
    for (int s = 0; s < nS; ++s) off[s + 1] += off[s];
    // reachable product states are few (about 3L) next to nS*(L+1): hash map, not a dense table
    std::unordered_map<int64_t, int32_t> id;
    id.reserve(8 * (size_t)(L + 2));
    std::vector<int64_t> order;
    order.reserve(8 * (size_t)(L + 2));
    auto key = [&](int q, int i) { return (int64_t)q * (L + 1) + i; };
    id.emplace(key(0, 0), 0);
    order.push_back(key(0, 0));
    std::vector<int32_t> arcs, aux, fin;
    for (size_t head = 0; head < order.size(); ++head) {
        int q = (int)(order[head] / (L + 1)), i = (int)(order[head] % (L + 1));
        for (int a = off[q]; a < off[q + 1]; ++a) {
            const int32_t* r = topo_arcs + 4 * (size_t)a;
            int ph = topo_aux[a];
            if (r[2] < 0) {
                if (i == L) { fin.push_back((int32_t)head); fin.push_back(r[3]); }
                continue;
            }
            int ni;
            if (ph == 0) ni = i;
            else if (i < L && ph == phones[i]) ni = i + 1;
            else continue;
            int64_t k = key(r[1], ni);
            auto it = id.find(k);
            int32_t nid;
            if (it == id.end()) { nid = (int32_t)order.size(); id.emplace(k, nid); order.push_back(k); }
            else nid = it->second;
            arcs.push_back((int32_t)head); arcs.push_back(nid); arcs.push_back(r[2]); arcs.push_back(r[3]);
            aux.push_back(ph);
        }
    }
    const int sf = (int)order.size();
    // final arcs are appended after all emitting arcs but must stay grouped by source state:
    // merge them in at the end of their source state's run.
    const int nE = (int)aux.size(), nF = (int)fin.size() / 2;
    *out_num_arcs = nE + nF;
    *out_num_states = sf + 1;
    if (!out_arcs) return WF_OK;
    if (cap < nE + nF || !out_aux) return wf_fail(WF_ENOSPC, "wf_compose_linear: arc capacity %d < %d", cap, nE + nF);
    int w = 0, f = 0, e = 0;
    while (e < nE || f < nF) {
        int se = e < nE ? arcs[4 * e] : INT32_MAX;
        int sfn = f < nF ? fin[2 * f] : INT32_MAX;
        if (se <= sfn) {
            memcpy(out_arcs + 4 * (size_t)w, &arcs[4 * (size_t)e], 16);
            out_aux[w] = aux[e];
            ++e;
        } else {
            out_arcs[4 * w] = fin[2 * f]; out_arcs[4 * w + 1] = sf; out_arcs[4 * w + 2] = -1; out_arcs[4 * w + 3] = fin[2 * f + 1];
            out_aux[w] = -1;
            ++f;
        }
        ++w;
    }
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
extern "C" int wf_topo_index_words(int32_t nA, int32_t nS, int64_t* words) {
    if (nA < 1 || nS < 2 || !words) return wf_fail(WF_EINVAL, "wf_topo_index_words: bad argument");
    *words = (int64_t)nS + 1 + 2 * (int64_t)nA;
    return WF_OK;
}

extern "C" int wf_topo_index_build(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t nA, int32_t nS,
                                   int32_t* index) {
    if (!topo_arcs || !topo_aux || nA < 1 || nS < 2 || !index) return wf_fail(WF_EINVAL, "wf_topo_index_build: bad argument");
    int32_t* soff = index;
    int32_t* perm = index + nS + 1;
    int32_t* key = perm + nA;
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
    const int32_t* soff = index;
    const int32_t* perm = index + nS + 1;
    const int32_t* key = perm + nA;
    PairMap id;
    std::vector<int64_t> order;
    std::vector<int32_t> arcs, aux;
    int64_t w = 0;
    bool fits = out_arcs && out_aux;
    out_arc_off[0] = 0;
    for (int b = 0; b < B; ++b) {
        const int32_t* seq = phones + phones_off[b];
        const int L = phones_off[b + 1] - phones_off[b];
        if (L < 0 || (L > 0 && !phones)) return wf_fail(WF_EINVAL, "wf_compose_linear_batch: transcript %d", b);
        for (int i = 0; i < L; ++i)
            if (seq[i] < 1 || (ctc_cross && seq[i] >= num_phones))
                return wf_fail(WF_ERANGE, "wf_compose_linear_batch: transcript %d position %d: phone %d out of range", b, i, seq[i]);
        id.reset(8 * (size_t)(L + 2));
        order.clear(); arcs.clear(); aux.clear();
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
            const int q = (int)(order[head] / (L + 1)), i = (int)(order[head] % (L + 1));
            const int lo = soff[q], hi = soff[q + 1];
            // runs of this state's arcs: final (key -1), phone 0, phone b[i]
            const int32_t* kb = key + lo;
            const int32_t* ke = key + hi;
            const int f1 = (int)(std::upper_bound(kb, ke, -1) - key);
            const int z1 = (int)(std::upper_bound(key + f1, ke, 0) - key);
            int p0 = hi, p1 = hi, nph = 0;
            if (i < L) {
                nph = seq[i];
                p0 = (int)(std::lower_bound(key + z1, ke, nph) - key);
                p1 = (int)(std::upper_bound(key + p0, ke, nph) - key);
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
        const int sf = (int)order.size();
        const int64_t n = (int64_t)aux.size();
        if (w + n > INT32_MAX) return wf_fail(WF_EUNSUPPORTED, "wf_compose_linear_batch: more than 2^31 arcs");
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
    if (!fits) return wf_fail(WF_ENOSPC, "wf_compose_linear_batch: arc capacity %lld < %lld", (long long)cap, (long long)w);
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
    PairMap id;
    std::vector<int64_t> order;
    std::vector<int32_t> arcs, aux, boff, cand, seen;
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
        order.clear(); arcs.clear(); aux.clear();
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
