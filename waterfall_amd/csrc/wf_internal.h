// Internal layout of the prepared graph blob shared by the host builder (wf_graph.cpp) and the
// HIP kernels (wf_kernels.hip).  All offsets are in int32 words from the start of the blob, so the
// same bytes are valid on the host and in HBM.
#pragma once
#include <stdint.h>

#define WF_BLOB_MAGIC 0x31474657  // "WFG1"
#define WF_HDR_WORDS 8
#define WF_INFO_WORDS 64
// A state whose in-degree (forward sweep) or out-degree (backward sweep) exceeds this is reduced
// cooperatively by the whole workgroup instead of by one lane.
#define WF_HEAVY_DEG 16
// The register-resident sweep keeps up to this many arcs per state in VGPRs.
#define WF_REG_DEG 4

// header words
enum { WF_H_MAGIC = 0, WF_H_NUM_FSAS = 1, WF_H_WORDS = 2, WF_H_MAX_LABEL = 3, WF_H_MAX_STATES = 4,
       WF_H_MAX_ARCS = 5 };

// per-FSA info words (blob[WF_HDR_WORDS + g*WF_INFO_WORDS + i])
enum {
    WF_I_S = 0,        // states (incl. super-final)
    WF_I_A = 1,        // emitting arcs
    WF_I_NF = 2,       // final arcs
    WF_I_K = 3,        // label columns: compact -> #distinct labels; direct -> max_label+1
    WF_I_DIRECT = 4,   // 1: column == label (emissions read straight from log_probs)
    WF_I_MAXIN = 5,
    WF_I_MAXOUT = 6,
    WF_I_NHEAVY_IN = 7,
    WF_I_NHEAVY_OUT = 8,
    WF_I_ARC_BASE = 9,  // index of this FSA's first arc in the FsaVec arc array
    WF_I_IN_OFF = 10,   // [S+1]  incoming-arc CSR (arcs sorted by dst, then original index)
    WF_I_IN_SRC = 11,   // [A]
    WF_I_IN_COL = 12,   // [A]
    WF_I_IN_W = 13,     // [A] float
    WF_I_IN_ARC = 14,   // [A] original arc index local to the FSA
    WF_I_OUT_OFF = 15,  // [S+1]  outgoing-arc CSR (emitting arcs in original order)
    WF_I_OUT_DST = 16,
    WF_I_OUT_COL = 17,
    WF_I_OUT_W = 18,
    WF_I_LAB_OFF = 19,  // [K+1]  arcs sorted by column
    WF_I_LAB_SRC = 20,
    WF_I_LAB_DST = 21,
    WF_I_LAB_W = 22,
    WF_I_LABELS = 23,   // [K] label id of each column (compact mode)
    WF_I_FIN_SRC = 24,  // [NF]
    WF_I_FIN_W = 25,
    WF_I_FIN_ARC = 26,
    WF_I_HEAVY_IN = 27,   // [NHEAVY_IN] state ids
    WF_I_HEAVY_OUT = 28,  // [NHEAVY_OUT]
    WF_I_MAX_LABEL = 29,
    WF_I_MAXLAB_DEG = 30,  // max arcs sharing one column
    WF_I_FINB_LOG = 31,    // [S] float: beta_T[s] = logsumexp of the final-arc weights leaving s
    WF_I_FINB_MAX = 32,    // [S] float: same in the tropical semiring
    WF_I_ARC_SRC = 33,     // [A+NF] source state of every original arc (local arc index)
    WF_I_NOVER_IN = 34,    // number of states with in-degree > WF_REG_DEG
    WF_I_OVER_IN = 35,     // one such state (the only one when NOVER_IN == 1)
    WF_I_NOVER_OUT = 36,   // same for out-degree
    WF_I_OVER_OUT = 37,
    WF_I_LAB_COL = 38,     // [A] column of every column-sorted arc
    WF_I_MINLEN = 39,      // fewest emitting arcs on any start -> final path (INT32_MAX: none): an
                           // utterance with fewer frames is structurally infeasible
    // ---- banded form (sweep_band_kernel): the non-final states in a topological order in which every
    // emitting arc goes from position p - d to position p with 0 <= d <= BAND_BW
    WF_I_BAND_BW = 40,     // band width, -1 when the graph has no such order (a cycle besides self-loops)
    WF_I_BAND_M = 41,      // most arcs sharing one (p, d) cell
    WF_I_BAND_ORDER = 42,  // [S-1] state at position p
    WF_I_BANDF = 43,       // [S-1][BW+1][M] x {column, weight bits}: arcs INTO the state at p from p - d
    WF_I_BANDB = 44,       // same for the backward sweep in reversed positions p' = S-2-p:
                           // arcs OUT OF the state at p' to p' - d (reversed), i.e. to position p + d
    WF_I_BAND_WE = 45,     // 1: every emitting arc has weight 0 ("weights elided": all training topologies)
    // ---- hub form (sweep_hub_kernel / post_hub_kernel): a topology used as denominator over a large table.  State 0 is
    // the hub; the other non-final states fall into NG groups of G consecutive states (group g = states 1 + g*G ..) that
    // are connected only to the hub and among themselves, all with the same arc pattern; the label of an arc of group g
    // is column C0[c] + g*CS for one of NC column classes c.  (mmictc: G = 2, classes p_0, p_1; 3state-skip: G = 2,
    // classes p_0, p_1, p_2.)  All weights 0, only the hub is final.
    WF_I_HUB_G = 46,       // states per group (0: the graph has no hub form)
    WF_I_HUB_NG = 47,
    WF_I_HUB_NC = 48,      // column classes per group (<= WF_HUB_MAX_NC)
    WF_I_HUB_C0 = 49,      // [NC] column of class c in group 0
    WF_I_HUB_CS = 50,      // column stride between consecutive groups
    WF_I_HUB_COEF = 51,    // float [(G+1)*(G+1)][NC]: number of arcs src -> dst with class c at [(dst*(G+1) + src)*NC + c];
                           // index G = the hub (hub -> hub: the hub self-loops labelled with a group's columns)
    WF_I_HUB_NX = 52,      // hub self-loops whose label belongs to no group (<blk>)
    WF_I_HUB_XCOL = 53,    // [NX] their columns
    WF_I_HUB_FINW = 54,    // float bits: weight of the hub's final arc
    WF_I_OUT_ARC = 55,     // [A] original arc index (local to the FSA) of every entry of the outgoing CSR
    WF_I_OUT_REC = 56,     // [A][4] {dst, column, weight bits, arc index} of every entry of the outgoing CSR in one 16-byte
                           // record (search_kernel: one random access per arc instead of four)
};
#define WF_HUB_MAX_G 3
#define WF_HUB_MAX_NC 4
// limits of the banded sweep (template instantiations in wf_kernels.hip)
#define WF_BAND_MAX_BW 4
#define WF_BAND_MAX_M 2
#define WF_BAND_MAX_R 4

// per-utterance table entry written by the host for each call (int64 words)
enum {
    WF_U_FSA = 0,
    WF_U_SEQ = 1,
    WF_U_START = 2,
    WF_U_T = 3,
    WF_U_ALPHA = 4,  // element offset of alpha[0][0] in the alpha/beta arena
    WF_U_BETA = 5,
    WF_U_E = 6,      // element offset of the gathered emission rows (compact mode)
    WF_U_ROW0 = 7,   // first row index of this utterance in the flattened (b,t) row list
    // copies of the FSA's info words that post kernels need (saves one dependent load)
    WF_U_S = 8, WF_U_A = 9, WF_U_K = 10, WF_U_DIRECT = 11, WF_U_MAXLABDEG = 12,
    WF_U_LAB_OFF = 13, WF_U_LAB_SRC = 14, WF_U_LAB_DST = 15, WF_U_LAB_COL = 16, WF_U_LAB_W = 17,
    WF_U_LABELS = 18,
    WF_U_OFFS = 19,  // element offset of this utterance's per-group scale offsets (<= T+1 doubles)
    WF_U_P = 20,     // element offset of the fp64 emission probabilities (linear-domain sweep)
    WF_U_PP = 21,    // their row pitch in doubles: K rounded up to even, so that rows are 16-byte aligned
    WF_U_SOFF = 22,  // number of states of the utterances before this one (big-graph sweep: alpha ping-pong buffers)
    WF_U_WORDS = 24,
};

// ---- sizes shared by the host compiler (wf_graph.cpp) and the device compiler (wf_compile.hip) ----------------------
#if defined(__HIPCC__)
#define WF_HD __host__ __device__
#else
#define WF_HD
#endif
WF_HD static inline size_t wf_pad4(size_t w) { return (w + 3) & ~(size_t)3; }
// Upper bound of the words an FSA of S states and A arcs takes in the blob (O(1): no pass over the arcs).
// K <= 2A + 1: compact columns K <= A; direct columns K = max_label + 1 <= 2 * (#distinct labels).
WF_HD static inline size_t wf_fsa_words_bound(size_t S, size_t A) {
    const size_t K = 2 * A + 1;
    size_t w = 0;
    w += wf_pad4(S + 1) + 4 * wf_pad4(A);                       // in_off, in_src, in_col, in_w, in_arc
    w += wf_pad4(S + 1) + 4 * wf_pad4(A) + wf_pad4(4 * A);      // out_off, out_dst, out_col, out_w, out_arc, out_rec
    w += wf_pad4(K + 1) + 4 * wf_pad4(A);                       // lab_off, lab_src, lab_dst, lab_w, lab_col
    w += wf_pad4(K);                                            // labels
    w += 3 * wf_pad4(A);                                        // fin_src, fin_w, fin_arc (NF <= A)
    w += 2 * wf_pad4(S);                                        // heavy_in, heavy_out
    w += 2 * wf_pad4(S) + wf_pad4(A);                           // finb_log, finb_max, arc_src
    if (S - 1 <= 64 * WF_BAND_MAX_R)                            // banded form is only built for graphs one wavefront can hold
        w += wf_pad4(S) + 2 * wf_pad4(S * (WF_BAND_MAX_BW + 1) * WF_BAND_MAX_M * 2);
    w += wf_pad4(WF_HUB_MAX_NC) + wf_pad4((WF_HUB_MAX_G + 1) * (WF_HUB_MAX_G + 1) * WF_HUB_MAX_NC) + wf_pad4(A);   // hub form
    return w;
}

// ---- topology tables of the device numerator compiler (wf_topo_device_build -> numerator_compile_kernel): int32 words,
// offsets from the start of the table
enum {
    WF_T_MAGIC = 0, WF_T_NA, WF_T_NS, WF_T_CROSS, WF_T_NPH /* phones 1 .. NPH index pdst_off */, WF_T_MAXPH /* of the index */,
    WF_T_OARCS /* [nA][4] */, WF_T_OAUX /* [nA] */, WF_T_OINDEX /* wf_topo_index_build's index */, WF_T_OPSTATE, WF_T_OPTOKEN /* ctc_cross */,
    WF_T_OPDOFF /* [NPH + 2]: phone p's destination states are pdst[pdoff[p] .. pdoff[p+1]) */, WF_T_OPDST,
    WF_T_MAXD /* most emitting arcs of one product state */, WF_T_WORDS, WF_T_NUMPH /* size of phone_state / phone_token */,
    WF_T_HDR = 16
};
#define WF_TOPO_MAGIC 0x31544657  // "WFT1"
// limits of the device compiler (graphs beyond them: status != 0, the caller compiles on the host)
#define WF_NC_MAX_POS 256      // transcript positions L + 1
#define WF_NC_MAX_CAND 8       // topology states alive at one position
#define WF_NC_MAX_S (64 * WF_BAND_MAX_R + 1)
#define WF_NC_MAX_A 1024       // emitting arcs
#define WF_NC_MAX_NF 64
#define WF_NC_MAX_K 1024
// per-utterance capacities by transcript length L and the topology's WF_T_MAXD
WF_HD static inline int wf_nc_state_cap(int L) { const int s = WF_NC_MAX_CAND * (L + 1) + 1; return s < WF_NC_MAX_S ? s : WF_NC_MAX_S; }
WF_HD static inline int wf_nc_arc_cap(int L, int maxd) {
    const long long a = (long long)(wf_nc_state_cap(L) - 1) * maxd;
    return (int)(a < WF_NC_MAX_A ? a : WF_NC_MAX_A) + WF_NC_MAX_NF;
}

