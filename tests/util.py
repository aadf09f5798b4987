"""Shared helpers for the tests (test infrastructure; may import oracle/)."""
import ctypes as C
import os

import numpy as np

from oracle import dwfst_oracle as o
from oracle import graphs as og

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PHONES = ["<eps>", "<SIL>", "AH", "B", "K", "S", "T"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def golden_graphs(z):
    """Oracle graphs stored in a manual_ctc_<topo>.npz fixture."""
    out = []
    off = z["g_arc_off"]
    for i in range(len(off) - 1):
        sl = slice(off[i], off[i + 1])
        out.append({"src": z["g_src"][sl], "dst": z["g_dst"][sl], "label": z["g_label"][sl],
                    "score": z["g_score"][sl], "aux": z["g_aux"][sl], "num_states": int(z["g_num_states"][i])})
    return out


def ctc_phones(V):
    return ["<eps>"] + ["p%d" % i for i in range(1, V)]


def graphs_to_fsa(graphs):
    """oracle graph dicts -> waterfall_amd.Fsa vector."""
    import waterfall_amd as wf
    fs = []
    for g in graphs:
        arcs = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
        fs.append(wf.Fsa(arcs, g["aux"], [0, arcs.shape[0]], [g["num_states"]]))
    return wf.create_fsa_vec(fs)


def c_oracle_batch(lib, graphs, lp, segs, gscale=None, use_double=False, want_grad=True, nthreads=1):
    """Runs oracle/dwfst_oracle.c on a batch. Returns (tot[B], grad like lp)."""
    B = len(graphs)
    off = np.zeros(B + 1, dtype=np.int32)
    for i, g in enumerate(graphs):
        off[i + 1] = off[i] + g["src"].size
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([g[k] for g in graphs]).astype(dt))
    src, dst, lab = cat("src", np.int32), cat("dst", np.int32), cat("label", np.int32)
    sc = cat("score", np.float32)
    ns = np.array([g["num_states"] for g in graphs], dtype=np.int32)
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    segs = np.ascontiguousarray(segs, dtype=np.int32)
    tot = np.zeros(B, dtype=np.float32)
    grad = np.zeros_like(lp)
    gs = None if gscale is None else np.ascontiguousarray(gscale, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)
    lib.oracle_graphloss_batch.restype = C.c_int
    lib.oracle_graphloss_batch.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                                                         C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                                         C.c_void_p, C.c_void_p]
    rc = lib.oracle_graphloss_batch(B, p(off), p(ns), p(src), p(dst), p(lab), p(sc), p(lp), lp.strides[0] // 4,
                                    lp.strides[1] // 4, lp.shape[2], p(segs), p(gs), int(use_double), int(want_grad),
                                    nthreads, p(tot), p(grad))
    assert rc == 0
    return tot, grad


def c_oracle_tot_f64(lib, graphs, lp, segs, nthreads=1):
    """oracle_tot_scores_f64: the fp64 log-semiring totals, not rounded to fp32."""
    B = len(graphs)
    off = np.zeros(B + 1, dtype=np.int32)
    for i, g in enumerate(graphs):
        off[i + 1] = off[i] + g["src"].size
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([g[k] for g in graphs]).astype(dt))
    src, dst, lab = cat("src", np.int32), cat("dst", np.int32), cat("label", np.int32)
    sc = cat("score", np.float32)
    ns = np.array([g["num_states"] for g in graphs], dtype=np.int32)
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    segs = np.ascontiguousarray(segs, dtype=np.int32)
    tot = np.zeros(B, dtype=np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.oracle_tot_scores_f64.restype = C.c_int
    lib.oracle_tot_scores_f64.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    rc = lib.oracle_tot_scores_f64(B, p(off), p(ns), p(src), p(dst), p(lab), p(sc), p(lp), lp.strides[0] // 4,
                                   lp.strides[1] // 4, p(segs), nthreads, p(tot))
    assert rc == 0
    return tot


def c_oracle_viterbi(lib, g, lp, T):
    lib.oracle_viterbi_f32.restype = C.c_float
    lib.oracle_viterbi_f32.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_void_p]
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    path = np.full(T + 1, -1, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    src, dst, lab = (np.ascontiguousarray(g[k], dtype=np.int32) for k in ("src", "dst", "label"))
    sc = np.ascontiguousarray(g["score"], dtype=np.float32)
    s = lib.oracle_viterbi_f32(g["num_states"], src.size, p(src), p(dst), p(lab), p(sc), p(lp), lp.strides[0] // 4, T,
                               p(path))
    return s, path


def random_case(seed, topo_name, B, T, L, P, scale=3.0, ragged=True, den=False):
    """Synthetic batch as in BASELINE.md section 3 (smaller sizes): returns dict with topo, graphs
    (waterfall_amd.Fsa), oracle graphs, lp [B,T,V] float32 numpy, segs int32 [B,3]."""
    import torch
    import waterfall_amd as wf
    rng = np.random.default_rng(seed)
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    lazy = wf.Topology(topo_name, phones)
    tokens = lazy.tokens
    big = topo_name == "ctc" and P > 200          # the full CTC topology is O(P^2) arcs
    topo = None if big else lazy.fsa
    V = len(tokens)
    g = torch.Generator().manual_seed(seed)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g) * scale, -1).numpy()
    if ragged:
        lens = np.sort(rng.integers(max(1, int(0.7 * T)), T + 1, size=B))[::-1].copy()
        lens[0] = T
    else:
        lens = np.full(B, T)
    segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), lens], 1).astype(np.int32)
    sil = topo_name.endswith("sil") or topo_name.endswith("sil3") or "noblk" in topo_name
    seqs = []
    for b in range(B):
        s = [int(x) + 2 for x in rng.integers(0, P, size=L)]
        if sil:
            s = [1] + s + [1]
        seqs.append(s)
    if den:
        fsa = wf.create_fsa_vec([topo] * B)
    else:
        fsa = wf.compile_ctc_k2_graphs(lazy if big else topo, seqs)
    return {"topo": topo, "tokens": tokens, "fsa": fsa, "ographs": fsa.as_oracle_graphs(), "lp": lp, "segs": segs,
            "seqs": seqs, "V": V}


from waterfall_amd.synth import synth_decoding_graph  # noqa: E402,F401  (shared with bench.py)
