"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference DWFST loss hot path.

What is restated (reference file:line, all under /root/reference):

* ``waterfall/graph.py:61-76``   loss = -tot_scores(intersect_dense(graph, dense)), reductions
                                 'none' | 'sum' | 'mean' (mean = mean_b(loss_b / target_lengths[b])).
* k2 (third-party, NOT vendored in the reference, version unpinned -- README.md:23,31-32):
  ``DenseFsaVec`` / ``intersect_dense`` / ``Fsa.get_tot_scores`` / ``shortest_path`` semantics as
  written down in SURVEY.md section 8(0).  Lattice state = (t, s); an arc with label >= 0 consumes
  frame t and scores ``fl32(w + lp[t, label])``; label -1 arcs are only taken after the last frame.
* ``waterfall/manual_ctc/ctc.py:9-91`` computes the same total score / gradient for the graph
  encoded as (trans, transform); `tests/golden/` pins this file against outputs of that code.

Parity status: pinned against (1) golden vectors generated in the build container from the
reference's importable ``waterfall.manual_ctc`` code (tests/golden/make_golden.py), (2)
``torch.nn.functional.ctc_loss`` and (3) brute-force path enumeration.  The k2-specific parts
(``output_beam`` pruning arithmetic, k2's tie-breaking and state numbering) are **parity unpinned**:
k2 is absent from the reference tree and from this image.

Conventions: a graph is a dict with int32 arrays ``src, dst, label, aux`` and float32 ``score``
(one entry per arc, arcs sorted by src), ``num_states``; start state 0, super-final state
``num_states-1``, which is entered only by label -1 arcs.
"""
import numpy as np

NEG_INF = -np.inf


def make_graph(arcs, num_states, aux=None):
    """arcs: iterable of (src, dst, label, score). Sorted by src (stable) like a k2 Fsa."""
    arcs = list(arcs)
    order = sorted(range(len(arcs)), key=lambda i: arcs[i][0])
    a = [arcs[i] for i in order]
    g = {
        "src": np.array([x[0] for x in a], dtype=np.int32),
        "dst": np.array([x[1] for x in a], dtype=np.int32),
        "label": np.array([x[2] for x in a], dtype=np.int32),
        "score": np.array([x[3] for x in a], dtype=np.float32),
        "num_states": int(num_states),
    }
    if aux is None:
        g["aux"] = np.where(g["label"] >= 0, 0, -1).astype(np.int32)
    else:
        g["aux"] = np.array([aux[i] for i in order], dtype=np.int32)
    return g


def _seg_lse(v, idx, n, dtype):
    """log-sum-exp of v grouped by idx into n bins; empty bin -> -inf."""
    m = np.full(n, NEG_INF, dtype=dtype)
    np.maximum.at(m, idx, v)
    ms = np.where(np.isfinite(m), m, 0).astype(dtype)
    s = np.zeros(n, dtype=dtype)
    with np.errstate(under="ignore"):
        np.add.at(s, idx, np.exp(v - ms[idx]))
    with np.errstate(divide="ignore"):
        return np.where(s > 0, ms + np.log(s), NEG_INF).astype(dtype)


def _split(g):
    em = g["label"] >= 0
    fin = ~em
    return em, fin


def arc_scores(g, lp_t):
    """fl32(w + lp[t,label]) for the emitting arcs of g (k2 materialises lattice scores in fp32)."""
    em, _ = _split(g)
    return (g["score"][em] + lp_t[g["label"][em]].astype(np.float32)).astype(np.float32)


def forward(g, lp, T, log_semiring=True, dtype=np.float64, keep=None):
    """alpha[t, s], t = 0..T and the total score.  SURVEY.md 8(0); graph.py:61-65.

    keep: optional bool [T, A_emitting] mask of lattice arcs that survive output_beam pruning.
    """
    S = g["num_states"]
    em, fin = _split(g)
    src, dst = g["src"][em], g["dst"][em]
    alpha = np.full((T + 1, S), NEG_INF, dtype=dtype)
    alpha[0, 0] = 0.0
    for t in range(T):
        x = arc_scores(g, lp[t]).astype(dtype)
        v = alpha[t, src] + x
        if keep is not None:
            v = np.where(keep[t], v, NEG_INF)
        if log_semiring:
            alpha[t + 1] = _seg_lse(v, dst, S, dtype)
        else:
            m = np.full(S, NEG_INF, dtype=dtype)
            np.maximum.at(m, dst, v)
            alpha[t + 1] = m
    vf = alpha[T, g["src"][fin]] + g["score"][fin].astype(dtype)
    if vf.size == 0:
        tot = NEG_INF
    elif log_semiring:
        tot = _seg_lse(vf, np.zeros(vf.size, dtype=np.int64), 1, dtype)[0]
    else:
        tot = vf.max()
    return alpha, dtype(tot)


def backward(g, lp, T, log_semiring=True, dtype=np.float64, keep=None):
    """beta[t, s], t = 0..T (beta[T] holds the final-arc weights). SURVEY.md 8(0)."""
    S = g["num_states"]
    em, fin = _split(g)
    src, dst = g["src"][em], g["dst"][em]
    beta = np.full((T + 1, S), NEG_INF, dtype=dtype)
    fsrc = g["src"][fin]
    fw = g["score"][fin].astype(dtype)
    if log_semiring:
        beta[T] = _seg_lse(fw, fsrc, S, dtype)
    else:
        np.maximum.at(beta[T], fsrc, fw)
    for t in range(T - 1, -1, -1):
        x = arc_scores(g, lp[t]).astype(dtype)
        v = x + beta[t + 1, dst]
        if keep is not None:
            v = np.where(keep[t], v, NEG_INF)
        if log_semiring:
            beta[t] = _seg_lse(v, src, S, dtype)
        else:
            m = np.full(S, NEG_INF, dtype=dtype)
            np.maximum.at(m, src, v)
            beta[t] = m
    return beta


def prune_mask(g, lp, T, output_beam, dtype=np.float64):
    """Lattice arcs (t, a) kept by intersect_dense(output_beam).  [k2-upstream, parity unpinned]

    SURVEY.md 8(0): with tropical alpha^, beta^ and best = tropical total, keep (t, a) iff
    alpha^_t[src] + fl32(w + lp) + beta^_{t+1}[dst] >= best - output_beam.
    """
    em, _ = _split(g)
    src, dst = g["src"][em], g["dst"][em]
    a, best = forward(g, lp, T, log_semiring=False, dtype=dtype)
    b = backward(g, lp, T, log_semiring=False, dtype=dtype)
    keep = np.zeros((T, int(em.sum())), dtype=bool)
    if not np.isfinite(best):
        return keep
    for t in range(T):
        x = arc_scores(g, lp[t]).astype(dtype)
        keep[t] = (a[t, src] + x + b[t + 1, dst]) >= best - output_beam
    return keep


def tot_and_grad(g, lp, T, output_beam=None, dtype=np.float64):
    """(tot, d tot / d lp [Tmax, V]) in the log semiring.  Rows t >= T are 0; infeasible -> (-inf, 0).

    Autograd of graph.py:61-66 (k2 _GetTotScoresFunction / _IntersectDenseFunction backward):
    posterior(t, a) = exp(alpha_t[src] + x + beta_{t+1}[dst] - tot) summed per label.
    """
    Tmax, V = lp.shape
    keep = None
    if output_beam is not None and np.isfinite(output_beam) and output_beam > 0:
        keep = prune_mask(g, lp, T, output_beam, dtype=np.float64)
    em, _ = _split(g)
    src, dst, lab = g["src"][em], g["dst"][em], g["label"][em]
    alpha, tot = forward(g, lp, T, True, dtype, keep)
    grad = np.zeros((Tmax, V), dtype=dtype)
    if not np.isfinite(tot):
        return tot, grad
    beta = backward(g, lp, T, True, dtype, keep)
    for t in range(T):
        x = arc_scores(g, lp[t]).astype(dtype)
        e = alpha[t, src] + x + beta[t + 1, dst] - tot
        if keep is not None:
            e = np.where(keep[t], e, NEG_INF)
        with np.errstate(under="ignore"):
            np.add.at(grad[t], lab, np.exp(e))
    return tot, grad


def arc_posteriors(g, lp, T, dtype=np.float64):
    """post[t, j] = exp(alpha_t[src] + x + beta_{t+1}[dst] - tot) for the j-th EMITTING arc (original arc order);
    zeros when the utterance is infeasible.  The quantity waterfall/manual_ctc/prior.py:49-55 normalises per frame when
    the graph is given one machine state per arc (oracle/graphs.py graph_to_trans_transform)."""
    em, _ = _split(g)
    src, dst = g["src"][em], g["dst"][em]
    alpha, tot = forward(g, lp, T, True, dtype)
    post = np.zeros((T, int(em.sum())), dtype=dtype)
    if not np.isfinite(tot):
        return post
    beta = backward(g, lp, T, True, dtype)
    for t in range(T):
        x = arc_scores(g, lp[t]).astype(dtype)
        with np.errstate(under="ignore"):
            post[t] = np.exp(alpha[t, src] + x + beta[t + 1, dst] - tot)
    return post


def state_posteriors(g, lp, T, dtype=np.float64):
    """log P(state s after consuming frame t) = alpha_{t+1}[s] + beta_{t+1}[s] - tot, [T, S] (prior.py:4-55 when machine
    states are graph states); -inf where impossible."""
    alpha, tot = forward(g, lp, T, True, dtype)
    beta = backward(g, lp, T, True, dtype)
    with np.errstate(invalid="ignore"):
        out = alpha[1:T + 1] + beta[1:T + 1] - tot
    return np.where(np.isnan(out), NEG_INF, out)


def scale_graph(g, eta):
    """The graph with every score multiplied by eta (eta-semiring of softmax_ctc.py:44-62: 1/eta * LSE(eta * x) is the
    log semiring on scores scaled by eta)."""
    h = dict(g)
    h["score"] = (g["score"].astype(np.float64) * eta).astype(np.float32)
    return h


def viterbi(g, lp, T, dtype=np.float32):
    """Tropical best path. Returns (score, arcs) with arcs = T emitting arc ids + the final arc id.

    Rule (north_star): ties broken by the lowest arc index, applied at every back-trace step:
    the predecessor arc of lattice state (t+1, d) is the lowest-index arc among those reaching
    the max, and the final arc is the lowest-index final arc reaching the max total.
    Arithmetic: x = fl32(w + lp), v = fl(alpha[src] + x) evaluated in `dtype` (fp32 default; fp64
    mirrors lattice.get_tot_scores(use_double_scores=True, log_semiring=False),
    utils/lattice_analysis.py:129).  Infeasible -> (-inf, []).
    """
    S = g["num_states"]
    A = g["src"].size
    ids = np.arange(A, dtype=np.int64)
    em, fin = _split(g)
    src, dst, eid = g["src"][em], g["dst"][em], ids[em]
    alpha = np.full(S, NEG_INF, dtype=dtype)
    alpha[0] = 0
    bp = np.full((T, S), -1, dtype=np.int64)
    for t in range(T):
        x = arc_scores(g, lp[t]).astype(dtype)
        v = (alpha[src] + x).astype(dtype)
        m = np.full(S, NEG_INF, dtype=dtype)
        np.maximum.at(m, dst, v)
        hit = (v == m[dst]) & np.isfinite(v)
        best = np.full(S, A, dtype=np.int64)
        np.minimum.at(best, dst[hit], eid[hit])
        bp[t] = np.where(best < A, best, -1)
        alpha = m
    fsrc, fid = g["src"][fin], ids[fin]
    vf = (alpha[fsrc] + g["score"][fin].astype(dtype)).astype(dtype)
    if vf.size == 0 or not np.isfinite(vf.max()):
        return dtype(NEG_INF), []
    score = vf.max()
    fa = int(fid[vf == score].min())
    path = [fa]
    s = int(g["src"][fa])
    for t in range(T - 1, -1, -1):
        a = int(bp[t, s])
        path.append(a)
        s = int(g["src"][a])
    assert s == 0
    path.reverse()
    return dtype(score), path



def beam_search(g, lp, T, search_beam=np.inf, frames=None):
    """Frame-synchronous token passing with a tropical beam: the restatement wf_search is tested against.
    [the role of k2.get_lattice's search at utils/lattice_analysis.py:125; k2's own beam arithmetic is parity unpinned]

    A state of frame t survives when its tropical score v satisfies v >= max_t - search_beam (fp32 arithmetic as in
    `viterbi`); only survivors are expanded.  Returns (score, arcs, tot_log, n_active):
    best path over the surviving lattice with the tie rule of `viterbi`, the log-semiring total (fp64) over all paths
    through surviving states, and the survivor count per frame (t = 0..T).  frames: optional list that receives the
    tropical forward scores of every frame (fp32 [S], -inf = not a survivor), t = 0..T (lattice_search reads them).
    """
    S = g["num_states"]
    A = g["src"].size
    ids = np.arange(A, dtype=np.int64)
    em, fin = _split(g)
    src, dst, eid = g["src"][em], g["dst"][em], ids[em]
    alpha = np.full(S, NEG_INF, dtype=np.float32)
    alpha[0] = 0
    la = np.full(S, NEG_INF, dtype=np.float64)
    la[0] = 0
    bp = np.full((T, S), -1, dtype=np.int64)
    n_active = [1]
    if frames is not None:
        frames.append(alpha.copy())
    for t in range(T):
        x = arc_scores(g, lp[t])
        with np.errstate(invalid="ignore"):
            v = (alpha[src] + x).astype(np.float32)
        m = np.full(S, NEG_INF, dtype=np.float32)
        np.maximum.at(m, dst, v)
        hit = (v == m[dst]) & np.isfinite(v)
        best = np.full(S, A, dtype=np.int64)
        np.minimum.at(best, dst[hit], eid[hit])
        bp[t] = np.where(best < A, best, -1)
        with np.errstate(invalid="ignore"):
            ln = _seg_lse(la[src] + x.astype(np.float64), dst, S, np.float64)
        if not np.isfinite(m).any():
            return np.float32(NEG_INF), [], NEG_INF, n_active + [0] * (T - t)
        keep = m >= np.float32(m.max() - np.float32(search_beam)) if np.isfinite(search_beam) else np.isfinite(m)
        alpha = np.where(keep, m, np.float32(NEG_INF)).astype(np.float32)
        la = np.where(keep, ln, NEG_INF)
        n_active.append(int(keep.sum()))
        if frames is not None:
            frames.append(alpha.copy())
    fsrc, fid = g["src"][fin], ids[fin]
    vf = (alpha[fsrc] + g["score"][fin]).astype(np.float32)
    if vf.size == 0 or not np.isfinite(vf.max()):
        return np.float32(NEG_INF), [], NEG_INF, n_active
    score = vf.max()
    fa = int(fid[vf == score].min())
    path = [fa]
    s = int(g["src"][fa])
    for t in range(T - 1, -1, -1):
        a = int(bp[t, s])
        path.append(a)
        s = int(g["src"][a])
    path.reverse()
    lf = la[fsrc] + g["score"][fin].astype(np.float64)
    tot_log = _seg_lse(lf, np.zeros(lf.size, dtype=np.int64), 1, np.float64)[0]
    return np.float32(score), path, float(tot_log), n_active


def lattice_search(g, lp, T, search_beam=np.inf, output_beam=np.inf):
    """beam_search followed by get_lattice's output_beam: the restatement wf_search_lattice is tested against.
    [utils/lattice_analysis.py:125 `k2.get_lattice(..., search_beam=, output_beam=)` and :128 `get_tot_scores(True, True)`
    of the lattice it returns; k2's own arithmetic is parity unpinned]

    The lattice of the search: every arc from a survivor of frame t to a survivor of frame t + 1, and the final arcs
    out of the survivors of frame T.  With alpha the tropical forward scores of the search and beta the tropical
    backward scores within that lattice, an arc stays when fl32(alpha[src] + fl32(x + beta[dst])) >= best - output_beam
    (x = fl32(w + lp); a final arc: fl32(alpha[src] + w)).  Returns (score, arcs, tot_log, lattice_arcs): best path as
    beam_search, the log-semiring total (fp64) over the arcs that stay, and their number.
    """
    fr = []
    score, path, _, _ = beam_search(g, lp, T, search_beam, frames=fr)
    if not np.isfinite(score):
        return score, path, NEG_INF, 0
    S = g["num_states"]
    em, fin = _split(g)
    src, dst = g["src"][em], g["dst"][em]
    thresh = np.float32(score - np.float32(output_beam)) if np.isfinite(output_beam) else np.float32(NEG_INF)
    fsrc, fw = g["src"][fin], g["score"][fin].astype(np.float32)
    aT = fr[T]
    with np.errstate(invalid="ignore"):
        live = np.isfinite(aT[fsrc]) & np.isfinite(fw)
        bt = np.full(S, NEG_INF, dtype=np.float32)
        np.maximum.at(bt, fsrc[live], fw[live])
        kept = live & ((aT[fsrc] + fw).astype(np.float32) >= thresh)
    narcs = int(kept.sum())
    bl = _seg_lse(fw[kept].astype(np.float64), fsrc[kept], S, np.float64)
    for t in range(T - 1, -1, -1):
        x = arc_scores(g, lp[t])
        a0, a1 = fr[t], fr[t + 1]
        with np.errstate(invalid="ignore"):
            vb = (x + bt[dst]).astype(np.float32)
            live = np.isfinite(a0[src]) & np.isfinite(a1[dst]) & np.isfinite(vb)
            kept = live & ((a0[src] + vb).astype(np.float32) >= thresh)
            nbt = np.full(S, NEG_INF, dtype=np.float32)
            np.maximum.at(nbt, src[live], vb[live])
            nbl = _seg_lse(x[kept].astype(np.float64) + bl[dst[kept]], src[kept], S, np.float64)
        narcs += int(kept.sum())
        bt, bl = nbt, nbl
    return score, path, float(bl[0]), narcs


def lattice_words(g, lp, T, words, search_beam=np.inf, output_beam=np.inf):
    """lattice_search's output lattice restricted to the paths whose word sequence is `words`: the restatement of
    `latticeW = k2.connect(k2.compose(lattice, W))` with W = k2.linear_fst(words, words) and of its
    get_tot_scores(True, True) / (True, False) (utils/lattice_analysis.py:137-139,161-162).  g["aux"]: word id per arc
    (0 = epsilon, -1 on final arcs).  Returns (tot_log fp64, best fp32): log-semiring and tropical totals over those paths
    (-inf when the lattice holds none).  [k2-upstream, parity unpinned]"""
    fr = []
    score, path, _, _ = beam_search(g, lp, T, search_beam, frames=fr)
    if not np.isfinite(score):
        return NEG_INF, np.float32(NEG_INF)
    S = g["num_states"]
    n = len(words)
    em, fin = _split(g)
    src, dst, aux = g["src"][em], g["dst"][em], g["aux"][em]
    thresh = np.float32(score - np.float32(output_beam)) if np.isfinite(output_beam) else np.float32(NEG_INF)
    fsrc, fw = g["src"][fin], g["score"][fin].astype(np.float32)
    aT = fr[T]
    with np.errstate(invalid="ignore"):
        live = np.isfinite(aT[fsrc]) & np.isfinite(fw)
        bt = np.full(S, NEG_INF, dtype=np.float32)
        np.maximum.at(bt, fsrc[live], fw[live])
        kept = live & ((aT[fsrc] + fw).astype(np.float32) >= thresh)
    L = np.full((S, n + 1), NEG_INF, dtype=np.float64)       # log-semiring backward score of (state, words consumed)
    M = np.full((S, n + 1), NEG_INF, dtype=np.float32)       # tropical
    L[:, n] = _seg_lse(fw[kept].astype(np.float64), fsrc[kept], S, np.float64)
    np.maximum.at(M[:, n], fsrc[kept], fw[kept])
    # arcs that consume word k (k + 1 afterwards), per k; epsilon arcs keep k
    eps = aux == 0
    for t in range(T - 1, -1, -1):
        x = arc_scores(g, lp[t])
        a0, a1 = fr[t], fr[t + 1]
        with np.errstate(invalid="ignore"):
            vb = (x + bt[dst]).astype(np.float32)
            live = np.isfinite(a0[src]) & np.isfinite(a1[dst]) & np.isfinite(vb)
            kept = live & ((a0[src] + vb).astype(np.float32) >= thresh)
            nbt = np.full(S, NEG_INF, dtype=np.float32)
            np.maximum.at(nbt, src[live], vb[live])
            nL = np.full((S, n + 1), NEG_INF, dtype=np.float64)
            nM = np.full((S, n + 1), NEG_INF, dtype=np.float32)
            for k in range(n + 1):
                sel = kept & eps
                vals = [x[sel].astype(np.float64) + L[dst[sel], k]]
                valm = [(x[sel] + M[dst[sel], k]).astype(np.float32)]
                srcs = [src[sel]]
                if k < n:
                    sw = kept & (aux == words[k])
                    vals.append(x[sw].astype(np.float64) + L[dst[sw], k + 1])
                    valm.append((x[sw] + M[dst[sw], k + 1]).astype(np.float32))
                    srcs.append(src[sw])
                v, vm, ss = np.concatenate(vals), np.concatenate(valm), np.concatenate(srcs)
                nL[:, k] = _seg_lse(v, ss, S, np.float64)
                np.maximum.at(nM[:, k], ss, vm)
        bt, L, M = nbt, nL, nM
    return float(L[0, 0]), np.float32(M[0, 0])


def graphloss(graphs, lp, segs, output_beam=None, reduction="sum", target_lengths=None,
              dtype=np.float64):
    """graph.py:79-118 restated: returns (loss, d loss / d lp [B,T,V]).

    graphs: list of B graphs; lp [B,T,V]; segs int [B,3] rows (seq_idx, start, dur).
    Infeasible utterances give loss=+inf and zero gradient (models.py:305-313 masks them).
    """
    B = len(graphs)
    losses = np.zeros(B, dtype=dtype)
    grad = np.zeros(lp.shape, dtype=dtype)
    for i in range(B):
        b, st, dur = (int(x) for x in segs[i])
        tot, gr = tot_and_grad(graphs[i], lp[b, st:], dur, output_beam, dtype)
        losses[i] = -tot
        if reduction == "mean":
            f = 1.0 / (B * float(target_lengths[i]))
        else:
            f = 1.0
        grad[b, st:] += -gr * f
    if reduction == "none":
        return losses.astype(np.float32), grad
    if reduction == "sum":
        return np.float32(losses.astype(np.float32).sum()), grad
    assert reduction == "mean"
    return np.float32((losses.astype(np.float32) / np.asarray(target_lengths, dtype=np.float32)).mean()), grad


# ----------------------------------------------------------------------------------------------
# brute force (independent known-answer check, tiny lattices only)
# ----------------------------------------------------------------------------------------------

def enumerate_paths(g, T, limit=2_000_000):
    """All arc sequences of T emitting arcs + one final arc from state 0 (lexicographic in arc id)."""
    A = g["src"].size
    out = [[] for _ in range(g["num_states"])]
    for a in range(A):
        out[int(g["src"][a])].append(a)
    paths = []

    def rec(s, t, pre):
        if len(paths) > limit:
            raise RuntimeError("too many paths")
        if t == T:
            for a in out[s]:
                if g["label"][a] < 0:
                    paths.append(pre + [a])
            return
        for a in out[s]:
            if g["label"][a] >= 0:
                rec(int(g["dst"][a]), t + 1, pre + [a])

    rec(0, 0, [])
    return paths


def brute_force(g, lp, T):
    """(log-sum total, grad [Tmax,V], best score (fp64), all best paths) by path enumeration."""
    paths = enumerate_paths(g, T)
    Tmax, V = lp.shape
    if not paths:
        return NEG_INF, np.zeros((Tmax, V)), NEG_INF, []
    sc = np.zeros(len(paths))
    for i, p in enumerate(paths):
        s = 0.0
        for t, a in enumerate(p[:-1]):
            s += float(np.float32(g["score"][a] + np.float32(lp[t, g["label"][a]])))
        s += float(g["score"][p[-1]])
        sc[i] = s
    m = sc.max()
    tot = m + np.log(np.exp(sc - m).sum())
    grad = np.zeros((Tmax, V))
    w = np.exp(sc - tot)
    for i, p in enumerate(paths):
        for t, a in enumerate(p[:-1]):
            grad[t, g["label"][a]] += w[i]
    best = [p for i, p in enumerate(paths) if sc[i] == m]
    return tot, grad, m, best
