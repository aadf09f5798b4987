"""The decoding-side k2 calls of ``utils/lattice_analysis.py:120-168`` with the same names and call shapes:

    lattice = get_lattice(log_prob, log_prob_len, TLG, search_beam=..., output_beam=...)     # :125
    lattice.get_tot_scores(True, True) / (True, False)                                       # :128-129
    one_best = one_best_decoding(lattice); get_aux_labels(one_best); one_best.labels         # :133-135
    ETLGW = intersect_dense(TLGW, dense_fsa_vec, 10); ETLGW.get_tot_scores(True, True)        # :147-149
    W = arc_sort(linear_fst(words, words)); latticeW = connect(compose(lattice, W)); TLGW = connect(compose(TLG, W))  # :137-140
    latticeW.get_tot_scores(True, True) / (True, False)                                      # :161-162
    shortest_path(lattice, True)                                                             # :79
    TLG = load_decoding_graph("data/lang/TLG.pt")                                            # :199, compile_TLG.py:122-129

k2 materialises the pruned lattice; here a *lattice* is the triple (graph, dense log-probs, beams) and is never
materialised:

* ``get_lattice`` -> the beam-pruned frame-synchronous search ``wf_search`` (graph.pruned_search): token passing with
  ``search_beam`` and ``min/max_active_states`` over decoding graphs of any size; it yields the best path, the tropical
  score and the log-semiring total of the pruned lattice in one pass.  ``search_beam=inf, max_active_states=0`` is the
  exact search.  ``output_beam`` of get_lattice is applied as k2 applies it, backwards over the finished lattice
  (wf_search_lattice: one more launch keeps the arcs with alpha + score + beta within output_beam of the best path):
  ``get_tot_scores(True, True)`` is the total of that smaller lattice; the best path and the tropical score do not
  depend on it.  ``output_beam=None`` or inf skips the backward launch (total over everything search_beam kept).
  k2's own beam arithmetic is parity unpinned (the reference holds no fixture), see INTEGRATION.md.
* ``intersect_dense(graphs, dense, output_beam)`` -> the exact sweeps (wf_tot_scores / wf_viterbi) over the whole
  product, as in training (graph.py:61-62).
"""
import math

import torch

from .fsa import DenseFsaVec, Fsa, create_fsa_vec
from .graph import best_path, tot_scores

from .graph import pruned_search

__all__ = ["DenseLattice", "OneBest", "get_lattice", "intersect_dense", "one_best_decoding", "shortest_path",
           "get_aux_labels", "load_decoding_graph", "save_decoding_graph", "linear_fst", "connect", "compose_words",
           "compose", "WordFst", "DenseLatticeW", "invert"]


class DenseLattice:
    """graph x dense log-probs (x beams), not materialised.  ``search`` = None: exact sweeps over the whole product;
    else dict(search_beam, min_active_states, max_active_states): queries go through the pruned search, run once per
    score precision and cached."""

    def __init__(self, graph: Fsa, dense: DenseFsaVec, search=None):
        self.graph, self.dense, self.search = graph, dense, search
        self._res = {}

    def _searched(self, use_double_scores: bool):
        key = bool(use_double_scores)
        if key not in self._res:
            self._res[key] = pruned_search(self.graph, self.dense, log_tot=True, use_double_scores=key, **self.search)
        return self._res[key]

    def get_tot_scores(self, use_double_scores: bool, log_semiring: bool):
        """k2.Fsa.get_tot_scores(use_double_scores, log_semiring) -- positional order as at lattice_analysis.py:128."""
        if self.search is None:
            return tot_scores(self.graph, self.dense, log_semiring=log_semiring, use_double_scores=use_double_scores)
        r = self._searched(use_double_scores)
        if log_semiring:
            return r["tot_log"]
        return r["scores"].double() if use_double_scores else r["scores"]


class OneBest:
    """Best path per utterance (k2 returns a linear FsaVec): ``labels`` per utterance end with -1 like the labels of
    k2's linear FSAs, ``aux_labels`` keep every arc's aux label (0 = epsilon, -1 on the final arc)."""

    def __init__(self, graph, res):
        arcs = res["arcs"].cpu().numpy()
        self.scores = res["scores"]
        self.arcs = res["arcs"]
        self.labels, self.aux_labels = [], []
        for row in arcs:
            ids = row[row >= 0]
            self.labels.append([int(x) for x in graph.arcs[ids, 2]])
            self.aux_labels.append([int(x) for x in graph.aux_labels[ids]] if graph.aux_labels is not None else [])

    def get_tot_scores(self, use_double_scores: bool, log_semiring: bool):
        """A single path: both semirings give the path's score (lattice_analysis.py:167-168)."""
        return self.scores.double() if use_double_scores else self.scores


def _dense_from(log_prob, log_prob_len):
    B = log_prob.shape[0]
    lens = torch.as_tensor(log_prob_len).reshape(-1).tolist()
    seg = torch.tensor([[i, 0, int(lens[i])] for i in range(B)], dtype=torch.int32)
    return DenseFsaVec(log_prob, seg)


def _as_vec(graph, B):
    if graph.utt2fsa is None and graph.num_fsas == 1 and B >= 1:
        return create_fsa_vec([graph] * B)      # one decoding graph for the batch: stored once (shared-graph sweep)
    return graph


def get_lattice(log_prob, log_prob_len, decoding_graph, search_beam=20, output_beam=8, min_active_states=30,
                max_active_states=10000, subsampling_factor=1):
    """k2.get_lattice (icefall-style signature; lattice_analysis.py:125).  log_prob: [B, T, V] on the GPU.
    The search itself runs at the first query (get_tot_scores / one_best_decoding)."""
    search = dict(search_beam=float(search_beam), min_active_states=int(min_active_states), max_active_states=int(max_active_states))
    if output_beam is not None and math.isfinite(float(output_beam)):
        search["output_beam"] = float(output_beam)   # (the backward pass over the finished lattice)
    return DenseLattice(_as_vec(decoding_graph, log_prob.shape[0]), _dense_from(log_prob, log_prob_len), search)


def intersect_dense(a_fsas, b_fsas, output_beam, max_states=15000000, max_arcs=1073741824):
    """k2.intersect_dense(graphs, dense_fsa_vec, output_beam) (graph.py:61-62, lattice_analysis.py:147,153)."""
    B = b_fsas.supervision_segments.shape[0]
    return DenseLattice(_as_vec(a_fsas, B), b_fsas)


class WordFst:
    """k2.linear_fst(labels, aux_labels) as lattice_analysis.py:137,143 uses it (W = k2.linear_fst(words, words)): per
    utterance a word sequence to restrict a lattice or a decoding graph to."""

    def __init__(self, words):
        self.words = [[int(w) for w in ws] for ws in words]


def linear_fst(labels, aux_labels=None):
    """k2.linear_fst (lattice_analysis.py:137,143).  labels: a list of word-id lists (one per utterance)."""
    if labels and not isinstance(labels[0], (list, tuple)):
        labels = [labels]
    return WordFst(labels)


def invert(x):
    """k2.invert as lattice_analysis.py:126 uses it -- twice in a row, to turn ragged aux labels into a tensor: the pair
    is the identity, and so is this (a lattice here is not materialised; aux labels are one id per arc already)."""
    return x


def connect(x):
    """k2.connect (lattice_analysis.py:139,140,145): compose() below only ever builds trimmed results."""
    return x


class DenseLatticeW:
    """`k2.connect(k2.compose(lattice, W))` (lattice_analysis.py:139), not materialised: the search's lattice and the word
    sequences; get_tot_scores runs wf_search_lattice_words."""

    def __init__(self, lattice, wfst):
        if lattice.search is None:
            raise ValueError("compose(lattice, W): the lattice of intersect_dense is the whole product -- compose the "
                             "decoding graph with W instead (lattice_analysis.py:140)")
        self.lattice, self.words = lattice, wfst.words
        self._res = {}

    def get_tot_scores(self, use_double_scores: bool, log_semiring: bool):
        from .graph import lattice_words
        key = bool(use_double_scores)
        if key not in self._res:
            s = self.lattice.search
            self._res[key] = lattice_words(self.lattice.graph, self.lattice.dense, self.words, s["search_beam"],
                                           s.get("output_beam"), s["min_active_states"], s["max_active_states"], key)
        r = self._res[key]
        if log_semiring:
            return r["tot_words"]
        return r["best_words"].double() if use_double_scores else r["best_words"]


def compose(a, b, treat_epsilons_specially=True):
    """k2.compose as utils/lattice_analysis.py:139,140,145 calls it, with W = linear_fst(words, words) on the right:
    compose(lattice, W) -> the lattice restricted to W (DenseLatticeW); compose(TLG, W) -> the decoding graph restricted to
    W (compose_words; one graph per word sequence)."""
    if not isinstance(b, WordFst):
        raise TypeError("compose: the right-hand side must be linear_fst(words, words)")
    if isinstance(a, DenseLattice):
        return DenseLatticeW(a, b)
    if isinstance(a, Fsa):
        gs = [compose_words(a, w) for w in b.words]
        return gs[0] if len(gs) == 1 else create_fsa_vec(gs)
    raise TypeError("compose: expected a lattice or a decoding graph on the left")


def compose_words(graph: Fsa, words):
    """`k2.connect(k2.compose(TLG, W))` for a linear word acceptor W (lattice_analysis.py:140,145): the decoding graph
    restricted to the paths whose non-epsilon aux labels spell `words`.  Product states (k words consumed, state), built
    with numpy and trimmed to the states that lie on a complete path (scipy BFS both ways); arcs stay sorted by source
    state, the start state stays 0 and the super-final state stays last.  Host-side graph work, like compile_TLG's."""
    import numpy as np
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order
    assert graph.num_fsas == 1 and graph.aux_labels is not None
    a = graph.arcs
    src, dst, lab, aux = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2], graph.aux_labels
    S = int(graph.fsa_num_states[0])
    n = len(words)
    em, fin = lab >= 0, lab < 0
    eps = np.nonzero(em & (aux <= 0))[0]
    parts_src, parts_dst, parts_arc = [], [], []
    for k in range(n + 1):
        parts_src.append(k * S + src[eps]); parts_dst.append(k * S + dst[eps]); parts_arc.append(eps)
        if k < n:
            w = np.nonzero(em & (aux == int(words[k])))[0]
            parts_src.append(k * S + src[w]); parts_dst.append((k + 1) * S + dst[w]); parts_arc.append(w)
    FINAL = (n + 1) * S
    f = np.nonzero(fin)[0]
    parts_src.append(n * S + src[f]); parts_dst.append(np.full(f.size, FINAL, dtype=np.int64)); parts_arc.append(f)
    ps, pd, pa = np.concatenate(parts_src), np.concatenate(parts_dst), np.concatenate(parts_arc)
    N = FINAL + 1
    ones = np.ones(ps.size, dtype=np.int8)
    fwd = breadth_first_order(csr_matrix((ones, (ps, pd)), shape=(N, N)), 0, directed=True, return_predecessors=False)
    bwd = breadth_first_order(csr_matrix((ones, (pd, ps)), shape=(N, N)), FINAL, directed=True, return_predecessors=False)
    keep = np.zeros(N, dtype=bool)
    keep[np.intersect1d(fwd, bwd)] = True
    if not (keep[0] and keep[FINAL]):
        # no path spells these words: a start state and a final state without arcs (every total is -inf)
        return Fsa(np.zeros((0, 4), dtype=np.int32), np.zeros(0, dtype=np.int32), [0, 0], [2])
    ok = keep[ps] & keep[pd]
    ps, pd, pa = ps[ok], pd[ok], pa[ok]
    new_id = np.cumsum(keep) - 1                     # increasing in (k, state): 0 stays first, FINAL last
    order = np.lexsort((pa, new_id[ps]))             # by source state, then by the original arc index
    ps, pd, pa = ps[order], pd[order], pa[order]
    arcs = np.stack([new_id[ps], new_id[pd], lab[pa].astype(np.int64), a[pa, 3].astype(np.int64)], 1).astype(np.int32)
    return Fsa(arcs, aux[pa].astype(np.int32), [0, arcs.shape[0]], [int(keep.sum())])


def one_best_decoding(lattice, use_double_scores=True):
    """k2.one_best_decoding (lattice_analysis.py:133,158); ties -> lowest arc index (north_star)."""
    if lattice.search is not None:
        return OneBest(lattice.graph, lattice._searched(False))
    return OneBest(lattice.graph, best_path(lattice.graph, lattice.dense, use_double_scores=bool(use_double_scores)))


def shortest_path(lattice, use_double_scores=True):
    """k2.shortest_path (lattice_analysis.py:79)."""
    return one_best_decoding(lattice, use_double_scores)


def get_aux_labels(one_best):
    """k2.get_aux_labels (lattice_analysis.py:80,134): the non-epsilon aux labels (word ids) of each best path."""
    return [[x for x in ax if x > 0] for ax in one_best.aux_labels]


def load_decoding_graph(path, device=None):
    """Reads a decoding graph saved as ``torch.save(TLG.as_dict(), f"{lang_dir}/TLG.pt")`` (utils/compile_TLG.py:122-129,
    read back at utils/lattice_analysis.py:199 with k2.Fsa.from_dict).  Only a loader that executes nothing from the file
    is used (``torch.load(weights_only=True)``): the dict must hold tensors -- ``arcs`` int32 [A, 4] (src, dst, label,
    score bits) and optionally ``aux_labels`` int32 [A], or the pair ``aux_labels.row_splits`` / ``aux_labels.values``
    for ragged aux labels (what save_decoding_graph writes).  A file whose ``aux_labels`` is a pickled k2.RaggedTensor
    (compile_TLG.py:85 makes it ragged) is refused by that loader; re-export it with tensors only."""
    try:
        d = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:   # the safe loader refuses custom classes (k2.RaggedTensor)
        raise ValueError("%s: not loadable with torch.load(weights_only=True) (%s); save the graph with tensor-only "
                         "fields, e.g. waterfall_amd.decode.save_decoding_graph" % (path, type(e).__name__)) from e
    if not isinstance(d, dict) or "arcs" not in d:
        raise ValueError("%s: expected the dict of k2.Fsa.as_dict() with an 'arcs' tensor" % path)
    arcs = d["arcs"].to(torch.int32).numpy()
    if arcs.ndim != 2 or arcs.shape[1] != 4:
        raise ValueError("%s: 'arcs' must be int32 [num_arcs, 4]" % path)
    aux = None
    if "aux_labels" in d and torch.is_tensor(d["aux_labels"]):
        aux = d["aux_labels"].to(torch.int32).numpy()
    elif "aux_labels.row_splits" in d:
        # ragged aux labels: at most one output symbol per arc survives remove_values_eq(0) in every graph the
        # reference builds (word-level L has one word per arc); longer rows are rejected rather than truncated
        import numpy as np
        rs = d["aux_labels.row_splits"].to(torch.int64).numpy()
        vals = d["aux_labels.values"].to(torch.int32).numpy()
        n = np.diff(rs)
        if (n > 1).any():
            raise ValueError("%s: arcs with more than one aux label are not supported" % path)
        aux = np.zeros(arcs.shape[0], dtype=np.int32)
        aux[n == 1] = vals
        aux[arcs[:, 2] == -1] = -1
    if aux is None:
        import numpy as np
        aux = np.where(arcs[:, 2] >= 0, 0, -1).astype("int32")
    if aux.shape[0] != arcs.shape[0]:
        raise ValueError("%s: %d aux labels for %d arcs" % (path, aux.shape[0], arcs.shape[0]))
    num_states = int(arcs[:, :2].max()) + 1 if arcs.shape[0] else 0
    return Fsa(arcs, aux, [0, arcs.shape[0]], [num_states])


def save_decoding_graph(graph: Fsa, path):
    """Tensor-only counterpart of ``torch.save(TLG.as_dict(), path)`` (compile_TLG.py:129) that load_decoding_graph and
    k2.Fsa.from_dict both read: {'arcs': int32 [A, 4], 'aux_labels': int32 [A]}."""
    d = {"arcs": torch.from_numpy(graph.arcs.copy())}
    if graph.aux_labels is not None:
        d["aux_labels"] = torch.from_numpy(graph.aux_labels.copy())
    torch.save(d, path)
