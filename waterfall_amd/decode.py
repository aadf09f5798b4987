"""The decoding-side k2 calls of ``utils/lattice_analysis.py:120-168`` with the same names and call shapes:

    lattice = get_lattice(log_prob, log_prob_len, TLG, search_beam=..., output_beam=...)     # :125
    lattice.get_tot_scores(True, True) / (True, False)                                       # :128-129
    one_best = one_best_decoding(lattice); get_aux_labels(one_best); one_best.labels         # :133-135
    ETLGW = intersect_dense(TLGW, dense_fsa_vec, 10); ETLGW.get_tot_scores(True, True)        # :147-149
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
           "get_aux_labels", "load_decoding_graph", "save_decoding_graph"]


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
