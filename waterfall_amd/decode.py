"""The decoding-side k2 calls of ``utils/lattice_analysis.py:120-168`` with the same names and call shapes:

    lattice = get_lattice(log_prob, log_prob_len, TLG, search_beam=..., output_beam=...)     # :125
    lattice.get_tot_scores(True, True) / (True, False)                                       # :128-129
    one_best = one_best_decoding(lattice); get_aux_labels(one_best); one_best.labels         # :133-135
    ETLGW = intersect_dense(TLGW, dense_fsa_vec, 10); ETLGW.get_tot_scores(True, True)        # :147-149
    shortest_path(lattice, True)                                                             # :79

k2 materialises the (pruned) lattice; here a *lattice* is the pair (graph, dense log-probs) and every query runs
the exact sweep over the whole product (wf_tot_scores / wf_viterbi; graphs of any size), so
``search_beam`` / ``output_beam`` / ``min_active_states`` / ``max_active_states`` are accepted and not needed: the
exact search visits a superset of what any beam keeps (best path and tropical score are those of an infinite beam;
the log-semiring total includes the mass k2's beams prune away).  Beam-pruned frame-synchronous search for graphs
too large for the exact sweep (10^7+ arcs) is SURVEY.md 8(f) rank 3 and not built.
"""
import torch

from .fsa import DenseFsaVec, Fsa, create_fsa_vec
from .graph import best_path, tot_scores

__all__ = ["DenseLattice", "OneBest", "get_lattice", "intersect_dense", "one_best_decoding", "shortest_path",
           "get_aux_labels"]


class DenseLattice:
    """graph x dense log-probs, not materialised."""

    def __init__(self, graph: Fsa, dense: DenseFsaVec):
        self.graph, self.dense = graph, dense

    def get_tot_scores(self, use_double_scores: bool, log_semiring: bool):
        """k2.Fsa.get_tot_scores(use_double_scores, log_semiring) -- positional order as at lattice_analysis.py:128."""
        return tot_scores(self.graph, self.dense, log_semiring=log_semiring, use_double_scores=use_double_scores)


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
    """k2.get_lattice (icefall-style signature; lattice_analysis.py:125).  log_prob: [B, T, V] on the GPU."""
    return DenseLattice(_as_vec(decoding_graph, log_prob.shape[0]), _dense_from(log_prob, log_prob_len))


def intersect_dense(a_fsas, b_fsas, output_beam, max_states=15000000, max_arcs=1073741824):
    """k2.intersect_dense(graphs, dense_fsa_vec, output_beam) (graph.py:61-62, lattice_analysis.py:147,153)."""
    B = b_fsas.supervision_segments.shape[0]
    return DenseLattice(_as_vec(a_fsas, B), b_fsas)


def one_best_decoding(lattice, use_double_scores=True):
    """k2.one_best_decoding (lattice_analysis.py:133,158); ties -> lowest arc index (north_star)."""
    return OneBest(lattice.graph, best_path(lattice.graph, lattice.dense, use_double_scores=False))


def shortest_path(lattice, use_double_scores=True):
    """k2.shortest_path (lattice_analysis.py:79)."""
    return one_best_decoding(lattice, use_double_scores)


def get_aux_labels(one_best):
    """k2.get_aux_labels (lattice_analysis.py:80,134): the non-epsilon aux labels (word ids) of each best path."""
    return [[x for x in ax if x > 0] for ax in one_best.aux_labels]
