"""Numerator-graph compiler: the k2 graph-algebra chain of the reference's ``ctc_k2`` branch

    trans_fsa = k2.linear_fsa(targets); trans_fsa = k2.remove_epsilon_and_add_self_loops(trans_fsa)
    trans_fsa = k2.arc_sort(trans_fsa)
    decoding_graph = k2.compose(topo, trans_fsa, treat_epsilons_specially=False)

(``waterfall/models.py:263-270``) as one specialised product construction done on the host by
``wf_compose_linear`` (include/wf_dwfst.h).  The same four names are kept so the call site ports
verbatim: ``linear_fsa`` only records the label sequences, the two middle steps are no-ops on that
record, and ``compose`` builds the product graphs.
"""
import ctypes as C

import numpy as np

from . import _lib
from .fsa import Fsa
from .topology import Topology


class LinearFsaVec:
    """Result of ``linear_fsa``: one label sequence per utterance (labels >= 1; 0 is epsilon)."""

    def __init__(self, seqs, self_loops=False):
        self.seqs = [np.ascontiguousarray(s, dtype=np.int32).reshape(-1) for s in seqs]
        for s in self.seqs:
            if s.size and s.min() < 1:
                raise ValueError("linear_fsa labels must be >= 1 (0 is epsilon)")
        self.self_loops = self_loops

    def to(self, device):
        return self


def linear_fsa(labels, device=None):
    """k2.linear_fsa(List[List[int]]) (models.py:263; datapipe.py:244 for word ids)."""
    if hasattr(labels, "tolist"):
        labels = labels.tolist()
    return LinearFsaVec(labels)


def remove_epsilon_and_add_self_loops(fsa):
    """k2.remove_epsilon_and_add_self_loops on a linear FSA: it has no epsilons, so this only adds the
    epsilon self-loops that let topology arcs with an <eps> output stay at the same position."""
    assert isinstance(fsa, LinearFsaVec)
    return LinearFsaVec(fsa.seqs, self_loops=True)


def arc_sort(fsa):
    return fsa


def compose(topo, b, treat_epsilons_specially=False):
    """k2.compose(topo, linear_with_self_loops, treat_epsilons_specially=False) (models.py:269-270).

    topo: the topology Fsa, or a lazy ``Topology`` (restricted per transcript: identical result,
    needed for the O(P^2) CTC topology at large P).
    Product states (q, i), numbered in BFS order from (0, 0); topology arc (q -> q', tok : ph) gives
    (q,i) -> (q',i) if ph == 0 and (q,i) -> (q',i+1) if ph == b[i]; final arcs from (q, L) when q is
    final.  Output labels = tokens, aux_labels = phones.
    """
    if treat_epsilons_specially:
        raise NotImplementedError("the reference always composes with treat_epsilons_specially=False")
    if not isinstance(b, LinearFsaVec) or not b.self_loops:
        raise TypeError("compose expects remove_epsilon_and_add_self_loops(linear_fsa(...)) on the right")
    lazy = isinstance(topo, Topology)
    if not lazy:
        assert topo.num_fsas == 1 and topo.utt2fsa is None, "topology must be a single FSA"
    lib = _lib.lib()
    arcs_l, aux_l, off, ns = [], [], [0], []
    for seq in b.seqs:
        t = topo.restricted(seq) if lazy else topo
        ta = t.arcs
        tx = t.aux_labels
        nA, nS = ta.shape[0], int(t.fsa_num_states[0])
        na, nst = C.c_int32(0), C.c_int32(0)
        cap = max(64, 8 * (len(seq) + 2))
        while True:
            oa = np.empty((cap, 4), dtype=np.int32)
            ox = np.empty(cap, dtype=np.int32)
            rc = lib.wf_compose_linear(ta.ctypes.data, tx.ctypes.data, nA, nS, seq.ctypes.data, len(seq),
                                       oa.ctypes.data, ox.ctypes.data, C.byref(na), C.byref(nst), cap)
            if rc == -2 and na.value > cap:  # WF_ENOSPC: sizes were still reported
                cap = na.value
                continue
            _lib.check(rc)
            break
        arcs_l.append(oa[:na.value])
        aux_l.append(ox[:na.value])
        off.append(off[-1] + na.value)
        ns.append(nst.value)
    return Fsa(np.concatenate(arcs_l, 0), np.concatenate(aux_l, 0), off, ns)


def compile_ctc_k2_graphs(topo, targets):
    """The whole ``ctc_k2`` numerator chain of models.py:263-270 in one call."""
    return compose(topo, arc_sort(remove_epsilon_and_add_self_loops(linear_fsa(targets))),
                   treat_epsilons_specially=False)
