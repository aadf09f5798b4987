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
import itertools
import os

import numpy as np

from . import _lib
from .fsa import Fsa
from .topology import Topology


class LinearFsaVec:
    """Result of ``linear_fsa``: one label sequence per utterance (labels >= 1; 0 is epsilon), kept as one flat int32
    array + offsets -- the form the native compiler takes."""

    def __init__(self, flat, off, self_loops=False):
        self.flat, self.off, self.self_loops = flat, off, self_loops

    @property
    def seqs(self):
        return [self.flat[self.off[i]:self.off[i + 1]] for i in range(len(self.off) - 1)]

    def __len__(self):
        return len(self.off) - 1

    def to(self, device):
        return self


def linear_fsa(labels, device=None):
    """k2.linear_fsa(List[List[int]]) (models.py:263; datapipe.py:244 for word ids).  Also takes a pair
    (flat int32 array, lengths) as a data loader would hold the batch's targets."""
    if isinstance(labels, tuple) and len(labels) == 2:
        flat = np.ascontiguousarray(labels[0], dtype=np.int32).reshape(-1)
        lens = np.asarray(labels[1], dtype=np.int64).reshape(-1)
    else:
        if hasattr(labels, "tolist"):
            labels = labels.tolist()
        if labels and not isinstance(labels[0], (list, tuple, np.ndarray)):
            labels = [labels]          # k2.linear_fsa(List[int]): a single sequence
        B = len(labels)
        lens = np.fromiter((len(s) for s in labels), dtype=np.int64, count=B)
        flat = np.fromiter(itertools.chain.from_iterable(labels), dtype=np.int32, count=int(lens.sum()))
    off = np.zeros(lens.size + 1, dtype=np.int32)
    np.cumsum(lens, out=off[1:])
    if flat.size and int(flat.min()) < 1:
        raise ValueError("linear_fsa labels must be >= 1 (0 is epsilon)")
    if int(off[-1]) != flat.size:
        raise ValueError("linear_fsa: lengths do not add up to the number of labels")
    return LinearFsaVec(flat, off)


def remove_epsilon_and_add_self_loops(fsa):
    """k2.remove_epsilon_and_add_self_loops on a linear FSA: it has no epsilons, so this only adds the
    epsilon self-loops that let topology arcs with an <eps> output stay at the same position."""
    assert isinstance(fsa, LinearFsaVec)
    return LinearFsaVec(fsa.flat, fsa.off, self_loops=True)


def arc_sort(fsa):
    return fsa


def host_threads():
    """Host threads the numerator compiler spreads a batch over (WF_HOST_THREADS overrides; default: the CPUs this
    process may use, at most 8)."""
    v = os.environ.get("WF_HOST_THREADS")
    if v:
        return max(1, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n))


def _topo_tables(topo):
    """Arc arrays + the (state, phone) index of wf_topo_index_build for a topology; cached on the object."""
    tab = getattr(topo, "_compose_tables", None)
    if tab is not None:
        return tab
    lib = _lib.lib()
    if isinstance(topo, Topology):
        tab = topo.compose_arcs()
    else:
        assert topo.num_fsas == 1 and topo.utt2fsa is None, "topology must be a single FSA"
        tab = {"arcs": np.ascontiguousarray(topo.arcs, dtype=np.int32), "aux": np.ascontiguousarray(topo.aux_labels, dtype=np.int32),
               "num_states": int(topo.fsa_num_states[0]), "ctc_cross": 0, "phone_state": None, "phone_token": None}
    nA, nS = tab["arcs"].shape[0], tab["num_states"]
    n = C.c_int64(0)
    _lib.check(lib.wf_topo_index_words(tab["arcs"].ctypes.data, tab["aux"].ctypes.data, nA, nS, C.byref(n)))
    index = np.empty(n.value, dtype=np.int32)
    _lib.check(lib.wf_topo_index_build(tab["arcs"].ctypes.data, tab["aux"].ctypes.data, nA, nS, index.ctypes.data))
    tab["index"] = index
    try:
        topo._compose_tables = tab
    except AttributeError:
        pass
    return tab


def compose(topo, b, treat_epsilons_specially=False):
    """k2.compose(topo, linear_with_self_loops, treat_epsilons_specially=False) (models.py:269-270), for the whole
    batch in one native call (wf_compose_linear_batch, include/wf_dwfst.h).

    topo: the topology Fsa, or a ``Topology`` (whose large-table CTC form keeps the O(P^2) cross arcs implicit:
    identical result, 25 M arcs at P = 5000 never built).
    Product states (q, i), numbered in BFS order from (0, 0); topology arc (q -> q', tok : ph) gives
    (q,i) -> (q',i) if ph == 0 and (q,i) -> (q',i+1) if ph == b[i]; final arcs from (q, L) when q is
    final.  Output labels = tokens, aux_labels = phones.
    """
    if treat_epsilons_specially:
        raise NotImplementedError("the reference always composes with treat_epsilons_specially=False")
    if isinstance(b, Fsa):
        return _compose_graphs(topo, b)
    if not isinstance(b, LinearFsaVec) or not b.self_loops:
        raise TypeError("compose expects remove_epsilon_and_add_self_loops(linear_fsa(...)) or an epsilon-free Fsa vector on the right")
    lib = _lib.lib()
    tab = _topo_tables(topo)
    B = len(b)
    if B == 0:
        raise ValueError("compose: empty batch")
    flat = b.flat if b.flat.size else np.zeros(1, dtype=np.int32)
    off = b.off
    arc_off = np.zeros(B + 1, dtype=np.int32)
    ns = np.zeros(B, dtype=np.int32)
    ps, pt = tab["phone_state"], tab["phone_token"]
    total = int(off[-1])
    cap = 8 * total + 16 * B + 64
    words = 64 + 48 * B + 40 * cap            # blob estimate; the call reports what it needs when this is short
    need = C.c_size_t(0)
    nthreads = host_threads()
    while True:
        oa = np.empty((cap, 4), dtype=np.int32)
        ox = np.empty(cap, dtype=np.int32)
        pin, blob = _alloc_blob(words)
        rc = lib.wf_numerator_compile(tab["arcs"].ctypes.data, tab["aux"].ctypes.data, tab["arcs"].shape[0], tab["num_states"],
                                      tab["index"].ctypes.data, tab["ctc_cross"],
                                      ps.ctypes.data if ps is not None else None, pt.ctypes.data if pt is not None else None,
                                      ps.size if ps is not None else 0, flat.ctypes.data, off.ctypes.data, B,
                                      oa.ctypes.data, ox.ctypes.data, cap, arc_off.ctypes.data, ns.ctypes.data,
                                      blob.ctypes.data, blob.nbytes, C.byref(need), nthreads)
        if rc == -2 and (arc_off[B] > cap or need.value > blob.nbytes):   # WF_ENOSPC: sizes were reported
            cap = max(cap, int(arc_off[B]))
            words = max(words, need.value // 4)
            continue
        _lib.check(rc)
        break
    n = int(arc_off[B])
    fsa = Fsa(oa[:n], ox[:n], arc_off, ns)
    fsa._blob_host = blob[:int(blob[2])]      # WF_H_WORDS: the extent that travels to the device
    fsa._blob_pin = pin
    return fsa


def _alloc_blob(words):
    """int32 buffer for a prepared blob: pinned (so that Fsa.blob_dev uploads it with one asynchronous copy; torch's
    caching host allocator recycles it only after that copy has finished) when a GPU is present, plain numpy otherwise."""
    import torch
    # pinned only where this process already talks to the GPU: a forked DataLoader worker that compiles graphs must not
    # initialise (and cannot use) the parent's CUDA context
    if torch.cuda.is_available() and torch.cuda.is_initialized() and not torch.cuda._is_in_bad_fork():
        pin = torch.empty(words, dtype=torch.int32, pin_memory=True)
        return pin, pin.numpy()
    return None, np.empty(words, dtype=np.int32)


def _compose_graphs(topo, b):
    """compose(topo, b) for a vector of arbitrary epsilon-free phone graphs b (labels = phones >= 1, aux_labels = word
    ids, one graph per utterance; the epsilon self-loops of remove_epsilon_and_add_self_loops are implied):
    datapipe.py:249-254.  One native call (wf_compose_fsa_batch)."""
    if b.utt2fsa is not None:
        raise ValueError("compose: the right-hand side must hold one graph per utterance")
    tab = _topo_tables(topo)
    if tab["ctc_cross"]:
        raise NotImplementedError("the factored large-table CTC topology composes with linear transcripts only")
    lib = _lib.lib()
    B = b.num_fsas
    ba = np.ascontiguousarray(b.arcs, dtype=np.int32)
    bx = np.ascontiguousarray(b.aux_labels, dtype=np.int32) if b.aux_labels is not None else None
    boff = np.ascontiguousarray(b.fsa_arc_offsets, dtype=np.int32)
    bns = np.ascontiguousarray(b.fsa_num_states, dtype=np.int32)
    arc_off = np.zeros(B + 1, dtype=np.int32)
    ns = np.zeros(max(B, 1), dtype=np.int32)
    cap = 16 * int(ba.shape[0]) + 64
    while True:
        oa = np.empty((cap, 4), dtype=np.int32)
        ox = np.empty(cap, dtype=np.int32)
        rc = lib.wf_compose_fsa_batch(tab["arcs"].ctypes.data, tab["aux"].ctypes.data, tab["arcs"].shape[0], tab["num_states"],
                                      tab["index"].ctypes.data, ba.ctypes.data, bx.ctypes.data if bx is not None else None,
                                      boff.ctypes.data, bns.ctypes.data, B, oa.ctypes.data, ox.ctypes.data, cap,
                                      arc_off.ctypes.data, ns.ctypes.data)
        if rc == -2 and arc_off[B] > cap:
            cap = int(arc_off[B])
            continue
        _lib.check(rc)
        break
    n = int(arc_off[B])
    return Fsa(oa[:n], ox[:n], arc_off, ns[:B])


def compile_ctc_k2_graphs(topo, targets):
    """The whole ``ctc_k2`` numerator chain of models.py:263-270 in one call."""
    return compose(topo, arc_sort(remove_epsilon_and_add_self_loops(linear_fsa(targets))),
                   treat_epsilons_specially=False)
