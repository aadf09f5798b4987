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


def compile_ctc_k2_graphs(topo, targets, device=None, stream=None):
    """The whole ``ctc_k2`` numerator chain of models.py:263-270 in one call.  device = a CUDA device: the graphs are
    built ON that device (compile_ctc_k2_graphs_device below); otherwise on the host."""
    if device is not None and str(device) != "cpu":
        return compile_ctc_k2_graphs_device(topo, targets, device, stream)
    return compose(topo, arc_sort(remove_epsilon_and_add_self_loops(linear_fsa(targets))),
                   treat_epsilons_specially=False)


# ---- the numerator compiler on the device (wf_compile.hip) ---------------------------------------

_COMPILE_STREAMS = {}


def _compile_stream(device):
    import torch
    s = _COMPILE_STREAMS.get(device.index)
    if s is None:
        s = _COMPILE_STREAMS[device.index] = torch.cuda.Stream(device)
    return s


def _topo_device_table(topo, device):
    """(host copy, device copy) of wf_topo_device_build's table for a topology; cached on the object per device."""
    import torch
    tab = _topo_tables(topo)
    cache = tab.setdefault("device_tables", {})
    key = str(device)
    if key not in cache:
        lib = _lib.lib()
        ps, pt = tab["phone_state"], tab["phone_token"]
        args = (tab["arcs"].ctypes.data, tab["aux"].ctypes.data, tab["arcs"].shape[0], tab["num_states"], tab["index"].ctypes.data,
                tab["index"].size, tab["ctc_cross"], ps.ctypes.data if ps is not None else None,
                pt.ctypes.data if pt is not None else None, ps.size if ps is not None else 0)
        n = C.c_int64(0)
        _lib.check(lib.wf_topo_device_build(*args, None, 0, C.byref(n)))
        host = np.empty(n.value, dtype=np.int32)
        _lib.check(lib.wf_topo_device_build(*args, host.ctypes.data, host.size, C.byref(n)))
        cache[key] = (host, torch.from_numpy(host).to(device))
    return cache[key]


class _DeviceCompileJob:
    """One batch in flight through numerator_compile_kernel: the device buffers, the pinned copies of what the host needs
    to plan the compute calls (blob header + per-FSA info words, arc offsets, states, status) and the event behind them."""

    def __init__(self, topo, lin, device, stream):
        import torch
        lib = _lib.lib()
        self.topo, self.lin, self.device = topo, lin, device
        self.B = B = len(lin)
        host_tab, dev_tab = _topo_device_table(topo, device)
        off = lin.off
        nb, cap, nw = C.c_size_t(0), C.c_int64(0), C.c_size_t(0)
        _lib.check(lib.wf_numerator_compile_device_bytes(host_tab.ctypes.data, off.ctypes.data, B, C.byref(nb), C.byref(cap), C.byref(nw)))
        self.stream = stream if stream is not None else _compile_stream(device)
        nhdr = (8 + 64 * B + 3) // 4 * 4
        with torch.cuda.device(device), torch.cuda.stream(self.stream):
            # (allocated under the compile stream: the caching allocator orders their reuse after this stream's work)
            flat = lin.flat if lin.flat.size else np.zeros(1, dtype=np.int32)
            host_in = torch.empty(flat.size + B + 1, dtype=torch.int32, pin_memory=True)
            host_in[:flat.size] = torch.from_numpy(flat)
            host_in[flat.size:] = torch.from_numpy(off)
            dev_in = host_in.to(device, non_blocking=True)
            self.blob = torch.empty(nb.value // 4, dtype=torch.int32, device=device)
            self.arcs_dev = torch.empty((max(cap.value, 1), 4), dtype=torch.int32, device=device)
            self.aux_dev = torch.empty(max(cap.value, 1), dtype=torch.int32, device=device)
            self.meta_dev = torch.empty(3 * B + 1, dtype=torch.int32, device=device)
            work = torch.empty(nw.value, dtype=torch.uint8, device=device)
            _lib.check(lib.wf_numerator_compile_device(host_tab.ctypes.data, C.c_void_p(dev_tab.data_ptr()), C.c_void_p(dev_in.data_ptr()),
                                                       C.c_void_p(dev_in[flat.size:].data_ptr()), off.ctypes.data, B,
                                                       C.c_void_p(self.blob.data_ptr()), nb.value, C.c_void_p(self.arcs_dev.data_ptr()),
                                                       C.c_void_p(self.aux_dev.data_ptr()), cap.value, C.c_void_p(self.meta_dev.data_ptr()),
                                                       C.c_void_p(work.data_ptr()), nw.value, C.c_void_p(self.stream.cuda_stream)))
            self.hdr_pin = torch.empty(nhdr, dtype=torch.int32, pin_memory=True)
            self.meta_pin = torch.empty(3 * B + 1, dtype=torch.int32, pin_memory=True)
            self.hdr_pin.copy_(self.blob[:nhdr], non_blocking=True)
            self.meta_pin.copy_(self.meta_dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(self.stream)
            self._keep = (host_in, dev_in, work, dev_tab)
        self.joined = set()

    def finish(self, fsa):
        """Waits for the header; adopts the device blob, or -- when an utterance was beyond the kernel's limits -- compiles
        the batch on the host instead."""
        self.event.synchronize()
        hdr = self.hdr_pin.numpy()
        if int(hdr[0]) == 0:        # the blob's magic word is 0: some status != 0
            host = compose(self.topo, self.lin)
            fsa._pending = None
            fsa.arcs, fsa.aux_labels = host.arcs, host.aux_labels
            fsa.fsa_arc_offsets, fsa.fsa_num_states = host.fsa_arc_offsets, host.fsa_num_states
            fsa._blob_host, fsa._blob_pin = host._blob_host, host._blob_pin
            fsa.compiled_on = "host (device compiler status %s)" % sorted(set(int(x) for x in self.meta_pin.numpy()[2 * self.B + 1:]) - {0})
            return
        fsa._blob_host = hdr
        fsa._blob_dev[str(self.device)] = self.blob

    def join_stream(self, device):
        import torch
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream not in self.joined and cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_event(self.event)
            self.blob.record_stream(cur)
            self.joined.add(cur.cuda_stream)

    def fetch_arrays(self, fsa):
        if fsa._blob_host is None:
            self.finish(fsa)
            if fsa._pending is None:
                return
        meta = self.meta_pin.numpy()
        B = self.B
        off = meta[:B + 1].copy()
        n = int(off[B])
        fsa.__dict__["_fsa_arc_offsets"] = off
        fsa.__dict__["_fsa_num_states"] = meta[B + 1:2 * B + 1].copy()
        fsa.__dict__["_arcs"] = self.arcs_dev[:n].cpu().numpy()
        fsa.__dict__["_aux_labels"] = self.aux_dev[:n].cpu().numpy()


def compile_ctc_k2_graphs_device(topo, targets, device, stream=None):
    """compile_ctc_k2_graphs on the DEVICE: the transcripts are uploaded (one small asynchronous copy) and
    numerator_compile_kernel builds the graphs and their prepared tables; the host does no graph work.  Returns at once; the
    Fsa waits for the kernel's header (a few hundred bytes per utterance, copied back asynchronously) only when a compute
    call plans with it, so a training loop that compiles step i + 1's graphs before it runs step i's loss hides the
    compiler behind the loss kernels (the compile kernels run on a stream of their own).  Batches the kernel cannot take
    (WF_NC_* limits in csrc/wf_internal.h) are compiled on the host, transparently."""
    import torch
    device = torch.device(device)
    lin = arc_sort(remove_epsilon_and_add_self_loops(linear_fsa(targets)))
    if len(lin) == 0:
        raise ValueError("compose: empty batch")
    fsa = Fsa.__new__(Fsa)
    fsa._pending = None
    fsa.utt2fsa = None
    fsa.device = device
    fsa._blob_host = None
    fsa._blob_pin = None
    fsa._blob_dev = {}
    fsa.compiled_on = "device"
    for name in ("_arcs", "_aux_labels", "_fsa_arc_offsets", "_fsa_num_states"):
        fsa.__dict__[name] = None
    fsa._pending = _DeviceCompileJob(topo, lin, device, stream)
    return fsa
