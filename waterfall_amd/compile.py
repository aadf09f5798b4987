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
import collections
import itertools
import weakref
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


def compose(topo, b, treat_epsilons_specially=None):
    """k2.compose(topo, linear_with_self_loops, treat_epsilons_specially=False) (models.py:269-270), for the whole
    batch in one native call (wf_compose_linear_batch, include/wf_dwfst.h).

    topo: the topology Fsa, or a ``Topology`` (whose large-table CTC form keeps the O(P^2) cross arcs implicit:
    identical result, 25 M arcs at P = 5000 never built).
    Product states (q, i), numbered in BFS order from (0, 0); topology arc (q -> q', tok : ph) gives
    (q,i) -> (q',i) if ph == 0 and (q,i) -> (q',i+1) if ph == b[i]; final arcs from (q, L) when q is
    final.  Output labels = tokens, aux_labels = phones.
    """
    from . import decode as _decode
    if isinstance(b, _decode.WordFst):   # k2.compose(lattice, W) / k2.compose(TLG, W) of utils/lattice_analysis.py:139-145
        return _decode.compose(topo, b)
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
    """The stream the compile kernels run on (one per device); device: a torch.device or its index."""
    import torch
    di = device if isinstance(device, int) else device.index
    s = _COMPILE_STREAMS.get(di)
    if s is None:
        s = _COMPILE_STREAMS[di] = torch.cuda.Stream(torch.device("cuda", di))
    return s


def _topo_device_table(topo, di):
    """(host copy, device copy, host address, device address) of wf_topo_device_build's table for a topology; cached on
    the object per device index."""
    import torch
    tab = _topo_tables(topo)
    cache = tab.get("device_tables")
    if cache is None:
        cache = tab["device_tables"] = {}
    key = di
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
        dev = torch.from_numpy(host).to(torch.device("cuda", di))
        cache[key] = (host, dev, host.ctypes.data, dev.data_ptr())
    return cache[key]


class _PinnedSlot:
    """Pinned staging for one job: the transcripts on their way up, the words the host plans with on their way back."""

    def __init__(self, n_in, n_out):
        import torch
        self.cap_in, self.cap_out = n_in, n_out
        self.pin_in = torch.empty(n_in, dtype=torch.int32, pin_memory=True)
        self.pin_out = torch.empty(n_out, dtype=torch.int32, pin_memory=True)
        self.np_in, self.np_out = self.pin_in.numpy(), self.pin_out.numpy()
        self.ptr_in, self.ptr_out = self.pin_in.data_ptr(), self.pin_out.data_ptr()
        self.job = 0            # the job that used it last (wf_numerator_compile_job's number)
        self.owner = None       # weak reference to that job's Python object, while it has not read its results out


_SLOTS = {}       # device index -> [ring of _PinnedSlot, next]
_NSLOTS = 8


def _take_slot(dev_index, n_in, n_out):
    ring = _SLOTS.get(dev_index)
    if ring is None:
        ring = _SLOTS[dev_index] = [[None] * _NSLOTS, 0]
    k = ring[1]
    ring[1] = (k + 1) % _NSLOTS
    slot = ring[0][k]
    if slot is not None:
        owner = slot.owner() if slot.owner is not None else None
        if owner is not None:
            owner.detach()          # the job before reads its results out of the slot (waits for it if need be)
        elif slot.job:
            _lib.check(_lib.lib().wf_compile_job_wait(dev_index, slot.job))    # its upload must have left the slot
        slot.owner, slot.job = None, 0
    if slot is None or slot.cap_in < n_in or slot.cap_out < n_out:
        def up(n):
            return max(4096, 1 << int(n - 1).bit_length())
        slot = ring[0][k] = _PinnedSlot(up(n_in), up(n_out))
    return slot


_BUF_POOL = {}     # (device index, bytes) -> deque of (device buffer, the job that used it last), oldest first
_POOL_MIN, _POOL_MAX = 2, 8


def _take_buf(di, nbytes, cs):
    """A device buffer for one job: the one released longest ago (whose readers have long finished: the compile stream's wait
    for them is a formality, and the new job starts beside the loss that is running), or a new one."""
    import torch
    q = _BUF_POOL.get((di, nbytes))
    if q is not None and len(q) >= _POOL_MIN:
        return q.popleft()
    # allocated under the compile stream: the caching allocator hands out only blocks that stream may write at once
    main = torch.cuda.current_stream(di)
    torch.cuda.set_stream(cs)
    try:
        return torch.empty(nbytes // 4, dtype=torch.int32, device=cs.device), 0
    finally:
        torch.cuda.set_stream(main)


_JOB_SIZES = {}    # (id of the topology's device table, B) -> (device bytes, host words) that hold ANY batch of B transcripts


class _DeviceCompileJob:
    """One batch in flight through numerator_compile_kernel (wf_numerator_compile_job): ONE device buffer (the blob, behind it
    the arc staging rows and the work space), a pinned slot, and the job number whose event the host or a stream waits on."""

    def __init__(self, topo, flat, lens, device, stream):
        import torch
        lib = _lib.lib()
        self.dev_index = di = device.index if device.index is not None else torch.cuda.current_device()
        self.topo, self.flat, self.lens = topo, flat, lens
        self.B = B = lens.size
        tabs = _topo_device_table(topo, di)
        host_ptr, dev_ptr = tabs[2], tabs[3]
        sizes = _JOB_SIZES.get((host_ptr, B))
        if sizes is None:
            nb, nw = C.c_size_t(0), C.c_size_t(0)
            _lib.check(lib.wf_numerator_compile_job_bytes(host_ptr, None, B, C.byref(nb), C.byref(nw), (C.c_int64 * 5)()))
            sizes = _JOB_SIZES[(host_ptr, B)] = (nb.value, nw.value)
        n = flat.size
        n_in = B + 1 + n
        self.slot = slot = _take_slot(di, n_in, sizes[1])
        np_in = slot.np_in
        np_in[0] = 0
        np.cumsum(lens, out=np_in[1:B + 1])
        np_in[B + 1:n_in] = flat
        if int(np_in[B]) != n:
            raise ValueError("linear_fsa: lengths do not add up to the number of labels")
        cs = stream if stream is not None else _compile_stream(di)
        self.compile_stream = cs.cuda_stream
        self.job = 0
        self.buf, reuse_of = _take_buf(di, sizes[0], cs)
        self.joined = []
        job = C.c_int64(0)
        self.layout = lay = (C.c_int64 * 5)()
        rc = lib.wf_numerator_compile_job(host_ptr, dev_ptr, slot.ptr_in, B, self.buf.data_ptr(), sizes[0], slot.ptr_out,
                                          sizes[1], di, None, self.compile_stream, reuse_of, C.byref(job), lay)
        if rc:
            linear_fsa((flat, lens))     # (labels < 1: the same ValueError as on the host path)
            _lib.check(rc)
        self.job = slot.job = job.value
        slot.owner = weakref.ref(self)
        self.hdr = self.meta = None
        self.device = cs.device

    def __del__(self):
        # the buffer goes back to the pool, with an event behind the streams that read the graphs
        try:
            if not self.job:
                return
            n = len(self.joined)
            if n:
                _lib.lib().wf_compile_job_release(self.dev_index, self.job, (C.c_void_p * n)(*self.joined), n)
            q = _BUF_POOL.get((self.dev_index, self.buf.numel() * 4))
            if q is None:
                q = _BUF_POOL[(self.dev_index, self.buf.numel() * 4)] = collections.deque()
            if len(q) < _POOL_MAX:
                q.append((self.buf, self.job))
            else:
                import torch
                for h in self.joined:      # dropped: the allocator must know who may still read it
                    self.buf.record_stream(torch.cuda.ExternalStream(h, device=self.device))
        except Exception:                  # (interpreter shutdown)
            pass

    def detach(self):
        """Waits for the job and copies its results out of the pinned slot (which then serves another job)."""
        if self.hdr is None:
            _lib.check(_lib.lib().wf_compile_job_wait(self.dev_index, self.job))
            M = self.layout[0]
            out = self.slot.np_out
            self.meta = out[:4 * self.B + 2].copy()
            self.hdr = out[M:M + self.layout[1]].copy()
            self.slot.owner = None
            self.slot = None

    def finish(self, fsa):
        """Waits for the header; adopts the device blob, or -- when an utterance was beyond the kernel's limits -- compiles
        the batch on the host instead."""
        self.detach()
        hdr = self.hdr
        if hdr[0] == 0:        # the blob's magic word is 0: some status != 0
            host = compose(self.topo, arc_sort(remove_epsilon_and_add_self_loops(linear_fsa((self.flat, self.lens)))))
            fsa._pending = None
            fsa.arcs, fsa.aux_labels = host.arcs, host.aux_labels
            fsa.fsa_arc_offsets, fsa.fsa_num_states = host.fsa_arc_offsets, host.fsa_num_states
            fsa._blob_host, fsa._blob_pin = host._blob_host, host._blob_pin
            fsa.compiled_on = "host (device compiler status %s)" % sorted(set(int(x) for x in self.meta[2 * self.B + 1:3 * self.B + 1]) - {0})
            return
        fsa._blob_host = hdr
        fsa._blob_dev[str(self.device)] = self.buf        # (the blob starts the buffer; the arc staging rows sit behind it)
        nd = int(self.meta[4 * self.B + 1])
        fsa._all_compact, fsa._all_direct = nd == 0, nd == self.B

    def join_stream(self, device):
        import torch
        h = torch._C._cuda_getCurrentRawStream(self.dev_index)
        if h not in self.joined:
            self.joined.append(h)
            if h != self.compile_stream:
                _lib.check(_lib.lib().wf_compile_job_join(self.dev_index, self.job, h))

    def fetch_arrays(self, fsa):
        if fsa._blob_host is None:
            self.finish(fsa)
            if fsa._pending is None:
                return
        B = self.B
        off = self.meta[:B + 1].copy()
        n = int(off[B])
        fsa.__dict__["_fsa_arc_offsets"] = off
        fsa.__dict__["_fsa_num_states"] = self.meta[B + 1:2 * B + 1].copy()
        # (the job has completed -- detach waited for it -- so this copy needs no stream order of its own)
        w0 = self.layout[3] // 4
        rows = self.buf[w0:w0 + 5 * self.layout[4]].cpu().numpy().reshape(-1, 5)
        so = self.meta[3 * B + 1:4 * B + 1]
        rows = np.concatenate([rows[so[b]:so[b] + off[b + 1] - off[b]] for b in range(B)]) if n else rows[:0]
        fsa.__dict__["_arcs"] = np.ascontiguousarray(rows[:, :4])
        fsa.__dict__["_aux_labels"] = np.ascontiguousarray(rows[:, 4])


def compile_ctc_k2_graphs_device(topo, targets, device, stream=None):
    """compile_ctc_k2_graphs on the DEVICE: numerator_compile_kernel reads the transcripts from a pinned host buffer and
    builds the graphs and their prepared tables; the host does no graph work and enqueues no copy.  Returns at once; the
    Fsa waits for the kernel's header (a few hundred bytes per utterance, written to pinned host memory by the kernel)
    only when a compute call plans with it, so a training loop that compiles step i + 1's graphs before it runs step i's
    loss hides the compiler behind the loss kernels (the compile kernel runs on a stream of its own).  Batches the kernel
    cannot take (WF_NC_* limits in csrc/wf_internal.h) are compiled on the host, transparently."""
    import torch
    if not isinstance(device, torch.device):
        device = torch.device(device)
    if isinstance(targets, tuple) and len(targets) == 2:      # (flat labels, lengths), as a data loader holds them
        flat = np.ascontiguousarray(targets[0], dtype=np.int32).reshape(-1)
        lens = np.asarray(targets[1]).reshape(-1)
    else:
        lin = linear_fsa(targets)
        flat, lens = lin.flat, np.diff(lin.off)
    if lens.size == 0:
        raise ValueError("compose: empty batch")
    fsa = Fsa.__new__(Fsa)
    d = fsa.__dict__
    d["_pending"] = None
    d["utt2fsa"] = None
    d["_blob_host"] = None
    d["_blob_pin"] = None
    d["_blob_dev"] = {}
    d["compiled_on"] = "device"
    d["_arcs"] = d["_aux_labels"] = d["_fsa_arc_offsets"] = d["_fsa_num_states"] = None
    job = _DeviceCompileJob(topo, flat, lens, device, stream)
    d["device"] = job.device
    d["_pending"] = job
    return fsa
