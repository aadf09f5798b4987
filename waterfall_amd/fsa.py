"""Host-side graph containers mirroring the k2 types that appear on the reference's loss path.

* ``Fsa`` (a vector of FSAs)  <- ``k2.Fsa`` / FsaVec as built at ``waterfall/utils/datapipe.py:180``
  (``from_openfst(txt, acceptor=False)``), ``waterfall/models.py:263-270`` and
  ``waterfall/rnnp.py:91-92`` (``create_fsa_vec([topo] * B)``).
* ``DenseFsaVec``             <- ``k2.DenseFsaVec(log_probs, supervision_segments)``
  (``waterfall/models.py:256-261``).  No copy of the network output is made.

Graph structure lives on the host (numpy); ``Fsa.to(device)`` uploads the *prepared* blob built by
``wf_graph_prepare`` (include/wf_dwfst.h) once and caches it.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class Fsa:
    """A vector of FSAs in k2's arc layout.

    arcs: int32 [A,4] = (src, dst, label, score bits); state ids local to each FSA; start state 0;
    the last state of each FSA is the super-final state entered by label -1 arcs only.
    aux_labels: int32 [A] (output labels: phones or words; 0 = epsilon, -1 on final arcs).
    utt2fsa: optional int32 [B]: utterance i uses FSA utt2fsa[i] (one shared denominator topology).
    """

    requires_grad = False  # reference asserts this (models.py:272)

    def __init__(self, arcs, aux_labels, fsa_arc_offsets, fsa_num_states, utt2fsa=None):
        self._pending = None      # device compiler (compile.py): the job whose results this object adopts on first use
        self.arcs = np.ascontiguousarray(arcs, dtype=np.int32).reshape(-1, 4)
        self.aux_labels = None if aux_labels is None else np.ascontiguousarray(aux_labels, dtype=np.int32)
        self.fsa_arc_offsets = np.ascontiguousarray(fsa_arc_offsets, dtype=np.int32)
        self.fsa_num_states = np.ascontiguousarray(fsa_num_states, dtype=np.int32)
        self.utt2fsa = None if utt2fsa is None else np.ascontiguousarray(utt2fsa, dtype=np.int32)
        self.device = torch.device("cpu")
        self._blob_host = None
        self._blob_pin = None     # pinned torch tensor behind _blob_host (numerator compiler): asynchronous upload
        self._blob_dev = {}

    # The FsaVec arrays of a graph vector built by the DEVICE compiler live on the device; the host copies are fetched the
    # first time something asks for them (best_path's label extraction, tests) -- the loss path never does.
    def _lazy(name):
        def get(self):
            if self._pending is not None and self.__dict__.get("_" + name) is None:
                self._pending.fetch_arrays(self)
            return self.__dict__.get("_" + name)

        def put(self, v):
            self.__dict__["_" + name] = v
        return property(get, put)

    arcs = _lazy("arcs")
    aux_labels = _lazy("aux_labels")
    fsa_arc_offsets = _lazy("fsa_arc_offsets")
    fsa_num_states = _lazy("fsa_num_states")
    del _lazy

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def from_arcs(cls, arcs, num_states, aux_labels=None):
        """arcs: iterable of (src, dst, label, score[float]); sorted by src (stable) like k2."""
        arcs = list(arcs)
        order = sorted(range(len(arcs)), key=lambda i: arcs[i][0])
        a = np.zeros((len(arcs), 4), dtype=np.int32)
        sc = np.zeros(len(arcs), dtype=np.float32)
        for j, i in enumerate(order):
            a[j, 0], a[j, 1], a[j, 2] = arcs[i][0], arcs[i][1], arcs[i][2]
            sc[j] = arcs[i][3]
        a[:, 3] = sc.view(np.int32)
        if aux_labels is None:
            aux = np.where(a[:, 2] >= 0, 0, -1).astype(np.int32)
        else:
            aux = np.asarray([aux_labels[i] for i in order], dtype=np.int32)
        return cls(a, aux, [0, len(arcs)], [num_states])

    @classmethod
    def from_openfst(cls, txt, acceptor=False, sym_in=None, sym_out=None):
        """Same text format as k2.Fsa.from_openfst (datapipe.py:175-180): ``src dst ilabel [olabel]
        [weight]`` arc lines and ``state [weight]`` final lines; score = -weight; every final state
        gets a label -1 arc to a new super-final state (the last state)."""
        arcs, aux, finals = [], [], []
        max_state = -1
        n_lab = 1 if acceptor else 2
        for line in txt.strip().splitlines():
            f = line.split()
            if not f:
                continue
            if len(f) <= 2:
                finals.append((int(f[0]), 0.0 - (float(f[1]) if len(f) == 2 else 0.0)))   # 0.0 - w: never -0.0
                max_state = max(max_state, int(f[0]))
                continue
            s, d = int(f[0]), int(f[1])
            il = sym_in[f[2]] if sym_in is not None else int(f[2])
            if acceptor:
                ol = il
            else:
                ol = sym_out[f[3]] if sym_out is not None else int(f[3])
            w = float(f[2 + n_lab]) if len(f) > 2 + n_lab else 0.0
            arcs.append((s, d, il, 0.0 - w))
            aux.append(ol)
            max_state = max(max_state, s, d)
        sf = max_state + 1
        for s, w in finals:
            arcs.append((s, sf, -1, w))
            aux.append(-1)
        return cls.from_arcs(arcs, sf + 1, aux)

    # ---- views ------------------------------------------------------------------------------
    @property
    def num_fsas(self):
        if self._pending is not None:
            return self._pending.B
        return len(self.fsa_num_states)

    @property
    def shape(self):
        n = self.num_fsas if self.utt2fsa is None else len(self.utt2fsa)
        return (n, None, None)

    @property
    def labels(self):
        return self.arcs[:, 2]

    @property
    def scores(self):
        return self.arcs[:, 3].view(np.float32)

    @property
    def num_arcs(self):
        return self.arcs.shape[0]

    def __getitem__(self, i):
        g = i if self.utt2fsa is None else int(self.utt2fsa[i])
        a0, a1 = self.fsa_arc_offsets[g], self.fsa_arc_offsets[g + 1]
        aux = None if self.aux_labels is None else self.aux_labels[a0:a1]
        return Fsa(self.arcs[a0:a1], aux, [0, a1 - a0], [self.fsa_num_states[g]])

    def as_oracle_graphs(self):
        """List of per-utterance dict graphs in the layout oracle/dwfst_oracle.py uses (tests only)."""
        out = []
        n = self.shape[0]
        for i in range(n):
            g = i if self.utt2fsa is None else int(self.utt2fsa[i])
            a0, a1 = self.fsa_arc_offsets[g], self.fsa_arc_offsets[g + 1]
            a = self.arcs[a0:a1]
            out.append({
                "src": a[:, 0].copy(), "dst": a[:, 1].copy(), "label": a[:, 2].copy(),
                "score": a[:, 3].copy().view(np.float32),
                "aux": (self.aux_labels[a0:a1].copy() if self.aux_labels is not None
                        else np.where(a[:, 2] >= 0, 0, -1).astype(np.int32)),
                "num_states": int(self.fsa_num_states[g]),
            })
        return out

    # ---- device ------------------------------------------------------------------------------
    def _graph_host_struct(self):
        gh = _lib._GraphHost()
        gh.num_fsas = self.num_fsas
        gh.arcs = self.arcs.ctypes.data
        gh.fsa_arc_offsets = self.fsa_arc_offsets.ctypes.data
        gh.fsa_num_states = self.fsa_num_states.ctypes.data
        return gh

    def blob_host(self):
        if self._blob_host is None and self._pending is not None:
            self._pending.finish(self)            # waits for the device compiler's header (or falls back to the host compiler)
        if self._blob_host is None:
            lib = _lib.lib()
            gh = self._graph_host_struct()
            n = C.c_size_t(0)
            _lib.check(lib.wf_graph_prepare_bytes(C.byref(gh), C.byref(n)))      # an upper bound, O(1) per graph
            blob = np.empty(n.value // 4, dtype=np.int32)
            _lib.check(lib.wf_graph_prepare(C.byref(gh), blob.ctypes.data, n.value))
            self._blob_host = blob[:int(blob[2])].copy()                          # WF_H_WORDS: what was used
        return self._blob_host

    def all_compact(self):
        """True when every FSA uses compact label columns (a numerator: few distinct labels out of a large table), i.e.
        the gradient w.r.t. log_probs has only K non-zeros per row."""
        v = getattr(self, "_all_compact", None)
        if v is None:
            from .csrc_layout import HDR_WORDS, INFO_WORDS, INFO
            b = self.blob_host()
            if self.__dict__.get("_all_compact") is not None:      # (the device compiler counted while it built them)
                return self._all_compact
            v = self._all_compact = bool((b[HDR_WORDS + INFO["DIRECT"]:HDR_WORDS + INFO_WORDS * self.num_fsas:INFO_WORDS] == 0).all())
        return v

    def all_direct(self):
        """True when every FSA's label columns are the labels themselves (a topology over the whole table: a denominator),
        i.e. its gradient rows are dense."""
        v = getattr(self, "_all_direct", None)
        if v is None:
            from .csrc_layout import HDR_WORDS, INFO_WORDS, INFO
            b = self.blob_host()
            if self.__dict__.get("_all_direct") is not None:
                return self._all_direct
            v = self._all_direct = bool((b[HDR_WORDS + INFO["DIRECT"]:HDR_WORDS + INFO_WORDS * self.num_fsas:INFO_WORDS] != 0).all())
        return v

    def accumulate_ok(self, seg, num_seqs, V):
        """Will wf_fb_logsum take WF_ACCUMULATE for these graphs (numerators of the probability-domain path)?"""
        try:
            self.workspace_bytes(seg, num_seqs, V, _lib.WF_ACCUMULATE, True)
            return True
        except _lib.WfError:
            return False

    def workspace_bytes(self, seg, num_seqs, V, flags, need_grad):
        """wf_fb_workspace_bytes, cached per (segments, N, V, flags): a training loop asks with the same values twice per
        step (numerator graphs are new objects every step; a denominator topology is asked thousands of times)."""
        key = (seg.tobytes(), num_seqs, V, flags, bool(need_grad), _lib.ENV_EPOCH)
        c = self.__dict__.setdefault("_ws_cache", {})
        n = c.get(key)
        if n is None:
            if len(c) > 64:
                c.clear()
            out = C.c_size_t(0)
            _lib.check(_lib.lib().wf_fb_workspace_bytes(self.blob_host().ctypes.data, None if self.utt2fsa is None else self.utt2fsa.ctypes.data,
                                                         seg.ctypes.data, seg.shape[0], num_seqs, V, flags, 1 if need_grad else 0, C.byref(out)))
            n = c[key] = out.value
        return n

    def blob_dev(self, device):
        device = torch.device(device)
        key = str(device)
        if self._pending is not None:
            self.blob_host()
            if self._pending is not None:
                self._pending.join_stream(device)   # the caller's stream waits for the compile kernels
        if key not in self._blob_dev:
            if self._pending is not None:
                # (a device-compiled vector: its host side holds the header words only, the tables exist on that device)
                raise RuntimeError(f"these graphs were compiled on {self._pending.device}, not {device}: compile them there "
                                   "(compile_ctc_k2_graphs(..., device=...)) or on the host")
            host = self.blob_host()
            if self._blob_pin is not None:
                dev = torch.empty(host.size, dtype=torch.int32, device=device)
                dev.copy_(self._blob_pin[:host.size], non_blocking=True)      # stream-ordered, no host wait
                self._blob_dev[key] = dev
            else:
                self._blob_dev[key] = torch.from_numpy(host).to(device)
        return self._blob_dev[key]

    def to(self, device):
        device = torch.device(device)
        if device.type != "cpu":
            self.blob_dev(device)
        self.device = device
        return self


def create_fsa_vec(fsas):
    """k2.create_fsa_vec. A list that repeats one object (rnnp.py:91-92: ``[topo] * B``) is stored
    once and shared by all utterances instead of being copied B times."""
    fsas = list(fsas)
    assert fsas, "empty list"
    if all(f is fsas[0] for f in fsas) and fsas[0].num_fsas == 1 and fsas[0].utt2fsa is None:
        f = fsas[0]
        v = Fsa(f.arcs, f.aux_labels, f.fsa_arc_offsets, f.fsa_num_states,
                utt2fsa=np.zeros(len(fsas), dtype=np.int32))
        v._blob_host = f._blob_host
        v._blob_dev = f._blob_dev  # share uploads with the topology itself
        v._shared_from = f         # (a device-compiled graph's buffer returns to the compiler's pool when f dies)
        return v
    arcs, aux, off, ns = [], [], [0], []
    for f in fsas:
        assert f.utt2fsa is None
        arcs.append(f.arcs)
        aux.append(f.aux_labels if f.aux_labels is not None else np.where(f.arcs[:, 2] >= 0, 0, -1))
        for g in range(f.num_fsas):
            off.append(off[-1] + int(f.fsa_arc_offsets[g + 1] - f.fsa_arc_offsets[g]))
            ns.append(int(f.fsa_num_states[g]))
    return Fsa(np.concatenate(arcs, 0), np.concatenate(aux, 0), off, ns)


class DenseFsaVec:
    """k2.DenseFsaVec(log_probs, supervision_segments) -- models.py:256-261.

    log_probs: float32 [N, T, V] on the GPU, already log-softmaxed, last dim contiguous.
    supervision_segments: int32 CPU tensor [B, 3], rows (sequence index, start frame, num frames).
    k2 materialises a [sum(T_b + 1), V + 1] copy with an extra column/row for the final symbol;
    here the pair (log_probs, segments) *is* the dense FSA vector.
    """

    def __init__(self, log_probs, supervision_segments):
        assert log_probs.dim() == 3, "log_probs must be [N, T, V]"
        assert log_probs.dtype == torch.float32
        seg = torch.as_tensor(supervision_segments)
        assert seg.dim() == 2 and seg.shape[1] == 3
        assert seg.device.type == "cpu", "supervision_segments must be a CPU tensor (as in k2)"
        assert seg.dtype == torch.int32
        if log_probs.stride(2) != 1:
            log_probs = log_probs.contiguous()
        N, T, V = log_probs.shape
        s = seg.numpy()
        # validated on every construction (three reductions over a [B,3] array): a cache keyed on the tensor's identity
        # would let a new tensor that reuses a freed one's id() through unchecked, and this is the guard in front of the
        # library's device reads (the C entry points repeat it: wf_fb_logsum / wf_viterbi / wf_search / wf_tot_scores)
        if s.size:
            assert int(s.min()) >= 0 and int(s[:, 0].max()) < N and int((s[:, 1].astype(np.int64) + s[:, 2]).max()) <= T, \
                "supervision_segments out of range"
        self.log_probs = log_probs
        self.supervision_segments = seg if seg.is_contiguous() else seg.contiguous()
        self.device = log_probs.device

    @property
    def shape(self):
        return (self.supervision_segments.shape[0], None, self.log_probs.shape[2])

    def to(self, device):
        return DenseFsaVec(self.log_probs.to(device), self.supervision_segments)
