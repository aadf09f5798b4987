"""Drop-in for ``waterfall/graph.py`` of the reference: same names, argument order, defaults and
return conventions (``GraphLoss`` graph.py:9-76, ``graphloss`` graph.py:79-118), computed by the HIP
library instead of ``k2.intersect_dense`` + ``Fsa.get_tot_scores``.

Extra entry points mirroring the other k2 calls on the path (utils/lattice_analysis.py):
``tot_scores`` (get_tot_scores, log or tropical) and ``best_path`` (shortest_path /
one_best_decoding + get_aux_labels).
"""
import ctypes as C
import math
import warnings

import numpy as np
import torch
from torch import nn

from . import _lib
from .fsa import DenseFsaVec, Fsa


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def _on(device):
    """Context that makes `device` current for the library's HIP calls; free when it already is (torch.cuda.device()
    costs ~10 us per entry, a third of the drop-in's host time per call)."""
    return _NULL if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def _check_inputs(graph, dense):
    if not isinstance(graph, Fsa) or not isinstance(dense, DenseFsaVec):
        raise TypeError("expected (waterfall_amd.Fsa, waterfall_amd.DenseFsaVec)")
    lp = dense.log_probs
    if lp.device.type != "cuda":
        raise _lib.WfError("log_probs must live on the GPU: waterfall_amd has no CPU path")
    seg = dense.supervision_segments.numpy()
    B = seg.shape[0]
    if graph.shape[0] != B:
        raise ValueError(f"decoding_graph has {graph.shape[0]} FSAs, dense_fsa_vec has {B} segments")
    return lp, seg, B


class _FbFunction(torch.autograd.Function):
    """tot[b] = log-semiring total score; backward = - arc posteriors folded per label."""

    @staticmethod
    def forward(ctx, log_probs, graph, dense, output_beam, use_double_scores):
        lp, seg, B = _check_inputs(graph, dense)
        lib = _lib.lib()
        dev = lp.device
        N, T, V = lp.shape
        # use_double_scores: fp64 totals, as lattice.get_tot_scores(use_double_scores=True) returns them (graph.py:66)
        flags = (_lib.WF_DOUBLE_ACC | _lib.WF_TOT_F64) if use_double_scores else 0
        need_grad = bool(ctx.needs_input_grad[0])
        if need_grad:
            flags |= _lib.WF_KEEP_AB
        beam = float(output_beam) if output_beam is not None else 0.0
        if beam > 0 and math.isfinite(beam):
            flags |= _lib.WF_PRUNED
        gh = graph.blob_host()
        gd = graph.blob_dev(dev)
        n = C.c_size_t(0)
        _lib.check(lib.wf_fb_workspace_bytes(_np_ptr(gh), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, N, V, flags,
                                              1 if need_grad else 0, C.byref(n)))
        ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
        tot = torch.empty(B, dtype=torch.float64 if use_double_scores else torch.float32, device=dev)
        grad = None
        if need_grad:
            # the gradient buffer is zero-filled beside the sweeps of this call (WF_PREZERO); backward only scatters
            grad = torch.empty_strided(lp.shape, lp.stride(), dtype=torch.float32, device=dev)
            flags |= _lib.WF_PREZERO
        with _on(dev):
            _lib.check(lib.wf_fb_logsum(_np_ptr(gh), _ptr(gd), _np_ptr(graph.utt2fsa), _ptr(lp), lp.stride(0),
                                        lp.stride(1), N, T, V, _np_ptr(seg), B, beam, flags, None, _ptr(tot),
                                        _ptr(grad) if grad is not None else None, _ptr(ws), n.value, _stream(dev)))
        if need_grad:
            ctx.graph, ctx.dense, ctx.flags, ctx.ws, ctx.ws_bytes = graph, dense, flags, ws, n.value
            ctx.grad = grad
            ctx.save_for_backward(lp, tot)
        return tot

    @staticmethod
    def backward(ctx, gtot):
        lp, tot = ctx.saved_tensors
        graph, dense = ctx.graph, ctx.dense
        lib = _lib.lib()
        dev = lp.device
        N, T, V = lp.shape
        seg = dense.supervision_segments.numpy()
        B = seg.shape[0]
        gs = gtot.to(torch.float32).contiguous()
        flags = ctx.flags
        grad, ctx.grad = ctx.grad, None
        if grad is None:   # a second backward through the same graph (retain_graph): fresh buffer, zero-filled by the call
            grad = torch.empty_strided(lp.shape, lp.stride(), dtype=torch.float32, device=dev)
            flags &= ~_lib.WF_PREZERO
        with _on(dev):
            _lib.check(lib.wf_fb_backward(_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)),
                                          _np_ptr(graph.utt2fsa), _ptr(lp), lp.stride(0), lp.stride(1), N, T, V,
                                          _np_ptr(seg), B, flags, _ptr(gs), _ptr(tot), _ptr(grad),
                                          _ptr(ctx.ws), ctx.ws_bytes, _stream(dev)))
        return grad, None, None, None, None


class _EagerLossFunction(torch.autograd.Function):
    """The whole of graph.py:61-76 as ONE autograd node: loss = reduction(-tot), computed by the FUSED wf_fb_logsum call
    (loss and d loss / d log_probs in one pass: no second C call, no separate row check), with the reduction's own factor
    -w_b folded into the gradient.  backward only multiplies the kept gradient by the upstream gradient
    (wf_grad_scale: nothing to do when that is 1, the usual loss.backward()).  Used for graphs with compact label
    columns (numerators: the gradient is K non-zeros per row); the others take _FbFunction."""

    @staticmethod
    def forward(ctx, log_probs, graph, dense, reduction, weights, beam=0.0):
        lp, seg, B = _check_inputs(graph, dense)
        lib = _lib.lib()
        dev = lp.device
        N, T, V = lp.shape
        gh, gd = graph.blob_host(), graph.blob_dev(dev)
        # output_beam (the "pruned" mode): the same fused call with WF_PRUNED -- tropical pass, keep masks, masked sweeps
        fl = _lib.WF_PRUNED if beam > 0 and math.isfinite(beam) else 0
        beam = float(beam) if fl else 0.0
        nbytes = graph.workspace_bytes(seg, N, V, fl, True)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        tot = torch.empty(B, dtype=torch.float32, device=dev)
        grad = torch.empty_like(lp) if lp.is_contiguous() else torch.empty_strided(lp.shape, lp.stride(), dtype=torch.float32, device=dev)
        # The kept gradient is d loss / d log_probs times the upstream factor this node EXPECTS: +1, but -1 for graphs
        # with dense gradient rows (a denominator: `numerator - denominator`, rnnp.py:100) -- backward's scaling pass over
        # the box (0.26 ms at cfg4 sizes) is then skipped in the usual case instead of paid in it.
        expect = -1.0 if graph.all_direct() else 1.0
        if weights is not None:
            gscale = weights if expect == 1.0 else weights.neg()               # d loss / d tot_b (x expect)
        else:
            gscale = _const(dev, B, -expect)
        st = _stream(dev)
        with _on(dev):
            _lib.check(lib.wf_fb_logsum(_np_ptr(gh), _ptr(gd), _np_ptr(graph.utt2fsa), _ptr(lp), lp.stride(0), lp.stride(1),
                                        N, T, V, _np_ptr(seg), B, beam, fl, _ptr(gscale), _ptr(tot), _ptr(grad), _ptr(ws),
                                        nbytes, st))
            if reduction == 'none':
                loss = tot.neg()
            else:
                loss = torch.empty((), dtype=torch.float32, device=dev)
                # 'mean': weights = -1 / (B * target_lengths) -> w = -weights
                w = None if weights is None else weights.neg()
                _lib.check(lib.wf_loss_reduce(_ptr(tot), _ptr(w), B, _ptr(loss), st))
        # everything backward needs is marshalled here, on the caller's thread: Python on the autograd engine's thread
        # runs several times slower (measured: 42 -> 16 us per backward)
        ctx.keep = (graph, dense, ws, gscale, seg, gd)
        ctx.beam = (beam, fl)
        ctx.expect = expect
        ctx.grad = grad
        ctx.call = [_np_ptr(gh), _ptr(gd), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, N, T, V, lp.stride(0), lp.stride(1), None,
                    1 if reduction == 'none' else 0, _ptr(grad), _ptr(ws), nbytes, st]
        ctx.save_for_backward(lp)
        return loss

    @staticmethod
    def backward(ctx, gout):
        grad, ctx.grad = ctx.grad, None
        call = ctx.call
        gs = gout if (gout.dtype == torch.float32 and gout.is_contiguous()) else gout.to(torch.float32).contiguous()
        if ctx.expect != 1.0:
            gs = gs.neg()   # relative to the factor the kept gradient already carries
        # (the engine runs this node with the forward's device and stream current)
        if grad is None:   # a second backward through a retained graph: the kept gradient was handed out, compute it again
            graph, dense, ws, gscale, seg, gd = ctx.keep
            (lp,) = ctx.saved_tensors
            grad = torch.empty_strided(lp.shape, lp.stride(), dtype=torch.float32, device=lp.device)
            tot = torch.empty(call[4], dtype=torch.float32, device=lp.device)
            with _on(lp.device):
                _lib.check(_lib.lib().wf_fb_logsum(call[0], call[1], call[2], _ptr(lp), call[8], call[9], call[5], call[6], call[7],
                                                   call[3], call[4], ctx.beam[0], ctx.beam[1], _ptr(gscale), _ptr(tot), _ptr(grad),
                                                   call[13], call[14], _stream(lp.device)))
            call = list(call)
            call[12] = _ptr(grad)
            call[15] = _stream(lp.device)
        call[10] = C.c_void_p(gs.data_ptr())
        _lib.check(_lib.lib().wf_grad_scale(*call))
        return grad, None, None, None, None, None


_NUMDEN_OVERLAP = False
_SIDE = {}


def set_num_den_overlap(on: bool):
    """graphloss_num_den: run the numerator's sweeps on a side stream beside the denominator's (they join before the
    numerator's posteriors are added).  Off by default: 1.81 -> 1.75 ms per step at cfg4 sizes is little for a second
    stream inside an autograd node (not capturable in a HIP graph as is)."""
    global _NUMDEN_OVERLAP
    _NUMDEN_OVERLAP = bool(on)


def _side_stream(dev):
    s = _SIDE.get(dev.index)
    if s is None:
        s = _SIDE[dev.index] = torch.cuda.Stream(dev)
    return s


class _NumDenFunction(torch.autograd.Function):
    """graphloss(numerator) - graphloss(denominator) on one DenseFsaVec (the 'k2' loss, rnnp.py:64-104) as ONE autograd
    node with ONE gradient buffer: the denominator's fused call writes every entry of the box, the numerator's adds its K
    entries per row (WF_ACCUMULATE: no zero-fill, no second box, no addition pass -- under autograd the two-call form
    pays the numerator's fill and the sum of two boxes, 0.4 ms of 2.2 at cfg4 sizes)."""

    @staticmethod
    def forward(ctx, log_probs, num, den, dense, reduction, weights):
        lp, seg, B = _check_inputs(num, dense)
        _check_inputs(den, dense)
        lib = _lib.lib()
        dev = lp.device
        N, T, V = lp.shape
        # (two workspaces: backward scales the box through the denominator's tables -- dense rows -- which the numerator's
        # call must not overwrite)
        nb_num = num.workspace_bytes(seg, N, V, _lib.WF_KEEP_AB if _NUMDEN_OVERLAP else 0, True)
        nb_den = den.workspace_bytes(seg, N, V, 0, True)
        ws_num = torch.empty(nb_num, dtype=torch.uint8, device=dev)
        ws_den = torch.empty(nb_den, dtype=torch.uint8, device=dev)
        tots = torch.empty(2, B, dtype=torch.float32, device=dev)
        grad = torch.empty_like(lp) if lp.is_contiguous() else torch.empty_strided(lp.shape, lp.stride(), dtype=torch.float32, device=dev)
        w = weights if weights is not None else _const(dev, B, 1.0)       # loss = sum_b w_b (tot_den_b - tot_num_b)
        wneg = w.neg() if weights is not None else _const(dev, B, -1.0)
        st = _stream(dev)

        def call(g, gscale, tot, flags, gr, ws, nbytes, stream):
            _lib.check(lib.wf_fb_logsum(_np_ptr(g.blob_host()), _ptr(g.blob_dev(dev)), _np_ptr(g.utt2fsa), _ptr(lp),
                                        lp.stride(0), lp.stride(1), N, T, V, _np_ptr(seg), B, 0.0, flags, _ptr(gscale),
                                        _ptr(tot), _ptr(gr) if gr is not None else None, _ptr(ws), nbytes, stream))
        with _on(dev):
            if _NUMDEN_OVERLAP:
                # the numerator's sweeps (2B small workgroups) on a side stream beside the denominator's (2B workgroups
                # that own a CU each): they meet again before the numerator's posteriors are added to the box
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                side.wait_stream(main)
                for t in (lp, ws_num, tots):
                    t.record_stream(side)
                call(num, wneg, tots[0], _lib.WF_KEEP_AB, None, ws_num, nb_num, C.c_void_p(side.cuda_stream))
                call(den, w, tots[1], 0, grad, ws_den, nb_den, st)
                main.wait_stream(side)
                _lib.check(lib.wf_fb_backward(_np_ptr(num.blob_host()), _ptr(num.blob_dev(dev)), _np_ptr(num.utt2fsa), _ptr(lp),
                                              lp.stride(0), lp.stride(1), N, T, V, _np_ptr(seg), B,
                                              _lib.WF_KEEP_AB | _lib.WF_ACCUMULATE, _ptr(wneg), _ptr(tots[0]), _ptr(grad),
                                              _ptr(ws_num), nb_num, st))
            else:
                call(den, w, tots[1], 0, grad, ws_den, nb_den, st)
                call(num, wneg, tots[0], _lib.WF_ACCUMULATE, grad, ws_num, nb_num, st)
        per_utt = (tots[1] - tots[0]) * w if weights is not None else tots[1] - tots[0]
        loss = per_utt if reduction == 'none' else per_utt.sum()
        ctx.grad = grad
        ctx.call = [_np_ptr(den.blob_host()), _ptr(den.blob_dev(dev)), _np_ptr(den.utt2fsa), _np_ptr(seg), B, N, T, V, lp.stride(0),
                    lp.stride(1), None, 1 if reduction == 'none' else 0, _ptr(grad), _ptr(ws_den), nb_den, st]
        ctx.keep = (num, den, dense, ws_den, seg)
        return loss

    @staticmethod
    def backward(ctx, gout):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            raise RuntimeError("graphloss_num_den: one backward per forward (the gradient was handed out); call it again "
                               "or use graphloss(num) - graphloss(den) with set_eager_gradient(False)")
        gs = gout if (gout.dtype == torch.float32 and gout.is_contiguous()) else gout.to(torch.float32).contiguous()
        call = ctx.call
        call[10] = C.c_void_p(gs.data_ptr())
        _lib.check(_lib.lib().wf_grad_scale(*call))
        return grad, None, None, None, None, None


def graphloss_num_den(num_graph: Fsa, den_graph: Fsa, dense_fsa_vec: DenseFsaVec, reduction: str = 'sum',
                      target_lengths: torch.Tensor = None, output_beam: float = 10) -> torch.Tensor:
    """``graphloss(num_graph, d, ...) - graphloss(den_graph, d, ...)`` -- the numerator / denominator loss of the
    reference's k2 branch (rnnp.py:64-104, conformer.py:405-430; same reduction and target_lengths in both calls there)
    -- with both gradients written into one buffer by the library.  Same values as the two-call form; falls back to it
    when the numerator graphs are beyond WF_ACCUMULATE's reach, under the "pruned" beam mode, or without a gradient."""
    assert reduction in ('none', 'mean', 'sum')
    lp = dense_fsa_vec.log_probs
    fused = (_beam_value(output_beam, den_graph, dense_fsa_vec) == 0.0 and _beam_value(output_beam, num_graph, dense_fsa_vec) == 0.0
             and lp.requires_grad and torch.is_grad_enabled() and lp.device.type == "cuda"
             and isinstance(num_graph, Fsa) and isinstance(den_graph, Fsa) and den_graph.all_direct()
             and num_graph.accumulate_ok(dense_fsa_vec.supervision_segments.numpy(), lp.shape[0], lp.shape[2]))
    if not fused:
        return (_graphloss(num_graph, dense_fsa_vec, output_beam, reduction, False, target_lengths)
                - _graphloss(den_graph, dense_fsa_vec, output_beam, reduction, False, target_lengths))
    weights = None
    if reduction == 'mean':
        B = dense_fsa_vec.supervision_segments.shape[0]
        weights = (1.0 / B) / target_lengths.to(device=lp.device, dtype=torch.float32)
    return _NumDenFunction.apply(lp, num_graph, den_graph, dense_fsa_vec, reduction, weights)


_CONST = {}


def _const(dev, n, value):
    key = (str(dev), n, value)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((n,), value, dtype=torch.float32, device=dev)
    return t


_EAGER = True


def set_eager_gradient(on: bool):
    """True (default): graphloss() on graphs with compact label columns computes loss AND gradient in its forward (the
    fused call) and keeps the gradient for backward -- one backward per forward.  False: the two-call split
    (forward: sweeps + row check; backward: posteriors), which supports retain_graph / double backward passes."""
    global _EAGER
    _EAGER = bool(on)


DEFAULT_BEAM_MODE = "auto"   # output_beam honoured (the reference's k2.intersect_dense(graph, dense, output_beam)) wherever
_BEAM_MODE = DEFAULT_BEAM_MODE  # the pruned lattice has a fast path; see set_output_beam_mode
_AUTO_WARNED = [False]


def set_output_beam_mode(mode: str):
    """How `output_beam` (graph.py:61-62 passes it to k2.intersect_dense) is honoured.

    "pruned": drop lattice arcs whose best path is more than output_beam below the best path (a
    tropical forward-backward pass, keep masks, then the masked sweeps and posteriors; on banded graphs -- every
    numerator, the small denominators -- all of it stays on the probability-domain path: 2.0x the exact step at
    cfg3 / cfg2 sizes).  This is the rule the reference's k2.intersect_dense call states (graph.py:61-62, default
    beam 10 at :81); k2 is not available to pin its arithmetic against ("parity unpinned").
    output_beam=None or inf: no pruning.  Graphs without that fast path (hub-form denominators: cfg4; graphs beyond
    the LDS sweeps; use_double_scores) take masked log-semiring sweeps: 33 ms instead of 2 for cfg4's
    numerator - denominator step.
    "auto" (the default): "pruned" for the graph vectors that have the fast path (wf_fb_pruned_is_fast), "exact" for
    the others, with one warning -- the reference's semantics wherever they do not cost an order of magnitude.
    "exact": sum over ALL lattice arcs whatever the beam -- the limit output_beam -> inf, half the cost.  Each
    pruned arc carries < e^-beam of the best path's mass, but an untrained model spreads such arcs over most of
    the lattice: against the oracle's pruned lattice the two modes agree inside the tolerances (1e-4 relative on
    the loss, 1e-3 absolute on the gradient) for beam 20 and for peaked posteriors, and differ by up to 5e-2 in
    single gradient entries for beam 10 on a noise posterior (0.19 at cfg3 size, 0.49 at cfg2 size:
    tests/test_gpu_parity.py::test_exact_mode_is_within_tolerance_of_the_reference_beams).
    """
    global _BEAM_MODE
    assert mode in ("auto", "exact", "pruned")
    _BEAM_MODE = mode


def _beam_value(output_beam, graph=None, dense=None, use_double_scores=False):
    """The beam wf_fb_logsum gets (0: no pruning) for this call under the current mode."""
    if _BEAM_MODE == "exact" or output_beam is None or not (output_beam > 0) or math.isinf(output_beam):
        return 0.0
    if _BEAM_MODE == "auto":
        if graph is None or not _pruned_is_fast(graph, dense, use_double_scores):
            if graph is not None and not _AUTO_WARNED[0]:
                _AUTO_WARNED[0] = True
                warnings.warn("waterfall_amd.graphloss: output_beam is not applied to graphs without a fast pruned path "
                              "(hub-form denominators, graphs beyond the LDS sweeps, use_double_scores): they get the sum "
                              "over all lattice arcs.  set_output_beam_mode('pruned') applies the beam to every graph "
                              "(masked log-semiring sweeps, ~15x slower on cfg4's denominators); 'exact' never prunes.")
            return 0.0
    return float(output_beam)


def _pruned_is_fast(graph, dense, use_double_scores):
    """wf_fb_pruned_is_fast for these graphs, cached on the graph object (the answer depends on the graphs' shapes)."""
    key = bool(use_double_scores)
    c = graph.__dict__.setdefault("_pruned_fast_cache", {})
    v = c.get(key)
    if v is None:
        seg = dense.supervision_segments.numpy()
        lp = dense.log_probs
        out = C.c_int32(0)
        _lib.check(_lib.lib().wf_fb_pruned_is_fast(_np_ptr(graph.blob_host()), _np_ptr(graph.utt2fsa), _np_ptr(seg), seg.shape[0],
                                                   lp.shape[0], lp.shape[2], _lib.WF_DOUBLE_ACC if use_double_scores else 0,
                                                   C.byref(out)))
        v = c[key] = bool(out.value)
    return v


class GraphLoss(nn.Module):
    """Same constructor and forward signature as the reference (graph.py:13-43)."""

    def __init__(self, output_beam: float, reduction: str = 'sum', use_double_scores: bool = False):
        super().__init__()
        assert reduction in ('none', 'mean', 'sum')
        self.output_beam = output_beam
        self.reduction = reduction
        self.use_double_scores = use_double_scores

    def forward(self, decoding_graph: Fsa, dense_fsa_vec: DenseFsaVec, target_lengths: torch.Tensor = None,
                max_states: int = 15000000, max_arcs: int = 549755813888) -> torch.Tensor:
        # max_states / max_arcs are accepted and ignored, exactly as in the reference (graph.py:42-43,61-62)
        return _graphloss(decoding_graph, dense_fsa_vec, self.output_beam, self.reduction, self.use_double_scores,
                          target_lengths)


_EAGER_DIRECT = True   # denominators (direct label columns) take the fused call too: no separate row-check pass


def _graphloss(decoding_graph, dense_fsa_vec, output_beam, reduction, use_double_scores, target_lengths):
    """graph.py:61-76.  One library call per batch; a batch in which a sequence is named by several supervision segments
    (k2.DenseFsaVec allows it; models.py:256-261 never builds one) is refused by that call (WF_EUNSUPPORTED: its gradient
    kernels walk one segment per sequence) and is then run in rounds, see _graphloss_rounds."""
    try:
        return _graphloss_once(decoding_graph, dense_fsa_vec, output_beam, reduction, use_double_scores, target_lengths)
    except _lib.WfError as e:
        if "more than one segment" not in str(e) or not isinstance(decoding_graph, Fsa):
            raise
    return _graphloss_rounds(decoding_graph, dense_fsa_vec, output_beam, reduction, use_double_scores, target_lengths)


def _select(graph, idx):
    """The FSAs of utterances `idx` as a vector over the SAME prepared tables (nothing is copied or prepared again)."""
    graph.blob_host()
    u = idx.astype(np.int32) if graph.utt2fsa is None else np.asarray(graph.utt2fsa)[idx].astype(np.int32)
    v = Fsa(graph.arcs, graph.aux_labels, graph.fsa_arc_offsets, graph.fsa_num_states, utt2fsa=np.ascontiguousarray(u))
    v._blob_host = graph._blob_host
    v._blob_dev = graph._blob_dev
    v._shared_from = graph
    return v


def _in_rounds(graph, dense, fn):
    """fn(graphs, dense) -> [B'] tensor, applied to round r = the r-th segment of every sequence; results in batch order."""
    seg = dense.supervision_segments.numpy()
    B = seg.shape[0]
    seen, rnd = {}, np.empty(B, dtype=np.int64)
    for i, q in enumerate(seg[:, 0].tolist()):
        rnd[i] = seen.get(q, 0)
        seen[q] = rnd[i] + 1
    parts, order = [], []
    for r in range(int(rnd.max()) + 1):
        idx = np.nonzero(rnd == r)[0]
        sub = DenseFsaVec(dense.log_probs, dense.supervision_segments[torch.from_numpy(idx)])
        parts.append(fn(_select(graph, idx), sub))
        order.append(idx)
    inv = np.argsort(np.concatenate(order), kind="stable")
    return torch.cat(parts)[torch.from_numpy(inv).to(parts[0].device)]


def _graphloss_rounds(graph, dense, output_beam, reduction, use_double_scores, target_lengths):
    """Several segments per sequence: round r holds the r-th segment of every sequence (one segment per sequence within a
    round: the ordinary call), the per-utterance losses are put back in the batch's order and autograd adds the rounds'
    gradient boxes.  One library call per round."""
    loss = _in_rounds(graph, dense, lambda g, d: _graphloss_once(g, d, output_beam, 'none', use_double_scores, None))
    if reduction == 'none':
        return loss
    if reduction == 'sum':
        return loss.sum()
    return (loss / target_lengths.to(loss.device)).mean()


def _graphloss_once(decoding_graph, dense_fsa_vec, output_beam, reduction, use_double_scores, target_lengths):
    """graph.py:61-76: loss = -tot_scores (float32), reduced 'none' | 'sum' | 'mean' (/ target_lengths, then batch mean)."""
    beam = _beam_value(output_beam, decoding_graph if isinstance(decoding_graph, Fsa) else None, dense_fsa_vec, use_double_scores)
    lp = dense_fsa_vec.log_probs
    if (_EAGER and not use_double_scores and lp.requires_grad and torch.is_grad_enabled()
            and isinstance(decoding_graph, Fsa) and (decoding_graph.all_compact() or _EAGER_DIRECT)):
        weights = None
        if reduction == 'mean':
            B = dense_fsa_vec.supervision_segments.shape[0]
            weights = (-1.0 / B) / target_lengths.to(device=lp.device, dtype=torch.float32)
        return _EagerLossFunction.apply(lp, decoding_graph, dense_fsa_vec, reduction, weights, beam)
    tot_scores = _FbFunction.apply(lp, decoding_graph, dense_fsa_vec, beam, use_double_scores)
    loss = -1 * tot_scores
    loss = loss.to(torch.float32)
    if reduction == 'none':
        return loss
    elif reduction == 'sum':
        return loss.sum()
    else:
        assert reduction == 'mean'
        loss = loss / target_lengths.to(loss.device)
        return loss.mean()


def graphloss(decoding_graph: Fsa, dense_fsa_vec: DenseFsaVec, output_beam: float = 10, reduction: str = 'sum',
              use_double_scores: bool = False, max_states: int = 15000000,
              target_lengths: torch.Tensor = None) -> torch.Tensor:
    """graph.py:79-118."""
    assert reduction in ('none', 'mean', 'sum')
    return _graphloss(decoding_graph, dense_fsa_vec, output_beam, reduction, use_double_scores, target_lengths)


def tot_scores(graph: Fsa, dense: DenseFsaVec, log_semiring: bool = True, use_double_scores: bool = False):
    """lattice.get_tot_scores(log_semiring=, use_double_scores=) over the exact intersection
    (utils/lattice_analysis.py:148-149).  No gradient.  Returns float32 [B] (float64 with double)."""
    try:
        return _tot_scores_once(graph, dense, log_semiring, use_double_scores)
    except _lib.WfError as e:
        if "more than one segment" not in str(e):
            raise
    return _in_rounds(graph, dense, lambda g, d: _tot_scores_once(g, d, log_semiring, use_double_scores))


def _tot_scores_once(graph, dense, log_semiring, use_double_scores):
    lp, seg, B = _check_inputs(graph, dense)
    lib = _lib.lib()
    dev = lp.device
    N, T, V = lp.shape
    flags = (_lib.WF_DOUBLE_ACC if use_double_scores else 0) | (0 if log_semiring else _lib.WF_TROPICAL)
    n = C.c_size_t(0)
    _lib.check(lib.wf_fb_workspace_bytes(_np_ptr(graph.blob_host()), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, N, V,
                                          flags, 0, C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    tot = torch.empty(B, dtype=torch.float64 if use_double_scores else torch.float32, device=dev)
    with _on(dev):
        _lib.check(lib.wf_tot_scores(_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)), _np_ptr(graph.utt2fsa),
                                     _ptr(lp), lp.stride(0), lp.stride(1), N, T, V, _np_ptr(seg), B, flags, _ptr(tot),
                                     _ptr(ws), n.value, _stream(dev)))
    return tot


def best_path(graph: Fsa, dense: DenseFsaVec, use_double_scores: bool = False):
    """k2.shortest_path / one_best_decoding (+ get_aux_labels): tropical best path per utterance.

    Returns dict(scores float32 [B] (device), arcs int32 [B, T+1] (device, -1 padded; indices into
    graph.arcs), labels list[list[int]] (emitting labels), aux_labels list[list[int]] (non-zero,
    non-final aux labels)).  Ties are broken by the lowest arc index at every back-trace step.
    """
    lp, seg, B = _check_inputs(graph, dense)
    lib = _lib.lib()
    dev = lp.device
    N, T, V = lp.shape
    flags = _lib.WF_DOUBLE_ACC if use_double_scores else 0
    n = C.c_size_t(0)
    _lib.check(lib.wf_viterbi_workspace_bytes(_np_ptr(graph.blob_host()), _np_ptr(graph.utt2fsa), _np_ptr(seg), B,
                                               V, flags, C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    scores = torch.empty(B, dtype=torch.float32, device=dev)
    arcs = torch.empty((B, T + 1), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(lib.wf_viterbi(_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)), _np_ptr(graph.utt2fsa),
                                  _ptr(lp), lp.stride(0), lp.stride(1), N, V, _np_ptr(seg), B, T, flags,
                                  _ptr(scores), _ptr(arcs), _ptr(ws), n.value, _stream(dev)))
    labels, aux = _path_labels(graph, arcs)
    return {"scores": scores, "arcs": arcs, "labels": labels, "aux_labels": aux}


def _path_labels(graph, arcs):
    a = arcs.cpu().numpy()
    labels, aux = [], []
    for i in range(a.shape[0]):
        ids = a[i][a[i] >= 0]
        lab = graph.arcs[ids, 2]
        labels.append([int(x) for x in lab if x >= 0])
        if graph.aux_labels is not None:
            ax = graph.aux_labels[ids]
            aux.append([int(x) for x in ax if x > 0])
        else:
            aux.append([])
    return labels, aux

def pruned_search(graph: Fsa, dense: DenseFsaVec, search_beam: float = 20.0, min_active_states: int = 30,
                  max_active_states: int = 10000, log_tot: bool = True, use_double_scores: bool = False,
                  return_active: bool = False, output_beam=None):
    """Beam-pruned frame-synchronous search (k2.get_lattice's search, utils/lattice_analysis.py:125; wf_search).

    Returns dict(scores, arcs, labels, aux_labels) like best_path, plus ``tot_log`` (log-semiring total of the pruned
    lattice; fp64 with use_double_scores) when log_tot, and ``n_active`` int32 [B, T+1] when return_active.
    search_beam = inf and max_active_states = 0 give the exact search.

    ``output_beam`` (get_lattice's, a float; None: not applied): after the search, one backward launch over the lattice it
    left keeps the arcs within output_beam of the best path (wf_search_lattice); ``tot_log`` is then the total of that
    smaller lattice and ``lattice_arcs`` (int64 [B]) its number of arcs.  The best path does not depend on it.
    """
    lp, seg, B = _check_inputs(graph, dense)
    lib = _lib.lib()
    dev = lp.device
    N, T, V = lp.shape
    flags = _lib.WF_DOUBLE_ACC if use_double_scores else 0
    lattice = output_beam is not None
    if lattice and not log_tot:
        raise ValueError("pruned_search: output_beam only changes tot_log; pass log_tot=True")
    n = C.c_size_t(0)
    if lattice:
        _lib.check(lib.wf_search_lattice_workspace_bytes(_np_ptr(graph.blob_host()), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, V,
                                                          int(max_active_states), C.byref(n)))
    else:
        _lib.check(lib.wf_search_workspace_bytes(_np_ptr(graph.blob_host()), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, V,
                                                  int(max_active_states), int(bool(log_tot)), C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    scores = torch.empty(B, dtype=torch.float32, device=dev)
    arcs = torch.empty((B, T + 1), dtype=torch.int32, device=dev)
    tot = torch.empty(B, dtype=torch.float64 if use_double_scores else torch.float32, device=dev) if log_tot else None
    nact = torch.empty((B, T + 1), dtype=torch.int32, device=dev) if return_active else None
    narcs = torch.empty(B, dtype=torch.int64, device=dev) if lattice else None
    with _on(dev):
        if lattice:
            _lib.check(lib.wf_search_lattice(_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)), _np_ptr(graph.utt2fsa),
                                             _ptr(lp), lp.stride(0), lp.stride(1), N, V, _np_ptr(seg), B, T,
                                             float(search_beam), float(output_beam), int(min_active_states),
                                             int(max_active_states), flags, _ptr(scores), _ptr(arcs), _ptr(tot),
                                             _ptr(narcs), _ptr(nact), _ptr(ws), n.value, _stream(dev)))
        else:
            _lib.check(lib.wf_search(_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)), _np_ptr(graph.utt2fsa),
                                     _ptr(lp), lp.stride(0), lp.stride(1), N, V, _np_ptr(seg), B, T, float(search_beam),
                                     int(min_active_states), int(max_active_states), flags, _ptr(scores), _ptr(arcs),
                                     _ptr(tot), _ptr(nact), _ptr(ws), n.value, _stream(dev)))
    res = {"scores": scores, "arcs": arcs}
    res["labels"], res["aux_labels"] = _path_labels(graph, arcs)
    if log_tot:
        res["tot_log"] = tot
    if return_active:
        res["n_active"] = nact
    if lattice:
        res["lattice_arcs"] = narcs
    return res



def lattice_words(graph: Fsa, dense: DenseFsaVec, words, search_beam: float = 20.0, output_beam: float = 8.0,
                  min_active_states: int = 30, max_active_states: int = 10000, use_double_scores: bool = True):
    """The search's output lattice restricted to a word sequence per utterance (wf_search_lattice_words):
    `latticeW = k2.connect(k2.compose(lattice, W))`, W = k2.linear_fst(words, words), utils/lattice_analysis.py:137-139.

    words: one list of word ids (> 0, the graph's aux labels) per utterance.  Returns dict(scores, arcs: the lattice's best
    path as pruned_search; tot_log: the lattice's total; tot_words / best_words: log-semiring and tropical totals over the
    lattice's paths that spell the words, -inf when it holds none).  output_beam None or inf: no backward pruning."""
    lp, seg, B = _check_inputs(graph, dense)
    if graph.aux_labels is None:
        raise ValueError("lattice_words: the graph has no aux labels (word ids)")
    if len(words) != B:
        raise ValueError(f"lattice_words: {len(words)} word sequences for {B} segments")
    lib = _lib.lib()
    dev = lp.device
    N, T, V = lp.shape
    flags = _lib.WF_DOUBLE_ACC if use_double_scores else 0
    ob = float("inf") if output_beam is None else float(output_beam)
    woff = np.zeros(B + 1, dtype=np.int32)
    woff[1:] = np.cumsum([len(w) for w in words])
    wflat = np.asarray([int(x) for w in words for x in w] or [0], dtype=np.int32)
    aux = graph.__dict__.get("_aux_dev", {}).get(str(dev))
    if aux is None:
        aux = torch.from_numpy(np.ascontiguousarray(graph.aux_labels, dtype=np.int32)).to(dev)
        graph.__dict__.setdefault("_aux_dev", {})[str(dev)] = aux
    n = C.c_size_t(0)
    _lib.check(lib.wf_search_lattice_words_workspace_bytes(_np_ptr(graph.blob_host()), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, V,
                                                            int(max_active_states), _np_ptr(wflat), _np_ptr(woff), C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    ft = torch.float64 if use_double_scores else torch.float32
    scores = torch.empty(B, dtype=torch.float32, device=dev)
    arcs = torch.empty((B, T + 1), dtype=torch.int32, device=dev)
    tot = torch.empty(B, dtype=ft, device=dev)
    totw = torch.empty(B, dtype=ft, device=dev)
    bestw = torch.empty(B, dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(lib.wf_search_lattice_words(_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)), _np_ptr(graph.utt2fsa),
                                               _ptr(lp), lp.stride(0), lp.stride(1), N, V, _np_ptr(seg), B, T,
                                               float(search_beam), ob, int(min_active_states), int(max_active_states), flags,
                                               _ptr(aux), _np_ptr(wflat), _np_ptr(woff), _ptr(scores), _ptr(arcs), _ptr(tot),
                                               _ptr(totw), _ptr(bestw), _ptr(ws), n.value, _stream(dev)))
    return {"scores": scores, "arcs": arcs, "tot_log": tot, "tot_words": totw, "best_words": bestw}
