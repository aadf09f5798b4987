"""Drop-in for the reference's ``waterfall/manual_ctc`` package (ctc.py:94-105 ``ctc_loss``,
softmax_ctc.py:97-110 ``ctc_softmax``, prior.py:4-55 ``compute_align``): same names, argument order,
reductions and gradient, computed by the HIP forward-backward kernels instead of the dense
``[B,T,S,V]`` / ``[B,S,S]`` log-sum-exp broadcasts (41 GB at B=32, T=500, S=129, V=5001).

The per-utterance state machine ``(trans [B,S,S], transform [B,S,V], trans_lengths [B])`` is turned into
a graph once per call (ctc.py:21-45 semantics): all mass starts in state 0 *before* frame 0; frame t
moves j -> i with weight ``trans[b,i,j]`` and emits the token(s) ``transform[b,i,:]`` allows for the
NEW state i; after the last frame one more, non-emitting transition must reach state
``trans_lengths[b]-1``.  Hence: graph states = machine states, an arc j -> i per allowed (i, j, token)
with label = token and score = trans + transform, and a final arc from every j that may enter the end
state, with score ``trans[b,end,j]``.  Entries at or below ``finfo(float32).min / 2`` are "not
allowed" (deprecated/datapipe_manual_ctc.py:257-258 pads with finfo.min).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .fsa import DenseFsaVec, Fsa, create_fsa_vec
from .graph import _FbFunction, _np_ptr, _ptr, _stream, _check_inputs

_THR = float(np.finfo(np.float32).min) / 2


def matrices_to_fsa(trans: torch.Tensor, trans_lengths: torch.Tensor, transform: torch.Tensor) -> Fsa:
    """(trans, trans_lengths, transform) -> FsaVec of B graphs (see the module docstring).

    Built where the matrices live: the allowed (source j, destination i) pairs of the whole batch are ONE masked
    ``nonzero`` over [B, S, S + 1] (column S = the final arc of source j), which already comes out in k2's arc order
    (utterance, source state, destination; the final arc last); the host receives the finished arc list (one copy) and
    only prepares the tables.  Rows of ``transform`` that allow several tokens (not produced by the reference's data
    pipe, whose rows are one-hot: deprecated/datapipe_manual_ctc.py:245-281) take the per-utterance loop below."""
    B, S, _ = trans.shape
    dev = trans.device
    allowed_tok = transform > _THR                       # [B,S,V] on the tensors' device
    ntok = allowed_tok.sum(-1)                           # [B,S]
    lens_h = trans_lengths.cpu().numpy().astype(np.int64)
    for b in range(B):
        if lens_h[b] < 1 or lens_h[b] > S:
            raise ValueError(f"trans_lengths[{b}] = {int(lens_h[b])} out of range for S = {S}")
    if bool((ntok > 1).any()):
        return _matrices_to_fsa_multi(trans, lens_h, transform, allowed_tok)
    tr = trans.detach().to(torch.float32)
    lens = trans_lengths.to(device=dev, dtype=torch.long)
    ar = torch.arange(S, device=dev)
    valid = ar[None, :] < lens[:, None]                  # [B,S] states of the machine
    tokw, tok = transform.detach().to(torch.float32).max(-1)   # [B,S]: weight and id of the row's token
    bidx = torch.arange(B, device=dev)
    # emitting arcs j -> i, indexed [b, j, i] so that nonzero() lists them by source state
    em = ((tr > _THR) & valid[:, :, None] & valid[:, None, :] & (ntok >= 1)[:, :, None]).transpose(1, 2)
    w_em = (tr + tokw[:, :, None]).transpose(1, 2)
    fin_row = tr[bidx, lens - 1]                         # [B, j] = trans[b, end, j]
    fin = (fin_row > _THR) & valid
    idx = torch.cat([em, fin[:, :, None]], 2).nonzero()  # [n, 3] = (b, j, d), d == S: the final arc
    ab, aj, ad = idx[:, 0], idx[:, 1], idx[:, 2]
    is_fin = ad == S
    adc = ad.clamp(max=S - 1)
    w = torch.where(is_fin, fin_row[ab, aj], w_em[ab, aj, adc])
    arcs = torch.stack([aj, torch.where(is_fin, lens[ab], ad), torch.where(is_fin, torch.full_like(ad, -1), tok[ab, adc]),
                        w.view(torch.int32).to(torch.long)], 1).to(torch.int32)
    counts = torch.bincount(ab, minlength=B)
    host = torch.cat([arcs.reshape(-1), counts.to(torch.int32)]).cpu().numpy()     # the one device -> host copy
    cnt = host[-B:].astype(np.int64)
    if (cnt == 0).any():
        raise ValueError(f"utterance {int(np.nonzero(cnt == 0)[0][0])}: the state machine has no allowed transition")
    off = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(cnt, out=off[1:])
    return Fsa(host[:-B].reshape(-1, 4), None, off, (lens_h + 1).astype(np.int32))


def _matrices_to_fsa_multi(trans, lens, transform, allowed_tok):
    """Rows of ``transform`` with several allowed tokens: one arc per (j, i, token), per-utterance on the host."""
    B, S, _ = trans.shape
    ntok = allowed_tok.sum(-1).cpu().numpy()
    tr = trans.detach().to(torch.float32).cpu().numpy()
    fsas = []
    for b in range(B):
        Sb = int(lens[b])
        t = tr[b, :Sb, :Sb]
        ii, jj = np.nonzero(t > _THR)                    # transition j -> i
        emits = ntok[b, ii] >= 1
        ii, jj = ii[emits], jj[emits]
        rows = []
        at = allowed_tok[b].cpu().numpy()
        tf = transform[b].detach().cpu().numpy()
        for i, j in zip(ii, jj):
            for c in np.nonzero(at[i])[0]:
                rows.append((j, i, c, t[i, j] + tf[i, c]))
        arcs_e = np.array(rows, dtype=np.float64).reshape(-1, 4)
        src, dst, lab, w = (arcs_e[:, 0].astype(np.int32), arcs_e[:, 1].astype(np.int32),
                            arcs_e[:, 2].astype(np.int32), arcs_e[:, 3].astype(np.float32))
        end = Sb - 1
        fj = np.nonzero(t[end, :] > _THR)[0].astype(np.int32)
        fsrc, fdst = fj, np.full(fj.size, Sb, dtype=np.int32)
        flab, fw = np.full(fj.size, -1, dtype=np.int32), t[end, fj].astype(np.float32)
        a_src = np.concatenate([src, fsrc]); a_dst = np.concatenate([dst, fdst])
        a_lab = np.concatenate([lab, flab]); a_w = np.concatenate([w, fw])
        # k2 layout: arcs grouped by source state (stable: emitting arcs in (j, i) order, then the final arc)
        order = np.argsort(a_src, kind="stable")
        arcs = np.stack([a_src[order], a_dst[order], a_lab[order], a_w[order].view(np.int32)], 1).astype(np.int32)
        if arcs.shape[0] == 0:
            raise ValueError(f"utterance {b}: the state machine has no allowed transition")
        fsas.append(Fsa(arcs, None, [0, arcs.shape[0]], [Sb + 1]))
    return create_fsa_vec(fsas)


def _dense(log_probs, input_lengths):
    B = log_probs.shape[0]
    lens = input_lengths.detach().cpu().to(torch.int32)
    seg = torch.stack([torch.arange(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32), lens], 1)
    return DenseFsaVec(log_probs, seg)


def _reduce(tot, trans_lengths, reduction):
    # ctc.py:57-73: every frame's normaliser equals the utterance's total log-likelihood
    if reduction == 'none':
        return -tot
    if reduction == 'sum':
        return -tot.sum()
    if reduction == 'mean':
        return -(tot / trans_lengths.to(tot.device)).mean()
    raise ValueError(reduction)


def ctc_loss(log_probs: torch.Tensor, input_lengths: torch.Tensor, trans: torch.Tensor,
             trans_lengths: torch.Tensor, transform: torch.Tensor, reduction='sum'):
    """waterfall/manual_ctc/ctc.py:94-105.  The gradient w.r.t. log_probs is -(state posteriors summed per
    token) times the upstream factor, for every reduction (the reference's own backward only works for 'sum')."""
    graph = matrices_to_fsa(trans, trans_lengths, transform)
    dense = _dense(log_probs, input_lengths)
    tot = _FbFunction.apply(log_probs, graph, dense, 0.0, False)
    return _reduce(tot, trans_lengths, reduction)


def ctc_softmax(log_probs: torch.Tensor, input_lengths: torch.Tensor, trans: torch.Tensor,
                transform: torch.Tensor, trans_lengths: torch.Tensor, eta=1.0, reduction='sum'):
    """waterfall/manual_ctc/softmax_ctc.py:97-110 (note the reference's argument order: transform before
    trans_lengths).  The eta-semiring 1/eta * LSE(eta * x) is the log semiring on scores scaled by eta:
    with y = eta * alpha, y_t[i] = LSE_j(eta * trans[i,j] + y_{t-1}[j]) + eta * e_t[i], so the loss is
    1/eta times the ordinary total score of (eta * log_probs, eta * trans); autograd carries the
    factors (d loss / d log_probs = -posteriors of the scaled model, softmax_ctc.py:80-92)."""
    eta = float(eta)
    graph = matrices_to_fsa(trans * eta, trans_lengths, transform * eta)
    scaled = log_probs * eta
    dense = _dense(scaled, input_lengths)
    tot = _FbFunction.apply(scaled, graph, dense, 0.0, False) / eta
    return _reduce(tot, trans_lengths, reduction)


def compute_align(log_probs: torch.Tensor, input_lengths: torch.Tensor, trans: torch.Tensor,
                  trans_lengths: torch.Tensor, transform: torch.Tensor):
    """waterfall/manual_ctc/prior.py:4-55: log of the normalised state posteriors, [B, T, S].
    Entries the reference leaves as garbage (frames >= input_lengths[b], states >= trans_lengths[b]) and
    impossible states are -inf here."""
    graph = matrices_to_fsa(trans, trans_lengths, transform)
    dense = _dense(log_probs.detach(), input_lengths)
    lp, seg, B = _check_inputs(graph, dense)
    lib = _lib.lib()
    dev = lp.device
    N, T, V = lp.shape
    S = trans.shape[-1]
    flags = _lib.WF_KEEP_AB
    gh, gd = graph.blob_host(), graph.blob_dev(dev)
    n = C.c_size_t(0)
    _lib.check(lib.wf_fb_workspace_bytes(_np_ptr(gh), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, N, V, flags, 1, C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    tot = torch.empty(B, dtype=torch.float32, device=dev)
    out = torch.empty((B, T, S + 1), dtype=torch.float32, device=dev)   # + the graph's super-final state
    with torch.cuda.device(dev):
        _lib.check(lib.wf_fb_logsum(_np_ptr(gh), _ptr(gd), _np_ptr(graph.utt2fsa), _ptr(lp), lp.stride(0), lp.stride(1),
                                    N, T, V, _np_ptr(seg), B, 0.0, flags, None, _ptr(tot), None, _ptr(ws), n.value,
                                    _stream(dev)))
        _lib.check(lib.wf_state_posteriors(_np_ptr(gh), _ptr(gd), _np_ptr(graph.utt2fsa), _np_ptr(seg), B, V, N, flags,
                                           _ptr(out), out.stride(0), out.stride(1), S + 1, T, _ptr(ws), n.value,
                                           _stream(dev)))
    return out[:, :, :S]
