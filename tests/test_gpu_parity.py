"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes), against the oracle.

Tolerances are north_star's: total log-likelihood within 1e-4 relative, gradients within 1e-3
absolute (fp32), Viterbi arc sequences bit-exact with ties broken by the lowest arc index.
"""
import numpy as np
import pytest
import torch

import waterfall_amd as wf
from oracle import dwfst_oracle as o
from tests import util

pytestmark = pytest.mark.gpu

TOT_RTOL = 1e-4
TOT_ATOL = 2e-5   # only matters for totals that are ~0 (a complete CTC denominator sums to log 1)
GRAD_ATOL = 1e-3
DEV = "cuda:0"

TOPO_CASES = ["2state", "2state-1", "3state-skip", "3state-skip-1", "3state-skip-2", "mmictc", "mmictc-1",
              "2state_sil", "mmictc_noblk"]


def run_hip(fsa, lp, segs, reduction="none", target_lengths=None, use_double=False, need_grad=True, eager=True):
    """graphloss + backward under autograd.  eager (the default of the drop-in): loss and gradient by the fused call in
    forward for graphs with compact label columns; eager=False: the forward / backward split (two C calls)."""
    x = torch.from_numpy(np.ascontiguousarray(lp)).to(DEV).requires_grad_(need_grad)
    dense = wf.DenseFsaVec(x, torch.from_numpy(np.ascontiguousarray(segs, dtype=np.int32)))
    wf.set_eager_gradient(eager)
    try:
        loss = wf.graphloss(fsa, dense, output_beam=10, reduction=reduction, use_double_scores=use_double,
                            target_lengths=target_lengths)
        grad = None
        if need_grad:
            (loss.sum() if reduction == "none" else loss).backward()
            grad = x.grad.cpu().numpy()
    finally:
        wf.set_eager_gradient(True)
    return loss.detach().cpu().numpy(), grad


def check_against_oracle(c, reduction="none", use_double=False, target_lengths=None):
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], reduction, target_lengths, use_double)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, reduction,
                              None if target_lengths is None else target_lengths.numpy())
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    return loss, grad


@pytest.mark.parametrize("name", ["manual_ctc_ctc_small.npz", "manual_ctc_ctc_medium.npz"])
def test_golden_ctc_from_reference_manual_ctc(name):
    z = util.load_golden(name)
    V = z["log_probs"].shape[-1]
    topo, _ = wf.build_topology("ctc", util.ctc_phones(V))
    off = z["target_off"]
    fsa = wf.compile_ctc_k2_graphs(topo, [z["targets"][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)])
    B = z["log_probs"].shape[0]
    segs = np.stack([np.arange(B), np.zeros(B), z["input_lengths"]], 1).astype(np.int32)
    loss, grad = run_hip(fsa, z["log_probs"], segs, "none")
    np.testing.assert_allclose(loss, z["loss_none"], rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, z["grad"], atol=GRAD_ATOL)
    loss_sum, _ = run_hip(fsa, z["log_probs"], segs, "sum")
    np.testing.assert_allclose(loss_sum, z["loss_sum"], rtol=TOT_RTOL, atol=TOT_ATOL)


@pytest.mark.parametrize("topo", TOPO_CASES)
def test_golden_topologies_from_reference_manual_ctc(topo):
    z = util.load_golden(f"manual_ctc_{topo}.npz")
    fsa = util.graphs_to_fsa(util.golden_graphs(z))
    B = z["log_probs"].shape[0]
    segs = np.stack([np.arange(B), np.zeros(B), z["input_lengths"]], 1).astype(np.int32)
    loss, grad = run_hip(fsa, z["log_probs"], segs, "none")
    np.testing.assert_allclose(loss, z["loss_none"], rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, z["grad"], atol=GRAD_ATOL)


@pytest.mark.parametrize("topo", sorted(wf.TOPOLOGIES))
@pytest.mark.parametrize("den", [False, True])
def test_every_topology_numerator_and_denominator(topo, den):
    if topo == "ctc" and den:
        P = 12  # the full CTC topology is O(P^2) arcs
    else:
        P = 9
    c = util.random_case(100 + len(topo), topo, B=4, T=50, L=7, P=P, den=den)
    check_against_oracle(c)


@pytest.mark.parametrize("eager", [True, False])
@pytest.mark.parametrize("reduction", ["none", "sum", "mean"])
def test_reductions_match_graph_py(reduction, eager):
    c = util.random_case(7, "2state", B=5, T=30, L=4, P=6)
    tl = torch.tensor([len(s) for s in c["seqs"]], dtype=torch.float32)
    tlen = tl if reduction == "mean" else None
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], reduction, tlen, eager=eager)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, reduction, None if tlen is None else tlen.numpy())
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)


def test_eager_gradient_upstream_factors_and_numerator_minus_denominator():
    # loss = 0.25 * (numerator - denominator) (models.py:338-340 shape): the numerator's eager gradient is scaled by the
    # upstream factor in backward; per-utterance factors with reduction 'none'
    c = util.random_case(15, "mmictc", B=4, T=36, L=5, P=7)
    d = util.random_case(15, "mmictc", B=4, T=36, L=5, P=7, den=True)
    x = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
    num = wf.graphloss(c["fsa"], dense, reduction="sum")
    den = wf.graphloss(d["fsa"], dense, reduction="sum")
    (0.25 * (num - den)).backward()
    wn, gn = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "sum")
    wd, gd = o.graphloss(d["ographs"], c["lp"], c["segs"], None, "sum")
    np.testing.assert_allclose(num.item(), wn, rtol=TOT_RTOL)
    np.testing.assert_allclose(den.item(), wd, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), 0.25 * (gn - gd), atol=GRAD_ATOL)
    x2 = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
    per = wf.graphloss(c["fsa"], wf.DenseFsaVec(x2, torch.from_numpy(c["segs"])), reduction="none")
    f = torch.tensor([1.0, -2.0, 0.5, 1.0], device=DEV)
    (per * f).sum().backward()
    _, g_none = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    np.testing.assert_allclose(x2.grad.cpu().numpy(), g_none * f.cpu().numpy()[:, None, None], atol=2 * GRAD_ATOL)


def test_double_scores():
    c = util.random_case(8, "mmictc", B=3, T=40, L=5, P=6)
    loss, grad = check_against_oracle(c, use_double=True)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    np.testing.assert_allclose(loss, want, rtol=2e-7)
    # (the totals are fp64; the gradient is the fast kernels' fp32 posterior arithmetic, 3e-5 here, as without the flag --
    # the fp64 log-semiring route, WF_NO_DOUBLE_FAST=1, stays within 2e-5)
    np.testing.assert_allclose(grad, wgrad, atol=1e-4)
    _, grad_slow = _with_env({"WF_NO_DOUBLE_FAST": "1"}, lambda: check_against_oracle(c, use_double=True))
    np.testing.assert_allclose(grad_slow, wgrad, atol=2e-5)


def test_no_grad_forward_only_and_shared_denominator():
    c = util.random_case(9, "3state-skip", B=6, T=33, L=4, P=10, den=True)
    assert c["fsa"].num_fsas == 1 and len(c["fsa"].utt2fsa) == 6   # [topo]*B stored once
    with torch.no_grad():
        loss, _ = run_hip(c["fsa"], c["lp"], c["segs"], "none", need_grad=False)
    want, _ = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)


def test_infeasible_utterance_gives_inf_loss_and_zero_grad():
    # models.py:305-313 (mask_inf): an utterance shorter than its transcript has no path
    c = util.random_case(10, "ctc", B=3, T=12, L=8, P=5, ragged=False)
    segs = c["segs"].copy()
    segs[1, 2] = 5    # 8 labels cannot fit in 5 frames
    loss, grad = run_hip(c["fsa"], c["lp"], segs, "none")
    assert np.isinf(loss[1]) and loss[1] > 0 and np.isfinite(loss[[0, 2]]).all()
    assert np.all(grad[1] == 0) and not np.isnan(grad).any()
    want, wgrad = o.graphloss(c["ographs"], c["lp"], segs, None, "none")
    np.testing.assert_allclose(loss[[0, 2]], want[[0, 2]], rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)


def test_ragged_segments_start_offsets_and_unused_sequences():
    # segments need not start at frame 0 nor cover every sequence (k2.DenseFsaVec semantics)
    c = util.random_case(12, "2state", B=3, T=40, L=3, P=5, ragged=False)
    lp = np.concatenate([c["lp"], c["lp"][:1]], 0)            # 4 sequences, the last one unused
    segs = np.array([[2, 5, 30], [0, 0, 40], [1, 10, 21]], dtype=np.int32)
    loss, grad = run_hip(c["fsa"], lp, segs, "none")
    want, wgrad = o.graphloss(c["ographs"], lp, segs, None, "none")
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    assert np.all(grad[3] == 0) and np.all(grad[2, :5] == 0) and np.all(grad[2, 35:] == 0)


def test_many_more_sequences_than_segments():
    # DenseFsaVec allows log_probs to hold far more sequences than the segments name (N = 300, B = 3): the workspace is
    # sized with the same num_seqs that the compute calls plan with
    c = util.random_case(14, "ctc", B=3, T=25, L=3, P=5, ragged=False)
    lp = np.log(np.full((300, 25, c["V"]), 1.0 / c["V"], dtype=np.float32))
    segs = np.array([[299, 0, 25], [7, 2, 20], [150, 0, 25]], dtype=np.int32)
    for i in range(3):
        lp[segs[i, 0]] = c["lp"][i]
    loss, grad = run_hip(c["fsa"], lp, segs, "none")
    want, wgrad = o.graphloss(c["ographs"], lp, segs, None, "none")
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    tot, fgrad = fused_logsum(c["fsa"], lp, segs)
    np.testing.assert_allclose(-tot, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(fgrad, wgrad, atol=GRAD_ATOL)
    assert np.count_nonzero(np.abs(fgrad).sum((1, 2))) == 3


def test_empty_transcript_and_single_frame():
    topo, tokens = wf.build_topology("ctc", util.ctc_phones(6))
    fsa = wf.compile_ctc_k2_graphs(topo, [[], [3], [2, 2]])
    g = torch.Generator().manual_seed(1)
    lp = torch.log_softmax(torch.randn(3, 4, len(tokens), generator=g), -1).numpy()
    segs = np.array([[0, 0, 4], [1, 0, 1], [2, 0, 3]], dtype=np.int32)
    loss, grad = run_hip(fsa, lp, segs, "none")
    want, wgrad = o.graphloss(fsa.as_oracle_graphs(), lp, segs, None, "none")
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    np.testing.assert_allclose(loss[0], -lp[0, :, 0].sum(), rtol=1e-6)   # blanks only


@pytest.mark.parametrize("topo,den", [("ctc", False), ("2state", False), ("mmictc", True), ("3state-skip-2", True),
                                      ("ctc", True)])
def test_viterbi_bit_exact(topo, den):
    c = util.random_case(40, topo, B=4, T=60, L=8, P=8, den=den)
    x = torch.from_numpy(c["lp"]).to(DEV)
    dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
    out = wf.best_path(c["fsa"], dense)
    scores = out["scores"].cpu().numpy()
    arcs = out["arcs"].cpu().numpy()
    trop = wf.tot_scores(c["fsa"], dense, log_semiring=False).cpu().numpy()
    base = c["fsa"].fsa_arc_offsets
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path = o.viterbi(g, c["lp"][i], T)
        gi = 0 if c["fsa"].utt2fsa is not None else i
        assert scores[i] == s and trop[i] == s                      # bit-exact fp32
        assert list(arcs[i, :T + 1] - base[gi]) == path
        assert np.all(arcs[i, T + 1:] == -1)
        lab = [int(g["label"][a]) for a in path[:-1]]
        assert out["labels"][i] == lab


def test_viterbi_ties_lowest_arc_index_and_infeasible():
    arcs = [(0, 0, 1, 0.0), (0, 0, 1, 0.0), (0, 1, 2, 0.0), (0, 1, 2, 0.0), (1, 2, -1, 0.0), (1, 2, -1, 0.0)]
    g = o.make_graph(arcs, 3)
    fsa = util.graphs_to_fsa([g, g])
    lp = np.log(np.full((2, 4, 3), 1 / 3, dtype=np.float32))
    segs = np.array([[0, 0, 4], [1, 0, 0]], dtype=np.int32)      # second utterance: T=0, needs >= 1 frame
    dense = wf.DenseFsaVec(torch.from_numpy(lp).to(DEV), torch.from_numpy(segs))
    out = wf.best_path(fsa, dense)
    a = out["arcs"].cpu().numpy()
    assert list(a[0]) == [0, 0, 0, 2, 4]
    assert np.all(a[1] == -1) and out["scores"][1].item() == -np.inf


def test_tot_scores_log_and_double():
    c = util.random_case(41, "2state", B=3, T=45, L=5, P=7)
    dense = wf.DenseFsaVec(torch.from_numpy(c["lp"]).to(DEV), torch.from_numpy(c["segs"]))
    want, _ = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    t32 = wf.tot_scores(c["fsa"], dense, True, False).cpu().numpy()
    t64 = wf.tot_scores(c["fsa"], dense, True, True).cpu().numpy()
    np.testing.assert_allclose(-t32, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(-t64, want, rtol=1e-6)
    assert t64.dtype == np.float64
    v64 = wf.tot_scores(c["fsa"], dense, False, True).cpu().numpy()
    for i, g in enumerate(c["ographs"]):
        s, _ = o.viterbi(g, c["lp"][i], int(c["segs"][i, 2]), dtype=np.float64)
        np.testing.assert_allclose(v64[i], s, rtol=1e-12)


def test_non_contiguous_log_probs_view():
    # encoder output sliced along V (stride(1) != V): strides travel through the ABI
    c = util.random_case(13, "ctc", B=2, T=20, L=3, P=4, ragged=False)
    V = c["V"]
    big = torch.zeros(2, 20, V + 3)
    big[:, :, :V] = torch.from_numpy(c["lp"])
    x = big.to(DEV)[:, :, :V].requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
    loss = wf.graphloss(c["fsa"], dense, reduction="sum")
    loss.backward()
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "sum")
    np.testing.assert_allclose(loss.item(), want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), wgrad, atol=GRAD_ATOL)


# ---- BASELINE.json sizes: C oracle where it finishes in seconds + size-independent properties ----

def _full_size(topo, P, L, den, seed=777, B=32, T=500):
    return util.random_case(seed, topo, B=B, T=T, L=L, P=P, den=den, ragged=True)


@pytest.mark.parametrize("topo,P,L,den", [("2state", 29, 80, False), ("2state", 29, 80, True),
                                          ("ctc", 5000, 64, False)])
def test_full_size_against_c_oracle_and_properties(oracle_lib, topo, P, L, den):
    c = _full_size(topo, P, L, den)
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    B = len(loss)
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, c["ographs"], c["lp"], c["segs"], gscale=[-1.0] * B,
                                        use_double=True, nthreads=8)
    np.testing.assert_allclose(loss, -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, grad_d, atol=GRAD_ATOL)
    for b in range(B):
        T = int(c["segs"][b, 2])
        np.testing.assert_allclose(grad[b, :T].sum(-1), -1.0, atol=2e-3)   # posteriors of one frame sum to 1
        assert np.all(grad[b, T:] == 0)
    assert np.all(grad <= 0)
    tot, fgrad = fused_logsum(c["fsa"], c["lp"], c["segs"])
    assert LAST_FALLBACKS[0] == 0          # the probability-domain kernels' own rows: nothing was redone in the log semiring
    np.testing.assert_allclose(-tot, -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(fgrad, grad_d, atol=GRAD_ATOL)


def test_full_size_mmictc_denominator_properties(oracle_lib):
    # cfg4 (LS BPE-5k mmictc denominator: S=10002, A=30002, fan-in(0)=15001) at reduced batch so the
    # test stays short; checked through properties + a float64 C-oracle pass on two utterances
    c = util.random_case(777, "mmictc", B=4, T=120, L=8, P=5000, den=True)
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    for b in range(4):
        T = int(c["segs"][b, 2])
        np.testing.assert_allclose(grad[b, :T].sum(-1), -1.0, atol=2e-3)
        assert np.all(grad[b, T:] == 0)
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, c["ographs"][:2], c["lp"][:2], c["segs"][:2], gscale=[-1.0] * 2,
                                        use_double=True, nthreads=2)
    np.testing.assert_allclose(loss[:2], -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad[:2], grad_d, atol=GRAD_ATOL)
    tot_f, grad_f = fused_logsum(c["fsa"], c["lp"], c["segs"])
    assert LAST_FALLBACKS[0] == 0          # sweep_hub_kernel's rows passed the check: no utterance was redone
    np.testing.assert_allclose(-tot_f[:2], -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad_f[:2], grad_d, atol=GRAD_ATOL)
    # default: probability-domain sweep_linbig_kernel.  The other implementations on the same batch: the
    # log-semiring sweep with this thread's arcs in registers / re-read from memory in every frame (same
    # arithmetic in the same order: bit-identical), the LDS-staged gradient kernel, the per-frame-launch sweep,
    # and the linear sweep with every utterance sent through the check -> fallback path
    run = lambda: run_hip(c["fsa"], c["lp"], c["segs"], "none")
    lg, gg = _with_env({"WF_NO_LINBIG_SWEEP": "1"}, run)
    np.testing.assert_allclose(lg, loss, rtol=1e-6)
    np.testing.assert_allclose(gg, grad, atol=2e-5)
    np.testing.assert_allclose(lg[:2], -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(gg[:2], grad_d, atol=GRAD_ATOL)
    l2, g2 = _with_env({"WF_NO_LINBIG_SWEEP": "1", "WF_NO_ARC_CACHE": "1"}, run)
    assert np.array_equal(lg, l2) and np.array_equal(gg, g2)
    l3, g3 = _with_env({"WF_NO_LINBIG_SWEEP": "1", "WF_POST_BIG_MIN_S": "1000000"}, run)
    np.testing.assert_allclose(l3, lg, rtol=1e-6)
    np.testing.assert_allclose(g3, gg, atol=1e-5)
    l4, g4 = _with_env({"WF_BIG_MIN_S": "1000"}, run)
    np.testing.assert_allclose(l4, lg, rtol=1e-6)
    np.testing.assert_allclose(g4, gg, atol=1e-5)
    l5, g5 = _with_env({"WF_FORCE_FALLBACK": "1"}, run)
    np.testing.assert_allclose(l5, lg, rtol=1e-6)
    np.testing.assert_allclose(g5, gg, atol=1e-5)
    # forward only (torch.no_grad() denominators): planned as both linear sweeps + row check without the posterior
    # kernel; the same totals as the gradient call, and as the log-semiring forward sweep
    fwd = lambda: run_hip(c["fsa"], c["lp"], c["segs"], "none", need_grad=False)[0]
    assert np.array_equal(fwd(), loss)
    np.testing.assert_allclose(_with_env({"WF_NO_FWD_VIA_FB": "1"}, fwd), loss, rtol=1e-6)
    np.testing.assert_allclose(_with_env({"WF_FORCE_FALLBACK": "1"}, fwd), loss, rtol=1e-6)


# ---- every sweep implementation behind wf_fb_logsum, on the same inputs ----------------------------
# default: banded wave sweep (numerators) / linear-domain LDS sweep; the environment hooks below force
# the other kernels and the check -> fallback path, so that each one is held to the same tolerances.

SWEEP_PATHS = {
    "table_by_copy": {"WF_NO_TAB_KERNARG": "1"},          # per-call tables through the pinned ring instead of kernel arguments
    "default": {},                                                        # banded sweep, row check inside post_small_kernel
    "check_kernel": {"WF_NO_POST_CHECK": "1"},                             # separate row-check kernel before the posterior kernel
    "fill_side_stream": {"WF_FILL_SIDE_STREAM": "1"},                      # gradient zero-fill on the side stream instead of inside the sweep's launch
    "band_then_fallback_check_kernel": {"WF_FORCE_FALLBACK": "1", "WF_NO_POST_CHECK": "1"},
    "no_gen_wave": {"WF_NO_GEN_SWEEP": "1"},                              # cyclic graphs <= 64 states: sweep_lin instead of the wave kernel's general mode
    "lin_lds": {"WF_NO_BAND_SWEEP": "1"},                                 # <= 64 states: the wave-resident variant (ds_bpermute)
    "lin_lds_no_wave": {"WF_NO_BAND_SWEEP": "1", "WF_NO_WAVE_SWEEP": "1"},  # state vector through the LDS whatever the size
    "log_generic": {"WF_NO_BAND_SWEEP": "1", "WF_NO_LIN_SWEEP": "1"},
    "band_then_fallback": {"WF_FORCE_FALLBACK": "1"},
    "lin_then_fallback": {"WF_NO_BAND_SWEEP": "1", "WF_FORCE_FALLBACK": "1"},
    "linbig": {"WF_NO_BAND_SWEEP": "1", "WF_NO_LIN_SWEEP": "2"},       # sweep_linbig_kernel (large denominators)
    "linbig_then_fallback": {"WF_NO_BAND_SWEEP": "1", "WF_NO_LIN_SWEEP": "2", "WF_FORCE_FALLBACK": "1"},
}


LAST_FALLBACKS = [None]   # fused_logsum: utterances of the last call that the log-semiring fallback redid


def fused_logsum(fsa, lp, segs, gscale=-1.0, beam=0.0):
    """wf_fb_logsum with grad != NULL (loss + gradient in one call: the path bench.py times).  Leaves the number of
    utterances the fallback redid in LAST_FALLBACKS[0] (a kernel whose rows fail the check is repaired by it: tests
    of a probability-domain kernel assert 0).  beam > 0: output_beam pruning (WF_PRUNED)."""
    import ctypes as C
    from waterfall_amd import _lib
    lib = _lib.lib()
    x = torch.from_numpy(np.ascontiguousarray(lp)).to(DEV)
    N, T, V = x.shape
    segs = np.ascontiguousarray(segs, dtype=np.int32)
    B = segs.shape[0]
    gh, gd = fsa.blob_host(), fsa.blob_dev(torch.device(DEV))
    p = lambda t: C.c_void_p(t.data_ptr())
    npp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)
    n = C.c_size_t(0)
    fl = _lib.WF_PRUNED if beam > 0 else 0
    _lib.check(lib.wf_fb_workspace_bytes(npp(gh), npp(fsa.utt2fsa), npp(segs), B, N, V, fl, 1, C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=DEV)
    tot = torch.empty(B, dtype=torch.float32, device=DEV)
    grad = torch.full_like(x, 7.0)   # every element must be overwritten
    gs = torch.full((B,), gscale, dtype=torch.float32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
    _lib.check(lib.wf_fb_logsum(npp(gh), p(gd), npp(fsa.utt2fsa), p(x), x.stride(0), x.stride(1), N, T, V, npp(segs), B,
                                float(beam), fl, p(gs), p(tot), p(grad), p(ws), n.value, st))
    cnt = C.c_int32(-1)
    _lib.check(lib.wf_debug_fallback_count(npp(gh), npp(fsa.utt2fsa), npp(segs), B, N, V, fl, 1, p(ws), n.value, st, C.byref(cnt)))
    LAST_FALLBACKS[0] = cnt.value
    torch.cuda.synchronize()
    return tot.cpu().numpy(), grad.cpu().numpy()


def _with_env(env, fn):
    """Runs fn with the WF_* test hooks set: the library reads them once per process, wf_debug_reload_env re-reads."""
    import os
    from waterfall_amd import _lib
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    _lib.reload_env()
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _lib.reload_env()


SWEEP_CASES = [
    # topo, den, L, T, P  -> banded instantiation it reaches by default
    ("ctc", False, 7, 50, 9),          # <2,2>
    ("ctc", False, 45, 130, 30),       # <2,2>, S = 92
    ("ctc", False, 80, 203, 40),       # <3,2>, T not a multiple of the chunk
    ("ctc", False, 100, 230, 40),      # <4,2>
    ("2state", False, 30, 100, 12),    # <2,2>
    ("3state-skip-2", False, 20, 90, 9),   # band 3 -> <4,4>
    ("mmictc", False, 20, 90, 9),      # band 4, two arcs per cell -> <4,4,2>
    ("2state_sil3", False, 10, 70, 9),  # three parallel arcs: not banded-eligible -> linear LDS sweep
    ("2state", True, 5, 60, 12),       # denominator: hub state, not banded -> linear LDS sweep with a cooperative state
    ("mmictc", True, 5, 33, 7),
    ("ctc", False, 3, 1, 5),           # single frame (infeasible for 3 labels)
    ("ctc", False, 2, 5, 5),
]


@pytest.mark.parametrize("path", sorted(SWEEP_PATHS))
@pytest.mark.parametrize("topo,den,L,T,P", SWEEP_CASES)
def test_sweep_paths_fused_and_autograd(path, topo, den, L, T, P):
    c = util.random_case(300 + L + T, topo, B=5, T=T, L=L, P=P, den=den, ragged=T > 10)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    ok = np.isfinite(want)

    def run():
        tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"])
        nfb = LAST_FALLBACKS[0]
        loss, agrad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
        loss2, agrad2 = run_hip(c["fsa"], c["lp"], c["segs"], "none", eager=False)
        return tot, grad, loss, agrad, loss2, agrad2, nfb
    tot, grad, loss, agrad, loss2, agrad2, nfb = _with_env(SWEEP_PATHS[path], run)
    # the kernel under test delivered the rows itself (a wrong row fails the check and the fallback repairs it silently)
    assert (nfb >= ok.sum()) if "fallback" in path else (nfb == 0)   # (forced: every feasible utterance is redone)
    for got_loss, got_grad in ((-tot, grad), (loss, agrad), (loss2, agrad2)):
        assert np.array_equal(np.isfinite(got_loss), ok)
        np.testing.assert_allclose(got_loss[ok], want[ok], rtol=TOT_RTOL, atol=TOT_ATOL)
        np.testing.assert_allclose(got_grad, wgrad, atol=GRAD_ATOL)
        assert not np.isnan(got_grad).any()


@pytest.mark.parametrize("path", sorted(SWEEP_PATHS))
def test_sweep_paths_weighted_arcs(path):
    # arc weights != 0 (LM-like scores): the banded sweep's weighted instantiation and the others
    c = util.random_case(77, "ctc", B=4, T=60, L=9, P=8)
    rng = np.random.default_rng(5)
    graphs = []
    for g in c["ographs"]:
        g = dict(g)
        g["score"] = (rng.standard_normal(g["score"].shape) * 0.7).astype(np.float32)
        graphs.append(g)
    fsa = util.graphs_to_fsa(graphs)
    want, wgrad = o.graphloss(graphs, c["lp"], c["segs"], None, "none")
    tot, grad = _with_env(SWEEP_PATHS[path], lambda: fused_logsum(fsa, c["lp"], c["segs"]))
    assert "fallback" in path or LAST_FALLBACKS[0] == 0   # (weighted arcs: some of the forced paths have no probability-domain sweep to fall back from)
    np.testing.assert_allclose(-tot, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)


def test_linear_sweep_underflow_is_caught_by_the_row_check():
    # Every path must start through labels ~900 nats below the frame's best label AMONG THE GRAPH'S OWN (the model
    # insists on the transcript's last label from the first frame on; c_t is the maximum over the graph's K columns):
    # exp(lp - c_t) of the labels the first frames need is 0, the probability-domain sweep dies, the row identity
    # fails, and the log-semiring fallback must deliver the exact result -- for that utterance only.
    topo, tokens = wf.build_topology("ctc", util.ctc_phones(6))
    fsa = wf.compile_ctc_k2_graphs(topo, [[1, 2, 3], [2, 2]])
    T, V = 40, len(tokens)
    lp = np.full((2, T, V), -900.0, dtype=np.float32)
    lp[:, :, 3] = 0.0          # label 3 ends the first transcript: reachable only after labels 1 and 2 at -900 each
    lp[1, :, :] = np.log(1.0 / V)   # the second utterance is ordinary
    segs = np.array([[0, 0, T], [1, 0, T]], dtype=np.int32)
    want, wgrad = o.graphloss(fsa.as_oracle_graphs(), lp, segs, None, "none")
    assert np.isfinite(want).all() and want[0] > 1500
    tot, grad = fused_logsum(fsa, lp, segs)
    assert LAST_FALLBACKS[0] == 1          # the first utterance only
    np.testing.assert_allclose(-tot, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    loss, agrad = run_hip(fsa, lp, segs, "none")
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(agrad, wgrad, atol=GRAD_ATOL)


# ---- output_beam pruning (k2.intersect_dense, graph.py:61-62; arithmetic parity-unpinned: SURVEY 8(0)) ----

@pytest.mark.parametrize("beam", [0.5, 2.0, 5.0, 10.0])
@pytest.mark.parametrize("topo,den", [("ctc", False), ("2state", False), ("mmictc", True), ("3state-skip", True)])
def test_output_beam_pruning_matches_oracle(beam, topo, den):
    c = util.random_case(500 + int(beam * 10), topo, B=4, T=40, L=6, P=7, den=den)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], beam, "none")
    wf.set_output_beam_mode("pruned")
    try:
        x = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
        dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
        loss = wf.graphloss(c["fsa"], dense, output_beam=beam, reduction="none")
        loss.sum().backward()
    finally:
        wf.set_output_beam_mode("exact")
    # the tropical pass runs in fp32 here and in fp64 in the oracle: an arc within rounding of the
    # threshold may fall on either side; it carries e^-beam of the best path's mass at most
    slack = 3e-3 if beam < 1 else 1e-3
    np.testing.assert_allclose(loss.detach().cpu().numpy(), want, rtol=TOT_RTOL, atol=slack)
    np.testing.assert_allclose(x.grad.cpu().numpy(), wgrad, atol=max(GRAD_ATOL, slack))
    exact, _ = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    if beam <= 2.0:
        assert np.all(want >= exact - 1e-9) and np.any(want > exact + 1e-3)   # pruning really removed mass


def test_output_beam_10_is_within_tolerance_of_exact():
    # the drop-in's default ("exact" mode) against what the reference's default beam would keep
    c = util.random_case(61, "2state", B=6, T=120, L=14, P=12)
    pruned, pgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], 10.0, "none")
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    np.testing.assert_allclose(loss, pruned, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, pgrad, atol=GRAD_ATOL)


ALL_TOPOLOGIES = ["ctc", "2state", "2state-1", "2state_sil", "2state_sil3", "3state-skip", "3state-skip-1",
                  "3state-skip-2", "mmictc", "mmictc_noblk", "mmictc-1"]


def _peaked_like_a_trained_model(c, mass=0.99):
    """log_probs with `mass` of every frame on the label of one alignment of the utterance's numerator graph (the best
    path under the noise posterior) and the rest spread evenly: what a trained model gives, where a beam matters most."""
    lp = np.empty_like(c["lp"])
    V = c["V"]
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        _, path = o.viterbi(g, c["lp"][i], T)
        lab = [int(g["label"][a]) for a in path[:-1]]
        lp[i] = np.log((1.0 - mass) / (V - 1))
        lp[i, np.arange(T), lab] = np.log(mass)
    return lp


@pytest.mark.parametrize("posterior", ["noise", "peaked"])
@pytest.mark.parametrize("beam", [10.0, 20.0])
@pytest.mark.parametrize("topo", ALL_TOPOLOGIES)
def test_exact_mode_is_within_tolerance_of_the_reference_beams(topo, beam, posterior):
    """graphloss(output_beam=10 | 20) in the "exact" mode sums over all lattice arcs where the reference prunes
    (graph.py:61-62; beams of the recipes' configs).  How far that is from the pruned lattice -- below the tolerance of
    north_star or not -- on every topology, both beams, a noise posterior and a peaked one, against the oracle.
    [k2's own beam arithmetic: parity unpinned]"""
    c = util.random_case(900 + ALL_TOPOLOGIES.index(topo), topo, B=3, T=60, L=8, P=9)
    lp = c["lp"] if posterior == "noise" else _peaked_like_a_trained_model(c)
    pruned, pgrad = o.graphloss(c["ographs"], lp, c["segs"], beam, "none")
    loss, grad = run_hip(c["fsa"], lp, c["segs"], "none")
    # Measured: inside north_star's tolerance (1e-4 relative on the loss, 1e-3 absolute on the gradient) everywhere except
    # beam 10 on the NOISE posterior, where an untrained model spreads e^-10-sized posteriors over so many lattice arcs
    # that pruning moves the loss by up to 5e-4 relative (0.09 nats at T = 60) and single gradient entries by up to 3.3e-2:
    # there the exact default and the reference's beam differ by that much ("output_beam").
    noisy10 = beam == 10.0 and posterior == "noise"
    np.testing.assert_allclose(loss, pruned, rtol=1e-3 if noisy10 else 1e-4, atol=1e-3)
    np.testing.assert_allclose(grad, pgrad, atol=5e-2 if noisy10 else 1e-3)
    assert np.all(pruned >= loss - 1e-3)      # pruning only removes mass: the pruned loss is never the smaller one


@pytest.mark.parametrize("beam", [10.0, 20.0])
@pytest.mark.parametrize("topo", ["2state", "mmictc", "3state-skip", "ctc"])
def test_exact_mode_vs_reference_beams_on_denominators(topo, beam):
    c = util.random_case(950, topo, B=3, T=60, L=8, P=9, den=True)
    pruned, pgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], beam, "none")
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    np.testing.assert_allclose(loss, pruned, rtol=1e-4, atol=2e-2 if beam == 10.0 else 1e-3)   # (noise posterior, see above)
    np.testing.assert_allclose(grad, pgrad, atol=2e-2 if beam == 10.0 else 1e-3)


def test_output_beam_pruning_on_a_graph_beyond_the_lds():
    """intersect_dense(TLGW, dense, 10) (lattice_analysis.py:147,153) in the literal "pruned" mode on a graph that only
    the per-frame launches hold: both passes (tropical rows, masked log-semiring rows) and the masked posteriors."""
    g, lp, segs = _big_case(15_000, B=3, T=30, seed=8)
    fsa = util.graphs_to_fsa([g, g, g])
    for beam in (3.0, 10.0):
        want, wgrad = o.graphloss([g, g, g], lp, segs, beam, "none")
        wf.set_output_beam_mode("pruned")
        try:
            x = torch.from_numpy(lp).to(DEV).requires_grad_(True)
            dense = wf.DenseFsaVec(x, torch.from_numpy(segs))
            loss = wf.graphloss(fsa, dense, output_beam=beam, reduction="none")
            loss.sum().backward()
        finally:
            wf.set_output_beam_mode("exact")
        np.testing.assert_allclose(loss.detach().cpu().numpy(), want, rtol=TOT_RTOL, atol=1e-3)
        np.testing.assert_allclose(x.grad.cpu().numpy(), wgrad, atol=GRAD_ATOL)
    exact, _ = o.graphloss([g, g, g], lp, segs, None, "none")
    want3, _ = o.graphloss([g, g, g], lp, segs, 3.0, "none")
    assert np.any(want3 > exact + 1e-3)     # the narrow beam really removed mass


def test_output_beam_fused_entry_point_needs_the_flag():
    import ctypes as C
    from waterfall_amd import _lib
    c = util.random_case(62, "ctc", B=2, T=20, L=3, P=5)
    lib = _lib.lib()
    x = torch.from_numpy(c["lp"]).to(DEV)
    segs = c["segs"]
    gh, gd = c["fsa"].blob_host(), c["fsa"].blob_dev(torch.device(DEV))
    n = C.c_size_t(0)
    npp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)
    _lib.check(lib.wf_fb_workspace_bytes(npp(gh), npp(c["fsa"].utt2fsa), npp(segs), 2, 2, x.shape[2], _lib.WF_PRUNED, 1, C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=DEV)
    tot = torch.empty(2, dtype=torch.float32, device=DEV)
    grad = torch.empty_like(x)
    args = lambda beam, flags: (npp(gh), C.c_void_p(gd.data_ptr()), npp(c["fsa"].utt2fsa), C.c_void_p(x.data_ptr()), x.stride(0),
                                x.stride(1), 2, x.shape[1], x.shape[2], npp(segs), 2, beam, flags, None,
                                C.c_void_p(tot.data_ptr()), C.c_void_p(grad.data_ptr()), C.c_void_p(ws.data_ptr()), n.value, None)
    assert lib.wf_fb_logsum(*args(5.0, 0)) == -1            # WF_EINVAL: beam without WF_PRUNED
    assert lib.wf_fb_logsum(*args(0.0, _lib.WF_PRUNED)) == -1
    _lib.check(lib.wf_fb_logsum(*args(5.0, _lib.WF_PRUNED)))
    torch.cuda.synchronize()
    want, wgrad = o.graphloss(c["ographs"], c["lp"], segs, 5.0, "none")
    np.testing.assert_allclose(-tot.cpu().numpy(), want, rtol=TOT_RTOL, atol=1e-3)
    np.testing.assert_allclose(-grad.cpu().numpy(), wgrad, atol=GRAD_ATOL)


# ---- waterfall/manual_ctc drop-in (ctc.py ctc_loss, softmax_ctc.py ctc_softmax, prior.py compute_align) ----

def _manual_ctc_inputs(z):
    """(trans, transform, trans_lengths) padded as deprecated/datapipe_manual_ctc.py:245-281 pads them."""
    from oracle import graphs as og
    V = z["log_probs"].shape[-1]
    off = z["target_off"]
    mats = [og.ctc_trans_transform(z["targets"][off[i]:off[i + 1]].tolist(), V) for i in range(len(off) - 1)]
    S = max(m[0].shape[0] for m in mats)
    B = len(mats)
    trans = np.full((B, S, S), og.FMIN, dtype=np.float32)
    transform = np.full((B, S, V), og.FMIN, dtype=np.float32)
    lens = np.zeros(B, dtype=np.int64)
    for b, (t, f) in enumerate(mats):
        s = t.shape[0]
        trans[b, :s, :s] = t
        transform[b, :s] = f
        lens[b] = s
    dev = torch.device(DEV)
    return (torch.from_numpy(trans).to(dev), torch.from_numpy(transform).to(dev), torch.from_numpy(lens).to(dev))


@pytest.mark.parametrize("name", ["manual_ctc_ctc_small.npz", "manual_ctc_ctc_medium.npz"])
def test_manual_ctc_dropin_against_reference_goldens(name):
    from waterfall_amd import manual_ctc
    z = util.load_golden(name)
    trans, transform, lens = _manual_ctc_inputs(z)
    il = torch.from_numpy(z["input_lengths"])
    x = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    loss = manual_ctc.ctc_loss(x, il, trans, lens, transform, reduction="sum")
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), z["grad"], atol=GRAD_ATOL)
    none = manual_ctc.ctc_loss(x.detach(), il, trans, lens, transform, reduction="none")
    np.testing.assert_allclose(none.cpu().numpy(), z["loss_none"], rtol=TOT_RTOL)
    mean = manual_ctc.ctc_loss(x.detach(), il, trans, lens, transform, reduction="mean")
    np.testing.assert_allclose(mean.item(), np.mean(z["loss_none"] / lens.cpu().numpy()), rtol=TOT_RTOL)
    # eta-semiring (softmax_ctc.py), reference argument order: (..., trans, transform, trans_lengths, eta)
    x2 = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    l2 = manual_ctc.ctc_softmax(x2, il, trans, transform, lens, eta=float(z["eta"]), reduction="sum")
    l2.backward()
    np.testing.assert_allclose(l2.item(), z["eta_loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x2.grad.cpu().numpy(), z["eta_grad"], atol=GRAD_ATOL)
    # posterior alignment (prior.py): compare as probabilities where the reference's entries are defined
    al = manual_ctc.compute_align(x.detach(), il, trans, lens, transform).cpu().numpy()
    ref = z["align"]
    for b in range(ref.shape[0]):
        T, S = int(z["input_lengths"][b]), int(lens[b])
        np.testing.assert_allclose(np.exp(al[b, :T, :S]), np.exp(ref[b, :T, :S]), atol=GRAD_ATOL)
        np.testing.assert_allclose(np.exp(al[b, :T, :S]).sum(-1), 1.0, atol=2e-3)
        assert np.all(np.isneginf(al[b, T:]))


@pytest.mark.parametrize("topo", ["2state", "mmictc", "3state-skip-2"])
def test_manual_ctc_dropin_arc_state_encodings(topo):
    # the same matrices the golden generator fed to the reference's ctc_loss (one machine state per arc,
    # non-emitting virtual start / end rows)
    from oracle import graphs as og
    from waterfall_amd import manual_ctc
    z = util.load_golden(f"manual_ctc_{topo}.npz")
    V = z["log_probs"].shape[-1]
    mats = [og.graph_to_trans_transform(g, V) for g in util.golden_graphs(z)]
    S = max(m[0].shape[0] for m in mats)
    B = len(mats)
    trans = np.full((B, S, S), og.FMIN, dtype=np.float32)
    transform = np.full((B, S, V), og.FMIN, dtype=np.float32)
    lens = np.zeros(B, dtype=np.int64)
    for b, (t, f) in enumerate(mats):
        s = t.shape[0]
        trans[b, :s, :s] = t
        transform[b, :s] = f
        lens[b] = s
    x = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    loss = manual_ctc.ctc_loss(x, torch.from_numpy(z["input_lengths"]), torch.from_numpy(trans).to(DEV),
                               torch.from_numpy(lens).to(DEV), torch.from_numpy(transform).to(DEV), reduction="sum")
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), z["grad"], atol=GRAD_ATOL)


# ---- graphs beyond the LDS-resident sweeps (decoding graphs): one launch per frame ----

def _big_case(S, B, T, V=301, seed=5):
    g = util.synth_decoding_graph(seed, S, V)
    gen = torch.Generator().manual_seed(seed)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 3.0, -1).numpy()
    lens = [T, max(1, T - 7), max(1, T // 2), 1][:B] + [T] * max(0, B - 4)
    segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), np.array(lens[:B])], 1).astype(np.int32)
    return g, lp, segs


@pytest.mark.parametrize("S", [12_001, 40_000])
def test_big_graph_viterbi_bit_exact(oracle_lib, S):
    g, lp, segs = _big_case(S, B=4, T=40)
    arcs_ = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    one = wf.Fsa(arcs_, g["aux"], [0, arcs_.shape[0]], [g["num_states"]])
    fsa = wf.create_fsa_vec([one] * 4)                 # one decoding graph shared by the batch
    assert fsa.utt2fsa is not None
    dense = wf.DenseFsaVec(torch.from_numpy(lp).to(DEV), torch.from_numpy(segs))
    out = wf.best_path(fsa, dense)
    scores, arcs = out["scores"].cpu().numpy(), out["arcs"].cpu().numpy()
    trop = wf.tot_scores(fsa, dense, log_semiring=False).cpu().numpy()
    for i in range(4):
        T = int(segs[i, 2])
        s, path = util.c_oracle_viterbi(oracle_lib, g, lp[i], T)
        assert scores[i] == s and trop[i] == s
        assert list(arcs[i, :T + 1]) == list(path)
        assert np.all(arcs[i, T + 1:] == -1)
    s_np, path_np = o.viterbi(g, lp[2], int(segs[2, 2]))     # numpy restatement agrees with the C one
    assert s_np == scores[2] and list(path_np) == list(arcs[2, :int(segs[2, 2]) + 1])


def test_big_graph_tot_scores_log_semiring_and_no_grad_loss():
    g, lp, segs = _big_case(15_000, B=3, T=30)
    fsa = util.graphs_to_fsa([g, g, g])                 # per-utterance graphs this time
    dense = wf.DenseFsaVec(torch.from_numpy(lp).to(DEV), torch.from_numpy(segs))
    want = np.array([o.forward(g, lp[i], int(segs[i, 2]))[1] for i in range(3)])
    t32 = wf.tot_scores(fsa, dense, True, False).cpu().numpy()
    t64 = wf.tot_scores(fsa, dense, True, True).cpu().numpy()
    np.testing.assert_allclose(t32, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(t64, want, rtol=1e-9)
    with torch.no_grad():
        loss = wf.graphloss(fsa, dense, reduction="none").cpu().numpy()
    np.testing.assert_allclose(loss, -want, rtol=TOT_RTOL, atol=TOT_ATOL)
    v64 = wf.tot_scores(fsa, dense, False, True).cpu().numpy()
    for i in range(3):
        s, _ = o.viterbi(g, lp[i], int(segs[i, 2]), dtype=np.float64)
        np.testing.assert_allclose(v64[i], s, rtol=1e-12)


def test_big_graph_loss_and_gradient_against_oracle():
    g, lp, segs = _big_case(15_000, B=3, T=30)
    fsa = util.graphs_to_fsa([g, g, g])
    x = torch.from_numpy(lp).to(DEV).requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(segs))
    loss = wf.graphloss(fsa, dense, reduction="none")
    loss.sum().backward()
    want, wgrad = o.graphloss([g, g, g], lp, segs, None, "none")
    np.testing.assert_allclose(loss.detach().cpu().numpy(), want, rtol=TOT_RTOL, atol=TOT_ATOL)
    grad = x.grad.cpu().numpy()
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    # occupation counts: every frame of every utterance carries posterior mass 1
    for i in range(3):
        T = int(segs[i, 2])
        np.testing.assert_allclose(-grad[i, :T].sum(-1), 1.0, rtol=1e-4)
        assert np.all(grad[i, T:] == 0)


@pytest.mark.parametrize("post_big", [0, 1])
@pytest.mark.parametrize("topo,den", [("ctc", False), ("mmictc", True), ("3state-skip", True), ("2state_sil", False)])
def test_per_frame_launch_sweep_forced_on_the_training_topologies(topo, den, post_big):
    c = util.random_case(61, topo, B=4, T=37, L=6, P=9, den=den)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")

    def run():
        x = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
        dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
        loss = wf.graphloss(c["fsa"], dense, reduction="none")
        loss.sum().backward()
        best = wf.best_path(c["fsa"], dense)
        return loss.detach().cpu().numpy(), x.grad.cpu().numpy(), best

    loss, grad, best = _with_env({"WF_BIG_MIN_S": "1", "WF_POST_BIG_MIN_S": "1" if post_big else "1000000"}, run)
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    base = c["fsa"].fsa_arc_offsets
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path = o.viterbi(g, c["lp"][i], T)
        gi = 0 if c["fsa"].utt2fsa is not None else i
        assert best["scores"][i].item() == s
        assert list(best["arcs"][i, :T + 1].cpu().numpy() - base[gi]) == path


def test_big_graph_shared_by_the_batch_two_groups_all_semirings():
    # 11 utterances on one decoding graph: two groups of kNB = 8 (the second partly filled), ragged lengths
    S, B, T = 12_001, 11, 26
    g, lp, _ = _big_case(S, B=B, T=T, seed=9)
    lens = np.array([T, T - 1, T - 3, 20, 17, 13, 9, 5, 2, 1, T], dtype=np.int64)
    segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), lens], 1).astype(np.int32)
    arcs_ = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    fsa = wf.create_fsa_vec([wf.Fsa(arcs_, g["aux"], [0, arcs_.shape[0]], [g["num_states"]])] * B)
    dense = wf.DenseFsaVec(torch.from_numpy(lp).to(DEV), torch.from_numpy(segs))

    def run():
        out = wf.best_path(fsa, dense)
        return (out["scores"].cpu().numpy(), out["arcs"].cpu().numpy(),
                wf.tot_scores(fsa, dense, log_semiring=False).cpu().numpy(),
                wf.tot_scores(fsa, dense, log_semiring=True).cpu().numpy(),
                wf.tot_scores(fsa, dense, True, True).cpu().numpy())

    sc, arcs, trop, logt, log64 = run()
    for i in range(B):
        Ti = int(lens[i])
        s, path = o.viterbi(g, lp[i], Ti)
        assert sc[i] == s and trop[i] == s
        if len(path):
            assert list(arcs[i, :Ti + 1]) == list(path) and np.all(arcs[i, Ti + 1:] == -1)
        else:                                   # no accepting path in so few frames
            assert s == -np.inf and np.all(arcs[i] == -1)
        want = o.forward(g, lp[i], Ti)[1]
        np.testing.assert_allclose(logt[i], want, rtol=TOT_RTOL, atol=TOT_ATOL)
        np.testing.assert_allclose(log64[i], want, rtol=1e-9)
    # one utterance per lane (the path of batches with different graphs): identical numbers
    sc2, arcs2, trop2, logt2, log642 = _with_env({"WF_NO_SHARED_BIG": "1"}, run)
    assert np.array_equal(sc, sc2) and np.array_equal(arcs, arcs2) and np.array_equal(trop, trop2)
    assert np.array_equal(logt, logt2) and np.array_equal(log64, log642)


def test_lattice_analysis_call_shapes_on_a_decoding_graph():
    # utils/lattice_analysis.py:120-135 with the same names; a word-labelled decoding graph (aux = word ids)
    g, lp, segs = _big_case(12_001, B=2, T=33, seed=21)
    rng = np.random.default_rng(3)
    g["aux"] = np.where(g["label"] >= 0, np.where(rng.random(g["label"].size) < 0.2, rng.integers(1, 50, g["label"].size), 0),
                        -1).astype(np.int32)
    arcs_ = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    TLG = wf.Fsa(arcs_, g["aux"], [0, arcs_.shape[0]], [g["num_states"]])
    x = torch.from_numpy(lp).to(DEV)
    lens = torch.tensor([33, 20])
    # (an unbounded search here: the lattice is the whole product; finite beams are tests/test_search.py's subject)
    lattice = wf.get_lattice(x, lens, TLG, search_beam=float("inf"), output_beam=8, min_active_states=0,
                             max_active_states=0)
    tot = lattice.get_tot_scores(True, True).cpu().numpy()
    best = lattice.get_tot_scores(True, False).cpu().numpy()
    one_best = wf.one_best_decoding(lattice)
    words = wf.get_aux_labels(one_best)
    for i, T in enumerate([33, 20]):
        s32, path32 = o.viterbi(g, lp[i], T)
        assert best[i] == float(s32)          # the search's tropical arithmetic is fp32 (k2's lattice arc scores are fp32)
        # (:128: the total of the lattice that output_beam = 8 leaves -- oracle.lattice_search states the rule)
        np.testing.assert_allclose(tot[i], o.lattice_search(g, lp[i], T, np.inf, 8.0)[2], rtol=1e-9)
        assert one_best.labels[i] == [int(g["label"][a]) for a in path32] and one_best.labels[i][-1] == -1
        assert words[i] == [int(g["aux"][a]) for a in path32 if g["aux"][a] > 0]
        assert one_best.get_tot_scores(True, False)[i].item() == float(s32)
    dense = wf.DenseFsaVec(x, torch.tensor([[0, 0, 33], [1, 0, 20]], dtype=torch.int32))
    again = wf.intersect_dense(TLG, dense, 10).get_tot_scores(True, True).cpu().numpy()   # the exact sweep
    whole = wf.get_lattice(x, lens, TLG, search_beam=float("inf"), output_beam=float("inf"), min_active_states=0,
                           max_active_states=0).get_tot_scores(True, True).cpu().numpy()
    np.testing.assert_allclose(again, whole, rtol=1e-9)              # nothing pruned: the whole product
    assert np.all(tot <= whole + 1e-9) and np.all(tot >= best - 1e-4)


def test_loss_stats_kernel_matches_the_elementwise_formula():
    from waterfall_amd.dist import fused_loss_stats
    rng = np.random.default_rng(4)
    for B in (1, 7, 300, 1000):
        t = rng.normal(-50, 20, size=B).astype(np.float32)
        t[rng.random(B) < 0.1] = -np.inf
        x = torch.from_numpy(t).to(DEV)
        out = fused_loss_stats(x, 1234).cpu().numpy()
        ok = np.isfinite(t)
        np.testing.assert_allclose(out[0], -t[ok].astype(np.float64).sum(), rtol=1e-5)
        assert out[1] == 1234 and out[2] == (~ok).sum()


def test_autograd_split_prezero_and_second_backward():
    # forward zero-fills the gradient buffer beside the sweeps (WF_PREZERO), backward only scatters; a second backward
    # through the retained graph takes a fresh buffer; a view with padded strides takes the strided fill
    c = util.random_case(71, "ctc", B=3, T=40, L=6, P=30)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "sum")
    V = c["V"]
    big = torch.zeros(3, 40, V + 5)
    big[:, :, :V] = torch.from_numpy(c["lp"])
    for x in (torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True), big.to(DEV)[:, :, :V].requires_grad_(True)):
        dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
        loss = wf.graphloss(c["fsa"], dense, reduction="sum")
        loss.backward(retain_graph=True)
        g1 = x.grad.clone()
        x.grad = None
        (2.0 * loss).backward()
        np.testing.assert_allclose(loss.item(), want, rtol=TOT_RTOL)
        np.testing.assert_allclose(g1.cpu().numpy(), wgrad, atol=GRAD_ATOL)
        np.testing.assert_allclose(x.grad.cpu().numpy(), 2 * wgrad, atol=2 * GRAD_ATOL)


# ---- round 2: reference-generated fixtures at realistic sizes; cfg4 / cfg5 at the benched sizes -------

def _segs_of(z):
    B = z["log_probs"].shape[0]
    return np.stack([np.arange(B), np.zeros(B), z["input_lengths"]], 1).astype(np.int32)


def _golden_fsa(z):
    if "g_src" in z:
        return util.graphs_to_fsa(util.golden_graphs(z))
    V = z["log_probs"].shape[-1]
    topo = wf.Topology("ctc", util.ctc_phones(V)) if V > 300 else wf.build_topology("ctc", util.ctc_phones(V))[0]
    off = z["target_off"]
    return wf.compile_ctc_k2_graphs(topo, [z["targets"][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)])


@pytest.mark.parametrize("name", ["manual_ctc_ctc_cfg2.npz", "manual_ctc_2state_cfg2.npz", "manual_ctc_ctc_v5001.npz"])
def test_golden_config_sizes_from_reference_manual_ctc(name):
    # the reference's own dense [B,T,S,V] forward-backward (fp32) at T = 500, L = 80 and at V = 5001
    z = util.load_golden(name)
    fsa = _golden_fsa(z)
    loss, grad = run_hip(fsa, z["log_probs"], _segs_of(z), "none")
    np.testing.assert_allclose(loss, z["loss_none"], rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, z["grad"], atol=GRAD_ATOL)
    np.testing.assert_allclose(loss.sum(), z["loss_sum"], rtol=TOT_RTOL)
    tot, fgrad = fused_logsum(fsa, z["log_probs"], _segs_of(z))
    np.testing.assert_allclose(-tot, z["loss_none"], rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(fgrad, z["grad"], atol=GRAD_ATOL)


@pytest.mark.parametrize("name", ["manual_ctc_ctc_cfg2.npz", "manual_ctc_ctc_v5001.npz"])
def test_manual_ctc_dropin_at_config_sizes(name):
    from waterfall_amd import manual_ctc
    z = util.load_golden(name)
    trans, transform, lens = _manual_ctc_inputs(z)
    il = torch.from_numpy(z["input_lengths"])
    x = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    loss = manual_ctc.ctc_loss(x, il, trans, lens, transform, reduction="sum")
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), z["grad"], atol=GRAD_ATOL)
    x2 = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    l2 = manual_ctc.ctc_softmax(x2, il, trans, transform, lens, eta=float(z["eta"]), reduction="sum")
    l2.backward()
    np.testing.assert_allclose(l2.item(), z["eta_loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x2.grad.cpu().numpy(), z["eta_grad"], atol=GRAD_ATOL)
    al = manual_ctc.compute_align(x.detach(), il, trans, lens, transform).cpu().numpy()
    for b in range(z["align"].shape[0]):        # the fixture keeps the alignment of the first utterance(s)
        T, S = int(z["input_lengths"][b]), int(lens[b])
        np.testing.assert_allclose(np.exp(al[b, :T, :S]), np.exp(z["align"][b, :T, :S]), atol=GRAD_ATOL)


@pytest.mark.parametrize("topo", ["2state", "mmictc", "3state-skip", "2state_sil3"])
def test_manual_ctc_dropin_eta_and_alignment_on_topologies(topo):
    # softmax_ctc.py / prior.py outputs of the reference on composed non-CTC graphs (one machine state per arc)
    from oracle import graphs as og
    from waterfall_amd import manual_ctc
    z = util.load_golden(f"manual_ctc_{topo}_eta_align.npz")
    V = z["log_probs"].shape[-1]
    mats = [og.graph_to_trans_transform(g, V) for g in util.golden_graphs(z)]
    S = max(m[0].shape[0] for m in mats)
    B = len(mats)
    trans = np.full((B, S, S), og.FMIN, dtype=np.float32)
    transform = np.full((B, S, V), og.FMIN, dtype=np.float32)
    lens = np.zeros(B, dtype=np.int64)
    for b, (t, f) in enumerate(mats):
        s = t.shape[0]
        trans[b, :s, :s] = t
        transform[b, :s] = f
        lens[b] = s
    trans, transform, lens_t = torch.from_numpy(trans).to(DEV), torch.from_numpy(transform).to(DEV), torch.from_numpy(lens).to(DEV)
    il = torch.from_numpy(z["input_lengths"])
    x = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    loss = manual_ctc.ctc_loss(x, il, trans, lens_t, transform, reduction="sum")
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), z["grad"], atol=GRAD_ATOL)
    x2 = torch.from_numpy(z["log_probs"]).to(DEV).requires_grad_(True)
    l2 = manual_ctc.ctc_softmax(x2, il, trans, transform, lens_t, eta=float(z["eta"]), reduction="sum")
    l2.backward()
    np.testing.assert_allclose(l2.item(), z["eta_loss_sum"], rtol=TOT_RTOL)
    np.testing.assert_allclose(x2.grad.cpu().numpy(), z["eta_grad"], atol=GRAD_ATOL)
    al = manual_ctc.compute_align(x.detach(), il, trans, lens_t, transform).cpu().numpy()
    for b in range(B):
        T, Sb = int(z["input_lengths"][b]), int(lens[b])
        # rows 1 .. Sb-2 are the arcs; the virtual start / end rows never hold mass after a frame
        np.testing.assert_allclose(np.exp(al[b, :T, 1:Sb - 1]), np.exp(z["align"][b, :T, 1:Sb - 1]), atol=GRAD_ATOL)
    # the graph itself (not its arc encoding) through graphloss
    loss_g, grad_g = run_hip(util.graphs_to_fsa(util.golden_graphs(z)), z["log_probs"], _segs_of(z), "none")
    np.testing.assert_allclose(loss_g, z["loss_none"], rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad_g, z["grad"], atol=GRAD_ATOL)


def test_full_size_3state_skip_denominator_properties(oracle_lib):
    # cfg4 (LS BPE-5k 3state-skip denominator: V=15001, S=10002, A=25002, fan-in(0)=10001), reduced batch / frames
    c = util.random_case(778, "3state-skip", B=4, T=120, L=8, P=5000, den=True)
    g0 = c["ographs"][0]
    assert c["V"] == 15001 and g0["num_states"] == 10002 and int((g0["label"] >= 0).sum()) == 25001
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    for b in range(4):
        T = int(c["segs"][b, 2])
        np.testing.assert_allclose(grad[b, :T].sum(-1), -1.0, atol=2e-3)
        assert np.all(grad[b, T:] == 0)
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, c["ographs"][:2], c["lp"][:2], c["segs"][:2], gscale=[-1.0] * 2,
                                        use_double=True, nthreads=2)
    np.testing.assert_allclose(loss[:2], -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad[:2], grad_d, atol=GRAD_ATOL)
    tot_f, grad_f = fused_logsum(c["fsa"], c["lp"], c["segs"])
    assert LAST_FALLBACKS[0] == 0          # sweep_hub_kernel (split ring at this V) delivered the rows itself
    np.testing.assert_allclose(-tot_f[:2], -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad_f[:2], grad_d, atol=GRAD_ATOL)
    run = lambda: run_hip(c["fsa"], c["lp"], c["segs"], "none")
    lg, gg = _with_env({"WF_NO_LINBIG_SWEEP": "1"}, run)          # log-semiring sweep on the same batch
    np.testing.assert_allclose(lg, loss, rtol=1e-6)
    np.testing.assert_allclose(gg, grad, atol=2e-5)
    l5, g5 = _with_env({"WF_FORCE_FALLBACK": "1"}, run)
    np.testing.assert_allclose(l5, loss, rtol=1e-6)
    np.testing.assert_allclose(g5, grad, atol=2e-5)
    fwd = lambda: run_hip(c["fsa"], c["lp"], c["segs"], "none", need_grad=False)[0]
    assert np.array_equal(fwd(), loss)


def test_full_size_3state_skip_numerator_against_c_oracle(oracle_lib):
    # cfg4's numerator as bench.py times it (V=15001, L=64, T=500), 8 of the 32 utterances
    c = _full_size("3state-skip", 5000, 64, False, B=8)
    assert c["V"] == 15001
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, c["ographs"], c["lp"], c["segs"], gscale=[-1.0] * 8,
                                        use_double=True, nthreads=8)
    np.testing.assert_allclose(loss, -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, grad_d, atol=GRAD_ATOL)
    tot, fgrad = fused_logsum(c["fsa"], c["lp"], c["segs"])
    np.testing.assert_allclose(-tot, -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(fgrad, grad_d, atol=GRAD_ATOL)


def test_full_size_mmictc_numerator_against_c_oracle(oracle_lib):
    c = _full_size("mmictc", 5000, 64, False, B=8)
    assert c["V"] == 10001
    loss, grad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, c["ographs"], c["lp"], c["segs"], gscale=[-1.0] * 8,
                                        use_double=True, nthreads=8)
    np.testing.assert_allclose(loss, -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, grad_d, atol=GRAD_ATOL)


@pytest.mark.parametrize("S,T", [(100_000, 500), (1_000_000, 48)])
def test_cfg5_benched_workload_bit_exact_against_c_oracle(oracle_lib, S, T):
    # bench.py's cfg5 workload (same generator, seed, B = 8, V = 5001); S = 1e5 at the benched T = 500, S = 1e6 at reduced T
    from waterfall_amd.synth import best_path_checksum, cfg5_workload
    B = 8
    g, fsa, lp, dense = cfg5_workload(torch.device(DEV), S, B, T, 5001)
    out = wf.best_path(fsa, dense)
    scores, arcs = out["scores"].cpu().numpy(), out["arcs"].cpu().numpy()
    trop = wf.tot_scores(fsa, dense, log_semiring=False).cpu().numpy()
    lp_h = lp.cpu().numpy()
    want_s, want_a = np.zeros(B, dtype=np.float32), np.full((B, T + 1), -1, dtype=np.int32)
    for i in range(B):
        s, path = util.c_oracle_viterbi(oracle_lib, g, lp_h[i], T)
        want_s[i] = s
        want_a[i, :len(path)] = path
    assert np.isfinite(want_s).all()
    assert np.array_equal(scores, want_s) and np.array_equal(trop, want_s)
    assert np.array_equal(arcs, want_a)
    assert best_path_checksum(scores, arcs) == best_path_checksum(want_s, want_a)


# ---- hub form: topologies used as denominators over a large table (sweep_hub_kernel / post_hub_kernel) ----------------

HUB_TOPOS = ["2state", "2state-1", "3state-skip", "3state-skip-1", "3state-skip-2", "mmictc", "mmictc-1"]
HUB_PATHS = {
    "hub": {"WF_NO_LIN_SWEEP": "2"},                                          # (below 1024 states the lane-per-state sweep would take it)
    "hub_check_kernel": {"WF_NO_LIN_SWEEP": "2", "WF_NO_POST_CHECK": "1"},
    "hub_then_fallback": {"WF_NO_LIN_SWEEP": "2", "WF_FORCE_FALLBACK": "1"},
    "hub_check_kernel_then_fallback": {"WF_NO_LIN_SWEEP": "2", "WF_NO_POST_CHECK": "1", "WF_FORCE_FALLBACK": "1"},
    "linbig": {"WF_NO_LIN_SWEEP": "2", "WF_NO_HUB_SWEEP": "1"},
    # the default cuts the sweeps in four launches (5/8 + 3 x 1/8 of the steps) with the posterior kernel of the ready frames
    # beside the next one; one launch + the posterior kernel after it, and another cut (utterances below 32 frames are never cut)
    "hub_single_launch": {"WF_NO_LIN_SWEEP": "2", "WF_NO_HUB_OVERLAP": "1"},
    "hub_cut_in_sevenths": {"WF_NO_LIN_SWEEP": "2", "WF_HUB_SEGS": "7:4,5,6,7", "WF_HUB_POST_LDS": "0"},
}


@pytest.mark.parametrize("path", sorted(HUB_PATHS))
@pytest.mark.parametrize("topo", HUB_TOPOS)
def test_hub_form_denominators(topo, path):
    from waterfall_amd.csrc_layout import HDR_WORDS, INFO_WORDS
    c = util.random_case(600 + len(topo), topo, B=4, T=45, L=4, P=300, den=True)
    info = c["fsa"].blob_host()[HDR_WORDS:HDR_WORDS + INFO_WORDS]
    assert info[46] in (2, 3) and info[47] == 300          # WF_I_HUB_G / WF_I_HUB_NG: the hub form was found
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")

    def run():
        tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"])
        loss, agrad = run_hip(c["fsa"], c["lp"], c["segs"], "none")
        with torch.no_grad():
            fwd, _ = run_hip(c["fsa"], c["lp"], c["segs"], "none", need_grad=False)
        return tot, grad, loss, agrad, fwd
    tot, grad, loss, agrad, fwd = _with_env(HUB_PATHS[path], run)
    nfb = _with_env(HUB_PATHS[path], lambda: (fused_logsum(c["fsa"], c["lp"], c["segs"]), LAST_FALLBACKS[0])[1])
    assert nfb == (4 if "fallback" in path else 0)          # the hub / linbig kernels' own rows passed the check
    for got_loss, got_grad in ((-tot, grad), (loss, agrad)):
        np.testing.assert_allclose(got_loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
        np.testing.assert_allclose(got_grad, wgrad, atol=GRAD_ATOL)
    np.testing.assert_allclose(fwd, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    for b in range(4):
        T = int(c["segs"][b, 2])
        np.testing.assert_allclose(grad[b, :T].sum(-1), -1.0, atol=2e-3)
        assert np.all(grad[b, T:] == 0)


@pytest.mark.parametrize("P", [900, 1700])
@pytest.mark.parametrize("topo", HUB_TOPOS)
def test_hub_split_ring(topo, P):
    """sweep_hub_kernel's split ring (rows too long for three LDS buffers: cfg4's 3state-skip, V = 15001) forced at a
    size the numpy oracle handles: 2 - 3 slots per lane, the last one partly filled and read from the back buffers."""
    c = util.random_case(700 + len(topo), topo, B=3, T=33, L=4, P=P, den=True)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    run = lambda: fused_logsum(c["fsa"], c["lp"], c["segs"])
    tot, grad = _with_env({"WF_HUB_FORCE_SPLIT": "1"}, run)
    assert LAST_FALLBACKS[0] == 0
    np.testing.assert_allclose(-tot, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    tot2, grad2 = _with_env({"WF_HUB_FORCE_SPLIT": "1", "WF_NO_HUB_SPLIT": "1"}, run)      # two whole-row buffers
    assert np.array_equal(tot, tot2) and np.array_equal(grad, grad2)
    tot3, grad3 = run()                                                                    # three whole-row buffers
    assert np.array_equal(tot, tot3) and np.array_equal(grad, grad3)


def test_label_grad_sums_kernel_matches_torch():
    """wf_label_grad_sums (the optional tail of the fused all-reduce buffer) on a strided gradient box."""
    from waterfall_amd.dist import fused_loss_stats
    g = torch.Generator().manual_seed(12)
    for N, T, V in ((3, 17, 5), (2, 90, 1025), (5, 41, 5001)):
        big = torch.randn(N, T + 3, V, generator=g).to(DEV)
        grad = big[:, 1:T + 1]                      # a view: rows strided, columns contiguous
        tot = torch.tensor([-1.0, -2.5, float("-inf")][:min(N, 3)], device=DEV)
        out = fused_loss_stats(tot, 77, grad=grad).cpu().numpy()
        ref = grad.double().sum((0, 1)).cpu().numpy()
        np.testing.assert_allclose(out[3:], ref, rtol=1e-4, atol=1e-4)
        assert out[1] == 77


@pytest.mark.parametrize("reduction", ["sum", "mean", "none"])
@pytest.mark.parametrize("topo,P", [("2state", 9), ("mmictc", 10), ("3state-skip", 300), ("mmictc", 300), ("2state_sil", 8)])
def test_graphloss_num_den_equals_the_two_call_form(topo, P, reduction):
    """graphloss_num_den (one autograd node, one gradient box: the denominator's call writes it, the numerator's adds
    its K entries per row with WF_ACCUMULATE) against graphloss(num) - graphloss(den) (rnnp.py:64-104) and the oracle;
    P = 300: the denominator takes the hub kernels."""
    c = util.random_case(900 + P, topo, B=4, T=37, L=5, P=P)
    den = wf.create_fsa_vec([c["topo"]] * 4)
    tl = torch.tensor([len(s) for s in c["seqs"]], dtype=torch.float32)
    seg = torch.from_numpy(c["segs"])
    up = torch.tensor([0.5, -2.0, 1.0, 3.0], device=DEV)

    def run(fn):
        x = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
        loss = fn(wf.DenseFsaVec(x, seg))
        (loss * up).sum().backward() if reduction == "none" else (loss * 0.7).backward()
        return loss.detach().cpu().numpy(), x.grad.cpu().numpy()
    two = lambda d: (wf.graphloss(c["fsa"], d, reduction=reduction, target_lengths=tl)
                     - wf.graphloss(den, d, reduction=reduction, target_lengths=tl))
    one = lambda d: wf.graphloss_num_den(c["fsa"], den, d, reduction=reduction, target_lengths=tl)
    l2, g2 = run(two)
    l1, g1 = run(one)
    np.testing.assert_allclose(l1, l2, rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(g1, g2, atol=2e-5)
    wf.set_num_den_overlap(True)     # the numerator's sweeps on a side stream (wf_fb_logsum(WF_KEEP_AB) + wf_fb_backward(WF_ACCUMULATE))
    try:
        l3, g3 = run(one)
    finally:
        wf.set_num_den_overlap(False)
    assert np.array_equal(l3, l1) and np.array_equal(g3, g1)
    # and against the oracle (sum): d loss / d lp = - posterior(num) + posterior(den)
    if reduction == "sum":
        wn, gn = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "sum")
        wd, gd = o.graphloss(den.as_oracle_graphs(), c["lp"], c["segs"], None, "sum")
        np.testing.assert_allclose(l1, wn - wd, rtol=TOT_RTOL, atol=TOT_ATOL)
        np.testing.assert_allclose(g1, 0.7 * (gn - gd), atol=GRAD_ATOL)


def test_accumulate_flag_limits():
    """WF_ACCUMULATE through the C ABI: refused (WF_EUNSUPPORTED, already by wf_fb_workspace_bytes) for graphs beyond
    post_small_kernel's reach, and with WF_PRUNED / a missing gradient."""
    from waterfall_amd import _lib
    c = util.random_case(31, "mmictc", B=2, T=20, L=3, P=300, den=True)          # a hub-form denominator: not accepted
    assert not c["fsa"].accumulate_ok(c["segs"], 2, c["V"])
    n = util.random_case(32, "mmictc", B=2, T=20, L=3, P=300)
    assert n["fsa"].accumulate_ok(n["segs"], 2, n["V"])


def test_hub_form_with_more_self_loop_labels_than_compute_lanes():
    """A hub-form graph whose hub has 901 self-loop labels outside the groups (more than sweep_hub_kernel's 768 compute
    lanes take one each): the 2state topology over 400 phones plus 900 filler labels looping on state 0."""
    from waterfall_amd.csrc_layout import HDR_WORDS, INFO_WORDS
    c = util.random_case(41, "2state", B=3, T=31, L=4, P=400, den=True)
    g = c["ographs"][0]
    V0 = c["V"]
    extra = 900
    k = 1                                               # after the hub's own first arc (its blank self-loop)
    ins = lambda a, vals: np.concatenate([a[:k], vals, a[k:]])
    g2 = {"src": ins(g["src"], np.zeros(extra, np.int32)), "dst": ins(g["dst"], np.zeros(extra, np.int32)),
          "label": ins(g["label"], np.arange(V0, V0 + extra, dtype=np.int32)),
          "score": ins(g["score"], np.zeros(extra, np.float32)), "aux": ins(g["aux"], np.arange(V0, V0 + extra, dtype=np.int32)),
          "num_states": g["num_states"]}
    V = V0 + extra
    gen = torch.Generator().manual_seed(5)
    lp = torch.log_softmax(torch.randn(3, 31, V, generator=gen) * 3.0, -1).numpy()
    fsa = util.graphs_to_fsa([g2] * 3)
    info = fsa.blob_host()[HDR_WORDS:HDR_WORDS + INFO_WORDS]
    assert info[46] == 2 and info[47] == 400 and info[52] == extra + 1          # WF_I_HUB_G / NG / NX
    want, wgrad = o.graphloss([g2] * 3, lp, c["segs"], None, "none")
    tot, grad = _with_env({"WF_NO_LIN_SWEEP": "2"}, lambda: fused_logsum(fsa, lp, c["segs"]))
    assert LAST_FALLBACKS[0] == 0
    np.testing.assert_allclose(-tot, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)


def test_graphloss_num_den_falls_back_to_the_two_call_form():
    """Where the fused pair does not apply -- no gradient wanted, the "pruned" beam mode -- graphloss_num_den is the
    two-call expression itself."""
    c = util.random_case(61, "2state", B=3, T=30, L=4, P=7)
    den = wf.create_fsa_vec([c["topo"]] * 3)
    seg = torch.from_numpy(c["segs"])
    x = torch.from_numpy(c["lp"]).to(DEV)
    with torch.no_grad():
        d = wf.DenseFsaVec(x, seg)
        a = wf.graphloss_num_den(c["fsa"], den, d, reduction="none")
        b = wf.graphloss(c["fsa"], d, reduction="none") - wf.graphloss(den, d, reduction="none")
    assert torch.equal(a, b) and not a.requires_grad
    wf.set_output_beam_mode("pruned")
    try:
        res = []
        for fn in (lambda d: wf.graphloss_num_den(c["fsa"], den, d, output_beam=6.0),
                   lambda d: wf.graphloss(c["fsa"], d, output_beam=6.0) - wf.graphloss(den, d, output_beam=6.0)):
            xg = x.clone().requires_grad_(True)
            l = fn(wf.DenseFsaVec(xg, seg))
            l.backward()
            res.append((l.item(), xg.grad.clone()))
    finally:
        wf.set_output_beam_mode("exact")
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("topo,B,T", [("mmictc", 32, 500), ("3state-skip", 32, 500), ("mmictc", 2, 1500)])
def test_cfg4_denominator_baseline_size_vs_fp64_oracle(oracle_lib, topo, B, T):
    # BASELINE.json configs[4] at its full size (P = 5000: S = 10 002, B = 32, T = 500, ragged) and a long T = 1500 (the error of
    # the hub kernels' fp32 recursion grows with the number of frames): sweep_hub_kernel + post_hub_kernel against the fp64 C
    # oracle on four utterances of the batch (first, second, middle, last: the longest and the shortest), properties on all
    c = util.random_case(779, topo, B=B, T=T, L=8, P=5000, den=True, ragged=B > 2)
    tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"])
    assert LAST_FALLBACKS[0] == 0          # the hub kernels' own rows: nothing was redone in the log semiring
    for b in range(B):
        Tb = int(c["segs"][b, 2])
        np.testing.assert_allclose(grad[b, :Tb].sum(-1), -1.0, atol=2e-3)
        assert np.all(grad[b, Tb:] == 0)
    pick = sorted(set([0, 1, B // 2, B - 1]))
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, [c["ographs"][b] for b in pick], c["lp"][pick],
                                        np.stack([np.arange(len(pick)), c["segs"][pick, 1], c["segs"][pick, 2]], 1).astype(np.int32),
                                        gscale=[-1.0] * len(pick), use_double=True, nthreads=4)
    np.testing.assert_allclose(-tot[pick], -tot_d, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad[pick], grad_d, atol=GRAD_ATOL)


def _tot_scores_f64(c):
    dense = wf.DenseFsaVec(torch.from_numpy(c["lp"]).to(DEV), torch.from_numpy(c["segs"]))
    return wf.tot_scores(c["fsa"].to(DEV), dense, log_semiring=True, use_double_scores=True)


@pytest.mark.parametrize("topo", sorted(wf.TOPOLOGIES))
def test_double_scores_fp64_totals_every_topology(oracle_lib, topo):
    # get_tot_scores(log_semiring=True, use_double_scores=True) (lattice_analysis.py:128,148) and graphloss(use_double_scores=
    # True): fp64 totals from the banded wave sweep (fp64 from the emission probabilities to the total) against the fp64 C
    # oracle's unrounded totals.  Topologies with arc weights: the oracle rounds w + lp to fp32 per arc as k2 does
    # (SURVEY 8(0)), the probability-domain kernels multiply exp(w) exp(lp) exactly -- a difference of 1e-7 per step.
    c = util.random_case(300 + len(topo), topo, B=4, T=50, L=7, P=9)
    want = util.c_oracle_tot_f64(oracle_lib, c["ographs"], c["lp"], c["segs"])
    weighted = bool(np.any(np.concatenate([g["score"][g["label"] >= 0] for g in c["ographs"]]) != 0))
    tot = _tot_scores_f64(c)
    assert tot.dtype == torch.float64
    np.testing.assert_allclose(tot.cpu().numpy(), want, rtol=2e-7 if weighted else 1e-12)
    # the same totals from the fp64 log-semiring sweep (every graph took it before; still the route of the large graphs)
    slow = _with_env({"WF_NO_DOUBLE_FAST": "1"}, lambda: _tot_scores_f64(c).cpu().numpy())
    np.testing.assert_allclose(slow, want, rtol=1e-12)
    # graphloss: the loss is float32 as graph.py:68 casts it; gradient against the oracle
    x = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
    loss = wf.graphloss(c["fsa"], wf.DenseFsaVec(x, torch.from_numpy(c["segs"])), reduction="none", use_double_scores=True)
    loss.sum().backward()
    _, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    np.testing.assert_allclose(loss.detach().cpu().numpy(), -want, rtol=2e-7)
    np.testing.assert_allclose(x.grad.cpu().numpy(), wgrad, atol=GRAD_ATOL)


def test_double_scores_cfg3_size_rtol_1e9(oracle_lib):
    # BASELINE.json configs[3] (ctc over a 5k table, B = 32, T = 500): fp64 totals of the fast kernels within 1e-9 relative of
    # the fp64 C oracle on every utterance
    c = util.random_case(4242, "ctc", B=32, T=500, L=64, P=5000)
    tot = _tot_scores_f64(c).cpu().numpy()
    want = util.c_oracle_tot_f64(oracle_lib, c["ographs"], c["lp"], c["segs"], nthreads=8)
    np.testing.assert_allclose(tot, want, rtol=1e-9)
    assert np.abs(tot - want).max() < 1e-9 * np.abs(want).max()
    # the fused loss + gradient call (bench.py's cfg3_double line) returns the same fp64 totals
    import bench
    w = bench.Workload("cfg3_ctc_bpe5k_num", 32, 500, torch.device(DEV), seed=777, double=True)
    w.step(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want2 = util.c_oracle_tot_f64(oracle_lib, w.fsa.as_oracle_graphs(), w.lp.cpu().numpy(), w.segs, nthreads=8)
    np.testing.assert_allclose(w.tot.cpu().numpy(), want2, rtol=1e-9)
    assert w.fallbacks(torch.cuda.current_stream().cuda_stream) == 0


def test_hub_sweep_in_segments_equals_the_single_launch():
    # cfg4-shaped denominator, ragged: the cut sweeps (state restored from the rows + hubsave, the total taken at the middle
    # frame) against one launch per direction -- same rows, totals bit-equal, gradient within the difference of the two
    # normalisers (fp32 rows at the middle frame vs the fp64 hub value at the end)
    c = util.random_case(781, "mmictc", B=6, T=200, L=8, P=1200, den=True)
    run = lambda: fused_logsum(c["fsa"], c["lp"], c["segs"])
    tot, grad = run()
    assert LAST_FALLBACKS[0] == 0
    tot1, grad1 = _with_env({"WF_NO_HUB_OVERLAP": "1"}, run)
    assert np.array_equal(tot, tot1)
    np.testing.assert_allclose(grad, grad1, rtol=2e-6, atol=1e-9)
    tot7, grad7 = _with_env({"WF_HUB_SEGS": "16:9,10,11,12,13,14,15,16"}, run)
    assert np.array_equal(tot, tot7)
    np.testing.assert_allclose(grad, grad7, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("topo,den,beam", [("ctc", False, None), ("mmictc", True, None), ("2state", False, 6.0)])
@pytest.mark.parametrize("reduction", ["sum", "none", "mean"])
def test_several_supervision_segments_per_sequence(topo, den, beam, reduction):
    """k2.DenseFsaVec lets several supervision segments name one sequence (models.py:256-261 builds one per sequence):
    one library call refuses such a batch (WF_EUNSUPPORTED) and the drop-in runs it in rounds; loss and gradient against
    the oracle, whose gradient adds the segments' posteriors into the sequence's rows."""
    c = util.random_case(71, topo, B=6, T=48, L=3, P=8, den=den)
    segs = np.array([[0, 0, 20], [2, 5, 30], [0, 24, 24], [1, 0, 48], [2, 36, 12], [0, 20, 4]], dtype=np.int32)
    lp = torch.from_numpy(c["lp"][:3]).to(DEV).requires_grad_(True)
    dense = wf.DenseFsaVec(lp, torch.from_numpy(segs))
    tl = torch.tensor([3, 3, 3, 3, 3, 3])
    old = wf.graph._BEAM_MODE
    wf.set_output_beam_mode("pruned")
    try:
        loss = wf.graphloss(c["fsa"], dense, output_beam=beam, reduction=reduction, target_lengths=tl)
        w = torch.arange(1, 7, dtype=torch.float32, device=DEV) if reduction == "none" else None
        (loss * w).sum().backward() if reduction == "none" else loss.backward()
    finally:
        wf.set_output_beam_mode(old)
    want, gwant = o.graphloss(c["ographs"], c["lp"][:3], segs, beam, reduction, target_lengths=tl.numpy())
    np.testing.assert_allclose(loss.detach().cpu().numpy(), want, rtol=TOT_RTOL, atol=TOT_ATOL)
    if reduction == "none":   # d (sum_i w_i loss_i): the oracle's gradient per utterance, weighted
        gwant = np.zeros_like(gwant)
        for i in range(6):
            gi = o.graphloss([c["ographs"][i]], c["lp"][:3], segs[i:i + 1], beam, "sum")[1]
            gwant += (i + 1) * gi
    np.testing.assert_allclose(lp.grad.cpu().numpy(), gwant, atol=GRAD_ATOL * (6 if reduction == "none" else 1))
    if reduction == "sum" and beam is None:   # the other autograd node (use_double_scores: wf_fb_logsum + wf_fb_backward)
        lp2 = torch.from_numpy(c["lp"][:3]).to(DEV).requires_grad_(True)
        l2 = wf.graphloss(c["fsa"], wf.DenseFsaVec(lp2, torch.from_numpy(segs)), output_beam=None, reduction="sum",
                          use_double_scores=True)
        l2.backward()
        np.testing.assert_allclose(l2.item(), want, rtol=TOT_RTOL, atol=TOT_ATOL)
        np.testing.assert_allclose(lp2.grad.cpu().numpy(), gwant, atol=GRAD_ATOL)
    # forward-only entry points take such batches as they are
    tot = wf.tot_scores(c["fsa"], wf.DenseFsaVec(lp.detach(), torch.from_numpy(segs)), True, True).cpu().numpy()
    bp = wf.best_path(c["fsa"], wf.DenseFsaVec(lp.detach(), torch.from_numpy(segs)))
    sr = wf.pruned_search(c["fsa"], wf.DenseFsaVec(lp.detach(), torch.from_numpy(segs)), search_beam=float("inf"),
                          min_active_states=0, max_active_states=0, log_tot=False)
    for i in range(6):
        x, T = c["lp"][segs[i, 0], segs[i, 1]:], int(segs[i, 2])
        np.testing.assert_allclose(tot[i], o.forward(c["ographs"][i], x, T)[1], rtol=1e-9)
        assert bp["scores"][i].item() == o.viterbi(c["ographs"][i], x, T)[0] == sr["scores"][i].item()
