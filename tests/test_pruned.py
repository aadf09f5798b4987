"""GPU tests (-m gpu) of output_beam pruning (k2.intersect_dense(graph, dense, output_beam), waterfall/graph.py:61-62,81)
on the banded probability-domain path: trop_band_kernel + band_mask_kernel + the MASK instantiations of
sweep_band_kernel / post_small_kernel, against oracle.prune_mask's lattice (k2's own arithmetic is parity
unpinned: the reference holds no fixture).  Every banded instantiation, the general-graph mode (small denominators),
the fallback routes, and the BASELINE-size batches the bench reports.
"""
import numpy as np
import pytest
import torch

import waterfall_amd as wf
from oracle import dwfst_oracle as o
from tests import util
from tests.test_gpu_parity import DEV, GRAD_ATOL, LAST_FALLBACKS, SWEEP_CASES, TOT_ATOL, TOT_RTOL, _with_env, fused_logsum

pytestmark = pytest.mark.gpu

PRUNED_PATHS = {
    "fast": {},                                                            # tropical band sweep, masks, masked band sweep, masked post_small
    "fast_check_kernel": {"WF_NO_POST_CHECK": "1"},                        # row check by check_kernel, masked log-semiring fallback sweep
    "fast_then_rescue": {"WF_FORCE_FALLBACK": "1"},                        # every utterance redone by the masked rescue_kernel
    "fast_then_fallback_sweep": {"WF_FORCE_FALLBACK": "1", "WF_NO_POST_CHECK": "1"},
    "fast_table_by_copy": {"WF_NO_TAB_KERNARG": "1"},
    "fast_fill_side_stream": {"WF_FILL_SIDE_STREAM": "1"},
    "log_semiring": {"WF_NO_PRUNED_FAST": "1"},                            # round 2's path: sweep_kernel<M_FB_MAX> + masked sweep_kernel
}


def autograd_pruned(fsa, lp, segs, beam, eager):
    x = torch.from_numpy(np.ascontiguousarray(lp)).to(DEV).requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(np.ascontiguousarray(segs, dtype=np.int32)))
    wf.set_output_beam_mode("pruned")
    wf.set_eager_gradient(eager)
    try:
        loss = wf.graphloss(fsa, dense, output_beam=beam, reduction="none")
        loss.sum().backward()
    finally:
        wf.set_output_beam_mode("exact")
        wf.set_eager_gradient(True)
    return loss.detach().cpu().numpy(), x.grad.cpu().numpy()


@pytest.mark.parametrize("path", sorted(PRUNED_PATHS))
@pytest.mark.parametrize("beam", [2.0, 10.0])
@pytest.mark.parametrize("topo,den,L,T,P", SWEEP_CASES)
def test_pruned_paths_against_oracle(path, beam, topo, den, L, T, P):
    c = util.random_case(300 + L + T, topo, B=5, T=T, L=L, P=P, den=den, ragged=T > 10)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], beam, "none")
    ok = np.isfinite(want)

    def run():
        tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"], beam=beam)
        nfb = LAST_FALLBACKS[0]
        loss, agrad = autograd_pruned(c["fsa"], c["lp"], c["segs"], beam, True)
        loss2, agrad2 = autograd_pruned(c["fsa"], c["lp"], c["segs"], beam, False)
        return tot, grad, loss, agrad, loss2, agrad2, nfb
    tot, grad, loss, agrad, loss2, agrad2, nfb = _with_env(PRUNED_PATHS[path], run)
    if "then" not in path:
        assert nfb == 0        # the masked probability-domain rows passed the row check: nothing was redone
    # the tropical pass runs in fp32 here and in fp64 in the oracle: an arc within rounding of the threshold may fall on
    # either side; it carries e^-beam of the best path's mass at most
    slack = 3e-3 if beam < 5 else 1e-3
    for got_loss, got_grad in ((-tot, grad), (loss, agrad), (loss2, agrad2)):
        assert np.array_equal(np.isfinite(got_loss), ok)
        np.testing.assert_allclose(got_loss[ok], want[ok], rtol=TOT_RTOL, atol=slack)
        np.testing.assert_allclose(got_grad, wgrad, atol=max(GRAD_ATOL, slack))
        assert not np.isnan(got_grad).any()


@pytest.mark.parametrize("beam", [1.0, 6.0])
def test_pruned_fast_path_equals_the_log_semiring_path_arc_for_arc(beam):
    # same fp32 tropical rows, same keep expression: both routes sum over the SAME lattice arcs, so they agree far inside
    # the oracle tolerance -- also with weighted arcs (LM-like scores)
    c = util.random_case(78, "ctc", B=4, T=70, L=9, P=8)
    rng = np.random.default_rng(6)
    graphs = []
    for g in c["ographs"]:
        g = dict(g)
        g["score"] = (rng.standard_normal(g["score"].shape) * 0.7).astype(np.float32)
        graphs.append(g)
    fsa = util.graphs_to_fsa(graphs)
    fast = fused_logsum(fsa, c["lp"], c["segs"], beam=beam)
    assert LAST_FALLBACKS[0] == 0
    slow = _with_env({"WF_NO_PRUNED_FAST": "1"}, lambda: fused_logsum(fsa, c["lp"], c["segs"], beam=beam))
    np.testing.assert_allclose(fast[0], slow[0], rtol=1e-5)
    np.testing.assert_allclose(fast[1], slow[1], atol=2e-4)
    want, wgrad = o.graphloss(graphs, c["lp"], c["segs"], beam, "none")
    np.testing.assert_allclose(-fast[0], want, rtol=TOT_RTOL, atol=3e-3)
    np.testing.assert_allclose(fast[1], wgrad, atol=3e-3)
    exact, _ = o.graphloss(graphs, c["lp"], c["segs"], None, "none")
    if beam <= 2.0:
        assert np.any(want > exact + 1e-3)     # the beam really removed mass


@pytest.mark.parametrize("beam", [10.0, 20.0])
@pytest.mark.parametrize("topo,P,L,den", [("ctc", 5000, 64, False), ("2state", 29, 80, False), ("2state", 29, 80, True)])
def test_pruned_at_baseline_size_against_oracle(topo, P, L, den, beam):
    """cfg3 (B=32, T=500, V=5001, CTC numerators L=64) and cfg2 (2state numerators L=80 / the 60-state denominator) with
    the reference's beams (graph.py:81 default 10; 20 in egs/wsj/asr_char/conf/train_k2.yaml:3): loss inside 1e-4
    relative, gradient inside 1e-3 absolute of the oracle's pruned lattice, and nothing redone by the fallback."""
    c = util.random_case(777, topo, B=32, T=500, L=L, P=P, den=den, ragged=True)
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], beam, "none")
    tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"], beam=beam)
    assert LAST_FALLBACKS[0] == 0
    np.testing.assert_allclose(-tot, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(grad, wgrad, atol=GRAD_ATOL)
    assert np.all(grad <= 0)
    for b in range(32):
        T = int(c["segs"][b, 2])
        np.testing.assert_allclose(grad[b, :T].sum(-1), -1.0, atol=2e-3)   # the kept arcs' posteriors of a frame sum to 1
        assert np.all(grad[b, T:] == 0)
    if beam == 10.0:
        exact, egrad = o.graphloss(c["ographs"][:4], c["lp"][:4], c["segs"][:4], None, "none")
        assert np.abs(egrad - wgrad[:4]).max() > 1e-3      # at this beam the pruned and the exact gradient really differ
    loss, agrad = autograd_pruned(c["fsa"], c["lp"], c["segs"], beam, True)
    np.testing.assert_allclose(loss, want, rtol=TOT_RTOL, atol=TOT_ATOL)
    np.testing.assert_allclose(agrad, wgrad, atol=GRAD_ATOL)


def test_auto_mode_prunes_where_the_fast_path_exists():
    # the drop-in's default: numerators (banded) honour output_beam, hub-form denominators get the exact sum and one warning
    import warnings
    from waterfall_amd import graph as g
    num = util.random_case(41, "mmictc", B=3, T=40, L=5, P=300)
    den = util.random_case(41, "mmictc", B=3, T=40, L=5, P=300, den=True)

    def loss_of(c, mode):
        wf.set_output_beam_mode(mode)
        try:
            x = torch.from_numpy(c["lp"]).to(DEV).requires_grad_(True)
            l = wf.graphloss(c["fsa"], wf.DenseFsaVec(x, torch.from_numpy(c["segs"])), output_beam=4.0, reduction="none")
            l.sum().backward()
            return l.detach().cpu().numpy(), x.grad.cpu().numpy()
        finally:
            wf.set_output_beam_mode("exact")
    g._AUTO_WARNED[0] = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        la, ga = loss_of(num, "auto")
        assert not w                                     # the numerators have the fast pruned path: no warning
        lp_, gp = loss_of(num, "pruned")
        le, ge = loss_of(num, "exact")
        assert np.array_equal(la, lp_) and np.array_equal(ga, gp) and not np.array_equal(la, le)
        da, dga = loss_of(den, "auto")
        assert len(w) == 1 and "output_beam" in str(w[0].message)
        de, dge = loss_of(den, "exact")
        dp, _ = loss_of(den, "pruned")
        assert np.array_equal(da, de) and np.array_equal(dga, dge) and not np.array_equal(da, dp)
