"""CPU tests: pin the oracle (oracle/) against the reference's own outputs and known answers.

Golden fixtures come from the reference's importable manual_ctc code and topology generator
scripts (tests/golden/make_golden.py); see SURVEY.md 8(c).
"""
import numpy as np
import pytest
import torch

from oracle import dwfst_oracle as o
from oracle import graphs as og
from tests import util

TOPO_CASES = ["2state", "2state-1", "3state-skip", "3state-skip-1", "3state-skip-2", "mmictc", "mmictc-1",
              "2state_sil", "mmictc_noblk"]


def _ctc_graphs(z):
    V = z["log_probs"].shape[-1]
    topo_txt = open(f"{util.GOLDEN}/topo_ctc.txt").read()  # structure check happens in test_host.py
    phones = util.ctc_phones(V)
    import waterfall_amd.topology as wt
    topo, tokens = wt.build_topology("ctc", phones)
    assert len(tokens) == V
    tg = topo.as_oracle_graphs()[0]
    off = z["target_off"]
    return [og.compose_topo_transcript(tg, z["targets"][off[i]:off[i + 1]].tolist()) for i in range(len(off) - 1)]


@pytest.mark.parametrize("name", ["manual_ctc_ctc_small.npz", "manual_ctc_ctc_medium.npz"])
def test_oracle_matches_reference_manual_ctc_on_ctc(name):
    z = util.load_golden(name)
    lp, lens = z["log_probs"], z["input_lengths"]
    graphs = _ctc_graphs(z)
    B = lp.shape[0]
    segs = np.stack([np.arange(B), np.zeros(B), lens], 1).astype(np.int32)
    loss, grad = o.graphloss(graphs, lp, segs, None, "none")
    # ctc.py:57-61: the reference's per-utterance loss is the frame-mean of identical totals
    np.testing.assert_allclose(loss, z["loss_none"], rtol=2e-5)
    np.testing.assert_allclose(loss.sum(), z["loss_sum"], rtol=2e-5)
    # ctc.py:76-91: gradient of the 'sum' loss = -posteriors (rows sum to -1 on valid frames)
    np.testing.assert_allclose(grad, z["grad"], atol=2e-5)
    for b in range(B):
        np.testing.assert_allclose(grad[b, :lens[b]].sum(-1), -1.0, atol=1e-6)
        assert np.all(grad[b, lens[b]:] == 0)


@pytest.mark.parametrize("topo", TOPO_CASES)
def test_oracle_matches_reference_manual_ctc_on_topologies(topo):
    z = util.load_golden(f"manual_ctc_{topo}.npz")
    lp, lens = z["log_probs"], z["input_lengths"]
    graphs = util.golden_graphs(z)
    B = lp.shape[0]
    segs = np.stack([np.arange(B), np.zeros(B), lens], 1).astype(np.int32)
    loss, grad = o.graphloss(graphs, lp, segs, None, "none")
    np.testing.assert_allclose(loss, z["loss_none"], rtol=2e-5)
    np.testing.assert_allclose(grad, z["grad"], atol=2e-5)


# ---- round-2 fixtures: the reference's manual_ctc code at realistic sizes (T = 500, L = 80; V = 5001) and its
# eta-semiring / alignment outputs on non-CTC topologies (tests/golden/make_golden.py make_round2) ----

def _segs_of(z):
    B = z["log_probs"].shape[0]
    return np.stack([np.arange(B), np.zeros(B), z["input_lengths"]], 1).astype(np.int32)


@pytest.mark.parametrize("name", ["manual_ctc_ctc_cfg2.npz", "manual_ctc_ctc_v5001.npz", "manual_ctc_2state_cfg2.npz"])
def test_oracle_matches_reference_manual_ctc_at_config_sizes(name):
    z = util.load_golden(name)
    graphs = util.golden_graphs(z) if "g_src" in z else _ctc_graphs(z)
    loss, grad = o.graphloss(graphs, z["log_probs"], _segs_of(z), None, "none")
    # the reference accumulates 500 frames in fp32 (ctc.py:41-45); the oracle in fp64
    np.testing.assert_allclose(loss, z["loss_none"], rtol=2e-5)
    np.testing.assert_allclose(loss.sum(), z["loss_sum"], rtol=2e-5)
    np.testing.assert_allclose(grad, z["grad"], atol=5e-4)   # measured 2.8e-4 at T = 500 (fp32 drift in the reference)
    if "align" in z:
        # prior.py:49-55 state posteriors of the first utterance(s).  The machine's 2L+1 states are the states of the
        # graph that the manual_ctc drop-in's host conversion builds from (trans, transform) -- same state ids
        from waterfall_amd import manual_ctc
        V = z["log_probs"].shape[-1]
        off = z["target_off"]
        for b in range(z["align"].shape[0]):
            tr, tf = og.ctc_trans_transform(z["targets"][off[b]:off[b + 1]].tolist(), V)
            fsa = manual_ctc.matrices_to_fsa(torch.from_numpy(tr)[None], torch.tensor([tr.shape[0]]), torch.from_numpy(tf)[None])
            g = fsa.as_oracle_graphs()[0]
            T = int(z["input_lengths"][b])
            sp = o.state_posteriors(g, z["log_probs"][b], T)
            np.testing.assert_allclose(np.exp(sp[:, :tr.shape[0]]), np.exp(z["align"][b, :T, :tr.shape[0]]), atol=5e-4)


@pytest.mark.parametrize("topo", ["2state", "mmictc", "3state-skip", "2state_sil3"])
def test_oracle_matches_reference_eta_semiring_and_alignment_on_topologies(topo):
    z = util.load_golden(f"manual_ctc_{topo}_eta_align.npz")
    lp, lens = z["log_probs"], z["input_lengths"]
    graphs = util.golden_graphs(z)
    segs = _segs_of(z)
    loss, grad = o.graphloss(graphs, lp, segs, None, "none")
    np.testing.assert_allclose(loss, z["loss_none"], rtol=2e-5)
    np.testing.assert_allclose(grad, z["grad"], atol=2e-5)
    # softmax_ctc.py:44-62: 1/eta * LSE(eta x) = the log semiring on the scaled model; its gradient w.r.t. log_probs
    # is minus the scaled model's posteriors (softmax_ctc.py:80-92)
    eta = float(z["eta"])
    l_eta, g_eta = o.graphloss([o.scale_graph(g, eta) for g in graphs], lp * np.float32(eta), segs, None, "none")
    np.testing.assert_allclose(l_eta.sum() / eta, z["eta_loss_sum"], rtol=2e-5)
    np.testing.assert_allclose(g_eta, z["eta_grad"], atol=2e-5)
    # prior.py:49-55 on the arc encoding: row 1 + j of the machine = the j-th emitting arc
    for b, g in enumerate(graphs):
        T = int(lens[b])
        post = o.arc_posteriors(g, lp[b], T)
        n = post.shape[1]
        np.testing.assert_allclose(np.exp(z["align"][b, :T, 1:1 + n]), post, atol=2e-5)


def test_oracle_matches_torch_ctc_loss():
    # independent known answer: the reference treats F.ctc_loss as the same loss (models.py:212-216)
    z = util.load_golden("manual_ctc_ctc_medium.npz")
    lp, lens = z["log_probs"], z["input_lengths"]
    graphs = _ctc_graphs(z)
    B = lp.shape[0]
    off = z["target_off"]
    x = torch.from_numpy(lp).double().requires_grad_(True)
    tl = torch.tensor(np.diff(off))
    loss = torch.nn.functional.ctc_loss(x.permute(1, 0, 2), torch.from_numpy(z["targets"]).long(),
                                        torch.from_numpy(lens), tl, reduction="none")
    loss.sum().backward()
    segs = np.stack([np.arange(B), np.zeros(B), lens], 1).astype(np.int32)
    mine, grad = o.graphloss(graphs, lp, segs, None, "none")
    np.testing.assert_allclose(mine, loss.detach().numpy(), rtol=1e-6)
    # F.ctc_loss differentiates w.r.t. logits-through-log_softmax convention: grad = exp(lp) - post
    g = x.grad.numpy()
    for b in range(B):
        T = lens[b]
        np.testing.assert_allclose(grad[b, :T], g[b, :T] - np.exp(lp[b, :T].astype(np.float64)), atol=1e-6)


@pytest.mark.parametrize("topo", ["ctc", "2state", "mmictc", "3state-skip"])
def test_oracle_matches_brute_force(topo):
    c = util.random_case(5, topo, B=2, T=6, L=2, P=3, ragged=False)
    for g, lp in zip(c["ographs"], c["lp"]):
        tot, grad = o.tot_and_grad(g, lp, 6)
        btot, bgrad, bbest, bpaths = o.brute_force(g, lp, 6)
        assert np.isfinite(btot)
        np.testing.assert_allclose(tot, btot, rtol=1e-12)
        np.testing.assert_allclose(grad, bgrad, atol=1e-12)
        score, path = o.viterbi(g, lp, 6, dtype=np.float64)
        np.testing.assert_allclose(score, bbest, rtol=1e-12)
        assert path == min(bpaths)  # lowest arc index at every step == lexicographically smallest here


def test_oracle_viterbi_tie_break_lowest_arc_index():
    # two parallel arcs with identical scores everywhere: the lower arc index must win each step
    arcs = [(0, 0, 1, 0.0), (0, 0, 1, 0.0), (0, 1, 2, 0.0), (0, 1, 2, 0.0), (1, 2, -1, 0.0), (1, 2, -1, 0.0)]
    g = o.make_graph(arcs, 3)
    lp = np.log(np.full((4, 3), 1 / 3, dtype=np.float32))
    score, path = o.viterbi(g, lp, 4)
    assert path == [0, 0, 0, 2, 4]
    _, _, _, best = o.brute_force(g, lp, 4)
    assert len(best) == 2 ** 5 and min(best) == path


def test_oracle_infeasible_is_minus_inf_and_zero_grad():
    c = util.random_case(3, "ctc", B=1, T=3, L=5, P=4, ragged=False)
    tot, grad = o.tot_and_grad(c["ographs"][0], c["lp"][0], 3)
    assert tot == -np.inf and np.all(grad == 0)
    score, path = o.viterbi(c["ographs"][0], c["lp"][0], 3)
    assert score == -np.inf and path == []


def test_oracle_finite_difference_gradient():
    c = util.random_case(11, "2state", B=1, T=7, L=2, P=3, ragged=False)
    g, lp = c["ographs"][0], c["lp"][0].astype(np.float64)
    # arc scores are rounded to fp32 inside the oracle (k2 lattice scores are fp32): use a step
    # large enough to dominate that rounding
    tot, grad = o.tot_and_grad(g, lp, 7)
    rng = np.random.default_rng(0)
    for _ in range(10):
        t, v = rng.integers(0, 7), rng.integers(0, lp.shape[1])
        d = np.zeros_like(lp)
        d[t, v] = 1e-2
        tp, _ = o.tot_and_grad(g, lp + d, 7)
        tm, _ = o.tot_and_grad(g, lp - d, 7)
        np.testing.assert_allclose((tp - tm) / 2e-2, grad[t, v], atol=2e-4)


def test_oracle_prune_mask_keeps_best_path_and_converges():
    c = util.random_case(21, "2state", B=1, T=20, L=4, P=5, ragged=False)
    g, lp = c["ographs"][0], c["lp"][0]
    exact, _ = o.tot_and_grad(g, lp, 20)
    best, _ = o.forward(g, lp, 20, log_semiring=False)[1], None
    t1, _ = o.tot_and_grad(g, lp, 20, output_beam=1e-3)
    t10, g10 = o.tot_and_grad(g, lp, 20, output_beam=10.0)
    t100, _ = o.tot_and_grad(g, lp, 20, output_beam=1000.0)
    assert best <= t1 <= t10 <= t100 + 1e-12
    np.testing.assert_allclose(t100, exact, rtol=1e-12)
    np.testing.assert_allclose(g10.sum(-1), 1.0, atol=1e-9)


@pytest.mark.parametrize("topo,den", [("ctc", False), ("2state", False), ("mmictc", True), ("3state-skip", True)])
def test_c_oracle_matches_numpy_oracle(oracle_lib, topo, den):
    c = util.random_case(31, topo, B=3, T=40, L=6, P=6, den=den)
    tot_c, grad_c = util.c_oracle_batch(oracle_lib, c["ographs"], c["lp"], c["segs"], gscale=[-1, -1, -1])
    loss, grad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    np.testing.assert_allclose(-tot_c, loss, rtol=1e-5)
    np.testing.assert_allclose(grad_c, grad, atol=5e-5)
    tot_d, grad_d = util.c_oracle_batch(oracle_lib, c["ographs"], c["lp"], c["segs"], gscale=[-1, -1, -1],
                                        use_double=True, nthreads=2)
    np.testing.assert_allclose(-tot_d, loss, rtol=1e-6)
    np.testing.assert_allclose(grad_d, grad, atol=1e-6)
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path = util.c_oracle_viterbi(oracle_lib, g, c["lp"][i], T)
        s2, path2 = o.viterbi(g, c["lp"][i], T)
        assert s == s2 and list(path) == path2


def test_beam_search_restatement_is_viterbi_and_forward_at_infinite_beam():
    """oracle.beam_search (the beam rule wf_search is tested against) collapses to the pinned `viterbi` / `forward`
    when nothing is pruned, and pruning only ever removes paths."""
    from tests import util
    c = util.random_case(77, "mmictc", B=2, T=25, L=5, P=6, den=True)
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path, tl, na = o.beam_search(g, c["lp"][i], T)
        s0, p0 = o.viterbi(g, c["lp"][i], T)
        assert s == s0 and path == p0
        np.testing.assert_allclose(tl, o.forward(g, c["lp"][i], T)[1], rtol=1e-12)
        s2, p2, tl2, na2 = o.beam_search(g, c["lp"][i], T, 2.0)
        assert s2 <= s and tl2 <= tl + 1e-12 and all(a <= b for a, b in zip(na2, na))


@pytest.mark.parametrize("topo,den", [("ctc", False), ("mmictc", True), ("3state-skip", True)])
def test_lattice_search_restatement_against_prune_mask(topo, den):
    """oracle.lattice_search (search + get_lattice's output_beam, what wf_search_lattice is tested against) against two
    independent statements: with output_beam = inf it is beam_search; with search_beam = inf its total is the forward
    sum over `prune_mask`'s arcs (the output_beam rule of intersect_dense, graph.py:61-62), and its arc count is that
    mask's plus the final arcs that stay."""
    from tests import util
    c = util.random_case(52, topo, B=3, T=30, L=7, P=8, den=den)
    for i, g in enumerate(c["ographs"]):
        T, lp = int(c["segs"][i, 2]), c["lp"][i]
        for sb in (np.inf, 4.0):
            s, path, tl, na = o.lattice_search(g, lp, T, sb, np.inf)
            s2, p2, tl2, _ = o.beam_search(g, lp, T, sb)
            assert s == s2 and path == p2
            np.testing.assert_allclose(tl, tl2, rtol=1e-12)
        fin = g["label"] < 0
        for ob in (2.0, 6.0):
            s, path, tl, na = o.lattice_search(g, lp, T, np.inf, ob)
            keep = o.prune_mask(g, lp, T, ob, dtype=np.float32)
            np.testing.assert_allclose(tl, o.forward(g, lp, T, True, keep=keep)[1], rtol=1e-12)
            assert 0 < na - int(keep.sum()) <= int(fin.sum())
            assert tl <= o.lattice_search(g, lp, T, np.inf, np.inf)[2] + 1e-12


def test_lattice_words_restatement_against_the_composed_graph():
    """oracle.lattice_words (the lattice restricted to a word sequence, what wf_search_lattice_words is tested against) with
    nothing pruned against an independent route: decode.compose_words builds `connect(compose(TLG, W))` as a graph and the
    pinned `forward` / `viterbi` run on it (utils/lattice_analysis.py:139-149)."""
    import torch
    import waterfall_amd as wf
    from waterfall_amd import decode
    from waterfall_amd.synth import synth_decoding_graph
    g = synth_decoding_graph(3, 300, 20)
    rng = np.random.default_rng(0)
    em = g["label"] >= 0
    g["aux"] = np.where(em, np.where(rng.random(em.size) < 0.3, rng.integers(1, 5, em.size), 0), -1).astype(np.int32)
    arcs = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    TLG = wf.Fsa(arcs, g["aux"], [0, arcs.shape[0]], [g["num_states"]])
    T = 12
    lp = torch.log_softmax(torch.randn(T, 20, generator=torch.Generator().manual_seed(1)) * 2, -1).numpy()
    s, path, tl, _ = o.lattice_search(g, lp, T)
    words = [int(g["aux"][a]) for a in path if g["aux"][a] > 0]
    for ws in (words, words[:-1], words + [1], []):
        tw, bw = o.lattice_words(g, lp, T, ws)
        gw = decode.compose_words(TLG, ws).as_oracle_graphs()[0]
        fw, vb = o.forward(gw, lp, T)[1], o.viterbi(gw, lp, T)[0]
        if np.isfinite(fw):
            np.testing.assert_allclose(tw, fw, rtol=1e-12)
        else:
            assert tw == -np.inf
        # (tropical totals: the backward sum associates from the right, viterbi from the left -- fp32 rounding apart)
        assert (bw == vb) if not np.isfinite(vb) else abs(float(bw) - float(vb)) <= 1e-5 * abs(float(vb))
        assert tw <= tl + 1e-9
    np.testing.assert_allclose(o.lattice_words(g, lp, T, words)[1], s, rtol=1e-6)   # the best path spells its own words
    assert decode.compose_words(TLG, [4] * 15).arcs.shape[0] == 0   # longer than any path: an empty graph
