"""GPU tests (-m gpu) of the beam-pruned frame-synchronous search wf_search (k2.get_lattice's role at
utils/lattice_analysis.py:125), through the C ABI.

What pins it: with an infinite beam it must equal the exact Viterbi sweep bit for bit (scores and arcs) and the exact
log-semiring total; with a finite beam it must equal oracle.beam_search (this repository's statement of the beam rule:
k2's own beam arithmetic is parity unpinned, the reference holds no fixture for it); max / min active states are checked
through their defining properties.
"""
import numpy as np
import pytest
import torch

import waterfall_amd as wf
from waterfall_amd import decode
from oracle import dwfst_oracle as o
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
INF = float("inf")

# Every test of this file runs on each search kernel: search_kernel (one workgroup per utterance), search_mw_kernel with
# four workgroups per utterance and every frame of >= 2 survivors shared between them (the graphs here are small: the
# default would run most frames on workgroup 0 alone), and what wf_search picks by itself.
VARIANTS = {"one_workgroup": {"WF_SEARCH_WGS": "1"},
            "four_workgroups_every_frame_shared": {"WF_SEARCH_WGS": "4", "WF_SEARCH_SOLO_MAX": "2"},
            "default": {}}


@pytest.fixture(autouse=True, params=sorted(VARIANTS))
def search_kernel_variant(request):
    import os
    env = VARIANTS[request.param]
    old = {k: os.environ.get(k) for k in ("WF_SEARCH_WGS", "WF_SEARCH_SOLO_MAX")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    yield request.param
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _case(topo, den, seed=50, B=4, T=50):
    c = util.random_case(seed, topo, B=B, T=T, L=7, P=8, den=den)
    dense = wf.DenseFsaVec(torch.from_numpy(c["lp"]).to(DEV), torch.from_numpy(c["segs"]))
    return c, dense


def _big(S, B=4, T=40, V=301, seed=5):
    g = util.synth_decoding_graph(seed, S, V)
    gen = torch.Generator().manual_seed(seed)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 3.0, -1).numpy()
    lens = ([T, max(1, T - 7), max(1, T // 2), 1] + [T] * B)[:B]
    segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), np.array(lens)], 1).astype(np.int32)
    arcs_ = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    one = wf.Fsa(arcs_, g["aux"], [0, arcs_.shape[0]], [g["num_states"]])
    fsa = wf.create_fsa_vec([one] * B)
    dense = wf.DenseFsaVec(torch.from_numpy(lp).to(DEV), torch.from_numpy(segs))
    return g, lp, segs, fsa, dense


@pytest.mark.parametrize("topo,den", [("ctc", False), ("2state", False), ("mmictc", True), ("3state-skip-2", True),
                                      ("ctc", True)])
def test_infinite_beam_equals_exact_sweeps(topo, den):
    c, dense = _case(topo, den)
    exact = wf.best_path(c["fsa"], dense)
    out = wf.pruned_search(c["fsa"], dense, search_beam=INF, min_active_states=0, max_active_states=0,
                           use_double_scores=True, return_active=True)
    assert torch.equal(out["scores"], exact["scores"])            # bit-exact fp32
    assert torch.equal(out["arcs"], exact["arcs"])
    assert out["labels"] == exact["labels"]
    tot = wf.tot_scores(c["fsa"], dense, True, True).cpu().numpy()
    np.testing.assert_allclose(out["tot_log"].cpu().numpy(), tot, rtol=1e-6, atol=1e-5)
    nact = out["n_active"].cpu().numpy()
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        want = o.beam_search(g, c["lp"][i], T)[3]
        assert list(nact[i, :T + 1]) == want                      # every reachable state survives
        assert np.all(nact[i, T + 1:] == 0)


@pytest.mark.parametrize("beam", [3.0, 8.0, 15.0])
@pytest.mark.parametrize("topo,den", [("ctc", False), ("mmictc", True), ("3state-skip", True), ("2state_sil", False)])
def test_finite_beam_against_oracle(topo, den, beam):
    c, dense = _case(topo, den, seed=51)
    out = wf.pruned_search(c["fsa"], dense, search_beam=beam, min_active_states=0, max_active_states=0,
                           use_double_scores=True, return_active=True)
    scores, arcs = out["scores"].cpu().numpy(), out["arcs"].cpu().numpy()
    tot, nact = out["tot_log"].cpu().numpy(), out["n_active"].cpu().numpy()
    base = c["fsa"].fsa_arc_offsets
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path, tl, na = o.beam_search(g, c["lp"][i], T, beam)
        gi = 0 if c["fsa"].utt2fsa is not None else i
        assert scores[i] == s
        assert list(arcs[i, :T + 1] - base[gi]) == path if path else np.all(arcs[i] == -1)
        assert list(nact[i, :T + 1]) == na
        if np.isfinite(tl):
            np.testing.assert_allclose(tot[i], tl, rtol=1e-6, atol=1e-5)
        else:
            assert tot[i] == -np.inf


@pytest.mark.parametrize("S", [12_001, 40_000])
def test_big_graph_search(oracle_lib, S):
    g, lp, segs, fsa, dense = _big(S)
    exact = wf.best_path(fsa, dense)
    out = wf.pruned_search(fsa, dense, search_beam=INF, min_active_states=0, max_active_states=0)
    assert torch.equal(out["scores"], exact["scores"]) and torch.equal(out["arcs"], exact["arcs"])
    for i in (0, 2):
        s, path = util.c_oracle_viterbi(oracle_lib, g, lp[i], int(segs[i, 2]))
        assert out["scores"][i].item() == s
        assert list(out["arcs"][i, :int(segs[i, 2]) + 1].cpu().numpy()) == list(path)
    # finite beam on utterance 1, against the numpy statement of the beam rule
    pr = wf.pruned_search(fsa, dense, search_beam=6.0, min_active_states=0, max_active_states=0, use_double_scores=True,
                          return_active=True)
    for i in range(4):
        Ti = int(segs[i, 2])
        s, path, tl, na = o.beam_search(g, lp[i], Ti, 6.0)
        assert pr["scores"][i].item() == s
        got = list(pr["arcs"][i, :Ti + 1].cpu().numpy())
        assert got == path if path else all(a == -1 for a in got)   # the beam may lose every complete path
        assert list(pr["n_active"][i, :Ti + 1].cpu().numpy()) == na
        if np.isfinite(tl):
            np.testing.assert_allclose(pr["tot_log"][i].item(), tl, rtol=1e-6)
        else:
            assert pr["tot_log"][i].item() == -np.inf


def test_beam_bounds_and_active_state_limits():
    g, lp, segs, fsa, dense = _big(20_000, B=3, T=60, seed=9)
    exact = wf.best_path(fsa, dense)["scores"].cpu().numpy()
    # (survivor sets are NOT nested in the beam: a wider beam can keep a state that raises a later frame's best score and
    # with it the cut-off, so only the bounds below hold for every beam)
    prev_score = None
    for beam in (2.0, 5.0, 10.0, 20.0, INF):
        r = wf.pruned_search(fsa, dense, search_beam=beam, min_active_states=0, max_active_states=0,
                             use_double_scores=True, return_active=True)
        sc, tot = r["scores"].cpu().numpy(), r["tot_log"].cpu().numpy()
        assert np.all(sc <= exact)                                  # a pruned search never beats the exact one
        assert np.all(tot >= sc.astype(np.float64) - 1e-6)          # log-sum >= its best path
        fin = np.isfinite(sc)
        assert np.all(np.isfinite(tot) == fin)
        prev_score = sc
    assert np.array_equal(prev_score, exact)
    # max_active_states: never more survivors than the limit; the result stays a valid path of the graph
    lim = wf.pruned_search(fsa, dense, search_beam=INF, min_active_states=0, max_active_states=200, return_active=True)
    na = lim["n_active"].cpu().numpy()
    assert na.max() <= 200 and np.all(lim["scores"].cpu().numpy() <= exact)
    arcs = lim["arcs"].cpu().numpy()
    for i in range(3):
        T = int(segs[i, 2])
        path = arcs[i, :T + 1]
        assert np.all(path >= 0)
        assert g["src"][path[0]] == 0 and g["label"][path[-1]] == -1
        assert np.all(g["dst"][path[:-1]] == g["src"][path[1:]])
        x = np.float32(0)
        for t, a in enumerate(path[:-1]):
            x = np.float32(x + np.float32(g["score"][a] + lp[i, t, g["label"][a]]))
        assert np.float32(x + g["score"][path[-1]]) == lim["scores"][i].item()
    # min_active_states keeps a narrow beam alive: at least min(min_active, candidates) survivors per frame
    narrow = wf.pruned_search(fsa, dense, search_beam=0.5, min_active_states=0, max_active_states=0, return_active=True)
    kept = wf.pruned_search(fsa, dense, search_beam=0.5, min_active_states=50, max_active_states=0, return_active=True)
    nn, nk = narrow["n_active"].cpu().numpy(), kept["n_active"].cpu().numpy()
    full = wf.pruned_search(fsa, dense, search_beam=INF, min_active_states=0, max_active_states=0,
                            return_active=True)["n_active"].cpu().numpy()
    T0 = int(segs[0, 2])
    assert np.all(nk[0, 3:T0 + 1] >= np.minimum(50, nn[0, 3:T0 + 1]))
    assert np.all(nk[0, 3:T0 + 1] >= 50) or np.any(full[0, 3:T0 + 1] < 50)
    assert np.all(kept["scores"].cpu().numpy() >= narrow["scores"].cpu().numpy())


def test_get_lattice_call_shapes_of_lattice_analysis():
    """utils/lattice_analysis.py:125-135 verbatim against the search; :147-149 (intersect_dense) stays exact."""
    g, lp, segs, fsa, dense = _big(12_001, B=2, T=30, seed=11)
    one = wf.Fsa(fsa.arcs, fsa.aux_labels, fsa.fsa_arc_offsets, fsa.fsa_num_states)
    x = torch.from_numpy(lp).to(DEV)
    lens = torch.tensor([int(segs[0, 2]), int(segs[1, 2])])
    lattice = decode.get_lattice(x, lens, one, search_beam=20.0, output_beam=6.0, max_active_states=2000)
    tot_log = lattice.get_tot_scores(True, True)
    tot_best = lattice.get_tot_scores(True, False)
    assert tot_log.dtype == torch.float64 and tot_best.dtype == torch.float64
    one_best = decode.one_best_decoding(lattice)
    ref = wf.pruned_search(fsa, dense, 20.0, 30, 2000, use_double_scores=False)
    assert torch.equal(one_best.scores, ref["scores"])
    assert torch.equal(tot_best, ref["scores"].double())
    assert decode.get_aux_labels(one_best) == ref["aux_labels"]
    assert torch.all(tot_log >= tot_best - 1e-6)
    exact = decode.intersect_dense(one, dense, 10)
    assert torch.all(exact.get_tot_scores(True, True) >= tot_log - 1e-6)          # pruning only removes mass
    assert torch.all(exact.get_tot_scores(True, False) >= tot_best - 1e-3)          # (fp64 exact vs fp32 search rounding)
    fp64 = decode.one_best_decoding(exact, use_double_scores=True)                # forwarded (VERDICT weak 16)
    assert fp64.scores.shape == (2,)


def test_hub_with_hundreds_of_outgoing_arcs_and_log_total():
    """A topology over 300 phones used as the decoding graph: the hub's ~600 outgoing arcs are dealt over the lanes
    (arc-parallel expansion), every frame touches every state."""
    c = util.random_case(60, "mmictc", B=3, T=40, L=4, P=300, den=True)
    dense = wf.DenseFsaVec(torch.from_numpy(c["lp"]).to(DEV), torch.from_numpy(c["segs"]))
    exact = wf.best_path(c["fsa"], dense)
    out = wf.pruned_search(c["fsa"], dense, INF, 0, 0, use_double_scores=True)
    assert torch.equal(out["scores"], exact["scores"]) and torch.equal(out["arcs"], exact["arcs"])
    tot = wf.tot_scores(c["fsa"], dense, True, True).cpu().numpy()
    np.testing.assert_allclose(out["tot_log"].cpu().numpy(), tot, rtol=1e-6, atol=1e-5)
    pr = wf.pruned_search(c["fsa"], dense, 5.0, 0, 0, use_double_scores=True, return_active=True)
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path, tl, na = o.beam_search(g, c["lp"][i], T, 5.0)
        assert pr["scores"][i].item() == s and list(pr["n_active"][i, :T + 1].cpu().numpy()) == na
        np.testing.assert_allclose(pr["tot_log"][i].item(), tl, rtol=1e-6)


def test_random_graphs_with_exact_ties_against_oracle():
    """tools/fuzz_search.py (40 seeded cases): random graphs with skewed out-degrees, states of thousands of arcs,
    quantised weights and emissions (exact ties on the arg-max and on the beam boundary), random beams; also the
    max_active_states properties.  Runs as a child process: the tool is a script."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-u", os.path.join(root, "tools", "fuzz_search.py"), "40", "7", "8000"], cwd=root,
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _fan_graph(n, weights, V):
    """start --label 1--> n states, each with a self-loop (label 1) and a final arc; arcs sorted by source state."""
    arcs = [(0, 1 + i, 1, float(weights[i])) for i in range(n)]
    for i in range(n):
        arcs.append((1 + i, 1 + i, 1, 0.0))
        arcs.append((1 + i, n + 1, -1, 0.0))
    return wf.Fsa.from_arcs(arcs, n + 2)


@pytest.mark.parametrize("tied", [True, False])
def test_max_active_with_more_tied_states_than_fit(tied):
    # the histogram's best bin alone holds more than max_active candidates -- exact ties (tied), or scores a
    # hair apart while one straggler stretches the histogram's range.  The cut must never drop the frame's best candidate:
    # the best path and score stay those of the exact search, run after run.
    n, cap, T, V = 600, 100, 12, 5
    rng = np.random.default_rng(3)
    w = np.zeros(n, dtype=np.float32) if tied else (-1e-5 * rng.permutation(n)).astype(np.float32)
    if not tied:
        w[-1] = -40.0       # the straggler: bins of 0.04, every other candidate in bin 0
    fsa = wf.create_fsa_vec([_fan_graph(n, w, V)] * 2).to(DEV)
    lp = torch.log_softmax(torch.randn(2, T, V, generator=torch.Generator().manual_seed(1)), -1).to(DEV)
    dense = wf.DenseFsaVec(lp, torch.tensor([[0, 0, T], [1, 0, T - 3]], dtype=torch.int32))
    exact = wf.best_path(fsa, dense)
    outs = [wf.pruned_search(fsa, dense, search_beam=INF, min_active_states=0, max_active_states=cap, return_active=True)
            for _ in range(3)]
    for out in outs:
        assert torch.equal(out["scores"], exact["scores"])
        assert torch.equal(out["arcs"], exact["arcs"])
        assert int(out["n_active"].max()) <= cap
    na = outs[0]["n_active"][0, 1:T + 1]
    assert int(na.min()) >= (cap if tied else cap // 2)      # (not tied: the refined bins cut within a bin of the cap)


# ---------------------------------------------------------------------------------------------
# output_beam of get_lattice (wf_search_lattice): the backward pass over the lattice the search left
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("output_beam", [1.5, 5.0, INF])
@pytest.mark.parametrize("search_beam", [INF, 8.0])
@pytest.mark.parametrize("topo,den", [("ctc", False), ("mmictc", True), ("3state-skip", True), ("2state_sil", False)])
def test_output_beam_against_oracle(topo, den, search_beam, output_beam):
    c, dense = _case(topo, den, seed=52)
    out = wf.pruned_search(c["fsa"], dense, search_beam=search_beam, min_active_states=0, max_active_states=0,
                           use_double_scores=True, output_beam=output_beam)
    plain = wf.pruned_search(c["fsa"], dense, search_beam=search_beam, min_active_states=0, max_active_states=0,
                             use_double_scores=True)
    assert torch.equal(out["scores"], plain["scores"]) and torch.equal(out["arcs"], plain["arcs"])   # the best path stays
    tot, narcs = out["tot_log"].cpu().numpy(), out["lattice_arcs"].cpu().numpy()
    base = c["fsa"].fsa_arc_offsets
    for i, g in enumerate(c["ographs"]):
        T = int(c["segs"][i, 2])
        s, path, tl, na = o.lattice_search(g, c["lp"][i], T, search_beam, output_beam)
        gi = 0 if c["fsa"].utt2fsa is not None else i
        assert out["scores"][i].item() == s
        assert list(out["arcs"][i, :T + 1].cpu().numpy() - base[gi]) == path if path else True
        assert narcs[i] == na
        if np.isfinite(tl):
            np.testing.assert_allclose(tot[i], tl, rtol=1e-9, atol=1e-9)
        else:
            assert tot[i] == -np.inf
    if output_beam == INF:   # nothing on a complete path is dropped: the forward total, summed the other way round
        np.testing.assert_allclose(tot, plain["tot_log"].cpu().numpy(), rtol=1e-6, atol=1e-5)
    else:
        assert np.all(tot <= plain["tot_log"].cpu().numpy() + 1e-9)
        assert np.all(tot >= out["scores"].cpu().numpy().astype(np.float64) - 1e-4)   # the best path is in the lattice


def test_output_beam_on_a_big_graph_with_active_state_limits():
    g, lp, segs, fsa, dense = _big(40_000, B=4, T=40, seed=13)
    pr = wf.pruned_search(fsa, dense, search_beam=7.0, min_active_states=0, max_active_states=0, use_double_scores=True,
                          output_beam=3.0)
    for i in range(4):
        Ti = int(segs[i, 2])
        s, path, tl, na = o.lattice_search(g, lp[i], Ti, 7.0, 3.0)
        assert pr["scores"][i].item() == s
        assert pr["lattice_arcs"][i].item() == na
        if np.isfinite(tl):
            np.testing.assert_allclose(pr["tot_log"][i].item(), tl, rtol=1e-9)
        else:
            assert pr["tot_log"][i].item() == -np.inf
    # with max_active the survivors are the kernel's choice: the backward pass must still be consistent with the forward one
    capped = wf.pruned_search(fsa, dense, search_beam=12.0, min_active_states=20, max_active_states=1500,
                              use_double_scores=True)
    same = wf.pruned_search(fsa, dense, search_beam=12.0, min_active_states=20, max_active_states=1500,
                            use_double_scores=True, output_beam=INF)
    assert torch.equal(capped["scores"], same["scores"]) and torch.equal(capped["arcs"], same["arcs"])
    np.testing.assert_allclose(same["tot_log"].cpu().numpy(), capped["tot_log"].cpu().numpy(), rtol=1e-6)
    tight = wf.pruned_search(fsa, dense, search_beam=12.0, min_active_states=20, max_active_states=1500,
                             use_double_scores=True, output_beam=2.0)
    assert torch.all(tight["tot_log"] <= same["tot_log"] + 1e-9)
    assert torch.all(tight["lattice_arcs"] <= same["lattice_arcs"])
    assert torch.all(tight["tot_log"] >= tight["scores"].double() - 1e-4)


def test_get_lattice_applies_output_beam():
    """lattice_analysis.py:125,128: the lattice total is over the lattice that output_beam leaves."""
    g, lp, segs, fsa, dense = _big(12_001, B=2, T=30, seed=11)
    one = wf.Fsa(fsa.arcs, fsa.aux_labels, fsa.fsa_arc_offsets, fsa.fsa_num_states)
    x = torch.from_numpy(lp).to(DEV)
    lens = torch.tensor([int(segs[0, 2]), int(segs[1, 2])])
    narrow = decode.get_lattice(x, lens, one, search_beam=20.0, output_beam=2.0, min_active_states=0, max_active_states=0)
    wide = decode.get_lattice(x, lens, one, search_beam=20.0, output_beam=INF, min_active_states=0, max_active_states=0)
    tn, tw = narrow.get_tot_scores(True, True), wide.get_tot_scores(True, True)
    assert torch.all(tn < tw)
    for i in range(2):
        Ti = int(segs[i, 2])
        np.testing.assert_allclose(tn[i].item(), o.lattice_search(g, lp[i], Ti, 20.0, 2.0)[2], rtol=1e-9)
        np.testing.assert_allclose(tw[i].item(), o.beam_search(g, lp[i], Ti, 20.0)[2], rtol=1e-6)
    assert torch.equal(decode.one_best_decoding(narrow).scores, decode.one_best_decoding(wide).scores)
    t32 = narrow.get_tot_scores(False, True)                     # fp32 totals: the same launch, rounded at the end
    assert t32.dtype == torch.float32
    np.testing.assert_allclose(t32.cpu().numpy(), tn.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("drop", [1, 2, 3])
def test_a_workgroup_that_never_arrives_is_survived(drop, search_kernel_variant):
    """The barriers between an utterance's workgroups are bounded: when a partner never arrives (here: the test hook makes
    the last workgroup leave at once, in the forward kernel (1), the backward one (2) or both) the waiting workgroups give
    the utterance up after ~0.1 s and the one-workgroup launches redo it -- same results as without the hook."""
    import os
    if search_kernel_variant != "four_workgroups_every_frame_shared":
        pytest.skip("the hook only acts on the multi-workgroup kernels")
    g, lp, segs, fsa, dense = _big(20_000, B=3, T=25, seed=17)
    kw = dict(search_beam=9.0, min_active_states=0, max_active_states=0, use_double_scores=True, return_active=True,
              output_beam=4.0)
    want = wf.pruned_search(fsa, dense, **kw)
    os.environ["WF_SEARCH_TEST_DROP"] = str(drop)
    try:
        got = wf.pruned_search(fsa, dense, **kw)
    finally:
        os.environ.pop("WF_SEARCH_TEST_DROP", None)
    for k in ("scores", "arcs", "n_active", "lattice_arcs"):
        assert torch.equal(got[k], want[k]), k
    np.testing.assert_allclose(got["tot_log"].cpu().numpy(), want["tot_log"].cpu().numpy(), rtol=1e-12)
    for i in range(3):
        Ti = int(segs[i, 2])
        s, path, tl, na = o.lattice_search(g, lp[i], Ti, 9.0, 4.0)
        assert got["scores"][i].item() == s and got["lattice_arcs"][i].item() == na
        if np.isfinite(tl):
            np.testing.assert_allclose(got["tot_log"][i].item(), tl, rtol=1e-9)


def test_output_beam_with_zero_frame_segments():
    """A segment of 0 frames: the lattice is the start state's final arcs (if any)."""
    arcs = [(0, 1, 1, -0.5), (0, 2, -1, -0.25), (1, 1, 1, -0.1), (1, 2, -1, -0.75)]
    fsa = wf.create_fsa_vec([wf.Fsa.from_arcs(arcs, 3)] * 3)
    lp = torch.log_softmax(torch.randn(3, 6, 4, generator=torch.Generator().manual_seed(2)), -1)
    segs = torch.tensor([[0, 0, 0], [1, 2, 4], [2, 6, 0]], dtype=torch.int32)
    dense = wf.DenseFsaVec(lp.to(DEV), segs)
    out = wf.pruned_search(fsa, dense, search_beam=5.0, min_active_states=0, max_active_states=0, use_double_scores=True,
                           output_beam=3.0)
    g = fsa.as_oracle_graphs()
    for i in range(3):
        b, st, T = (int(x) for x in segs[i])
        s, path, tl, na = o.lattice_search(g[i], lp[b, st:].numpy(), T, 5.0, 3.0)
        assert out["scores"][i].item() == s and out["lattice_arcs"][i].item() == na
        np.testing.assert_allclose(out["tot_log"][i].item(), tl, rtol=1e-9)
    assert out["scores"][0].item() == -0.25 and out["lattice_arcs"][0].item() == 1


# ---------------------------------------------------------------------------------------------
# lattice_analysis.py:137-172: the lattice (and the decoding graph) restricted to a word sequence
# ---------------------------------------------------------------------------------------------

def _word_graph(S, B, T, seed, V=301):
    g, lp, segs, fsa, dense = _big(S, B=B, T=T, seed=seed, V=V)
    rng = np.random.default_rng(seed)
    g["aux"] = np.where(g["label"] >= 0, np.where(rng.random(g["label"].size) < 0.25, rng.integers(1, 40, g["label"].size), 0),
                        -1).astype(np.int32)
    arcs_ = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    one = wf.Fsa(arcs_, g["aux"], [0, arcs_.shape[0]], [g["num_states"]])
    return g, lp, segs, one, wf.create_fsa_vec([one] * B), dense


@pytest.mark.parametrize("search_beam,output_beam", [(7.0, 3.0), (INF, 5.0), (9.0, INF)])
def test_lattice_restricted_to_words_against_oracle(search_beam, output_beam):
    g, lp, segs, one, fsa, dense = _word_graph(12_001, B=4, T=26, seed=31)
    kw = dict(search_beam=search_beam, output_beam=output_beam, min_active_states=0, max_active_states=0)
    base = wf.pruned_search(fsa, dense, use_double_scores=True, **kw)
    words = [[w for w in ws] for ws in base["aux_labels"]]                 # the best path's words
    words[1] = words[1][:-1] + [39, 39, 39] if words[1] else [39]            # a sequence the lattice may not hold
    out = wf.lattice_words(fsa, dense, words, **kw)
    assert torch.equal(out["scores"], base["scores"]) and torch.equal(out["arcs"], base["arcs"])
    np.testing.assert_allclose(out["tot_log"].cpu().numpy(), base["tot_log"].cpu().numpy(), rtol=1e-12)
    for i in range(4):
        Ti = int(segs[i, 2])
        tw, bw = o.lattice_words(g, lp[i], Ti, words[i], search_beam, output_beam)
        assert out["best_words"][i].item() == bw
        if np.isfinite(tw):
            np.testing.assert_allclose(out["tot_words"][i].item(), tw, rtol=1e-9)
        else:
            assert out["tot_words"][i].item() == -np.inf
        if i != 1 and np.isfinite(base["scores"][i].item()):
            # the best path spells its own words (backward association of the fp32 sum: rounding apart)
            np.testing.assert_allclose(bw, base["scores"][i].item(), rtol=1e-6)
            assert tw <= base["tot_log"][i].item() + 1e-9


def test_lattice_analysis_ratios_with_the_reference_calls():
    """utils/lattice_analysis.py:125-172 with the same names: the three ratios of analyse_one."""
    import waterfall_amd as k2
    g, lp, segs, TLG, _, _ = _word_graph(12_001, B=1, T=30, seed=33)
    log_prob = torch.from_numpy(lp[:1]).to(DEV)
    log_prob_len = torch.tensor([30])
    dense_fsa_vec = k2.DenseFsaVec(log_prob, torch.tensor([[0, 0, 30]], dtype=torch.int32))
    lattice = k2.get_lattice(log_prob, log_prob_len, TLG, search_beam=12.0, output_beam=6.0, min_active_states=0,
                             max_active_states=0)
    lattice = k2.invert(k2.invert(lattice))
    lattice_total_score = lattice.get_tot_scores(True, True)
    one_best = k2.one_best_decoding(lattice)
    one_best_aux_label = k2.get_aux_labels(one_best)
    W = k2.linear_fst(one_best_aux_label, one_best_aux_label)
    W = k2.arc_sort(W)
    latticeW = k2.connect(k2.compose(lattice, W))
    TLGW = k2.connect(k2.compose(TLG, W))
    ETLGW = k2.intersect_dense(TLGW, dense_fsa_vec, 10)
    TLG_W_total_score = ETLGW.get_tot_scores(True, True)
    lattice_W_total_score = latticeW.get_tot_scores(True, True)
    lattice_W_best_score = latticeW.get_tot_scores(True, False)
    r1 = np.exp((lattice_W_total_score - lattice_total_score).item())
    r2 = np.exp((lattice_W_total_score - TLG_W_total_score).item())
    r3 = np.exp((lattice_W_best_score - lattice_W_total_score).item())
    assert 0 < r1 <= 1 + 1e-9 and 0 < r2 <= 1 + 1e-9 and 0 < r3 <= 1 + 1e-9
    words = one_best_aux_label[0]
    tw, bw = o.lattice_words(g, lp[0], 30, words, 12.0, 6.0)
    np.testing.assert_allclose(lattice_W_total_score.item(), tw, rtol=1e-9)
    assert lattice_W_best_score.item() == float(bw)
    np.testing.assert_allclose(float(bw), one_best.get_tot_scores(True, False)[0].item(), rtol=1e-6)
    gw = TLGW.as_oracle_graphs()[0]
    np.testing.assert_allclose(TLG_W_total_score.item(), o.forward(gw, lp[0], 30)[1], rtol=1e-9)
    np.testing.assert_allclose(TLG_W_total_score.item(), o.lattice_words(g, lp[0], 30, words)[0], rtol=1e-9)   # two routes
    np.testing.assert_allclose(lattice_total_score.item(), o.lattice_search(g, lp[0], 30, 12.0, 6.0)[2], rtol=1e-9)


@pytest.mark.parametrize("drop", [4, 7])
def test_words_pass_survives_a_workgroup_that_never_arrives(drop, search_kernel_variant):
    import os
    if search_kernel_variant != "four_workgroups_every_frame_shared":
        pytest.skip("the hook only acts on the multi-workgroup kernels")
    g, lp, segs, one, fsa, dense = _word_graph(12_001, B=3, T=22, seed=35)
    kw = dict(search_beam=8.0, output_beam=4.0, min_active_states=0, max_active_states=0)
    words = wf.pruned_search(fsa, dense, use_double_scores=True, **kw)["aux_labels"]
    want = wf.lattice_words(fsa, dense, words, **kw)
    os.environ["WF_SEARCH_TEST_DROP"] = str(drop)
    try:
        got = wf.lattice_words(fsa, dense, words, **kw)
    finally:
        os.environ.pop("WF_SEARCH_TEST_DROP", None)
    for k in ("scores", "arcs", "best_words"):
        assert torch.equal(got[k], want[k]), k
    np.testing.assert_allclose(got["tot_words"].cpu().numpy(), want["tot_words"].cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(got["tot_log"].cpu().numpy(), want["tot_log"].cpu().numpy(), rtol=1e-12)
    for i in range(3):
        tw, bw = o.lattice_words(g, lp[i], int(segs[i, 2]), words[i], 8.0, 4.0)
        assert got["best_words"][i].item() == bw
        if np.isfinite(tw):
            np.testing.assert_allclose(got["tot_words"][i].item(), tw, rtol=1e-9)
