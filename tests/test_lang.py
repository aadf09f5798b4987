"""Lang-dir reader / writer (datapipe.py:69-271 without Kaldi / OpenFst) and BASELINE.md cfg1: the yesno CTC recipe
(phones and lexicon of egs/yesno/asr/input, B = 8 utterances of 8 words, T = 100)."""
import numpy as np
import pytest
import torch

from oracle import dwfst_oracle as o
import waterfall_amd as wf

# contents of /root/reference/egs/yesno/asr/input/phones.txt and lexicon.txt (data, 11 lines)
YESNO_PHONES = ["<UNK>", "<SIL>", "Y", "E", "S", "N", "O"]
YESNO_LEXICON = [("<SIL>", ["<SIL>"]), ("<UNK>", ["<UNK>"]), ("YES", ["Y", "E", "S"]), ("NO", ["N", "O"])]
DEV = "cuda:0"


@pytest.mark.parametrize("topo", sorted(wf.TOPOLOGIES))
def test_lang_dir_round_trip(tmp_path, topo):
    d = wf.write_lang_dir(str(tmp_path), YESNO_LEXICON, YESNO_PHONES, topo)
    lang = wf.Lang(d, load_topo=True)
    t = wf.Topology(topo, ["<eps>"] + YESNO_PHONES)
    assert np.array_equal(lang.topo.arcs, t.fsa.arcs) and np.array_equal(lang.topo.aux_labels, t.fsa.aux_labels)
    assert lang.tokens == t.tokens and lang.num_nn_output == len(t.tokens)
    assert lang.words[0] == "<eps>" and lang.phones[0] == "<eps>" and lang.phones[-1] == "#0"
    yes, no = lang.word2idx["YES"], lang.word2idx["NO"]
    assert lang.get_pids_list_from_wids_list([[yes, no]]) == [[lang.phone2idx[p] for p in "YESNO"]]
    assert lang.load_lexicon() is None          # neither k2/L_inv.pt nor L.fst.txt: the align lexicon stands in


def _yesno_batch(tmp_path, seed=777, B=8, T=100):
    d = wf.write_lang_dir(str(tmp_path), YESNO_LEXICON, YESNO_PHONES, "ctc")
    lang = wf.Lang(d, load_topo=True)
    assert lang.num_nn_output == 7 and lang.tokens[0] == "<blk>"          # P = 6, V = 7
    rng = np.random.default_rng(seed)
    wids = [[lang.word2idx["YES" if x else "NO"] for x in rng.integers(0, 2, size=8)] for _ in range(B)]
    graphs = lang.compile_training_graph(wids)
    g = torch.Generator().manual_seed(seed)
    lp = torch.log_softmax(torch.randn(B, T, lang.num_nn_output, generator=g) * 3.0, -1)
    segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), np.full(B, T)], 1).astype(np.int32)
    return lang, wids, graphs, lp, segs


def test_cfg1_yesno_ctc_numerators_equal_torch_ctc_loss(tmp_path):
    lang, wids, graphs, lp, segs = _yesno_batch(tmp_path)
    pids = lang.get_pids_list_from_wids_list(wids)
    assert all(16 <= len(p) <= 24 for p in pids)                           # L ~ 20
    want, wgrad = o.graphloss(graphs.as_oracle_graphs(), lp.numpy(), segs, None, "none")
    # the ctc topology over k2/phones.txt is standard CTC on the token ids of the phones
    tok = [[lang.token2idx[lang.idx2phone[p]] for p in ps] for ps in pids]
    x = lp.clone().requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(x.permute(1, 0, 2), torch.tensor([t for ts in tok for t in ts]),
                                       torch.full((8,), 100), torch.tensor([len(t) for t in tok]), blank=0,
                                       reduction="none")
    np.testing.assert_allclose(want, ref.detach().numpy(), rtol=1e-5)
    ref.sum().backward()
    # torch's ctc backward assumes its input is a log_softmax output and returns exp(lp) - occupation
    np.testing.assert_allclose(wgrad, (x.grad - lp.exp()).numpy(), atol=3e-4)   # torch computes in fp32


@pytest.mark.gpu
def test_cfg1_yesno_on_the_gpu_numerator_and_denominator(tmp_path):
    lang, wids, graphs, lp, segs = _yesno_batch(tmp_path)
    x = lp.to(DEV).requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(segs))
    num = wf.graphloss(graphs, dense, output_beam=10, reduction="none")
    den = wf.graphloss(wf.create_fsa_vec([lang.topo] * 8), dense, output_beam=10, reduction="none")
    (num.sum() - den.sum()).backward()
    wn, gn = o.graphloss(graphs.as_oracle_graphs(), lp.numpy(), segs, None, "none")
    tg = lang.topo.as_oracle_graphs()[0]
    wd, gd = o.graphloss([tg] * 8, lp.numpy(), segs, None, "none")
    np.testing.assert_allclose(num.detach().cpu().numpy(), wn, rtol=1e-4)
    np.testing.assert_allclose(den.detach().cpu().numpy(), wd, rtol=1e-4, atol=1e-4)   # the CTC topology accepts every path: tot = 0
    np.testing.assert_allclose(x.grad.cpu().numpy(), gn - gd, atol=1e-3)


# ---- word-level graphs: alternative pronunciations + optional silence (datapipe.py:244-254 via L_inv) --------------

LEX_VARIANTS = YESNO_LEXICON + [("YES", ["Y", "E"]), ("O", ["O"])]


def _accepted_strings(g, limit=200000):
    """All label strings of an acyclic oracle graph."""
    out_arcs = {}
    for s, d, l in zip(g["src"], g["dst"], g["label"]):
        out_arcs.setdefault(int(s), []).append((int(d), int(l)))
    res, stack = set(), [(0, ())]
    while stack:
        s, pre = stack.pop()
        for d, l in out_arcs.get(s, []):
            if l < 0:
                res.add(pre)
            else:
                stack.append((d, pre + (l,)))
        assert len(res) < limit
    return res


def test_word_graphs_accept_exactly_the_lexicon_expansions(tmp_path):
    d = wf.write_lang_dir(str(tmp_path), LEX_VARIANTS, YESNO_PHONES, "2state_sil")
    lang = wf.Lang(d, load_topo=True)
    P = lang.phone2idx
    wids = [[lang.word2idx["YES"], lang.word2idx["NO"]]]
    g = lang.word_graphs(wids, "<SIL>").as_oracle_graphs()[0]
    want = set()
    sil = [(), (P["<SIL>"],)]
    for a in sil:
        for yes in ((P["Y"], P["E"], P["S"]), (P["Y"], P["E"])):
            for b in sil:
                for c in sil:
                    want.add(a + yes + b + (P["N"], P["O"]) + c)
    assert _accepted_strings(g) == want
    # word ids sit on the first arc of each pronunciation
    words = sorted(int(x) for x in g["aux"] if x > 0)
    assert words == sorted([lang.word2idx["YES"]] * 4 + [lang.word2idx["NO"]] * 2)
    # weights: log(0.5) on silence arcs / where silence is skipped
    gw = lang.word_graphs(wids, "<SIL>", 0.5).as_oracle_graphs()[0]
    assert set(gw["score"].tolist()) == {0.0, float(np.float32(np.log(0.5)))}


@pytest.mark.parametrize("topo", ["ctc", "2state", "2state_sil", "2state_sil3", "mmictc", "mmictc_noblk", "3state-skip-2"])
def test_general_compose_equals_oracle_and_linear_case(tmp_path, topo):
    from oracle import graphs as og
    d = wf.write_lang_dir(str(tmp_path), LEX_VARIANTS, YESNO_PHONES, topo)
    lang = wf.Lang(d, load_topo=True)
    wids = [[lang.word2idx[w] for w in ws] for ws in (["YES", "NO"], ["O"], ["NO", "O", "YES"], [])]
    tg = lang.topo.as_oracle_graphs()[0]
    for sp in (None, 0.3):
        bg = lang.word_graphs(wids, "<SIL>", sp)
        g = lang.compile_training_graph(wids, optional_silence="<SIL>", sil_prob=sp)
        for got, bo in zip(g.as_oracle_graphs(), bg.as_oracle_graphs()):
            want = og.compose_topo_fsa(tg, bo)
            assert got["num_states"] == want["num_states"]
            for k in ("src", "dst", "label", "aux", "score"):
                assert np.array_equal(got[k], want[k]), (topo, k)
    pids = lang.get_pids_list_from_wids_list(wids[:3])
    lin = wf.create_fsa_vec([wf.Fsa.from_arcs([(i, i + 1, p, 0.0) for i, p in enumerate(ps)] + [(len(ps), len(ps) + 1, -1, 0.0)],
                                              len(ps) + 2, list(ps) + [-1]) for ps in pids])
    a, b = wf.compose(lang.topo, lin), wf.compile_ctc_k2_graphs(lang.topo, pids)
    assert np.array_equal(a.arcs, b.arcs) and np.array_equal(a.aux_labels, b.aux_labels)
    assert np.array_equal(a.fsa_num_states, b.fsa_num_states)


@pytest.mark.gpu
def test_word_level_numerators_on_the_gpu(tmp_path):
    d = wf.write_lang_dir(str(tmp_path), LEX_VARIANTS, YESNO_PHONES, "2state_sil")
    lang = wf.Lang(d, load_topo=True)
    rng = np.random.default_rng(11)
    names = ["YES", "NO", "O"]
    wids = [[lang.word2idx[names[x]] for x in rng.integers(0, 3, size=n)] for n in (6, 3, 9, 1)]
    graphs = lang.compile_training_graph(wids, optional_silence="<SIL>", sil_prob=0.5)
    B, T, V = 4, 70, lang.num_nn_output
    gen = torch.Generator().manual_seed(3)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 2.0, -1)
    segs = np.array([[0, 0, 70], [1, 0, 41], [2, 0, 66], [3, 0, 9]], dtype=np.int32)
    x = lp.to(DEV).requires_grad_(True)
    loss = wf.graphloss(graphs, wf.DenseFsaVec(x, torch.from_numpy(segs)), reduction="none")
    loss.sum().backward()
    want, wgrad = o.graphloss(graphs.as_oracle_graphs(), lp.numpy(), segs, None, "none")
    np.testing.assert_allclose(loss.detach().cpu().numpy(), want, rtol=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), wgrad, atol=1e-3)
    best = wf.best_path(graphs, wf.DenseFsaVec(x.detach(), torch.from_numpy(segs)))
    # the aux labels along the best path spell the word sequence
    assert best["aux_labels"] == wids


# ---- the default route of compile_training_graph is the reference's (datapipe.py:233-256): every pronunciation, L.fst's
# ---- optional silences.  k2's arc weights after remove_epsilon / its numbering: parity unpinned (no k2 here). ---------

def _weighted_language(g):
    """phone string -> log-sum of the scores of its paths (acyclic oracle graph)."""
    out_arcs = {}
    for s, d, l, w in zip(g["src"], g["dst"], g["label"], g["score"]):
        out_arcs.setdefault(int(s), []).append((int(d), int(l), float(w)))
    res, stack = {}, [(0, (), 0.0)]
    while stack:
        s, pre, w = stack.pop()
        for d, l, x in out_arcs.get(s, []):
            if l < 0:
                res[pre] = np.logaddexp(res.get(pre, -np.inf), w + x)
            else:
                stack.append((d, pre + (l,), w + x))
    return res


def test_default_training_graph_has_every_pronunciation_and_the_optional_silence(tmp_path):
    d = wf.write_lang_dir(str(tmp_path / "a"), LEX_VARIANTS, YESNO_PHONES, "2state_sil", optional_silence="<SIL>")
    lang = wf.Lang(d, load_topo=True)
    wids = [[lang.word2idx["YES"], lang.word2idx["NO"]], [lang.word2idx["O"]]]
    got = lang.compile_training_graph(wids)                                   # no arguments: the reference's call
    want = wf.compose(lang.topo, lang.word_graphs(wids, "<SIL>", 0.5))
    assert np.array_equal(got.arcs, want.arcs) and np.array_equal(got.aux_labels, want.aux_labels)
    # the old short cut (last pronunciation only, no silence) is a strictly narrower graph
    narrow = wf.compile_ctc_k2_graphs(lang.topo, lang.get_pids_list_from_wids_list(wids))
    assert got.fsa_num_states[0] > narrow.fsa_num_states[0]
    # variants but no optional silence in the lang dir: still both pronunciations
    d2 = wf.write_lang_dir(str(tmp_path / "b"), LEX_VARIANTS, YESNO_PHONES, "ctc")
    lang2 = wf.Lang(d2, load_topo=True)
    P = lang2.phone2idx
    bg = lang2.word_graphs([[lang2.word2idx["YES"]]], None).as_oracle_graphs()[0]
    assert _accepted_strings(bg) == {(P["Y"], P["E"], P["S"]), (P["Y"], P["E"])}
    g2 = lang2.compile_training_graph([[lang2.word2idx["YES"]]])
    w2 = wf.compose(lang2.topo, lang2.word_graphs([[lang2.word2idx["YES"]]], None))
    assert np.array_equal(g2.arcs, w2.arcs)
    # one pronunciation per word and no silence: the linear route, identical graphs
    g3 = lang2.compile_training_graph([[lang2.word2idx["NO"], lang2.word2idx["O"]]])
    w3 = wf.compile_ctc_k2_graphs(lang2.topo, [[P["N"], P["O"], P["O"]]])
    assert np.array_equal(g3.arcs, w3.arcs) and np.array_equal(g3.fsa_num_states, w3.fsa_num_states)


def test_lexicon_transducer_route_equals_the_align_lexicon_route(tmp_path):
    """load_lexicon (L.fst.txt, then the k2/L_inv.pt dict) + intersect / invert / remove_epsilon on the host against the
    graphs built straight from the align lexicon: the same weighted set of phone strings per utterance."""
    d = wf.write_lang_dir(str(tmp_path), LEX_VARIANTS, YESNO_PHONES, "2state_sil", optional_silence="<SIL>", sil_prob=0.3,
                          write_L=True)
    lang = wf.Lang(d, load_topo=True, load_lexicon=True)
    assert lang.L is not None and lang.L_inv is not None
    names = ["YES", "NO", "O"]
    wids = [[lang.word2idx[n] for n in ws] for ws in (["YES", "NO"], ["O"], ["NO", "O", "YES"], [])]
    via_L = lang._lexicon_graphs(wids).as_oracle_graphs()
    direct = lang.word_graphs(wids, "<SIL>", 0.3).as_oracle_graphs()
    for a, b, ws in zip(via_L, direct, wids):
        la, lb = _weighted_language(a), _weighted_language(b)
        assert set(la) == set(lb) and len(la) >= 2
        for k in la:
            np.testing.assert_allclose(la[k], lb[k], atol=1e-5)
        assert {int(x) for x in a["aux"] if x > 0} == {int(x) for x in b["aux"] if x > 0} == set(ws)
    g = lang.compile_training_graph(wids)                 # default: the loaded transducer's route
    assert g.num_fsas == 4
    # the L_inv.pt dict layout (torch.save(L_inv.as_dict()), datapipe.py:204), weights-only
    torch.save({"arcs": torch.from_numpy(lang.L_inv.arcs.copy()), "aux_labels": torch.from_numpy(lang.L_inv.aux_labels.copy())},
               str(tmp_path / "k2" / "L_inv.pt"))
    again = wf.Lang(d, load_topo=True, load_lexicon=True)
    assert np.array_equal(again.L_inv.arcs, lang.L_inv.arcs) and np.array_equal(again.L.aux_labels, lang.L.aux_labels)
    for a, b in zip(again._lexicon_graphs(wids).as_oracle_graphs(), via_L):
        assert np.array_equal(a["label"], b["label"]) and np.array_equal(a["score"], b["score"])
