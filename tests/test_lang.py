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
    with pytest.raises(NotImplementedError):
        lang.load_lexicon()


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
