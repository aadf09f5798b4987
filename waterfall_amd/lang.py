"""Lang-dir reader of the reference (``waterfall/utils/datapipe.py:69-271`` ``class Lang``) without Kaldi / OpenFst
binaries, plus a writer for the files it reads (what ``utils/prepare_<topo>.sh`` leaves under ``lang/k2``).

    lang = Lang(lang_dir, load_topo=True)
    graphs = lang.compile_training_graph(word_ids_list, device)        # models.py:294-295
    den = create_fsa_vec([lang.topo] * B)                              # models.py:321-322

Same attribute names as the reference (``words, phones, tokens, num_nn_output, word2idx, idx2word, phone2idx,
idx2phone, token2idx, idx2token, lexicon, topo``).  Differences, all because neither OpenFst nor k2 exist here:

* ``load_topo`` reads ``k2/T.fst.txt`` -- the text ``fstprint k2/T.fst`` prints (the reference pipes the binary
  through the ``fstprint`` executable, datapipe.py:175-180); ``write_lang_dir`` writes that file, and on a box with
  OpenFst it is one ``fstprint`` away.
* ``compile_training_graph`` expands words through ``phones/align_lexicon.int`` (``get_pids_list_from_wids_list``,
  datapipe.py:258-271) and composes the phone sequence with the topology in one native call.  The reference goes
  through ``L_inv`` (datapipe.py:244-254), which adds L.fst's optional silences and alternative pronunciations;
  that graph needs k2's ``intersect`` / ``remove_epsilon`` and is not rebuilt here (second
  half: parity unpinned, no k2 to pin it against).
"""
import os

import numpy as np

from .compile import compile_ctc_k2_graphs
from .fsa import Fsa
from .topology import Topology, TOPOLOGIES

__all__ = ["Lang", "write_lang_dir", "read_list"]


def read_list(path):
    """First column of every line (datapipe.py:20-31)."""
    with open(path) as f:
        return [line.strip().split()[0] for line in f if line.strip()]


class Lang:
    def __init__(self, lang_dir, load_topo=False, load_lexicon=False):
        self._lang_dir = lang_dir
        self.words = read_list(os.path.join(lang_dir, "words.txt"))          # with <eps> and disambig
        self.phones = read_list(os.path.join(lang_dir, "phones.txt"))        # with <eps> and disambig
        self.tokens = read_list(os.path.join(lang_dir, "k2", "tokens.txt"))  # the NN outputs
        self.num_nn_output = len(self.tokens)
        self.word2idx = {w: i for i, w in enumerate(self.words)}
        self.idx2word = dict(enumerate(self.words))
        self.phone2idx = {p: i for i, p in enumerate(self.phones)}
        self.idx2phone = dict(enumerate(self.phones))
        self.token2idx = {t: i for i, t in enumerate(self.tokens)}
        self.idx2token = dict(enumerate(self.tokens))
        self.topo = None
        self.load_align_lexicon()
        if load_topo:
            self.load_topo()
        if load_lexicon:
            self.load_lexicon()

    def load_align_lexicon(self):
        """phones/align_lexicon.int: ``wid wid pid pid ...`` (datapipe.py:133-147; a later line for the same word
        replaces an earlier one, as there)."""
        self.lexicon = {}
        with open(os.path.join(self._lang_dir, "phones", "align_lexicon.int")) as f:
            for line in f:
                lc = line.strip().split()
                if lc:
                    self.lexicon[int(lc[0])] = [int(x) for x in lc[2:]]

    def load_topo(self):
        """k2/T.fst.txt -> self.topo (k2.Fsa.from_openfst(text, acceptor=False), datapipe.py:168-181)."""
        path = os.path.join(self._lang_dir, "k2", "T.fst.txt")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: print the topology with `fstprint {self._lang_dir}/k2/T.fst > {path}` "
                                    "(OpenFst) or write the lang dir with waterfall_amd.lang.write_lang_dir")
        with open(path) as f:
            self.topo = Fsa.from_openfst(f.read(), acceptor=False)
        return self.topo

    def load_lexicon(self):
        raise NotImplementedError("L_inv (k2/L_inv.pt or L.fst, datapipe.py:183-204) needs k2 / OpenFst; "
                                  "compile_training_graph here expands words through phones/align_lexicon.int")

    def get_pids_list_from_wids_list(self, wids_list):
        """datapipe.py:258-271."""
        return [[p for w in wids for p in self.lexicon[w]] for wids in wids_list]

    def compile_training_graph(self, word_ids_list, device=None, optional_silence=None, sil_prob=None):
        """Numerator graphs of a batch of word-id sequences (models.py:294-295).

        Default: words -> phones through the align lexicon (last pronunciation of a word, as self.lexicon keeps it),
        then topo o linear(phones) for the whole batch in one native call.
        optional_silence=<phone symbol or id>: the phone-level graph of each utterance is built as Kaldi's L.fst would
        give it through L_inv (datapipe.py:244-254) -- every pronunciation of every word (all lines of
        align_lexicon.int) in parallel, an optional silence phone before the first word and after every word -- and
        composed with the topology (wf_compose_fsa_batch).  sil_prob: None = unweighted arcs; else log(sil_prob) on a
        silence arc and log(1 - sil_prob) where it is skipped (make_lexicon_fst.pl).  The arc weights and state
        numbering of k2's route (intersect / invert / remove_epsilon) are not pinned here: k2 is not available."""
        if self.topo is None:
            self.load_topo()
        if optional_silence is None:
            g = compile_ctc_k2_graphs(self.topo, self.get_pids_list_from_wids_list(word_ids_list))
        else:
            from .compile import compose
            g = compose(self.topo, self.word_graphs(word_ids_list, optional_silence, sil_prob))
        return g.to(device) if device is not None else g

    def load_pronunciations(self):
        """All lines of phones/align_lexicon.int: word id -> list of pronunciations."""
        prons = {}
        with open(os.path.join(self._lang_dir, "phones", "align_lexicon.int")) as f:
            for line in f:
                lc = line.strip().split()
                if lc:
                    p = [int(x) for x in lc[2:]]
                    if p not in prons.setdefault(int(lc[0]), []):
                        prons[int(lc[0])].append(p)
        return prons

    def word_graphs(self, word_ids_list, optional_silence, sil_prob=None):
        """Epsilon-free phone-level graphs of word sequences (labels = phones, aux_labels = word ids on the first arc
        of a pronunciation): states b_i (after word i) and s_i (after word i and a silence); see compile_training_graph."""
        import math
        from .fsa import create_fsa_vec
        sil = self.phone2idx[optional_silence] if isinstance(optional_silence, str) else int(optional_silence)
        w_sil = math.log(sil_prob) if sil_prob else 0.0
        w_no = math.log(1.0 - sil_prob) if sil_prob else 0.0
        prons = self.load_pronunciations()
        out = []
        for wids in word_ids_list:
            n = len(wids)
            arcs, aux = [], []
            nxt = [2 * (n + 1)]                  # b_i = 2 i, s_i = 2 i + 1, then the states inside pronunciations

            def new_state():
                nxt[0] += 1
                return nxt[0] - 1
            for i in range(n + 1):
                arcs.append((2 * i, 2 * i + 1, sil, w_sil)); aux.append(0)
            for i, w in enumerate(wids):
                for p in prons[w]:
                    if not p:
                        raise ValueError(f"word {w} has an empty pronunciation")
                    end = 2 * (i + 1)
                    first_dst = end if len(p) == 1 else new_state()
                    arcs.append((2 * i, first_dst, p[0], w_no)); aux.append(w)
                    arcs.append((2 * i + 1, first_dst, p[0], 0.0)); aux.append(w)
                    cur = first_dst
                    for j in range(1, len(p)):
                        d = end if j == len(p) - 1 else new_state()
                        arcs.append((cur, d, p[j], 0.0)); aux.append(0)
                        cur = d
            fin = nxt[0]
            arcs.append((2 * n, fin, -1, w_no)); aux.append(-1)
            arcs.append((2 * n + 1, fin, -1, 0.0)); aux.append(-1)
            out.append(Fsa.from_arcs(arcs, fin + 1, aux))
        return create_fsa_vec(out)


def write_lang_dir(lang_dir, lexicon, phones, topo, oov="<UNK>"):
    """Writes the files ``Lang`` reads, following Kaldi's ``prepare_lang.sh`` id conventions and
    ``utils/prepare_<topo>.sh`` (e.g. prepare_ctc.sh:20-39, prepare_2state.sh:20-38):

    lexicon: list of (word, [phones]) -- lexicon.txt lines; phones: phone symbols in phones.txt order (no <eps>);
    topo: a name in waterfall_amd.TOPOLOGIES.
      words.txt        <eps> 0, the words sorted, #0, <s>, </s>
      phones.txt       <eps> 0, the phones, #0 (disambiguation symbols last)
      oov.txt / oov.int
      phones/align_lexicon.txt|int     ``w w p1 p2 ...``
      k2/phones.txt    phones.txt without the disambiguation symbols (same ids)
      k2/tokens.txt    the NN outputs, <blk> (or <SIL>) = 0
      k2/T.fst.txt     the training topology as ``fstprint`` text: arcs ``src dst token phone``, arc-sorted by phone
    """
    if topo not in TOPOLOGIES:
        raise KeyError(f"unknown topology {topo!r}")
    os.makedirs(os.path.join(lang_dir, "phones"), exist_ok=True)
    os.makedirs(os.path.join(lang_dir, "k2"), exist_ok=True)
    words = ["<eps>"] + sorted({w for w, _ in lexicon}) + ["#0", "<s>", "</s>"]
    plist = ["<eps>"] + list(phones)
    w2i = {w: i for i, w in enumerate(words)}
    p2i = {p: i for i, p in enumerate(plist)}

    def dump(rel, lines):
        with open(os.path.join(lang_dir, rel), "w") as f:
            f.write("".join(line + "\n" for line in lines))

    dump("words.txt", [f"{w} {i}" for i, w in enumerate(words)])
    dump("phones.txt", [f"{p} {i}" for i, p in enumerate(plist)] + [f"#0 {len(plist)}"])
    dump("oov.txt", [oov])
    dump("oov.int", [str(w2i[oov])] if oov in w2i else [])
    dump("phones/align_lexicon.txt", [" ".join([w, w] + list(ps)) for w, ps in lexicon])
    dump("phones/align_lexicon.int", [" ".join(str(x) for x in [w2i[w], w2i[w]] + [p2i[p] for p in ps]) for w, ps in lexicon])
    dump("k2/phones.txt", [f"{p} {i}" for i, p in enumerate(plist)])
    t = Topology(topo, plist)
    dump("k2/tokens.txt", [f"{tok} {i}" for i, tok in enumerate(t.tokens)])
    dump("k2/topo_name", [topo])
    f = t.fsa
    S = int(f.fsa_num_states[0])
    lines = []
    for (s, d, lab, _), aux in zip(f.arcs.tolist(), f.aux_labels.tolist()):
        lines.append(f"{s}\t{d}\t{lab}\t{aux}" if lab >= 0 else f"{s}")
    assert all(d == S - 1 for (_, d, lab, _) in f.arcs.tolist() if lab < 0)
    dump("k2/T.fst.txt", lines)
    return lang_dir
