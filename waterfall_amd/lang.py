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
* ``load_lexicon`` reads ``k2/L_inv.pt`` (the dict ``torch.save(L_inv.as_dict())`` leaves, datapipe.py:183-204) with
  ``torch.load(weights_only=True)``, or ``L.fst.txt`` (the text ``fstprint L.fst`` prints), and keeps ``L`` / ``L_inv``
  as Fsa objects.  Without either file the lexicon transducer is implied by ``phones/align_lexicon.int`` and
  ``phones/optional_silence.int`` (what Kaldi's make_lexicon_fst.pl builds L.fst from).
* ``compile_training_graph(word_ids_list, device)`` follows datapipe.py:233-256: every pronunciation of every word and
  L.fst's optional silences are in the numerator graph.  With a loaded ``L_inv`` the chain intersect(L_inv, words) ->
  invert -> remove_epsilon -> compose(topo, .) is carried out on the host (``_lexicon_graphs``); otherwise the same
  phone-level graphs are built directly from the align lexicon (``word_graphs``).  Only a lexicon with ONE pronunciation
  per word and no optional silence takes the linear short cut (one native call, identical graphs).  k2's arc weights
  after remove_epsilon and its state numbering are parity unpinned (no k2 here); the accepted phone sequences and
  their total weights are tested against a brute-force expansion (tests/test_lang.py).
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
        """k2/L_inv.pt -> self.L_inv / self.L (datapipe.py:183-204), else L.fst.txt (`fstprint L.fst`), else None: the
        lexicon implied by phones/align_lexicon.int is then used by compile_training_graph."""
        import torch
        pt = os.path.join(self._lang_dir, "k2", "L_inv.pt")
        txt = os.path.join(self._lang_dir, "L.fst.txt")
        self.L = self.L_inv = None
        if os.path.exists(pt):
            try:
                d = torch.load(pt, map_location="cpu", weights_only=True)
            except Exception as e:
                raise ValueError(f"{pt}: not loadable with torch.load(weights_only=True) ({type(e).__name__})") from e
            arcs = d["arcs"].to(torch.int32).numpy()
            aux = d["aux_labels"].to(torch.int32).numpy()
            ns = int(arcs[:, :2].max()) + 1
            self.L_inv = Fsa(arcs, aux, [0, arcs.shape[0]], [ns])
            self.L = _invert(self.L_inv)
        elif os.path.exists(txt):
            with open(txt) as f:
                self.L = Fsa.from_openfst(f.read(), acceptor=False)
            self.L_inv = _invert(self.L)
        return self.L_inv

    def get_pids_list_from_wids_list(self, wids_list):
        """datapipe.py:258-271."""
        return [[p for w in wids for p in self.lexicon[w]] for wids in wids_list]

    def _optional_silence(self):
        path = os.path.join(self._lang_dir, "phones", "optional_silence.int")
        if os.path.exists(path):
            with open(path) as f:
                t = f.read().split()
            if t:
                return int(t[0])
        return None

    def compile_training_graph(self, word_ids_list, device=None, optional_silence="auto", sil_prob="auto"):
        """Numerator graphs of a batch of word-id sequences (models.py:294-295, datapipe.py:233-256): every
        pronunciation of every word, optional silence before the first word and after every word.

        * a loaded L_inv (load_lexicon=True and k2/L_inv.pt or L.fst.txt present): the reference's chain on that transducer;
        * else the lexicon of phones/align_lexicon.int (all its lines) with the optional-silence phone of
          phones/optional_silence.int (sil_prob 0.5, prepare_lang.sh's default; ``optional_silence=None`` switches the
          silence off, a phone symbol / id selects another one; sil_prob=None: unweighted);
        * one pronunciation per word and no silence: topo o linear(phones), one native call (the same graphs).
        """
        if self.topo is None:
            self.load_topo()
        from .compile import compose
        if getattr(self, "L_inv", None) is not None and optional_silence == "auto":
            g = compose(self.topo, self._lexicon_graphs(word_ids_list))
            return g.to(device) if device is not None else g
        sil = self._optional_silence() if optional_silence == "auto" else optional_silence
        if isinstance(sil, str):
            sil = self.phone2idx[sil]
        if sil_prob == "auto":
            sil_prob = 0.5 if sil is not None else None
        prons = self.load_pronunciations()
        used = {w for wids in word_ids_list for w in wids}
        if sil is None and all(len(prons[w]) == 1 for w in used):
            g = compile_ctc_k2_graphs(self.topo, [[p for w in wids for p in prons[w][0]] for wids in word_ids_list])
        else:
            g = compose(self.topo, self.word_graphs(word_ids_list, sil, sil_prob))
        return g.to(device) if device is not None else g

    def load_pronunciations(self):
        """All lines of phones/align_lexicon.int: word id -> list of pronunciations (cached)."""
        prons = getattr(self, "_prons", None)
        if prons is not None:
            return prons
        prons = {}
        with open(os.path.join(self._lang_dir, "phones", "align_lexicon.int")) as f:
            for line in f:
                lc = line.strip().split()
                if lc:
                    p = [int(x) for x in lc[2:]]
                    if p not in prons.setdefault(int(lc[0]), []):
                        prons[int(lc[0])].append(p)
        self._prons = prons
        return prons

    def word_graphs(self, word_ids_list, optional_silence, sil_prob=None):
        """Epsilon-free phone-level graphs of word sequences (labels = phones, aux_labels = word ids on the first arc
        of a pronunciation), the language of make_lexicon_fst.pl's L.fst restricted to the word sequence: states b_i
        (after word i) and, with an optional silence, s_i (after word i and a silence)."""
        import math
        from .fsa import create_fsa_vec
        sil = None if optional_silence is None else (
            self.phone2idx[optional_silence] if isinstance(optional_silence, str) else int(optional_silence))
        w_sil = math.log(sil_prob) if (sil_prob and sil is not None) else 0.0
        w_no = math.log(1.0 - sil_prob) if (sil_prob and sil is not None) else 0.0
        prons = self.load_pronunciations()
        out = []
        for wids in word_ids_list:
            n = len(wids)
            arcs, aux = [], []
            nxt = [2 * (n + 1)]                  # b_i = 2 i, s_i = 2 i + 1, then the states inside pronunciations

            def new_state():
                nxt[0] += 1
                return nxt[0] - 1
            if sil is not None:
                for i in range(n + 1):
                    arcs.append((2 * i, 2 * i + 1, sil, w_sil)); aux.append(0)
            for i, w in enumerate(wids):
                for p in prons[w]:
                    if not p:
                        raise ValueError(f"word {w} has an empty pronunciation")
                    end = 2 * (i + 1)
                    first_dst = end if len(p) == 1 else new_state()
                    arcs.append((2 * i, first_dst, p[0], w_no)); aux.append(w)
                    if sil is not None:
                        arcs.append((2 * i + 1, first_dst, p[0], 0.0)); aux.append(w)
                    cur = first_dst
                    for j in range(1, len(p)):
                        d = end if j == len(p) - 1 else new_state()
                        arcs.append((cur, d, p[j], 0.0)); aux.append(0)
                        cur = d
            fin = nxt[0]
            arcs.append((2 * n, fin, -1, w_no)); aux.append(-1)
            if sil is not None:
                arcs.append((2 * n + 1, fin, -1, 0.0)); aux.append(-1)
            out.append(_connect(arcs, aux, fin + 1))
        return out[0] if len(out) == 1 else create_fsa_vec(out)     # (one graph per utterance, never a shared one)

    def _lexicon_graphs(self, word_ids_list):
        """datapipe.py:244-252 on the host: intersect(L_inv, linear(words) + epsilon self-loops), invert (phones in,
        words out), remove the phone epsilons (tropical closure, as k2.remove_epsilon), trim.  One epsilon-free phone
        graph per utterance, aux_labels = word ids."""
        from .fsa import create_fsa_vec
        L = self.L
        idx = getattr(self, "_L_index", None)
        if idx is None:
            S = int(L.fsa_num_states[0])
            by_word = {}            # (state, word id) -> arcs; word 0 = no output
            finals = {}
            sc = L.scores
            for a, (s, d, ph, _) in enumerate(L.arcs.tolist()):
                if ph == -1:
                    finals[s] = max(finals.get(s, -np.inf), float(sc[a]))
                else:
                    by_word.setdefault((s, int(L.aux_labels[a])), []).append((d, ph, float(sc[a])))
            idx = self._L_index = (by_word, finals, S)
        by_word, finals, _ = idx
        out = []
        for wids in word_ids_list:
            n = len(wids)
            ids, arcs = {(0, 0): 0}, []       # product states (L state, words consumed); arcs (src, dst, phone, word, w)
            queue = [(0, 0)]
            while queue:
                q, i = queue.pop()
                src = ids[(q, i)]
                steps = [(0, i)] + ([(wids[i], i + 1)] if i < n else [])
                for word, j in steps:
                    for d, ph, w in by_word.get((q, word), ()):
                        if (d, j) not in ids:
                            ids[(d, j)] = len(ids)
                            queue.append((d, j))
                        arcs.append((src, ids[(d, j)], ph, word, w))
            fin = [(ids[(q, n)], w) for q, w in finals.items() if (q, n) in ids]
            out.append(_remove_epsilon(arcs, fin, len(ids)))
        return out[0] if len(out) == 1 else create_fsa_vec(out)


def _invert(f):
    """k2.Fsa.invert for tensor aux labels: labels <-> aux_labels (final arcs keep -1 / -1)."""
    arcs = f.arcs.copy()
    aux = f.aux_labels.copy()
    em = arcs[:, 2] >= 0
    lab = arcs[:, 2].copy()
    arcs[em, 2] = aux[em]
    aux[em] = lab[em]
    return Fsa(arcs, aux, f.fsa_arc_offsets, f.fsa_num_states)


def _connect(arcs, aux, num_states):
    """Keeps the states that lie on a start -> final path (k2.connect) and renumbers them; the final state stays last."""
    fwd, bwd = {}, {}
    for s, d, _, _ in arcs:
        fwd.setdefault(s, []).append(d)
        bwd.setdefault(d, []).append(s)

    def reach(start, nb):
        seen, st = {start}, [start]
        while st:
            for y in nb.get(st.pop(), ()):
                if y not in seen:
                    seen.add(y)
                    st.append(y)
        return seen
    keep = reach(0, fwd) & reach(num_states - 1, bwd)
    if 0 not in keep:
        raise ValueError("the word sequence has no path through the lexicon")
    order = sorted(keep)
    ren = {s: i for i, s in enumerate(order)}
    ka = [(ren[s], ren[d], lab, w) for (s, d, lab, w) in arcs if s in keep and d in keep]
    kx = [x for (s, d, _, _), x in zip(arcs, aux) if s in keep and d in keep]
    return Fsa.from_arcs(ka, len(order), kx)


def _remove_epsilon(arcs, finals, num_states):
    """Epsilon removal on the input (phone) side in the tropical semiring: state s gets, for every state r in its
    epsilon closure (best score c), r's non-epsilon arcs with score c + w, and r's final weight.  A word label on an
    epsilon arc (none in L.fst as make_lexicon_fst.pl writes it) would be lost and is refused."""
    eps, real = {}, {}
    for s, d, ph, word, w in arcs:
        if ph == 0:
            if word != 0:
                raise NotImplementedError("a word label on an epsilon-input arc of the lexicon")
            eps.setdefault(s, []).append((d, w))
        else:
            real.setdefault(s, []).append((d, ph, word, w))
    fin_w = {}
    for s, w in finals:
        fin_w[s] = max(fin_w.get(s, -np.inf), w)
    sf = num_states
    out, aux = [], []
    for s in range(num_states):
        best = {s: 0.0}
        st = [s]
        while st:
            r = st.pop()
            for d, w in eps.get(r, ()):
                c = best[r] + w
                if c > best.get(d, -np.inf):
                    best[d] = c
                    st.append(d)
        f = -np.inf
        for r, c in best.items():
            for d, ph, word, w in real.get(r, ()):
                out.append((s, d, ph, c + w)); aux.append(word)
            if r in fin_w:
                f = max(f, c + fin_w[r])
        if f > -np.inf:
            out.append((s, sf, -1, f)); aux.append(-1)
    return _connect(out, aux, sf + 1)


def write_lang_dir(lang_dir, lexicon, phones, topo, oov="<UNK>", optional_silence=None, sil_prob=0.5, write_L=False):
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
    optional_silence: a phone symbol -> phones/optional_silence.txt|int; write_L: also L.fst.txt, the text of the lexicon
    transducer as utils/lang/make_lexicon_fst.pl builds it (start 0, loop 1, silence 2; word label on the first phone).
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
    if optional_silence is not None:
        dump("phones/optional_silence.txt", [optional_silence])
        dump("phones/optional_silence.int", [str(p2i[optional_silence])])
    if write_L:
        import math
        lines, nxt = [], 3
        if optional_silence is not None:
            c_no, c_sil = -math.log(1.0 - sil_prob), -math.log(sil_prob)
            lines += [f"0\t1\t0\t0\t{c_no}", f"0\t2\t0\t0\t{c_sil}", f"2\t1\t{p2i[optional_silence]}\t0"]
        else:
            c_no = c_sil = 0.0
            lines += ["0\t1\t0\t0"]
        for w, ps in lexicon:
            cur = 1
            for j, p in enumerate(ps):
                wl = w2i[w] if j == 0 else 0
                if j == len(ps) - 1:
                    lines.append(f"{cur}\t1\t{p2i[p]}\t{wl}\t{c_no}")
                    if optional_silence is not None:
                        lines.append(f"{cur}\t2\t{p2i[p]}\t{wl}\t{c_sil}")
                else:
                    lines.append(f"{cur}\t{nxt}\t{p2i[p]}\t{wl}")
                    cur = nxt
                    nxt += 1
        lines.append("1\t0")
        dump("L.fst.txt", lines)
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
