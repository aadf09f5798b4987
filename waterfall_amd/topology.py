"""Token topologies of the reference, as arc lists (no OpenFst needed).

Each ``_arcs_<name>`` function restates one generator script under ``/root/reference/utils`` (the
`_k2` variants are the *training* topologies); ``build_topology`` then applies the
symbol-table convention of the matching ``utils/prepare_<name>.sh`` (e.g. ``prepare_2state.sh:20-38``,
``prepare_ctc.sh:20-39``): token ids = NN output columns with ``<blk>`` (or ``<SIL>``) = 0, phone ids =
position in ``k2/phones.txt`` with ``<eps>`` = 0, arcs sorted per state by output label
(``fstarcsort --sort_type=olabel``), then ``k2.Fsa.from_openfst(acceptor=False)`` semantics
(``waterfall/utils/datapipe.py:168-181``): label = token, aux = phone, score = 0, one label -1 arc
from every final state to the appended super-final state.

phones: list of phone symbols in id order as in ``k2/phones.txt`` (index 0 is ``<eps>``; ``<SIL>``
may be present and keeps its id; ``#n`` disambiguation symbols must not be present).
"""
from .fsa import Fsa

EPS = "<eps>"
BLK = "<blk>"
SIL = "<SIL>"


def _real(phones):
    return [p for p in phones if p not in (EPS, SIL) and not p.startswith("#")]


def _arcs_ctc(phones):
    """utils/get_token_fst_ctc.py:25-53 (ignore_silence_between_words=True). All states final."""
    arcs = [(0, 0, BLK, EPS)]
    nodes = []
    x = 1
    for p in phones:
        if p in (EPS, BLK, SIL):
            continue
        arcs += [(0, x, p, p), (x, x, p, EPS), (x, 0, BLK, EPS)]
        nodes.append((p, x))
        x += 1
    for p, nid in nodes:
        for other in range(1, len(nodes) + 1):
            if other != nid:
                arcs.append((other, nid, p, p))
    return arcs, list(range(len(nodes) + 1))


def _arcs_2state(phones):
    """utils/get_token_fst_2state_k2.py:18-35."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in _real(phones):
        arcs += [(0, x, p + "_0", p), (x, 0, p + "_1", EPS), (x, x, p + "_0", EPS),
                 (x, x + 1, p + "_1", EPS), (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        x += 2
    return arcs, [0]


def _arcs_2state_1(phones):
    """utils/get_token_fst_2state-1_k2.py:18-34 (no x_0 self-loop)."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in _real(phones):
        arcs += [(0, x, p + "_0", p), (x, 0, p + "_1", EPS), (x, x + 1, p + "_1", EPS),
                 (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        x += 2
    return arcs, [0]


def _arcs_2state_sil(phones, nsil):
    """utils/get_token_fst_2state_k2_sil.py:19-40 (nsil=1) and ..._sil3.py:19-48 (nsil=3)."""
    arcs = []
    x = 1
    for p in phones:
        if p == EPS or p.startswith("#"):
            continue
        if p == SIL:
            toks = [SIL] if nsil == 1 else [SIL + "_%d" % i for i in range(nsil)]
            arcs += [(0, 0, t, SIL) for t in toks]
            arcs += [(0, x, t, SIL) for t in toks]
            arcs += [(x, x, t, EPS) for t in toks]
            arcs += [(x, 0, t, EPS) for t in toks]
            x += 1
        else:
            arcs += [(0, x, p + "_0", p), (x, 0, p + "_1", EPS), (x, x, p + "_0", EPS),
                     (x, x + 1, p + "_1", EPS), (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
            x += 2
    return arcs, [0]


def _arcs_3state_skip(phones):
    """utils/get_token_fst_3state-skip_k2.py:18-34."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in _real(phones):
        arcs += [(0, x, p + "_0", p), (x, 0, p + "_2", EPS), (x, x + 1, p + "_1", EPS),
                 (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_2", EPS)]
        x += 2
    return arcs, [0]


def _arcs_3state_skip_1(phones):
    """utils/get_token_fst_3state-skip-1_k2.py:18-38."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in _real(phones):
        arcs += [(0, x, p + "_0", p), (x, 0, p + "_2", EPS), (x, x + 2, p + "_2", EPS),
                 (x + 2, x + 2, p + "_2", EPS), (x + 2, 0, p + "_2", EPS), (x, x + 1, p + "_1", EPS),
                 (x + 1, x + 1, p + "_1", EPS), (x + 1, x + 2, p + "_2", EPS), (x + 1, 0, p + "_2", EPS)]
        x += 3
    return arcs, [0]


def _arcs_3state_skip_2(phones):
    """utils/get_token_fst_3state-skip-2_k2.py:18-39."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in _real(phones):
        arcs += [(0, x, p + "_0", p), (x, x, p + "_0", EPS), (x, x + 2, p + "_2", EPS),
                 (x + 2, x + 2, p + "_2", EPS), (x + 2, 0, p + "_2", EPS), (x, 0, p + "_2", EPS),
                 (x, x + 1, p + "_1", EPS), (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_2", EPS),
                 (x + 1, x + 2, p + "_2", EPS)]
        x += 3
    return arcs, [0]


def _arcs_mmictc(phones):
    """utils/get_token_fst_mmictc_k2.py:23-54 (default mode, with <blk>)."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in phones:
        if p in (EPS, SIL) or p.startswith("#"):
            continue
        arcs += [(0, 0, p + "_0", p), (0, x, p + "_0", p), (x, 0, p + "_1", EPS),
                 (x, x + 1, p + "_1", EPS), (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        x += 2
    return arcs, [0]


def _arcs_mmictc_noblk(phones):
    """utils/get_token_fst_mmictc_k2.py:33-52 with argv[2] == 'no_blk' (<SIL> replaces <blk>)."""
    arcs = []
    x = 1
    for p in phones:
        if p == EPS or p.startswith("#"):
            continue
        if p == SIL:
            arcs += [(0, 0, SIL, SIL), (0, x, SIL, SIL), (x, x, SIL, EPS), (x, 0, SIL, EPS)]
            x += 1
            continue
        arcs += [(0, 0, p + "_0", p), (0, x, p + "_0", p), (x, 0, p + "_1", EPS),
                 (x, x + 1, p + "_1", EPS), (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        x += 2
    return arcs, [0]


def _arcs_mmictc_1(phones):
    """utils/get_token_fst_mmictc-1_k2.py:20-38."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in _real(phones):
        arcs += [(0, 0, p + "_0", p), (0, x, p + "_0", p), (x, x, p + "_0", EPS), (x, 0, p + "_1", EPS),
                 (x, x + 1, p + "_1", EPS), (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        x += 2
    return arcs, [0]


def _tokens(phones, first, nsub):
    """Token table of utils/prepare_<topo>.sh, k2/tokens.txt: `first` symbols then p_0..p_{nsub-1}."""
    toks = list(first)
    for p in _real(phones):
        if nsub == 0:
            toks.append(p)
        else:
            toks += [p + "_%d" % i for i in range(nsub)]
    return toks


TOPOLOGIES = {
    # name: (arc generator, leading tokens, sub-tokens per phone)
    "ctc": (_arcs_ctc, [BLK], 0),
    "2state": (_arcs_2state, [BLK], 2),
    "2state-1": (_arcs_2state_1, [BLK], 2),
    "2state_sil": (lambda ph: _arcs_2state_sil(ph, 1), [SIL], 2),
    "2state_sil3": (lambda ph: _arcs_2state_sil(ph, 3), [SIL + "_0", SIL + "_1", SIL + "_2"], 2),
    "3state-skip": (_arcs_3state_skip, [BLK], 3),
    "3state-skip-1": (_arcs_3state_skip_1, [BLK], 3),
    "3state-skip-2": (_arcs_3state_skip_2, [BLK], 3),
    "mmictc": (_arcs_mmictc, [BLK], 2),
    "mmictc_noblk": (_arcs_mmictc_noblk, [SIL], 2),
    "mmictc-1": (_arcs_mmictc_1, [BLK], 2),
}


def _arcs_blk(phones, mmictc):
    """deprecated/get_token_fst_2state_blk_k2.py:19-37 and deprecated/get_token_fst_mmictc_blk_k2.py:20-39: the 2state /
    mmictc arcs for EVERY phone of a phones.txt that still holds the disambiguation symbols (<SIL> is an ordinary phone
    here), plus a self-loop '#k : #k' on state 0 per disambiguation symbol (for decoding-graph construction)."""
    arcs = [(0, 0, BLK, EPS)]
    x = 1
    for p in phones:
        if p in (EPS, BLK):
            continue
        if p.startswith("#"):
            arcs.append((0, 0, p, p))
        elif mmictc:
            arcs += [(0, 0, p + "_0", p), (0, x, p + "_0", p), (x, 0, p + "_1", EPS), (x, x + 1, p + "_1", EPS),
                     (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        else:
            arcs += [(0, x, p + "_0", p), (x, 0, p + "_1", EPS), (x, x, p + "_0", EPS), (x, x + 1, p + "_1", EPS),
                     (x + 1, x + 1, p + "_1", EPS), (x + 1, 0, p + "_1", EPS)]
        x += 2           # (the generators advance the state counter for disambiguation symbols too)
    return arcs, [0]


# The reference keeps these two generators under deprecated/: available as text (topology_text) for building decoding
# graphs with disambiguation symbols, not as training topologies (no tokens table of their own in the recipes).
DEPRECATED_TOPOLOGIES = {
    "2state_blk": lambda ph: _arcs_blk(ph, False),
    "mmictc_blk": lambda ph: _arcs_blk(ph, True),
}


def topology_text(name, phones):
    """The OpenFst text the reference generator prints (symbolic labels), one arc per line."""
    arcs, finals = (TOPOLOGIES[name][0] if name in TOPOLOGIES else DEPRECATED_TOPOLOGIES[name])(phones)
    lines = ["%d %d %s %s" % a for a in arcs] + ["%d" % f for f in finals]
    return "\n".join(lines) + "\n"


def _assemble(name, phones_subset, tok_id, ph_id):
    gen = TOPOLOGIES[name][0]
    arcs, finals = gen(phones_subset)
    rows = [(s, d, tok_id[t], ph_id[o]) for (s, d, t, o) in arcs]
    # fstarcsort --sort_type=olabel: per state by output label (stable, then input label)
    order = sorted(range(len(rows)), key=lambda i: (rows[i][0], rows[i][3], rows[i][2]))
    n_states = max(max(r[0], r[1]) for r in rows) + 1
    sf = n_states
    out, aux = [], []
    for i in order:
        s, d, t, o = rows[i]
        out.append((s, d, t, 0.0))
        aux.append(o)
    for f in finals:
        out.append((f, sf, -1, 0.0))
        aux.append(-1)
    return Fsa.from_arcs(out, sf + 1, aux)


class Topology:
    """A token topology over a phone table, materialised on demand.

    ``.fsa`` is the full T.fst (what ``Lang.load_topo`` loads, datapipe.py:168-181).  For large
    tables the full CTC topology is O(P^2) arcs (25 M at P = 5000), so numerator
    compilation uses ``restricted(phone_ids)``: the same generator run on the phones of one
    transcript only, with token / phone ids of the full tables.  Composing a transcript with the
    restricted topology gives exactly the graph the full topology gives (arcs that can never match
    the transcript are the only ones left out; relative arc order is preserved).
    """

    def __init__(self, name, phones):
        if name not in TOPOLOGIES:
            raise KeyError(f"unknown topology {name!r}; known: {sorted(TOPOLOGIES)}")
        self.name = name
        self.phones = list(phones)
        _, first, nsub = TOPOLOGIES[name]
        self.tokens = _tokens(self.phones, first, nsub)
        self._tok_id = {t: i for i, t in enumerate(self.tokens)}
        self._ph_id = {p: i for i, p in enumerate(self.phones)}
        self._fsa = None

    @property
    def num_nn_output(self):
        return len(self.tokens)

    @property
    def fsa(self):
        if self._fsa is None:
            self._fsa = _assemble(self.name, self.phones, self._tok_id, self._ph_id)
        return self._fsa

    # the full CTC topology has P^2 + 2P + 1 arcs: beyond this many phones numerator compilation uses its factored form
    LAZY_CTC_PHONES = 64

    def compose_arcs(self, factored=None):
        """Arc tables for compile.compose / wf_compose_linear_batch: the full topology, or (CTC over a large phone
        table, or factored=True) the CTC topology without the arcs "state of another phone -> state of phone p, p : p"
        (utils/get_token_fst_ctc.py:37-40), which phone_state / phone_token then describe."""
        import numpy as np
        if factored is None:
            factored = self.name == "ctc" and len(self.phones) > self.LAZY_CTC_PHONES
        if not factored:
            f = self.fsa
            return {"arcs": np.ascontiguousarray(f.arcs, dtype=np.int32), "aux": np.ascontiguousarray(f.aux_labels, dtype=np.int32),
                    "num_states": int(f.fsa_num_states[0]), "ctc_cross": 0, "phone_state": None, "phone_token": None}
        if self.name != "ctc":
            raise ValueError("only the ctc topology has a factored form")
        keep = [p for p in self.phones if p not in (EPS, BLK, SIL)]      # get_token_fst_ctc.py:31-33
        n = len(keep)
        ph = np.array([self._ph_id[p] for p in keep], dtype=np.int32)
        tok = np.array([self._tok_id[p] for p in keep], dtype=np.int32)
        blk = self._tok_id[BLK]
        x = np.arange(1, n + 1, dtype=np.int32)
        sf = n + 1
        # per state sorted by (phone, token) as prepare_ctc.sh's fstarcsort leaves it, the final arc last
        o0 = np.lexsort((tok, ph))
        rows0 = np.concatenate([[[0, 0, blk, 0]], np.stack([np.zeros(n, np.int32), x[o0], tok[o0], ph[o0]], 1), [[0, sf, -1, -1]]])
        first_blk = blk < tok                                           # (x -> 0, blk) before (x -> x, p) iff blk's id is smaller
        r_a = np.stack([x, np.where(first_blk, 0, x), np.where(first_blk, blk, tok), np.zeros(n, np.int32)], 1)
        r_b = np.stack([x, np.where(first_blk, x, 0), np.where(first_blk, tok, blk), np.zeros(n, np.int32)], 1)
        r_f = np.stack([x, np.full(n, sf, np.int32), np.full(n, -1, np.int32), np.full(n, -1, np.int32)], 1)
        rows = np.concatenate([rows0, np.stack([r_a, r_b, r_f], 1).reshape(-1, 4)]).astype(np.int32)
        arcs = np.ascontiguousarray(np.stack([rows[:, 0], rows[:, 1], rows[:, 2], np.zeros(len(rows), np.int32)], 1), dtype=np.int32)
        aux = np.ascontiguousarray(rows[:, 3], dtype=np.int32)
        phone_state = np.zeros(len(self.phones), dtype=np.int32)
        phone_token = np.zeros(len(self.phones), dtype=np.int32)
        phone_state[ph] = x
        phone_token[ph] = tok
        return {"arcs": arcs, "aux": aux, "num_states": n + 2, "ctc_cross": 1, "phone_state": phone_state, "phone_token": phone_token}

    def restricted(self, phone_ids):
        keep = set(int(i) for i in phone_ids)
        sub = [p for i, p in enumerate(self.phones) if i in keep or p in (EPS, SIL)]
        return _assemble(self.name, sub, self._tok_id, self._ph_id)


def build_topology(name, phones):
    """Returns (topo: Fsa, tokens: list[str]).  len(tokens) = Lang.num_nn_output (datapipe.py:112-114)."""
    t = Topology(name, phones)
    return t.fsa, t.tokens
