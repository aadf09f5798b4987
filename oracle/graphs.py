"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of how the reference builds its graphs.

* ``from_openfst_text``: what ``k2.Fsa.from_openfst(txt, acceptor=False)`` does with the text that
  ``waterfall/utils/datapipe.py:168-181`` feeds it ([k2-upstream] semantics):
  labels = ilabel, aux = olabel, score = -weight, every final state gets a label -1 arc into a new
  super-final state (the last state id).
* ``compose_topo_transcript``: ``k2.compose(topo, remove_epsilon_and_add_self_loops(linear_fsa(t)),
  treat_epsilons_specially=False)`` as used by ``waterfall/models.py:263-270``; rule written in
  SURVEY.md 8(0) "Composition rule".  States are discovered breadth first and numbered position-major
  (k2's own numbering is parity unpinned).
* ``ctc_trans_transform``: ``utils/make_transition_mat.py:27-57`` and
  ``utils/make_transform_mat.py:10-36`` (those modules import kaldiio and cannot be imported).
* ``graph_to_trans_transform``: encodes any graph as the (trans, transform) pair that
  ``waterfall/manual_ctc/ctc.py:9-73`` consumes, one manual_ctc state per arc.
"""
from collections import deque

import numpy as np

from .dwfst_oracle import make_graph


def from_openfst_text(txt, sym_in=None, sym_out=None):
    """Parse OpenFst text (src dst ilabel olabel [weight] / final [weight]) into an oracle graph."""
    arcs, aux, finals = [], [], []
    max_state = -1
    for line in txt.strip().splitlines():
        f = line.split()
        if not f:
            continue
        if len(f) <= 2:
            s = int(f[0])
            w = float(f[1]) if len(f) == 2 else 0.0
            finals.append((s, -w))
            max_state = max(max_state, s)
            continue
        s, d = int(f[0]), int(f[1])
        il = sym_in[f[2]] if sym_in is not None else int(f[2])
        ol = sym_out[f[3]] if sym_out is not None else int(f[3])
        w = float(f[4]) if len(f) > 4 else 0.0
        arcs.append((s, d, il, -w))
        aux.append(ol)
        max_state = max(max_state, s, d)
    sf = max_state + 1
    for s, w in finals:
        arcs.append((s, sf, -1, w))
        aux.append(-1)
    return make_graph(arcs, sf + 1, aux)


def topo_final_states(g):
    return set(int(s) for s, l in zip(g["src"], g["label"]) if l < 0)


def compose_topo_transcript(topo, phones):
    """topo: graph with label=token, aux=phone; phones: list of phone ids (>= 1)."""
    L = len(phones)
    out = [[] for _ in range(topo["num_states"])]
    for a in range(topo["src"].size):
        out[int(topo["src"][a])].append(a)
    ids = {(0, 0): 0}
    order = [(0, 0)]
    q = deque([(0, 0)])
    arcs, aux = [], []
    fin = []
    while q:
        s, i = q.popleft()
        for a in out[s]:
            lab, ph, d, w = int(topo["label"][a]), int(topo["aux"][a]), int(topo["dst"][a]), float(topo["score"][a])
            if lab < 0:
                if i == L:
                    fin.append((ids[(s, i)], w))
                continue
            if ph == 0:
                ns = (d, i)
            elif i < L and ph == phones[i]:
                ns = (d, i + 1)
            else:
                continue
            if ns not in ids:
                ids[ns] = len(order)
                order.append(ns)
                q.append(ns)
            arcs.append((ids[(s, i)], ids[ns], lab, w))
            aux.append(ph)
    sf = len(order)
    for s, w in fin:
        arcs.append((s, sf, -1, w))
        aux.append(-1)
    return make_graph(_position_major(arcs, order, sf), sf + 1, aux)


def _position_major(arcs, order, sf):
    """States are discovered breadth first but NUMBERED by (right-hand state, topology state) -- position-major for a
    linear transcript, the numbering of the product's compilers (wf_graph.cpp renumber_position_major; k2's own numbering
    in compose is parity unpinned).  Arcs keep their order within a state (make_graph's stable sort by source)."""
    new = {old: i for i, old in enumerate(sorted(range(len(order)), key=lambda j: (order[j][1], order[j][0])))}
    new[sf] = sf
    return [(new[s], new[d], lab, w) for s, d, lab, w in arcs]


def compose_topo_fsa(topo, b):
    """compose(topo, b, treat_epsilons_specially=False) for an arbitrary epsilon-free right-hand acceptor / transducer
    b (label = phone >= 1, aux = word id or 0, label -1 arcs into its last state) that carries an epsilon self-loop on
    every state (k2.remove_epsilon_and_add_self_loops, datapipe.py:249-254).  Product states (q, r) in BFS order from
    (0, 0); per product state the topology's arcs in order, and per topology arc with phone p the arcs of r labelled p
    in b's order; topology arcs with phone 0 pair with the self-loop.  Output label = token, aux = b's aux (phone when
    b has none), score = sum.  A product final arc needs a final arc on both sides."""
    tout = [[] for _ in range(topo["num_states"])]
    for a in range(topo["src"].size):
        tout[int(topo["src"][a])].append(a)
    bout = [[] for _ in range(b["num_states"])]
    for a in range(b["src"].size):
        bout[int(b["src"][a])].append(a)
    baux = b.get("aux")
    ids = {(0, 0): 0}
    order = [(0, 0)]
    q = deque([(0, 0)])
    arcs, aux, fin = [], [], []
    while q:
        s, r = q.popleft()
        for a in tout[s]:
            lab, ph, d, w = int(topo["label"][a]), int(topo["aux"][a]), int(topo["dst"][a]), np.float32(topo["score"][a])
            if lab < 0:
                for e in bout[r]:
                    if int(b["label"][e]) < 0:
                        fin.append((ids[(s, r)], np.float32(w + np.float32(b["score"][e]))))
                continue
            if ph == 0:
                nxt = [((d, r), w, 0)]
            else:
                nxt = [((d, int(b["dst"][e])), np.float32(w + np.float32(b["score"][e])),
                        int(baux[e]) if baux is not None else ph) for e in bout[r] if int(b["label"][e]) == ph]
            for ns, ww, ax in nxt:
                if ns not in ids:
                    ids[ns] = len(order)
                    order.append(ns)
                    q.append(ns)
                arcs.append((ids[(s, r)], ids[ns], lab, float(ww)))
                aux.append(ax)
    sf = len(order)
    # final arcs go to the end of their source state's arcs (as in compose_topo_transcript after make_graph's sort)
    for s, w in fin:
        arcs.append((s, sf, -1, float(w)))
        aux.append(-1)
    return make_graph(_position_major(arcs, order, sf), sf + 1, aux)


FMIN = np.finfo(np.float32).min


def ctc_trans_transform(target, num_token):
    L = len(target)
    S = 2 * L + 1
    trans = np.full((S, S), FMIN, dtype=np.float32)
    for i in range(L):
        trans[2 * i, 2 * i] = 0.0
        if 2 * i - 1 >= 0:
            trans[2 * i, 2 * i - 1] = 0.0
        trans[2 * i + 1, 2 * i + 1] = 0.0
        trans[2 * i + 1, 2 * i] = 0.0
        if 2 * i - 1 >= 0 and target[i] != target[i - 1]:
            trans[2 * i + 1, 2 * i - 1] = 0.0
    trans[S - 1, S - 1] = 0.0
    trans[S - 1, S - 2] = 0.0
    transform = np.full((S, num_token), FMIN, dtype=np.float32)
    for i, item in enumerate(target):
        transform[2 * i, 0] = 0.0
        transform[2 * i + 1, item] = 0.0
    transform[S - 1, 0] = 0.0
    return trans, transform


def graph_to_trans_transform(g, num_token):
    """One manual_ctc state per emitting arc, plus a virtual start row 0 and virtual end row S-1.

    ctc.py:34 starts all mass in row 0 *before* frame 0 and ctc.py:39 ends in row S-1 *after* the
    last frame, so: trans[a+1, 0] = 0 if src(a) is the start state; trans[a2+1, a1+1] = 0 if
    dst(a1) == src(a2); trans[S-1, a+1] = w_final if dst(a) has a final arc.  The arc's own weight
    is folded into its incoming transitions (topology weights are 0 in the reference anyway).
    Rows 0 and S-1 get a transform row that is FMIN everywhere (they never emit).
    """
    em = np.nonzero(g["label"] >= 0)[0]
    fin = np.nonzero(g["label"] < 0)[0]
    n = em.size
    S = n + 2
    trans = np.full((S, S), FMIN, dtype=np.float32)
    transform = np.full((S, num_token), FMIN, dtype=np.float32)
    fw = {}
    for a in fin:
        fw[int(g["src"][a])] = float(g["score"][a])
    for j, a2 in enumerate(em):
        transform[j + 1, g["label"][a2]] = 0.0
        if g["src"][a2] == 0:
            trans[j + 1, 0] = g["score"][a2]
        for i, a1 in enumerate(em):
            if g["dst"][a1] == g["src"][a2]:
                trans[j + 1, i + 1] = g["score"][a2]
        if int(g["dst"][a2]) in fw:
            trans[S - 1, j + 1] = fw[int(g["dst"][a2])]
    return trans, transform
