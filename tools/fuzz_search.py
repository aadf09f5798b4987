"""Randomised parity run (GPU) of the beam-pruned search wf_search against oracle.beam_search, and of wf_search_lattice
(the search followed by get_lattice's output_beam) against oracle.lattice_search, and of wf_search_lattice_words against
oracle.lattice_words: random graphs (skewed
out-degrees, states with thousands of arcs, fan-in hubs, dead ends), quantised weights and emissions (many exact ties:
the lowest-arc-index rule and the `>= best - beam` boundary are hit all the time), random beams and lengths.
usage: python tools/fuzz_search.py [cases] [seed] [max_states]   (test infrastructure: imports oracle/)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import dwfst_oracle as o  # noqa: E402
import waterfall_amd as wf  # noqa: E402


def random_graph(rng, S, V, quant):
    n = S - 1
    deg = np.minimum(rng.geometric(0.45, size=n), 12).astype(np.int64)
    if rng.integers(2):
        deg[rng.integers(0, n, size=max(1, n // 50))] = 0                      # dead ends
    for _ in range(int(rng.integers(0, 3))):                                    # states with very many arcs
        deg[rng.integers(0, n)] = int(rng.integers(100, 5000))
    deg[0] = max(deg[0], 1)
    src = np.repeat(np.arange(n), deg)
    m = src.size
    near = np.clip(src + rng.integers(-2, 6, size=m), 0, n - 1)
    far = rng.integers(0, n, size=m)
    hubs = rng.integers(0, n, size=4)
    pick = rng.random(m)
    dst = np.where(pick < 0.6, near, np.where(pick < 0.9, far, hubs[rng.integers(0, 4, size=m)]))
    lab = rng.integers(0, V, size=m)
    sc = -0.5 * rng.integers(0, 8, size=m) if quant else -rng.random(m) * 4.0
    fin = np.unique(np.concatenate([rng.integers(0, n, size=max(1, n // 15)), [n - 1]]))
    fsc = -0.5 * rng.integers(0, 4, size=fin.size) if quant else -rng.random(fin.size)
    src = np.concatenate([src, fin]); dst = np.concatenate([dst, np.full(fin.size, n)])
    lab = np.concatenate([lab, np.full(fin.size, -1)]); sc = np.concatenate([sc, fsc])
    order = np.argsort(src, kind="stable")
    return {"src": src[order].astype(np.int32), "dst": dst[order].astype(np.int32), "label": lab[order].astype(np.int32),
            "score": sc[order].astype(np.float32), "aux": lab[order].astype(np.int32), "num_states": S}


def to_fsa(g):
    arcs = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    return wf.Fsa(arcs, g["aux"], [0, arcs.shape[0]], [g["num_states"]])


n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
smax = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
bad = 0
for case in range(n):
    quant = bool(rng.integers(2))
    V = int(rng.integers(2, 200))
    B, T = int(rng.integers(1, 5)), int(rng.integers(1, 40))
    shared = bool(rng.integers(2))
    S = int(np.exp(rng.uniform(np.log(3), np.log(smax))))
    gs = [random_graph(rng, S, V, quant)] * B if shared else \
        [random_graph(rng, int(np.exp(rng.uniform(np.log(3), np.log(smax)))), V, quant) for _ in range(B)]
    one = to_fsa(gs[0])
    fsa = wf.create_fsa_vec([one] * B if shared else [to_fsa(g) for g in gs])
    if quant:
        lp = (-0.5 * rng.integers(0, 12, size=(B, T, V))).astype(np.float32)
    else:
        lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32)) * 3.0, -1).numpy()
    lens = rng.integers(1, T + 1, size=B)
    lens[0] = T
    segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), lens], 1).astype(np.int32)
    dense = wf.DenseFsaVec(torch.from_numpy(lp).cuda(), torch.from_numpy(segs))
    beam = float(rng.choice([0.5, 1.0, 2.0, 3.5, 6.0, 10.0, 20.0, np.inf]))
    out = wf.pruned_search(fsa, dense, search_beam=beam, min_active_states=0, max_active_states=0,
                           use_double_scores=True, return_active=True)
    scores, arcs = out["scores"].cpu().numpy(), out["arcs"].cpu().numpy()
    tot, nact = out["tot_log"].cpu().numpy(), out["n_active"].cpu().numpy()
    base = fsa.fsa_arc_offsets
    why = []
    for i, g in enumerate(gs):
        Ti = int(segs[i, 2])
        s, path, tl, na = o.beam_search(g, lp[i], Ti, beam)
        gi = 0 if fsa.utt2fsa is not None else i
        if not (scores[i] == s):
            why.append(f"score[{i}] {scores[i]} != {s}")
        got = list(arcs[i, :Ti + 1] - base[gi]) if path else []
        if path and got != path:
            why.append(f"path[{i}]")
        if not path and not np.all(arcs[i] == -1):
            why.append(f"path[{i}] should be empty")
        if list(nact[i, :Ti + 1]) != na:
            why.append(f"n_active[{i}]")
        if np.isfinite(tl):
            if not np.isclose(tot[i], tl, rtol=1e-6, atol=1e-5):
                why.append(f"tot[{i}] {tot[i]} != {tl}")
        elif tot[i] != -np.inf:
            why.append(f"tot[{i}] {tot[i]} != -inf")
    # get_lattice's output_beam over the same search (wf_search_lattice) against oracle.lattice_search
    ob = float(rng.choice([0.5, 1.0, 2.5, 6.0, np.inf]))
    lat = wf.pruned_search(fsa, dense, search_beam=beam, min_active_states=0, max_active_states=0,
                           use_double_scores=True, output_beam=ob)
    ltot, larcs = lat["tot_log"].cpu().numpy(), lat["lattice_arcs"].cpu().numpy()
    if not (torch.equal(lat["scores"], out["scores"]) and torch.equal(lat["arcs"], out["arcs"])):
        why.append("output_beam changed the best path")
    for i, g in enumerate(gs):
        Ti = int(segs[i, 2])
        s, path, tl, na = o.lattice_search(g, lp[i], Ti, beam, ob)
        if larcs[i] != na:
            why.append(f"lattice_arcs[{i}] {larcs[i]} != {na} (output_beam {ob})")
        if np.isfinite(tl):
            if not np.isclose(ltot[i], tl, rtol=1e-9, atol=1e-9):
                why.append(f"lattice tot[{i}] {ltot[i]} != {tl} (output_beam {ob})")
        elif ltot[i] != -np.inf:
            why.append(f"lattice tot[{i}] {ltot[i]} != -inf")
    # ... and that lattice restricted to a word sequence (wf_search_lattice_words) against oracle.lattice_words: the best
    # path's words (aux labels: the arc labels here, 0 = epsilon), sometimes cut short or with a wrong word in them
    words = []
    for ws in lat["aux_labels"]:
        ws = list(ws)
        r = rng.integers(4)
        if r == 1 and ws:
            ws = ws[:-1]
        elif r == 2:
            ws = ws + [int(rng.integers(1, V + 1))]
        elif r == 3 and ws:
            ws[int(rng.integers(len(ws)))] = int(rng.integers(1, V + 1))
        words.append(ws)
    lw = wf.lattice_words(fsa, dense, words, search_beam=beam, output_beam=ob, min_active_states=0, max_active_states=0)
    tw_, bw_ = lw["tot_words"].cpu().numpy(), lw["best_words"].cpu().numpy()
    for i, g in enumerate(gs):
        Ti = int(segs[i, 2])
        tw, bw = o.lattice_words(g, lp[i], Ti, words[i], beam, ob)
        if not (bw_[i] == bw):
            why.append(f"best_words[{i}] {bw_[i]} != {bw}")
        if np.isfinite(tw):
            if not np.isclose(tw_[i], tw, rtol=1e-9, atol=1e-9):
                why.append(f"tot_words[{i}] {tw_[i]} != {tw}")
        elif tw_[i] != -np.inf:
            why.append(f"tot_words[{i}] {tw_[i]} != -inf")
    # active-state limits: properties only (the oracle states the beam rule alone)
    cap = int(rng.integers(1, 300))
    lim = wf.pruned_search(fsa, dense, search_beam=beam, min_active_states=0, max_active_states=cap, return_active=True)
    if lim["n_active"].cpu().numpy()[:, 1:].max() > cap:
        why.append("max_active exceeded")
    la, ls = lim["arcs"].cpu().numpy(), lim["scores"].cpu().numpy()
    for i, g in enumerate(gs):
        Ti = int(segs[i, 2])
        gi = 0 if fsa.utt2fsa is not None else i
        if np.isfinite(ls[i]):
            p = la[i, :Ti + 1] - base[gi]
            x = np.float32(0)
            okp = p.min() >= 0 and g["src"][p[0]] == 0 and g["label"][p[-1]] == -1 and np.all(g["dst"][p[:-1]] == g["src"][p[1:]])
            if okp:
                for t, a in enumerate(p[:-1]):
                    x = np.float32(x + np.float32(g["score"][a] + lp[i, t, g["label"][a]]))
                okp = np.float32(x + g["score"][p[-1]]) == ls[i]
            if not okp:
                why.append(f"max_active path[{i}] invalid")
            if ls[i] > scores[i] and beam == np.inf:
                why.append(f"max_active score[{i}] above the exact one")
    if why:
        bad += 1
        print("MISMATCH", case, dict(quant=quant, V=V, B=B, T=T, shared=shared, S=[g["num_states"] for g in gs], beam=beam, cap=cap),
              why[:6], flush=True)
    elif case % 10 == 0:
        print("ok", case, flush=True)
print(f"{n} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
