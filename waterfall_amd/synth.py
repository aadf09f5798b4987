"""Synthetic inputs shared by bench.py and the tests; numpy only."""
import numpy as np


def synth_decoding_graph(seed, S, V, hubs=4, hub_fan_in=300):
    """Synthetic k2-style epsilon-free decoding graph as BASELINE.md cfg5 describes it (A = 3 S emitting arcs):
    per state a self-loop, an arc to the next state and a random jump, LM-like negative weights; a few hub states
    (word-boundary-like) with a large fan-in; ~1 % of the states final.  Returns an oracle graph dict."""
    rng = np.random.default_rng(seed)
    n = S - 1                                   # real states; S-1 is the super-final state
    st = np.arange(n, dtype=np.int32)
    jump = rng.integers(0, n, size=n).astype(np.int32)
    hub_ids = rng.choice(n, size=hubs, replace=False).astype(np.int32)
    into_hub = rng.choice(n, size=min(n, hubs * hub_fan_in), replace=False)
    jump[into_hub] = hub_ids[rng.integers(0, hubs, size=into_hub.size)]
    src = np.concatenate([st, st, st])
    dst = np.concatenate([st, np.minimum(st + 1, n - 1), jump])
    lab = rng.integers(0, V, size=3 * n).astype(np.int32)
    sc = (-rng.random(3 * n) * 4.0).astype(np.float32)
    fin = np.unique(np.concatenate([rng.choice(n, size=max(1, n // 100), replace=False), [n - 1]])).astype(np.int32)
    src = np.concatenate([src, fin]); dst = np.concatenate([dst, np.full(fin.size, n, dtype=np.int32)])
    lab = np.concatenate([lab, np.full(fin.size, -1, dtype=np.int32)])
    sc = np.concatenate([sc, (-rng.random(fin.size)).astype(np.float32)])
    order = np.argsort(src, kind="stable")      # k2 keeps arcs sorted by source state
    return {"src": src[order].astype(np.int32), "dst": dst[order].astype(np.int32), "label": lab[order],
            "score": sc[order], "aux": lab[order].copy(), "num_states": S}


def cfg5_workload(dev, S, B=8, T=500, V=5001, seed=777, peaked=False, peak=10.0):
    """BASELINE.md cfg5 as bench.py times it and tests/test_gpu_parity.py checks it: one synthetic TLG-like graph
    (A = 3 S) shared by B utterances, log_softmax(randn * 3) drawn on the device.  Returns (graph dict, Fsa vector,
    log_probs [B,T,V] on dev, DenseFsaVec).

    peaked: the posterior of a trained model instead of noise -- per utterance a complete path of the graph is planted
    (the exact best path under the noise posterior) and log_probs = log_softmax(randn + 10 * onehot(its label)), about
    0.7 of every frame's mass on the planted label: the workload of the beam-pruned search (a beam has nothing to hold
    on to in pure noise).  peak: the planted label's logit boost (10: a confident model, a search beam of 20 keeps a few
    dozen states; 4-6: the 10^3 - 10^4 active states of real decoding)."""
    import torch
    from .fsa import DenseFsaVec, Fsa, create_fsa_vec
    g = synth_decoding_graph(seed, S, V)
    arcs = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
    one = Fsa(arcs, g["aux"], [0, arcs.shape[0]], [g["num_states"]])
    fsa = create_fsa_vec([one] * B)
    gen = torch.Generator(device=dev).manual_seed(seed)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=gen, device=dev) * 3.0, -1)
    dense = DenseFsaVec(lp, torch.tensor([[i, 0, T] for i in range(B)], dtype=torch.int32))
    if peaked:
        from .graph import best_path
        arcs_best = best_path(fsa, dense)["arcs"][:, :T].cpu().numpy()
        lab = torch.from_numpy(g["label"][arcs_best].astype(np.int64)).to(dev)
        z = torch.randn(B, T, V, generator=gen, device=dev)
        z.scatter_add_(2, lab.unsqueeze(-1), torch.full((B, T, 1), float(peak), device=dev))
        lp = torch.log_softmax(z, -1)
        dense = DenseFsaVec(lp, torch.tensor([[i, 0, T] for i in range(B)], dtype=torch.int32))
    return g, fsa, lp, dense


def best_path_checksum(scores, arcs):
    """(sum of the fp32 best scores in fp64, a position-weighted sum of the arc ids mod 2^61-1): what bench.py prints
    for cfg5 and the parity test recomputes from the oracle's paths."""
    sc = float(np.asarray(scores, dtype=np.float64).sum())
    a = np.asarray(arcs, dtype=np.int64)
    w = (np.arange(a.size, dtype=np.int64).reshape(a.shape) % 1000003) + 1
    return sc, int(((a + 2) * w % 2305843009213693951).sum() % 2305843009213693951)
