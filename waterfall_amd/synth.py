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
