"""Randomised parity run (GPU): random topology / sizes / ragged lengths / sweep path against the numpy oracle.
usage: python tools/fuzz_gpu.py [cases] [seed] [big|hub]   (test infrastructure: imports oracle/)
big: larger cases (more kernel instantiations); hub: denominators over 128..2400 phones (sweep_hub_kernel / post_hub_kernel
and their fallbacks).  A third of the cases run with a finite output_beam (pruned lattice) against the oracle's."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import util  # noqa: E402
from oracle import dwfst_oracle as o  # noqa: E402
import waterfall_amd as wf  # noqa: E402

PATHS = [{}, {"WF_NO_BAND_SWEEP": "1"}, {"WF_NO_BAND_SWEEP": "1", "WF_NO_LIN_SWEEP": "1"}, {"WF_FORCE_FALLBACK": "1"},
         {"WF_NO_BAND_SWEEP": "1", "WF_NO_LIN_SWEEP": "2"}, {"WF_BIG_MIN_S": "1"}, {"WF_BIG_MIN_S": "1", "WF_POST_BIG_MIN_S": "1"},
         {"WF_NO_GATHER_SMALL": "1"}]
HUB_PATHS = [{}, {"WF_HUB_FORCE_SPLIT": "1"}, {"WF_HUB_FORCE_SPLIT": "1", "WF_NO_HUB_PATTERN": "1"}, {"WF_NO_HUB_PATTERN": "1"}, {"WF_NO_HUB_SWEEP": "1"}, {"WF_FORCE_FALLBACK": "1"}, {"WF_NO_TAB_KERNARG": "1"}]
mode = sys.argv[3] if len(sys.argv) > 3 else ""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
topos = sorted(wf.TOPOLOGIES)
bad = 0
for case in range(n):
    topo = topos[rng.integers(len(topos))]
    den = bool(rng.integers(2))
    big = mode == "big"
    B, T, L, P = int(rng.integers(1, 10)), int(rng.integers(1, 300 if big else 90)), int(rng.integers(0, 90 if big else 13)), int(rng.integers(2, 45 if big else 13))
    env = PATHS[rng.integers(len(PATHS))]
    if mode == "hub":
        den, B, T, P = True, int(rng.integers(1, 5)), int(rng.integers(1, 70)), int(rng.integers(128, 2400))
        while topo == "ctc":   # the full CTC topology has O(P^2) arcs
            topo = topos[rng.integers(len(topos))]
        env = HUB_PATHS[rng.integers(len(HUB_PATHS))]
    beam = [None, None, 4.0, 9.0][rng.integers(4)]
    c = util.random_case(int(rng.integers(1 << 30)), topo, B=B, T=T, L=L, P=P, den=den, ragged=bool(rng.integers(2)))
    for k in list(os.environ):
        if k.startswith("WF_"):
            del os.environ[k]
    os.environ.update(env)
    from waterfall_amd import _lib
    _lib.reload_env()   # the library reads the hooks once per process
    if rng.integers(2):   # start offsets, unused sequences, permuted sequence indices
        N, pad = B + int(rng.integers(0, 3)), int(rng.integers(0, 6))
        lp2 = np.log(np.full((N, T + pad, c["V"]), 1.0 / c["V"], dtype=np.float32))
        perm = rng.permutation(N)[:B]
        segs2 = c["segs"].copy()
        for i in range(B):
            st = int(rng.integers(0, T + pad - segs2[i, 2] + 1))
            lp2[perm[i], st:st + segs2[i, 2]] = c["lp"][i, :segs2[i, 2]]
            segs2[i, 0], segs2[i, 1] = perm[i], st
        c["lp"], c["segs"] = lp2, segs2
    only = os.environ.get("FUZZ_ONLY")   # replay one case of a run (the draws above are made for every case)
    if only is not None and case != int(only):
        continue
    x = torch.from_numpy(c["lp"]).cuda().requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(c["segs"]))
    wf.set_output_beam_mode("pruned" if beam else "exact")
    loss = wf.graphloss(c["fsa"], dense, output_beam=beam or 10, reduction="none")
    fin = torch.isfinite(loss)
    torch.where(fin, loss, torch.zeros_like(loss)).sum().backward()
    want, wgrad = o.graphloss(c["ographs"], c["lp"], c["segs"], beam, "none")
    got, g = loss.detach().cpu().numpy(), x.grad.cpu().numpy()
    same = lambda w, wg: np.allclose(got, w, rtol=1e-4, atol=1e-4) and np.allclose(g, np.where(np.isfinite(wg), wg, 0), atol=1e-3) \
        and np.all(np.isfinite(g))
    ok = same(want, wgrad)
    if not ok and beam:
        # Lattice arcs whose best path sits on the beam's edge to within the fp32 rounding of the tropical scores (the
        # oracle prunes in fp64) may fall on either side.  The kept set grows with the beam and the total with the set,
        # so the result must lie between the oracle's for a beam a hair narrower and a hair wider.
        eps = 2e-3
        (wn, gn), (ww, gw) = (o.graphloss(c["ographs"], c["lp"], c["segs"], beam + d, "none") for d in (-eps, eps))
        z = lambda a: np.where(np.isfinite(a), a, 0)
        ok = bool(np.all((got <= wn + 1e-4 * np.abs(wn) + 1e-4) & (got >= ww - 1e-4 * np.abs(ww) - 1e-4)) and np.all(np.isfinite(g))
                  and np.all(np.abs(g - z(wgrad)) <= np.abs(z(gn) - z(gw)).max() + 1e-3))
        if ok:
            print("  (case", case, "lies between the oracle's results at beam", beam, "-+", eps, ": arcs on the beam's edge)", flush=True)
    best = wf.best_path(c["fsa"], dense)
    for i, gr in enumerate(c["ographs"]):
        sq, st, tt = (int(v) for v in c["segs"][i])
        s, path = o.viterbi(gr, c["lp"][sq, st:st + tt], tt)
        if best["scores"][i].item() != s and not (np.isinf(s) and np.isinf(best["scores"][i].item())):
            ok = False
    if ok and not den and not beam and c["topo"] is not None and rng.integers(2):
        # the numerator - denominator pair as one node (WF_ACCUMULATE) against the two-call form, same inputs
        dg = wf.create_fsa_vec([c["topo"]] * B)
        res = []
        for fn in (lambda d: wf.graphloss_num_den(c["fsa"], dg, d, reduction="sum"),
                   lambda d: wf.graphloss(c["fsa"], d, reduction="sum") - wf.graphloss(dg, d, reduction="sum")):
            x2 = torch.from_numpy(c["lp"]).cuda().requires_grad_(True)
            l2 = fn(wf.DenseFsaVec(x2, torch.from_numpy(c["segs"])))
            if torch.isfinite(l2):
                (l2 * -1.5).backward()
            res.append((l2.item(), None if x2.grad is None else x2.grad.cpu().numpy()))
        (la, ga), (lb, gb) = res
        if np.isfinite(lb):
            ok = bool(np.isclose(la, lb, rtol=1e-5, atol=1e-3) and np.allclose(ga, gb, atol=2e-5))
        else:
            ok = not np.isfinite(la) or np.isnan(la) == np.isnan(lb)
        if not ok:
            print("  (graphloss_num_den differs from the two-call form)", la, lb, flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", case, topo, den, B, T, L, P, env, beam, np.abs(got - want).max(),
              np.abs(g - np.where(np.isfinite(wgrad), wgrad, 0)).max(), flush=True)
    elif case % 20 == 0:
        print("ok", case, flush=True)
print(f"{n} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
