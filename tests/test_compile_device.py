"""GPU tests (-m gpu) of the numerator graph compiler on the device (wf_compile.hip numerator_compile_kernel; SURVEY.md
8(f) rank 1: models.py:263-270 / datapipe.py:233-256 rebuild the numerator graphs on every training step): for every
topology the device-built graph vector equals the host compiler's (wf_numerator_compile) -- FsaVec arrays, per-FSA info
words, every prepared table, word for word -- and the loss computed from it is bit-identical."""
import numpy as np
import pytest
import torch

import waterfall_amd as wf
from waterfall_amd.csrc_layout import HDR_WORDS, INFO, INFO_WORDS
from tests import util
from tests.test_gpu_parity import DEV, fused_logsum

pytestmark = pytest.mark.gpu

# info words that hold blob offsets, with the length of the table behind each as a function of the FSA's sizes
TABLES = {
    "IN_OFF": lambda d: d["S"] + 1, "IN_SRC": lambda d: d["A"], "IN_COL": lambda d: d["A"], "IN_W": lambda d: d["A"], "IN_ARC": lambda d: d["A"],
    "HEAVY_IN": lambda d: d["S"],
    "OUT_OFF": lambda d: d["S"] + 1, "OUT_DST": lambda d: d["A"], "OUT_COL": lambda d: d["A"], "OUT_W": lambda d: d["A"], "OUT_ARC": lambda d: d["A"],
    "OUT_REC": lambda d: 4 * d["A"], "HEAVY_OUT": lambda d: d["S"],
    "LAB_OFF": lambda d: d["K"] + 1, "LAB_SRC": lambda d: d["A"], "LAB_DST": lambda d: d["A"], "LAB_W": lambda d: d["A"], "LABELS": lambda d: d["K"],
    "LAB_COL": lambda d: d["A"],
    "FIN_SRC": lambda d: d["NF"], "FIN_W": lambda d: d["NF"], "FIN_ARC": lambda d: d["NF"],
    "FINB_LOG": lambda d: d["S"], "FINB_MAX": lambda d: d["S"], "ARC_SRC": lambda d: d["A"] + d["NF"],
    "BAND_ORDER": lambda d: d["S"] - 1 if d["BAND_BW"] >= 0 else 0,
    "BANDF": lambda d: 2 * (d["S"] - 1) * (d["BAND_BW"] + 1) * d["BAND_M"] if d["BAND_BW"] >= 0 else 0,
    "BANDB": lambda d: 2 * (d["S"] - 1) * (d["BAND_BW"] + 1) * d["BAND_M"] if d["BAND_BW"] >= 0 else 0,
}
OFFSET_WORDS = set(TABLES) | {"HUB_C0", "HUB_COEF", "HUB_XCOL"}


def assert_same_graph_vector(dev_fsa, host_fsa):
    hb = host_fsa.blob_host()
    dh = dev_fsa.blob_host()
    assert dev_fsa.compiled_on == "device"
    db = dev_fsa.blob_dev(torch.device(DEV)).cpu().numpy()
    B = host_fsa.num_fsas
    assert dev_fsa.num_fsas == B and int(db[1]) == B and db[0] == hb[0]
    assert np.array_equal(dh, db[:dh.size])               # the pinned prefix the host plans with is the blob's
    assert list(db[3:6]) == list(hb[3:6])                  # max label / states / arcs of the batch
    assert np.array_equal(dev_fsa.fsa_arc_offsets, host_fsa.fsa_arc_offsets)
    assert np.array_equal(dev_fsa.fsa_num_states, host_fsa.fsa_num_states)
    assert np.array_equal(dev_fsa.arcs, host_fsa.arcs)
    assert np.array_equal(dev_fsa.aux_labels, host_fsa.aux_labels)
    names = {v: k for k, v in INFO.items()}
    for g in range(B):
        hi = hb[HDR_WORDS + g * INFO_WORDS:HDR_WORDS + (g + 1) * INFO_WORDS]
        di = db[HDR_WORDS + g * INFO_WORDS:HDR_WORDS + (g + 1) * INFO_WORDS]
        d = {k: int(hi[v]) for k, v in INFO.items()}
        for w in range(INFO_WORDS):
            name = names.get(w, "word%d" % w)
            if name in OFFSET_WORDS:
                assert (hi[w] == 0) == (di[w] == 0), (g, name)
                assert hi[w] - hi[INFO["IN_OFF"]] == di[w] - di[INFO["IN_OFF"]] or hi[w] == 0, (g, name)   # the same place in its region
            else:
                assert hi[w] == di[w], (g, name, int(hi[w]), int(di[w]))
        for name, length in TABLES.items():
            n = length(d)
            ho, do = int(hi[INFO[name]]), int(di[INFO[name]])
            assert np.array_equal(hb[ho:ho + n], db[do:do + n]), (g, name)
            pad = (-n) % 4
            assert not db[do + n:do + n + pad].any(), (g, name, "padding")


CASES = [(name, 9) for name in sorted(wf.TOPOLOGIES)] + [("ctc", 5000), ("mmictc", 5000), ("3state-skip", 300)]


@pytest.mark.parametrize("name,P", CASES)
def test_device_compiler_equals_host_compiler(name, P):
    rng = np.random.default_rng(len(name) * 31 + P)
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    topo = wf.Topology(name, phones)
    lo = 1 if (name.endswith("sil") or name.endswith("sil3") or "noblk" in name) else 2
    seqs = [[int(x) for x in rng.integers(lo, P + 2, size=n)] for n in (5, 1, 0, 17, 48, 3, 40)]
    seqs.append([lo + 1] * 6)                      # one phone repeated: the CTC topology has no direct arc between equal phones
    seqs.append([lo, lo, lo + 1, lo + 1, lo, lo])  # repeats in pairs
    host = wf.compile_ctc_k2_graphs(topo, seqs)
    dev = wf.compile_ctc_k2_graphs(topo, seqs, device=DEV)
    assert_same_graph_vector(dev, host)
    # the loss from the device-built graphs: the same kernels on the same tables -> the same bits
    B = len(seqs)
    T = 90
    V = topo.num_nn_output
    g = torch.Generator().manual_seed(5)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g) * 3, -1).numpy()
    segs = np.stack([np.arange(B), np.zeros(B), np.full(B, T)], 1).astype(np.int32)
    want = fused_logsum(host, lp, segs)
    got = fused_logsum(wf.compile_ctc_k2_graphs(topo, seqs, device=DEV), lp, segs)
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])


def test_device_compiler_at_cfg3_size_and_host_fallback():
    P, L, B = 5000, 64, 32
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    topo = wf.Topology("ctc", phones)
    rng = np.random.default_rng(777)
    flat = rng.integers(2, P + 2, size=B * L).astype(np.int32)
    host = wf.compile_ctc_k2_graphs(topo, (flat, np.full(B, L)))
    dev = wf.compile_ctc_k2_graphs(topo, (flat, np.full(B, L)), device=DEV)
    assert_same_graph_vector(dev, host)
    # a transcript beyond the kernel's 256 positions: the batch is compiled on the host instead, same result
    seqs = [[int(x) for x in rng.integers(2, 40, size=n)] for n in (300, 5)]
    small = wf.Topology("2state", ["<eps>", "<SIL>"] + ["p%d" % i for i in range(40)])
    h2 = wf.compile_ctc_k2_graphs(small, seqs)
    d2 = wf.compile_ctc_k2_graphs(small, seqs, device=DEV)
    d2.blob_host()
    assert d2.compiled_on.startswith("host")
    assert np.array_equal(d2.arcs, h2.arcs) and np.array_equal(d2.fsa_num_states, h2.fsa_num_states)
    x = torch.randn(2, 400, small.num_nn_output).log_softmax(-1)
    segs = np.array([[0, 0, 400], [1, 0, 400]], dtype=np.int32)
    want = fused_logsum(h2, x.numpy(), segs)
    got = fused_logsum(d2, x.numpy(), segs)
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
