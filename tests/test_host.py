"""CPU tests of the host layer: topologies vs the reference generators' output, the numerator graph
compiler vs the oracle, graph preparation, and the C ABI surface (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import waterfall_amd as wf
from waterfall_amd import _lib
from oracle import graphs as og
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", sorted(wf.TOPOLOGIES))
def test_topology_text_equals_reference_generator_output(name):
    # tests/golden/topo_<name>.txt is the stdout of the reference's utils/get_token_fst_*_k2.py
    want = open(os.path.join(util.GOLDEN, f"topo_{name}.txt")).read().split("\n")
    got = wf.topology_text(name, util.PHONES).split("\n")
    norm = lambda ls: [" ".join(l.split()) for l in ls if l.strip()]
    assert norm(got) == norm(want)


@pytest.mark.parametrize("name", ["2state_blk", "mmictc_blk"])
def test_deprecated_blk_topology_text_equals_reference_generator_output(name):
    # deprecated/get_token_fst_{2state,mmictc}_blk_k2.py run on a phones.txt with disambiguation symbols (make_golden.py)
    phones = [l.split()[0] for l in open(os.path.join(util.GOLDEN, "phones_disambig.txt"))]
    want = open(os.path.join(util.GOLDEN, f"topo_{name}.txt")).read().split("\n")
    got = wf.topology_text(name, phones).split("\n")
    norm = lambda ls: [" ".join(l.split()) for l in ls if l.strip()]
    assert norm(got) == norm(want)


@pytest.mark.parametrize("name,S,A", [("ctc", lambda P: 1 + P, lambda P: 1 + 3 * P + P * (P - 1)),
                                      ("2state", lambda P: 1 + 2 * P, lambda P: 1 + 6 * P),
                                      ("2state-1", lambda P: 1 + 2 * P, lambda P: 1 + 5 * P),
                                      ("3state-skip", lambda P: 1 + 2 * P, lambda P: 1 + 5 * P),
                                      ("3state-skip-1", lambda P: 1 + 3 * P, lambda P: 1 + 9 * P),
                                      ("3state-skip-2", lambda P: 1 + 3 * P, lambda P: 1 + 10 * P),
                                      ("mmictc", lambda P: 1 + 2 * P, lambda P: 1 + 6 * P),
                                      ("mmictc-1", lambda P: 1 + 2 * P, lambda P: 1 + 7 * P)])
def test_topology_sizes_match_survey_table(name, S, A):
    P = 29
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    topo, tokens = wf.build_topology(name, phones)
    nfinal = (1 + P) if name == "ctc" else 1
    assert topo.fsa_num_states[0] == S(P) + 1            # + super-final
    assert topo.num_arcs == A(P) + nfinal                # + label -1 arcs
    assert topo.labels.max() == len(tokens) - 1 and set(topo.labels[topo.labels >= 0]) == set(range(len(tokens)))


def test_from_openfst_matches_oracle_parser():
    txt = "0 1 1 5 0.5\n1 1 2 0\n1 0.25\n0 0 3 0 1.5\n"
    f = wf.Fsa.from_openfst(txt, acceptor=False)
    g = og.from_openfst_text(txt)
    h = f.as_oracle_graphs()[0]
    for k in ("src", "dst", "label", "aux", "score"):
        assert np.array_equal(g[k], h[k]), k
    assert h["num_states"] == 3 and list(h["label"]) == [1, 3, 2, -1]
    assert list(h["score"]) == [-0.5, -1.5, 0.0, -0.25]


@pytest.mark.parametrize("name", sorted(wf.TOPOLOGIES))
def test_compose_matches_oracle_compose(name):
    topo, tokens = wf.build_topology(name, util.PHONES)
    tg = topo.as_oracle_graphs()[0]
    sil = "sil" in name or "noblk" in name
    seqs = [[2, 2, 3], [6], [4, 5, 6, 2, 2, 2, 3]]
    if sil:
        seqs = [[1] + s + [1] for s in seqs]
    vec = wf.compile_ctc_k2_graphs(topo, seqs)
    for got, seq in zip(vec.as_oracle_graphs(), seqs):
        want = og.compose_topo_transcript(tg, seq)
        assert got["num_states"] == want["num_states"]
        for k in ("src", "dst", "label", "aux", "score"):
            assert np.array_equal(got[k], want[k]), (name, k)


def test_compose_empty_transcript_and_ctc_sizes():
    topo, _ = wf.build_topology("ctc", util.ctc_phones(12))
    vec = wf.compile_ctc_k2_graphs(topo, [[], [3, 3], list(range(1, 11))])
    S = vec.fsa_num_states
    assert S[0] == 2            # only (0,0) + super-final: accepts blanks only
    # ctc numerators S ~ 2L+2
    assert S[1] == 2 * 2 + 2 and S[2] == 2 * 10 + 2


def test_prepared_blob_is_consistent():
    from waterfall_amd.csrc_layout import INFO, HDR_WORDS, INFO_WORDS
    c = util.random_case(1, "mmictc", B=2, T=5, L=3, P=20)
    for fsa in (c["fsa"], wf.create_fsa_vec([c["topo"]] * 2)):
        b = fsa.blob_host()
        for g in range(fsa.num_fsas):
            I = b[HDR_WORDS + g * INFO_WORDS: HDR_WORDS + (g + 1) * INFO_WORDS]
            S, A, K = I[INFO["S"]], I[INFO["A"]], I[INFO["K"]]
            arcs = fsa.arcs[fsa.fsa_arc_offsets[g]:fsa.fsa_arc_offsets[g + 1]]
            em = arcs[arcs[:, 2] >= 0]
            assert A == len(em) and S == fsa.fsa_num_states[g]
            in_off = b[I[INFO["IN_OFF"]]: I[INFO["IN_OFF"]] + S + 1]
            in_src = b[I[INFO["IN_SRC"]]: I[INFO["IN_SRC"]] + A]
            in_arc = b[I[INFO["IN_ARC"]]: I[INFO["IN_ARC"]] + A]
            assert in_off[0] == 0 and in_off[-1] == A and np.all(np.diff(in_off) >= 0)
            for s in range(S):
                ids = in_arc[in_off[s]:in_off[s + 1]]
                assert np.all(arcs[ids, 1] == s) and np.all(np.diff(ids) > 0)
                assert np.array_equal(arcs[ids, 0], in_src[in_off[s]:in_off[s + 1]])
            lab_off = b[I[INFO["LAB_OFF"]]: I[INFO["LAB_OFF"]] + K + 1]
            assert lab_off[-1] == A
            labels = b[I[INFO["LABELS"]]: I[INFO["LABELS"]] + K]
            if I[INFO["DIRECT"]]:
                assert K == em[:, 2].max() + 1
            else:
                assert np.array_equal(labels, np.unique(em[:, 2]))
            nh = I[INFO["NHEAVY_IN"]]
            heavy = set(b[I[INFO["HEAVY_IN"]]: I[INFO["HEAVY_IN"]] + nh].tolist())
            assert heavy == set(np.nonzero(np.diff(in_off) > 16)[0].tolist())


def test_graph_validation_errors():
    bad = wf.Fsa(np.array([[0, 5, 1, 0]], dtype=np.int32), None, [0, 1], [2])
    with pytest.raises(_lib.WfError, match="state out of range"):
        bad.blob_host()
    bad = wf.Fsa(np.array([[0, 1, 1, 0]], dtype=np.int32), None, [0, 1], [2])
    with pytest.raises(_lib.WfError, match="only label -1 arcs may enter the final state"):
        bad.blob_host()
    bad = wf.Fsa(np.array([[0, 0, -1, 0]], dtype=np.int32), None, [0, 1], [2])
    with pytest.raises(_lib.WfError, match="must enter the final state"):
        bad.blob_host()


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "wf_dwfst.h")).read()
    declared = set(re.findall(r"\b(wf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"wf_graph_host"}
    lib = _lib.lib()
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.wf_version() >= 1
    # error path through the ABI: null graph -> WF_EINVAL + message, no exception
    n = C.c_size_t(0)
    assert lib.wf_graph_prepare_bytes(None, C.byref(n)) == -1
    assert b"null" in lib.wf_last_error()


def test_no_cpu_fallback_in_product():
    # the product must not import the oracle, and compute calls must refuse CPU tensors
    import torch
    src = ""
    for fn in os.listdir(os.path.join(ROOT, "waterfall_amd")):
        if fn.endswith(".py"):
            src += open(os.path.join(ROOT, "waterfall_amd", fn)).read()
    assert "import oracle" not in src and "from oracle" not in src
    c = util.random_case(2, "ctc", B=2, T=6, L=2, P=3)
    dense = wf.DenseFsaVec(torch.from_numpy(c["lp"]), torch.from_numpy(c["segs"]))
    with pytest.raises(_lib.WfError, match="no CPU path"):
        wf.graphloss(c["fsa"], dense)


@pytest.mark.parametrize("name", sorted(wf.TOPOLOGIES))
def test_lazy_restricted_topology_composes_to_the_same_graph(name):
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(12)]
    lazy = wf.Topology(name, phones)
    sil = "sil" in name or "noblk" in name
    seqs = [[5, 5, 9, 2], [13], [3, 8, 3, 8, 3]]
    if sil:
        seqs = [[1] + s + [1] for s in seqs]
    a = wf.compile_ctc_k2_graphs(lazy.fsa, seqs)
    b = wf.compile_ctc_k2_graphs(lazy, seqs)
    assert np.array_equal(a.arcs, b.arcs) and np.array_equal(a.aux_labels, b.aux_labels)
    assert np.array_equal(a.fsa_num_states, b.fsa_num_states)


def test_layout_mirror_in_sync_with_header():
    import re
    from waterfall_amd import csrc_layout as L
    h = open(os.path.join(ROOT, "waterfall_amd", "csrc", "wf_internal.h")).read()
    assert {k: int(v) for k, v in re.findall(r"WF_I_([A-Z_0-9]+) = (\d+)", h)} == L.INFO
    assert int(re.search(r"#define WF_INFO_WORDS (\d+)", h).group(1)) == L.INFO_WORDS


def _compose_one_by_one(topo, seqs):
    """wf_compose_linear (the per-utterance entry point) on the full topology."""
    lib = _lib.lib()
    ta, tx = topo.arcs, topo.aux_labels
    out = []
    for seq in seqs:
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        na, ns = C.c_int32(0), C.c_int32(0)
        _lib.check(lib.wf_compose_linear(ta.ctypes.data, tx.ctypes.data, ta.shape[0], int(topo.fsa_num_states[0]),
                                         seq.ctypes.data, len(seq), None, None, C.byref(na), C.byref(ns), 0))
        oa = np.empty((na.value, 4), dtype=np.int32)
        ox = np.empty(na.value, dtype=np.int32)
        _lib.check(lib.wf_compose_linear(ta.ctypes.data, tx.ctypes.data, ta.shape[0], int(topo.fsa_num_states[0]),
                                         seq.ctypes.data, len(seq), oa.ctypes.data, ox.ctypes.data, C.byref(na),
                                         C.byref(ns), na.value))
        out.append((oa, ox, ns.value))
    return out


@pytest.mark.parametrize("name", sorted(wf.TOPOLOGIES))
def test_batched_compiler_equals_per_utterance_compose(name):
    rng = np.random.default_rng(5)
    P = 9
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    sil = "sil" in name or "noblk" in name
    seqs = [[int(x) + 2 for x in rng.integers(0, P, size=L)] for L in (0, 1, 2, 5, 9, 17, 40)]
    if sil:
        seqs = [[1] + s + [1] for s in seqs]
    t = wf.Topology(name, phones)
    want = _compose_one_by_one(t.fsa, seqs)
    variants = [wf.Topology(name, phones), wf.Topology(name, phones).fsa]
    if name == "ctc":                       # the factored form used for large phone tables (implicit O(P^2) arcs)
        f = wf.Topology(name, phones)
        f.LAZY_CTC_PHONES = 0
        assert f.compose_arcs()["ctc_cross"] == 1 and f.compose_arcs()["arcs"].shape[0] == 4 * P + 2
        variants.append(f)
    for v in variants:
        got = wf.compile_ctc_k2_graphs(v, seqs)
        for i, (oa, ox, ns) in enumerate(want):
            a0, a1 = got.fsa_arc_offsets[i], got.fsa_arc_offsets[i + 1]
            assert np.array_equal(got.arcs[a0:a1], oa) and np.array_equal(got.aux_labels[a0:a1], ox), (name, i)
            assert got.fsa_num_states[i] == ns


def test_batched_compiler_large_table_ctc_and_errors():
    # BPE-5k sized table: the factored CTC topology has 4P+2 arcs instead of P^2+2P+1 (25 M)
    P = 5000
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    t = wf.Topology("ctc", phones)
    seqs = [[2, 2, 5001, 77], [4999]]
    vec = wf.compile_ctc_k2_graphs(t, seqs)
    assert list(vec.fsa_num_states) == [2 * 4 + 2, 2 * 1 + 2]
    small = wf.Topology("ctc", ["<eps>", "<SIL>", "a", "b", "c"])          # same transcript shape on a tiny table
    ref = wf.compile_ctc_k2_graphs(small.fsa, [[2, 2, 4, 3], [3]])
    assert np.array_equal(vec.arcs[:, :2], ref.arcs[:, :2])                  # same structure, other token ids
    with pytest.raises(_lib.WfError, match="out of range"):
        wf.compile_ctc_k2_graphs(t, [[5002]])
    with pytest.raises(ValueError, match=">= 1"):
        wf.compile_ctc_k2_graphs(t, [[0, 3]])
    # arc capacity too small: sizes are reported, the caller retries (compile.compose does)
    lib = _lib.lib()
    tab = __import__("waterfall_amd.compile", fromlist=["_topo_tables"])._topo_tables(small)
    flat = np.array([2, 2, 4, 3], dtype=np.int32)
    off = np.array([0, 4], dtype=np.int32)
    arc_off = np.zeros(2, dtype=np.int32)
    ns = np.zeros(1, dtype=np.int32)
    oa = np.empty((2, 4), dtype=np.int32)
    ox = np.empty(2, dtype=np.int32)
    rc = lib.wf_compose_linear_batch(tab["arcs"].ctypes.data, tab["aux"].ctypes.data, tab["arcs"].shape[0], tab["num_states"],
                                     tab["index"].ctypes.data, 0, None, None, 0, flat.ctypes.data, off.ctypes.data, 1,
                                     oa.ctypes.data, ox.ctypes.data, 2, arc_off.ctypes.data, ns.ctypes.data)
    assert rc == -2 and arc_off[1] == ref.fsa_arc_offsets[1] and ns[0] == ref.fsa_num_states[0]


class _RaggedStandIn:   # stands for a pickled k2.RaggedTensor (a class the weights-only loader must refuse)
    pass


def test_decoding_graph_file_round_trip_weights_only(tmp_path):
    """TLG.pt layout of utils/compile_TLG.py:129 (torch.save(TLG.as_dict())): tensor-only dicts load with the
    weights-only loader; anything that needs unpickling a class is refused with a clear error."""
    import torch
    from waterfall_amd import decode
    import waterfall_amd as wf
    arcs = [(0, 1, 1, -0.5), (1, 1, 1, 0.0), (1, 2, 2, -1.25), (2, 3, -1, 0.0)]
    g = wf.Fsa.from_arcs(arcs, 4, aux_labels=[7, 0, 9, -1])
    p = tmp_path / "TLG.pt"
    decode.save_decoding_graph(g, p)
    h = decode.load_decoding_graph(p)
    assert np.array_equal(h.arcs, g.arcs) and np.array_equal(h.aux_labels, g.aux_labels)
    assert list(h.fsa_num_states) == [4]
    # ragged aux labels as two tensors
    torch.save({"arcs": torch.from_numpy(g.arcs.copy()), "aux_labels.row_splits": torch.tensor([0, 1, 1, 2, 2]),
                "aux_labels.values": torch.tensor([7, 9], dtype=torch.int32)}, p)
    h = decode.load_decoding_graph(p)
    assert list(h.aux_labels) == [7, 0, 9, -1]

    torch.save({"arcs": torch.from_numpy(g.arcs.copy()), "aux_labels": _RaggedStandIn()}, p)
    with pytest.raises(ValueError, match="weights_only"):
        decode.load_decoding_graph(p)


def test_output_beam_default_mode_is_the_reference_semantics():
    """graphloss(..., output_beam=10) (graph.py:81) prunes by default, as k2.intersect_dense(graph, dense, output_beam)
    does (graph.py:61-62); "exact" (what the autouse fixture selects for the parity tests) ignores the beam."""
    from waterfall_amd import graph as g
    assert g.DEFAULT_BEAM_MODE == "auto"       # pruned wherever the pruned lattice has a fast path (GPU test: test_pruned.py)
    assert g._BEAM_MODE == "exact" and g._beam_value(10) == 0.0          # inside a test: the fixture's mode
    wf.set_output_beam_mode("pruned")
    try:
        assert g._beam_value(10) == 10.0 and g._beam_value(None) == 0.0 and g._beam_value(float("inf")) == 0.0
    finally:
        wf.set_output_beam_mode("exact")


def test_numerator_compiler_works_in_a_forked_child():
    """A torch DataLoader worker / multiprocessing child forked AFTER the parent used the compiler's thread pool inherits
    the pool object without its threads: the pool must notice and start its own instead of waiting for them forever."""
    import signal
    topo = wf.Topology("2state", ["<eps>", "<SIL>"] + ["p%d" % i for i in range(9)])
    seqs = [[2, 3, 4], [5, 5], [6, 7, 8, 9], [2], [3, 4], [10, 9, 8], [4, 4, 4], [7]]
    want = wf.compile_ctc_k2_graphs(topo, seqs)          # the parent's pool now has worker threads
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            signal.alarm(60)                              # a deadlock ends the child, not the test run
            got = wf.compile_ctc_k2_graphs(topo, seqs)
            code = 0 if (np.array_equal(got.arcs, want.arcs) and np.array_equal(got.aux_labels, want.aux_labels)
                         and np.array_equal(got.fsa_arc_offsets, want.fsa_arc_offsets)) else 2
        finally:
            os.write(w, bytes([code]))
            os._exit(code)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    out = os.read(r, 1)
    os.close(r)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0 and out == b"\x00"


def test_manual_ctc_matrices_to_fsa_vectorised_equals_the_loop():
    # the batched masked-nonzero build of the graph (one-hot transform rows) against the per-utterance loop kept for rows
    # with several tokens: the same FsaVec, arc for arc
    import torch
    from oracle import graphs as og
    from waterfall_amd import manual_ctc
    rng = np.random.default_rng(5)
    V = 11
    mats = [og.ctc_trans_transform(rng.integers(1, V, size=n).tolist(), V) for n in (5, 1, 8, 3)]
    S = max(m[0].shape[0] for m in mats)
    B = len(mats)
    trans = np.full((B, S, S), og.FMIN, dtype=np.float32)
    transform = np.full((B, S, V), og.FMIN, dtype=np.float32)
    lens = np.zeros(B, dtype=np.int64)
    for b, (t, f) in enumerate(mats):
        s_ = t.shape[0]
        trans[b, :s_, :s_] = t + rng.normal(size=t.shape).astype(np.float32) * (t > og.FMIN / 2)   # weighted transitions
        transform[b, :s_] = f
        lens[b] = s_
    trans[0, 1, 7] = 0.5          # a transition out of a padded... no: inside utterance 0 (S_0 = 11): a backward jump
    tt, tf, tl = torch.from_numpy(trans), torch.from_numpy(transform), torch.from_numpy(lens)
    a = manual_ctc.matrices_to_fsa(tt, tl, tf)
    b_ = manual_ctc._matrices_to_fsa_multi(tt, lens, tf, tf > manual_ctc._THR)
    assert np.array_equal(a.arcs, b_.arcs)
    assert np.array_equal(a.fsa_arc_offsets, b_.fsa_arc_offsets) and np.array_equal(a.fsa_num_states, b_.fsa_num_states)
    with pytest.raises(ValueError):
        manual_ctc.matrices_to_fsa(tt, torch.tensor([0, 3, 17, 7]), tf)


def test_lattice_analysis_graph_side_calls_on_the_host():
    """utils/lattice_analysis.py:137-145 on the host: W = arc_sort(linear_fst(words, words)); connect(compose(TLG, W)) is
    the decoding graph restricted to the word sequence (trimmed, arcs sorted by source, start 0, super-final last)."""
    import waterfall_amd as k2
    # start -a:1-> 1 -b:0-> 2 -c:2-> 3 -> final ; and a bypass 0 -d:3-> 3
    arcs = [(0, 1, 5, -0.5), (0, 3, 6, -1.0), (1, 2, 7, -0.25), (2, 3, 8, -0.125), (3, 4, -1, 0.0)]
    TLG = k2.Fsa.from_arcs(arcs, 5, aux_labels=[1, 3, 0, 2, -1])
    W = k2.arc_sort(k2.linear_fst([[1, 2]], [[1, 2]]))
    TLGW = k2.connect(k2.compose(TLG, W))
    assert isinstance(TLGW, k2.Fsa) and TLGW.num_fsas == 1
    a = TLGW.arcs
    assert a.shape[0] == 4 and list(a[:, 2]) == [5, 7, 8, -1]              # the bypass (word 3) is gone
    assert list(TLGW.aux_labels) == [1, 0, 2, -1]
    assert np.all(np.diff(a[:, 0]) >= 0) and a[0, 0] == 0 and a[-1, 1] == int(TLGW.fsa_num_states[0]) - 1
    assert k2.compose(TLG, k2.linear_fst([[3]], [[3]])).arcs.shape[0] == 2   # 0 -d:3-> 3 -> final
    assert k2.compose(TLG, k2.linear_fst([[2, 1]], [[2, 1]])).arcs.shape[0] == 0
    two = k2.compose(TLG, k2.linear_fst([[1, 2], [3]], [[1, 2], [3]]))
    assert two.num_fsas == 2
    assert k2.invert(k2.invert(TLG)) is TLG
