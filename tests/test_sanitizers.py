"""CPU test: the host side of the library (waterfall_amd/csrc/wf_graph.cpp -- prepare, topology index, numerator compiler,
composers: ~1000 lines of index arithmetic on caller buffers) built with gcc's AddressSanitizer + UndefinedBehavior-
Sanitizer and driven by tests/san/san_driver.cpp on the inputs below; every result must also equal what the production
library returns.  A sanitizer report aborts the driver (-fno-sanitize-recover=all) and fails the test."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import waterfall_amd as wf
from waterfall_amd import _lib
from waterfall_amd.compile import _topo_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHONES = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(7)]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("san") / "san_driver")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
           "-static-libubsan", "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "san", "san_driver.cpp"),
           os.path.join(ROOT, "waterfall_amd", "csrc", "wf_graph.cpp"), "-o", out, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


class Stream:
    def __init__(self):
        self.parts = []

    def op(self, code):
        self.parts.append(np.array([code], dtype=np.int32))

    def scalar(self, v):
        self.parts.append(np.array([v], dtype=np.int32))

    def arr(self, a):
        a = np.zeros(0, dtype=np.int32) if a is None else np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
        self.parts.append(np.array([a.size], dtype=np.int32))
        self.parts.append(a)

    def write(self, path):
        np.concatenate(self.parts).tofile(path)


def read_arrays(path):
    d = np.fromfile(path, dtype=np.int32)
    out, i = [], 0
    while i < d.size:
        n = int(d[i])
        out.append(d[i + 1:i + 1 + n])
        i += 1 + n
    return out


def run(driver, stream, tmp_path):
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    stream.write(fin)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, "sanitizer driver failed (rc %d):\n%s" % (r.returncode, r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
    return read_arrays(fout)


def seqs_for(name, rng, B, Lmax):
    sil = name.endswith("sil") or name.endswith("sil3") or "noblk" in name
    out = []
    for b in range(B):
        s = [int(x) + 2 for x in rng.integers(0, 7, size=int(rng.integers(0 if b == 1 else 1, Lmax + 1)))]
        out.append(([1] + s + [1]) if sil else s)
    return out


def test_prepare_index_compile_and_compose_under_asan_ubsan(driver, tmp_path):
    rng = np.random.default_rng(5)
    names = ["ctc", "2state", "2state_sil3", "3state-skip", "3state-skip-2", "mmictc", "mmictc_noblk", "mmictc-1"]
    st, want = Stream(), []
    for name in names:
        topo = wf.Topology(name, PHONES)
        f = topo.fsa
        # 1: prepare of the topology (a denominator: direct columns, heavy start state, hub form for some)
        st.op(1); st.arr(f.arcs); st.arr(f.fsa_arc_offsets); st.arr(f.fsa_num_states)
        want.append(("prepare " + name, [np.array([0]), f.blob_host()]))
        # 2: the numerator compiler on a ragged batch (an empty transcript included), 1 and 3 host threads
        seqs = seqs_for(name, rng, B=5, Lmax=9)
        tab = _topo_tables(f)
        g = wf.compile_ctc_k2_graphs(f, seqs)
        flat = np.array([x for s in seqs for x in s], dtype=np.int32)
        off = np.cumsum([0] + [len(s) for s in seqs]).astype(np.int32)
        for nthreads in (1, 3):
            st.op(2); st.arr(tab["arcs"]); st.arr(tab["aux"]); st.scalar(tab["num_states"]); st.scalar(0)
            st.arr(None); st.arr(None); st.arr(flat); st.arr(off); st.scalar(nthreads)
            want.append(("numerator %s x%d" % (name, nthreads),
                         [np.array([0]), g.arcs.reshape(-1), g.aux_labels, g.fsa_arc_offsets, g.fsa_num_states, None,
                          np.array([0]), np.array([1])]))
        # 4: one transcript through the two-call single-graph composer
        st.op(4); st.arr(tab["arcs"]); st.arr(tab["aux"]); st.scalar(tab["num_states"]); st.arr(seqs[0])
        one = wf.compile_ctc_k2_graphs(f, [seqs[0]])
        want.append(("compose_linear " + name, [np.array([0]), one.arcs.reshape(-1), one.aux_labels, one.fsa_num_states[:1]]))
    # the factored large-table CTC topology (cross arcs implied)
    big = wf.Topology("ctc", ["<eps>", "<SIL>"] + ["p%d" % i for i in range(300)])
    tabf = big.compose_arcs(factored=True)
    seqs = [[int(x) + 2 for x in rng.integers(0, 300, size=n)] for n in (12, 1, 30)]
    g = wf.compile_ctc_k2_graphs(big, seqs)
    st.op(2); st.arr(tabf["arcs"]); st.arr(tabf["aux"]); st.scalar(tabf["num_states"]); st.scalar(1)
    st.arr(tabf["phone_state"]); st.arr(tabf["phone_token"])
    st.arr(np.array([x for s in seqs for x in s])); st.arr(np.cumsum([0] + [len(s) for s in seqs])); st.scalar(2)
    want.append(("numerator factored ctc", [np.array([0]), g.arcs.reshape(-1), g.aux_labels, g.fsa_arc_offsets,
                                            g.fsa_num_states, None, np.array([0]), np.array([1])]))
    # 3: arbitrary right-hand graphs (two pronunciations in parallel, then a shared tail)
    topo = wf.Topology("2state", PHONES).fsa
    tab = _topo_tables(topo)
    b1 = wf.Fsa.from_arcs([(0, 1, 3, -0.5), (0, 2, 4, -1.0), (1, 3, 5, 0.0), (2, 3, 5, 0.0), (3, 4, -1, 0.0)], 5,
                          aux_labels=[11, 12, 0, 0, -1])
    b2 = wf.Fsa.from_arcs([(0, 1, 2, 0.0), (1, 2, -1, 0.0)], 3, aux_labels=[7, -1])
    bv = wf.create_fsa_vec([b1, b2])
    got = wf.compile.compose(topo, bv)
    st.op(3); st.arr(tab["arcs"]); st.arr(tab["aux"]); st.scalar(tab["num_states"])
    st.arr(bv.arcs); st.arr(bv.aux_labels); st.arr(bv.fsa_arc_offsets); st.arr(bv.fsa_num_states)
    want.append(("compose_fsa", [np.array([0]), got.arcs.reshape(-1), got.aux_labels, got.fsa_arc_offsets, got.fsa_num_states]))
    # malformed graphs: an error code, never a wild access
    bad = [
        (np.array([[0, 5, 1, 0], [0, 1, -1, 0]]), [0, 2], [2]),      # state out of range
        (np.array([[0, 0, -2, 0], [0, 1, -1, 0]]), [0, 2], [2]),     # label < -1
        (np.array([[1, 0, 1, 0], [0, 1, -1, 0]]), [0, 2], [2]),      # arc leaves the final state
        (np.array([[0, 1, 1, 0]]), [0, 1], [2]),                     # emitting arc into the final state
        (np.array([[0, 0, 1, 0]]), [0, 1], [1]),                     # fewer than two states
        (np.array([[0, 0, 1, 0]]), [1, 0], [2]),                     # offsets not monotone
    ]
    for arcs, off, ns in bad:
        st.op(1); st.arr(arcs); st.arr(off); st.arr(ns)
        want.append(("malformed", None))
    out = run(driver, st, tmp_path)
    i = 0
    for what, arrays in want:
        if arrays is None:
            assert out[i].size == 1 and out[i][0] < 0, what     # refused with an error code
            i += 1
            continue
        for a in arrays:
            if a is not None:
                assert np.array_equal(out[i], np.asarray(a).reshape(-1)), "%s: output %d differs" % (what, i)
            i += 1
    assert i == len(out)
