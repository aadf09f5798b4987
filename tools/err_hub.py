#!/usr/bin/env python3
"""Error of the hub kernels (fp32 recursion, fp32 rows) against the fp64 C oracle on cfg4's denominators at full graph
size, for T = 500 and a long T = 1500 (the fp32 recursion's error grows with the number of frames)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_gpu_parity import fused_logsum
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
for topo, T in (("mmictc", 500), ("3state-skip", 500), ("mmictc", 1500)):
    c = util.random_case(779, topo, B=2, T=T, L=8, P=5000, den=True, ragged=False)
    tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"])
    tot_d, grad_d = util.c_oracle_batch(lib, c["ographs"], c["lp"], c["segs"], gscale=[-1.0] * 2, use_double=True, nthreads=2)
    rows = np.abs(grad.sum(-1) + 1.0).max()
    print(topo, "T=%d" % T, "max rel tot err %.3g" % np.max(np.abs(tot - tot_d) / np.abs(tot_d)), "abs tot err %.3g" % np.max(np.abs(tot - tot_d)),
          "max abs grad err %.3g" % np.max(np.abs(grad - grad_d)), "max |row sum + 1| %.3g" % rows, flush=True)
