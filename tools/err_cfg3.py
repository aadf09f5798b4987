#!/usr/bin/env python3
"""Max error of the HIP path against the fp64 C oracle at BASELINE sizes (cfg3 / cfg2)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_gpu_parity import fused_logsum
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
for topo, P, L, den in (("ctc", 5000, 64, False), ("2state", 29, 80, False), ("2state", 29, 80, True)):
    c = util.random_case(777, topo, B=32, T=500, L=L, P=P, den=den, ragged=True)
    tot, grad = fused_logsum(c["fsa"], c["lp"], c["segs"])
    tot_d, grad_d = util.c_oracle_batch(lib, c["ographs"], c["lp"], c["segs"], gscale=[-1.0] * 32, use_double=True, nthreads=16)
    print(topo, "den" if den else "num", "max rel tot err %.3g" % np.max(np.abs(tot - tot_d) / np.abs(tot_d)),
          "max abs grad err %.3g" % np.max(np.abs(grad - grad_d)))
