import time, numpy as np
import waterfall_amd as wf
rng=np.random.default_rng(0)
P=5000;L=64;B=32
phones=["<eps>","<SIL>"]+["p%d"%i for i in range(P)]
for topo in ["ctc","2state","mmictc","3state-skip"]:
    lazy=wf.Topology(topo,phones)
    seqs=[[int(x)+2 for x in rng.integers(0,P,size=L)] for _ in range(B)]
    wf.compile_ctc_k2_graphs(lazy,seqs)
    tc=[];tp=[]
    for _ in range(20):
        t0=time.perf_counter(); fsa=wf.compile_ctc_k2_graphs(lazy,seqs); t1=time.perf_counter(); gh=fsa.blob_host(); t2=time.perf_counter()
        tc.append(t1-t0); tp.append(t2-t1)
    print(topo,"compile %.3f ms, prepare %.3f ms"%(np.median(tc)*1e3,np.median(tp)*1e3))
