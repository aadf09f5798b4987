"""GPU test (-m gpu) of the N > 1 path with the PRODUCT kernels: two ranks (fresh processes, gloo
rendezvous on 127.0.0.1, both on the one GPU this pool gives a job) each run graphloss + backward on the shard
waterfall_amd.dist.shard_batch deals them and pack / all-reduce the fused statistics buffer with fused_loss_stats; the
job-wide numbers and the reassembled per-utterance losses and gradients must equal the single-process run of the same
batch.  Functional only: two processes time-slice the device (RCCL over xGMI needs a multi-GPU node)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import util

pytestmark = pytest.mark.gpu


def _case():
    c = util.random_case(1234, "2state", B=7, T=40, L=5, P=6, ragged=True)
    c["segs"][6, 2] = 3            # one infeasible utterance (fewer frames than its transcript needs)
    return c


def _run(c, idx, with_den):
    """graphloss(numerator) [- graphloss(denominator)] + backward on the utterances idx; returns (tot [n], grad [n,T,V])."""
    import waterfall_amd as wf
    wf.set_output_beam_mode("exact")
    dev = torch.device("cuda:0")
    seqs = [c["seqs"][i] for i in idx]
    fsa = wf.compile_ctc_k2_graphs(c["topo"], seqs)
    segs = c["segs"][idx].copy()
    segs[:, 0] = np.arange(len(idx))
    x = torch.from_numpy(np.ascontiguousarray(c["lp"][idx])).to(dev).requires_grad_(True)
    dense = wf.DenseFsaVec(x, torch.from_numpy(segs))
    loss = wf.graphloss(fsa, dense, reduction="none")
    if with_den:
        den = wf.create_fsa_vec([c["topo"]] * len(idx))
        loss = loss - wf.graphloss(den, dense, reduction="none")
    finite = torch.isfinite(loss)
    torch.where(finite, loss, torch.zeros_like(loss)).sum().backward()      # mask_inf (models.py:305-313)
    return loss.detach(), x.grad, segs


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    import waterfall_amd as wf
    from waterfall_amd.dist import fused_loss_stats, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = _case()
        mine = shard_batch(c["segs"][:, 2], world, rank)
        loss, grad, segs = _run(c, mine, with_den=False)
        buf = fused_loss_stats(-loss.contiguous(), int(segs[:, 2].sum()), grad=grad.contiguous())
        torch.cuda.synchronize()
        q.put((rank, buf.cpu().numpy(), np.asarray(mine), loss.cpu().numpy(), grad.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_run_the_product_kernels_and_agree_with_one_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = _case()
    B = c["segs"].shape[0]
    loss1, grad1, _ = _run(c, np.arange(B), with_den=False)
    loss1, grad1 = loss1.cpu().numpy(), grad1.cpu().numpy()
    ok = np.isfinite(loss1)
    assert (~ok).sum() == 1
    want = np.concatenate([[loss1[ok].sum(), c["segs"][:, 2].sum(), 1.0], grad1.reshape(-1, grad1.shape[-1]).sum(0)])
    loss2 = np.empty_like(loss1)
    grad2 = np.empty_like(grad1)
    for rank, buf, mine, loss, grad in got:
        np.testing.assert_allclose(buf, want, rtol=1e-5, atol=1e-4)        # every rank holds the job-wide statistics
        loss2[mine] = loss
        grad2[mine] = grad
    assert np.array_equal(np.sort(np.concatenate([m for _, _, m, _, _ in got])), np.arange(B))
    # the shards' results are the single-process results, utterance by utterance (the same kernels on the same rows)
    assert np.array_equal(np.isfinite(loss2), ok)
    np.testing.assert_allclose(loss2[ok], loss1[ok], rtol=1e-6)
    np.testing.assert_allclose(grad2, grad1, atol=1e-6)
