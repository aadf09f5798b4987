"""N > 1 path on CPU: two gloo ranks shard one batch, each computes its shard's total scores
with the oracle (test infrastructure: the product path needs a GPU), and the one fused all-reduce per step
must reproduce the single-process numbers."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dwfst_oracle as o
import waterfall_amd as wf
from waterfall_amd.dist import shard_batch, fused_loss_stats, STATS_FIELDS
from . import util


def test_shard_batch_partitions_every_batch():
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 32, 33):
        lens = rng.integers(1, 500, size=n)
        for world in (1, 2, 3, 8):
            for mode in ("round_robin", "contiguous"):
                parts = [shard_batch(lens, world, r, mode) for r in range(world)]
                got = np.sort(np.concatenate(parts))
                assert np.array_equal(got, np.arange(n))
                sizes = [len(p) for p in parts]
                assert max(sizes) - min(sizes) <= 1
    # round robin balances the longest utterance of each shard: rank r gets the r-th longest first
    lens = np.array([10, 50, 30, 40, 20, 60])
    assert [int(lens[shard_batch(lens, 2, r)[0]]) for r in range(2)] == [60, 50]
    assert list(lens[shard_batch(lens, 2, 0, "contiguous")]) == [60, 50, 40]


def _case():
    c = util.random_case(99, "ctc", B=6, T=24, L=4, P=5, ragged=True)
    # one infeasible utterance: fewer frames than its transcript needs
    c["segs"][5, 2] = 2
    return c


def _rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = _case()
        mine = shard_batch(c["segs"][:, 2], world, rank)
        graphs = [c["ographs"][i] for i in mine]
        segs = c["segs"][mine].copy()
        segs[:, 0] = np.arange(len(mine))
        lp = c["lp"][mine]
        tot = np.array([o.forward(g, lp[i], int(segs[i, 2]))[1] for i, g in enumerate(graphs)], dtype=np.float32)
        _, grad = o.graphloss(graphs, lp, segs, None, "none")
        grad = np.where(np.isfinite(grad), grad, 0.0).astype(np.float32)
        buf = torch.empty(3 + lp.shape[-1])
        out = fused_loss_stats(torch.from_numpy(tot), int(segs[:, 2].sum()), grad=torch.from_numpy(grad), out=buf)
        assert out.data_ptr() == buf.data_ptr()
        q.put((rank, out.numpy().copy(), mine))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_reproduce_the_single_process_statistics():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = _case()
    tot = np.array([o.forward(g, c["lp"][i], int(c["segs"][i, 2]))[1] for i, g in enumerate(c["ographs"])])
    ok = np.isfinite(tot)
    assert (~ok).sum() == 1
    _, grad = o.graphloss(c["ographs"], c["lp"], c["segs"], None, "none")
    grad = np.where(np.isfinite(grad), grad, 0.0)
    want = np.concatenate([[-tot[ok].sum(), c["segs"][:, 2].sum(), 1.0], grad.reshape(-1, grad.shape[-1]).sum(0)])
    for rank, buf, mine in got:
        np.testing.assert_allclose(buf, want, rtol=1e-5, atol=1e-4)       # every rank holds the job-wide numbers
    assert np.array_equal(np.sort(np.concatenate([m for _, _, m in got])), np.arange(6))
    assert STATS_FIELDS == ("loss_sum", "n_frames", "n_infeasible")


def test_fused_loss_stats_without_a_process_group_is_local():
    tot = torch.tensor([-3.0, float("-inf"), -1.5])
    out = fused_loss_stats(tot, 77)
    assert out.tolist() == [4.5, 77.0, 1.0]
