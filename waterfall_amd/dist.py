"""Multi-GPU side of the hot path: utterances are independent, so a batch is sharded over
one process per GPU with NO data-path collective; the only exchange is one fused all-reduce per step of the
scalars the training loop logs / normalises by (the reference gets the same numbers from Lightning's DDP
reductions around graph.py:66-76).  The dense gradient stays sharded exactly as under DDP (it feeds each
rank's own encoder backward).  Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPU
box, "gloo" in the CPU tests.
"""
import numpy as np
import torch

__all__ = ["shard_batch", "fused_loss_stats", "STATS_FIELDS"]

STATS_FIELDS = ("loss_sum", "n_frames", "n_infeasible")


def shard_batch(lengths, world, rank, mode="round_robin"):
    """Indices (into the batch) of the utterances rank `rank` of `world` processes.

    The reference sorts a batch by length, longest first (datapipe.py:592-615).  "round_robin" deals that order
    out like cards, so every rank gets the same mix of long and short utterances (the sweep's duration is set by
    the longest utterance of a shard); "contiguous" cuts the sorted order into `world` runs of equal count.
    Every utterance lands on exactly one rank; ranks differ by at most one utterance.
    """
    lengths = np.asarray(lengths)
    n = lengths.shape[0]
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    order = np.argsort(-lengths, kind="stable")
    if mode == "round_robin":
        return order[rank::world]
    if mode == "contiguous":
        base, extra = divmod(n, world)
        lo = rank * base + min(rank, extra)
        return order[lo:lo + base + (1 if rank < extra else 0)]
    raise ValueError(f"unknown shard mode {mode!r}")


def fused_loss_stats(tot, n_frames, group=None, grad=None, out=None, async_op=False):
    """One all-reduce(SUM) of [loss_sum, n_frames, n_infeasible (, sum_{b,t} grad[b,t,:])] over the job.

    tot: [B_local] total scores of this rank's shard (-inf = infeasible utterance: counted, not summed, as the
    reference's loss would otherwise be inf for the whole job).  n_frames: number of frames of the shard.
    grad: optional [B_local, T, V] gradient whose per-label column sums are appended (V floats).
    out: optional preallocated float32 buffer of the right size on tot's device (no allocation in the step).
    async_op: start the all-reduce without making the current stream wait for it; returns (buffer, work) and the
    caller calls work.wait() before it reads or reuses the buffer (bench.py alternates two buffers, so the collective
    of step i overlaps the sweeps of step i + 1: the statistics are logging data, not on the data path).
    Returns the reduced buffer; `dict(zip(STATS_FIELDS, buf[:3].tolist()))` names the scalars.
    """
    import torch.distributed as dist
    V = grad.shape[-1] if grad is not None else 0
    if out is None:
        out = torch.empty(3 + V, dtype=torch.float32, device=tot.device)
    elif out.numel() != 3 + V or out.dtype != torch.float32:
        raise ValueError(f"out must be float32 with {3 + V} elements")
    if tot.device.type == "cuda" and tot.dtype == torch.float32 and tot.is_contiguous():
        # one launch instead of eight elementwise ones (they cost as much as a quarter of a cfg3 step)
        import ctypes as C
        from . import _lib
        with torch.cuda.device(tot.device):
            _lib.check(_lib.lib().wf_loss_stats(C.c_void_p(tot.data_ptr()), tot.numel(), float(n_frames),
                                                C.c_void_p(out.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(tot.device).cuda_stream)))
    else:
        ok = torch.isfinite(tot)
        out[0] = -torch.where(ok, tot, torch.zeros_like(tot)).sum()
        out[1] = float(n_frames)
        out[2] = (~ok).sum()
    if V and grad.device.type == "cuda" and grad.dtype == torch.float32 and grad.dim() == 3 and grad.stride(2) == 1:
        # one streaming kernel straight into the tail of the buffer (no [B*T, V] torch reduction, no temporary)
        import ctypes as C
        from . import _lib
        with torch.cuda.device(grad.device):
            _lib.check(_lib.lib().wf_label_grad_sums(C.c_void_p(grad.data_ptr()), grad.stride(0), grad.stride(1),
                                                     grad.shape[0], grad.shape[1], V, C.c_void_p(out[3:].data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream(grad.device).cuda_stream)))
    elif V:
        out[3:] = grad.reshape(-1, V).sum(0)
    work = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(out, group=group, async_op=async_op)
    return (out, work) if async_op else out
