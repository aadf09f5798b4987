#!/usr/bin/env python3
"""bench.py -- DWFST loss+grad throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic log-probs already resident in
HBM: emission gather (probabilities) -> concurrent alpha/beta sweeps (with the gradient zero-fill
beside them on a side stream) -> row check -> posterior/gradient rows (loss + gradient w.r.t.
log_probs, reduction 'sum'), called through the C ABI (wf_fb_logsum).  The headline workload
is BASELINE.json configs[2] (the configuration the metric is quoted on: B=32, T=500, V=5k, CTC
numerator graphs, "cfg3" in BASELINE.md); with N GPUs every rank processes its own B=32 batch
(weak scaling, utterances are independent; --scaling strong deals ONE batch of 32 over the ranks, BASELINE.json's
"B=32 sharded over 8 GPUs") and the step ends with ONE RCCL all-reduce of the fused (loss_sum, n_frames,
n_infeasible [, V per-label gradient sums with --grad-sums]) buffer.  The timed region of exactly K steps is repeated
until the regions add up to --min-region-ms (100); value / ms_per_step are the median region, *_min the fastest.
(WF_BENCH_BACKEND=gloo runs the N > 1 path with several ranks on ONE GPU: a functional rehearsal only -- two processes
time-slice the device with a quantum of ~0.1 s per switch, so its timings mean nothing.)

Prints one JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, measured live with HIP
events on the launch stream), "cpu_baseline" (oracle/dwfst_oracle.c, a scalar C port of the same
algorithm, timed on the host cores of this box on a bounded sample), "kernels", "other_workloads".
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import waterfall_amd as wf  # noqa: E402
from waterfall_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
KERNEL_NAMES = {"gather": "gather_lin_small_kernel", "sweep": "sweep_band_kernel", "post": "post_small_kernel",
                "zero": "sweep_band_kernel (fill workgroups of the same launch)", "check": "rescue_kernel"}
PROFILE_TAG = "r03"
PMC_FILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
STATS_FILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_bench_cfg3_kernel_stats.txt")
META_FILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_meta.json")          # {"kernel_source_sha256": ...} at profiling time
KERNEL_SOURCES = [os.path.join(ROOT, "waterfall_amd", "csrc", f) for f in ("wf_kernels.hip", "wf_internal.h")]   # (the profiled kernels' sources)


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def profiles_current():
    """The committed rocprofv3 summaries describe THIS kernel source: their recorded source hash equals the tree's.  When
    it does not (a kernel changed and the profiles were not regenerated) every field read from them is nulled."""
    try:
        return json.load(open(META_FILE))["kernel_source_sha256"] == kernel_source_hash()
    except (OSError, KeyError, ValueError):
        return False


def _pmc_kernel(ks, kernel):
    """Entry of the PMC summary for one of the step's kernels (matched by the stem of its name: the table-in-arguments
    variant of the gather kernel carries a suffix)."""
    name = KERNEL_NAMES.get(kernel, "").split(" ")[0]
    hit = [v for k, v in ks.items() if name and k.startswith(name.replace("_kernel", ""))]
    return hit[0] if hit else None


def pmc_step_traffic(is_default_workload):
    """HBM bytes ONE STEP moves according to the committed PMC passes: the sum over the step's kernels (median launch of
    each), or None without a current measurement."""
    if not is_default_workload or not os.path.exists(PMC_FILE) or not profiles_current():
        return None
    ks = json.load(open(PMC_FILE))["kernels"]
    tot = 0.0
    for kernel in ("gather", "sweep", "post", "check"):
        k = _pmc_kernel(ks, kernel)
        if k is None:
            return None
        tot += k["hbm_bytes_per_launch_fetch_doubled"]
    return tot


def pmc_traffic(kernel, is_default_workload):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as read), or None when there is no
    measurement for this workload and kernel source."""
    if not is_default_workload or not os.path.exists(PMC_FILE) or not profiles_current():
        return None
    k = _pmc_kernel(json.load(open(PMC_FILE))["kernels"], kernel)
    return k["hbm_bytes_per_launch_fetch_doubled"] if k else None


def rocprof_avg_us(kernel, is_default_workload):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this command, for
    comparison with the live HIP-event interval; None when the summary is of another kernel source."""
    name = KERNEL_NAMES.get(kernel, "").split(" ")[0]
    if not is_default_workload or not name or not os.path.exists(STATS_FILE) or not profiles_current():
        return None
    for line in open(STATS_FILE):   # tools/prof_summary.py columns: calls total_us avg_us pct kernel
        if name in line:
            return float(line.split()[2])
    return None


WORKLOADS = {
    # name: (topology, P, L, kind) -- BASELINE.md section 3
    "cfg3_ctc_bpe5k_num": dict(topo="ctc", P=5000, L=64, den=False,
                               desc="LS BPE-5k CTC numerator (ctc_k2 path), V=5001, L=64"),
    "cfg2_2state_char_num": dict(topo="2state", P=29, L=80, den=False,
                                 desc="WSJ char 2state numerator, V=59, L=80"),
    "cfg2_2state_char_den": dict(topo="2state", P=29, L=80, den=True,
                                 desc="WSJ char 2state denominator (topology x B), V=59"),
    "cfg4_mmictc_bpe5k_den": dict(topo="mmictc", P=5000, L=64, den=True,
                                  desc="LS BPE-5k mmictc denominator, V=10001, S=10002, A=30002"),
    "cfg4_mmictc_bpe5k_num": dict(topo="mmictc", P=5000, L=64, den=False,
                                  desc="LS BPE-5k mmictc numerator, V=10001, L=64"),
    "cfg4_3state-skip_bpe5k_num": dict(topo="3state-skip", P=5000, L=64, den=False,
                                       desc="LS BPE-5k 3state-skip numerator, V=15001, L=64"),
    "cfg4_3state-skip_bpe5k_den": dict(topo="3state-skip", P=5000, L=64, den=True,
                                       desc="LS BPE-5k 3state-skip denominator, V=15001, S=10002, A=25002"),
}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _np(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


class Workload:
    """Synthetic batch (seed 777 + rank) resident on one GPU."""

    def __init__(self, name, B, T, dev, seed, ragged=False, shard=None, beam=0.0, double=False):
        """shard = (rank, world): strong scaling -- the batch of B utterances is generated as a whole and this rank keeps
        the utterances waterfall_amd.dist.shard_batch deals it (round robin over the length-sorted order).
        beam > 0: output_beam pruning (k2.intersect_dense's output_beam, graph.py:61-62,81): WF_PRUNED.
        double: use_double_scores=True (WF_DOUBLE_ACC | WF_TOT_F64: fp64 totals)."""
        w = WORKLOADS[name]
        self.name, self.T, self.dev = name, T, dev
        self.beam = float(beam)
        self.flags = (_lib.WF_PRUNED if beam > 0 else 0) | ((_lib.WF_DOUBLE_ACC | _lib.WF_TOT_F64) if double else 0)
        P, L = w["P"], w["L"]
        rng = np.random.default_rng(seed)
        phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
        lazy = wf.Topology(w["topo"], phones)
        self.V = V = lazy.num_nn_output
        self.seqs = [[int(x) + 2 for x in rng.integers(0, P, size=L)] for _ in range(B)]
        lens = np.full(B, T)
        if ragged:
            lens = np.sort(rng.integers(int(0.7 * T), T + 1, size=B))[::-1].copy()
            lens[0] = T
        g = torch.Generator(device=dev).manual_seed(seed)
        keep = None
        if shard is not None:
            keep = np.sort(wf.dist.shard_batch(lens, shard[1], shard[0]))
            self.seqs = [self.seqs[i] for i in keep]
            lens = lens[keep]
            full = torch.randn(B, T, V, generator=g, device=dev)
            self.lp = torch.log_softmax(full[torch.from_numpy(keep).to(dev)] * 3.0, -1)
            del full
            B = len(keep)
        else:
            self.lp = torch.log_softmax(torch.randn(B, T, V, generator=g, device=dev) * 3.0, -1)
        self.B = B
        if w["den"]:
            self.fsa = wf.create_fsa_vec([lazy.fsa] * B)
        else:
            self.fsa = wf.compile_ctc_k2_graphs(lazy, self.seqs)
        self.segs = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), lens], 1).astype(np.int32)
        self.grad = torch.empty_like(self.lp)
        self.tot = torch.empty(B, dtype=torch.float64 if double else torch.float32, device=dev)
        self.gscale = torch.full((B,), -1.0, dtype=torch.float32, device=dev)
        self.gh = self.fsa.blob_host()
        self.gd = self.fsa.blob_dev(dev)
        n = C.c_size_t(0)
        lib = _lib.lib()
        _lib.check(lib.wf_fb_workspace_bytes(_np(self.gh), _np(self.fsa.utt2fsa), _np(self.segs), B, B, V, self.flags, 1, C.byref(n)))
        self.ws_bytes = n.value
        self.ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
        self.frames = int(lens.sum())
        # per-utterance graph sizes for the byte / arc-update accounting
        g2f = self.fsa.utt2fsa if self.fsa.utt2fsa is not None else np.arange(B)
        A = np.diff(self.fsa.fsa_arc_offsets)[g2f]
        S = self.fsa.fsa_num_states[g2f]
        self.arc_updates = int((A * lens).sum())          # per sweep direction
        self.S_frames = int((S * lens).sum())
        from waterfall_amd.csrc_layout import HDR_WORDS, INFO_WORDS, INFO
        K = np.array([self.gh[HDR_WORDS + g * INFO_WORDS + INFO["K"]] for g in g2f])
        direct = np.array([self.gh[HDR_WORDS + g * INFO_WORDS + INFO["DIRECT"]] for g in g2f])
        self.K_frames = int((K * lens).sum())
        self.compact_K_frames = int((K * lens * (1 - direct)).sum())
        self.S_mean, self.A_mean, self.K_mean = float(S.mean()), float(A.mean()), float(K.mean())

    def step(self, stream):
        lib = _lib.lib()
        lp = self.lp
        _lib.check(lib.wf_fb_logsum(_np(self.gh), _ptr(self.gd), _np(self.fsa.utt2fsa), _ptr(lp), lp.stride(0),
                                    lp.stride(1), self.B, self.T, self.V, _np(self.segs), self.B, self.beam, self.flags,
                                    _ptr(self.gscale), _ptr(self.tot), _ptr(self.grad), _ptr(self.ws),
                                    self.ws_bytes, C.c_void_p(stream)))

    def fallbacks(self, stream):
        """Utterances of the last step that the log-semiring fallback redid (0: the timed path is the probability-domain
        kernels' own work, not a repair of it)."""
        n = C.c_int32(-1)
        _lib.check(_lib.lib().wf_debug_fallback_count(_np(self.gh), _np(self.fsa.utt2fsa), _np(self.segs), self.B, self.B, self.V,
                                                      self.flags, 1, _ptr(self.ws), self.ws_bytes, C.c_void_p(stream), C.byref(n)))
        return int(n.value)

    # ---- algorithmic bytes (DESIGN.md "Kernels"): what each kernel must move, per batch --------
    def kernel_bytes(self, lin=False):
        V = self.V
        if lin:   # linear-domain sweep: emissions are staged as fp64 probabilities
            gather = (4 + 8) * self.K_frames + 4 * self.compact_K_frames   # read lp, write P (fp64) [+ raw E for compact graphs]
            sweep = 2 * 8 * self.K_frames + 2 * 8 * self.S_frames       # P read by both sweeps + alpha/beta rows out
        else:
            gather = 8 * self.compact_K_frames                      # read 4 B + write 4 B per gathered element
            sweep = 2 * 4 * self.K_frames + 2 * 8 * self.S_frames   # emissions read by both sweeps + alpha/beta rows out
        check = 2 * 8 * self.S_frames                           # alpha and beta rows read once more
        zero = 4 * V * self.B * self.T                          # zero-fill of the gradient box (side stream)
        direct_K_frames = self.K_frames - self.compact_K_frames
        post = 2 * 8 * self.S_frames + 4 * self.K_frames + 4 * self.compact_K_frames + 4 * direct_K_frames  # alpha/beta in + emissions in + values out
        # SURVEY.md 8(d) whole-path figure: 4V (lp row) + 4V (grad row) + 8S (alpha spill + reload) per frame
        path = (8 * V) * self.frames + 8 * self.S_frames
        return {"gather": gather, "sweep": sweep, "post": post, "zero": zero, "check": check, "path_survey_8d": path}


def time_steps(wl, steps, warmup, dist, stream, fused_buf, min_region_s=0.0, max_rounds=200):
    """Times EXACTLY `steps` steps between (barrier + synchronize) brackets, and repeats that timed region until the
    regions add up to min_region_s (a 2.7 ms region is at the mercy of one scheduling hiccup).  Returns the list of the
    regions' durations in seconds (every rank runs the same number of regions: the count is agreed on after the first)."""
    bufs = [fused_buf, fused_buf.clone()]
    works = [None, None]
    count = [0]
    with_grad = fused_buf.numel() > 3

    def one():
        wl.step(stream)
        if dist is not None:
            # single fused all-reduce per step: [loss_sum, n_frames, n_infeasible (, per-label gradient sums)], started
            # asynchronously on one of two buffers so that it overlaps the next step's sweeps (logging data: nothing on
            # the data path waits for it)
            j = count[0] & 1
            count[0] += 1
            if works[j] is not None:
                works[j].wait()
            if os.environ.get("WF_BENCH_SYNC_ALLREDUCE"):   # (diagnostic: the collective on the launch stream)
                wf.dist.fused_loss_stats(wl.tot, wl.frames, out=bufs[j], grad=wl.grad if with_grad else None)
                return
            _, works[j] = wf.dist.fused_loss_stats(wl.tot, wl.frames, out=bufs[j], async_op=True,
                                                   grad=wl.grad if with_grad else None)

    def drain():
        for j in (0, 1):
            if works[j] is not None:
                works[j].wait()
                works[j] = None

    def region():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        drain()          # every step's all-reduce is inside the timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dbg = bool(os.environ.get("WF_BENCH_DEBUG"))
    if dbg:
        import faulthandler
        faulthandler.dump_traceback_later(40, exit=False, file=sys.stderr)
    for _ in range(warmup):
        one()
    drain()
    if dbg:
        print("warmup done", file=sys.stderr, flush=True)
    times = [region()]
    if dbg:
        print("first region", times, file=sys.stderr, flush=True)
    rounds = int(min(max_rounds, max(1, np.ceil(min_region_s / max(times[0], 1e-6)))))
    if dist is not None:   # the same number of regions on every rank
        r = torch.tensor([rounds], dtype=torch.int64, device=wl.lp.device)
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        rounds = int(r.item())
    for _ in range(rounds - 1):
        times.append(region())
    return times


def kernel_times(wl, stream, reps=20):
    """Average per-kernel duration over `reps` timed steps: HIP events recorded by the library on
    the launch stream around each kernel.  The gradient zero-fill normally rides inside the banded sweep's launch
    (extra workgroups): its own duration is measured in a second set of steps with the fill sent to the side stream
    (WF_FILL_SIDE_STREAM=1, the library's measurement hook)."""
    lib = _lib.lib()

    def run():
        lib.wf_timing_enable(1)
        acc = np.zeros(6)
        ms = (C.c_float * 8)()
        for _ in range(reps):
            wl.step(stream)
            _lib.check(lib.wf_timing_read(ms))
            acc += np.array(list(ms)[:6])
        lib.wf_timing_enable(0)
        return dict(zip(("gather", "sweep", "post", "zero", "check", "sweep_device_span"), (acc / reps).tolist()))

    kt = run()
    span = kt.pop("sweep_device_span")
    kt["fill_in_sweep"] = bool(kt["zero"] < 0 and kt["check"] >= 0 and wl.fsa.all_compact())
    kt["sweep_device_span"] = span
    if kt["zero"] < 0 and kt["check"] >= 0:
        old = os.environ.get("WF_FILL_SIDE_STREAM")
        os.environ["WF_FILL_SIDE_STREAM"] = "1"
        _lib.reload_env()
        try:
            kt["zero"] = run()["zero"]
        finally:
            if old is None:
                os.environ.pop("WF_FILL_SIDE_STREAM", None)
            else:
                os.environ["WF_FILL_SIDE_STREAM"] = old
            _lib.reload_env()
    return kt


def cpu_baseline(wl, budget_s=12.0):
    """oracle/dwfst_oracle.c (scalar C port, fp32 accumulators as graph.py:83's default) on this
    box's host cores, OpenMP over utterances, on the same batch; bounded to ~budget_s seconds."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(so)
    from tests import util
    graphs = wl.fsa.as_oracle_graphs()
    lp = wl.lp.cpu().numpy()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    nthreads = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)), wl.B))
    t0 = time.perf_counter()
    util.c_oracle_batch(lib, graphs, lp, wl.segs, gscale=[-1.0] * wl.B, nthreads=nthreads)
    first = time.perf_counter() - t0
    reps = max(1, min(50, int(budget_s / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        util.c_oracle_batch(lib, graphs, lp, wl.segs, gscale=[-1.0] * wl.B, nthreads=nthreads)
    dt = (time.perf_counter() - t0) / reps
    out = {"value": wl.frames / dt, "unit": "utterance-frames/s", "cores": nthreads, "kind": "port",
           "sample": f"{reps} passes over the same batch ({wl.B}x{wl.T} frames, {wl.name}), "
                     f"oracle/dwfst_oracle.c fp32, OpenMP over utterances; k2 itself is not installable offline",
           "ms_per_batch": dt * 1e3}
    if WORKLOADS[wl.name]["topo"] == "ctc" and not WORKLOADS[wl.name]["den"]:
        # the reference's own equivalent 'builtin_ctc' loss (rnnp.py:107-108) on the same inputs
        x = torch.from_numpy(lp).requires_grad_(True)
        tg = torch.tensor([[p - 1 for p in s] for s in wl.seqs])
        il = torch.from_numpy(wl.segs[:, 2].astype(np.int64))
        tl = torch.full((wl.B,), tg.shape[1], dtype=torch.long)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 4.0:
            x.grad = None
            torch.nn.functional.ctc_loss(x.permute(1, 0, 2), tg, il, tl, reduction="sum").backward()
            n += 1
        out["torch_ctc_loss_cpu_frames_per_s"] = wl.frames * n / (time.perf_counter() - t0)
        out["torch_threads"] = torch.get_num_threads()
    return out


def _wall_and_host(fn, steps, warmup=3):
    """(wall ms per step, host-issue ms per step): the loop is timed until it has issued everything (host) and until
    the GPU has finished it (wall)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3


def dropin_times(wl, steps):
    """What a caller of the reference gets per step: k2.DenseFsaVec(...) -> graphloss(reduction='sum') -> backward()
    (models.py:256-277), on the resident graph and with fresh numerator graphs every step (linear_fsa ->
    remove_epsilon_and_add_self_loops -> arc_sort -> compose, models.py:263-270, + upload), next to the fused C call."""
    w = WORKLOADS[wl.name]
    segs_t = torch.from_numpy(wl.segs)
    x = wl.lp.detach().requires_grad_(True)

    def dropin(fsa=None):
        x.grad = None
        dense = wf.DenseFsaVec(x, segs_t)
        loss = wf.graphloss(fsa if fsa is not None else wl.fsa, dense, output_beam=10, reduction="sum")
        loss.backward()

    out = {}
    # the drop-in's default honours output_beam (the reference's semantics); "exact" is the mode of the headline
    wf.set_output_beam_mode("exact")
    try:
        wall, host = _wall_and_host(dropin, steps)
    finally:
        wf.set_output_beam_mode(wf.graph.DEFAULT_BEAM_MODE)
    out["dropin"] = {"ms_per_step": wall, "host_issue_ms_per_step": host,
                     "what": "DenseFsaVec + graphloss(reduction='sum') + backward() under autograd, resident graph, "
                             "set_output_beam_mode('exact') (the headline's mode: the sum over all lattice arcs)"}
    wall, host = _wall_and_host(dropin, steps)
    out["dropin_pruned_beam10"] = {"ms_per_step": wall, "host_issue_ms_per_step": host,
                                   "what": "the same lines with the drop-in's default mode: graphloss(output_beam=10) prunes the "
                                           "lattice as k2.intersect_dense(output_beam=10) states (graph.py:61-62,81)"}
    if w["den"]:
        return out
    wf.set_output_beam_mode("exact")   # (the fresh-graph lines below compare with the headline: exact)
    # fresh transcripts every step: `steps + 3` different batches drawn beforehand (flat int32 labels + lengths, as a data
    # loader holds them)
    rng = np.random.default_rng(4242)
    P, L, B = w["P"], w["L"], wl.B
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    topo = wf.Topology(w["topo"], phones)
    batches = [(rng.integers(2, P + 2, size=B * L).astype(np.int32), np.full(B, L)) for _ in range(steps + 3)]
    it = [0]

    def compile_only():
        b = batches[it[0] % len(batches)]
        it[0] += 1
        return wf.compile_ctc_k2_graphs(topo, b)

    t0 = time.perf_counter()
    for _ in range(steps):
        compile_only()
    out["compile_host_ms"] = (time.perf_counter() - t0) / steps * 1e3

    def fresh_dropin():
        dropin(compile_only())

    lib = _lib.lib()
    stream = torch.cuda.current_stream(wl.dev).cuda_stream

    def fresh_fused():
        fsa = compile_only()
        gh, gd = fsa.blob_host(), fsa.blob_dev(wl.dev)
        lp = wl.lp
        n = C.c_size_t(0)     # the workspace depends on the batch's graphs (distinct labels per utterance)
        _lib.check(lib.wf_fb_workspace_bytes(_np(gh), None, _np(wl.segs), wl.B, wl.B, wl.V, 0, 1, C.byref(n)))
        ws = torch.empty(n.value, dtype=torch.uint8, device=wl.dev)
        _lib.check(lib.wf_fb_logsum(_np(gh), _ptr(gd), None, _ptr(lp), lp.stride(0), lp.stride(1), wl.B, wl.T, wl.V,
                                    _np(wl.segs), wl.B, 0.0, 0, _ptr(wl.gscale), _ptr(wl.tot), _ptr(wl.grad), _ptr(ws),
                                    n.value, C.c_void_p(stream)))

    wall, host = _wall_and_host(fresh_dropin, steps)
    out["fresh_graphs_dropin_host_compiler"] = {"ms_per_step": wall, "host_issue_ms_per_step": host,
                                                "what": "new transcripts every step: compile_ctc_k2_graphs on the host (compose + prepare, "
                                                        f"{wf.compile.host_threads()} host threads) + asynchronous upload + the drop-in loss + backward"}
    wall, host = _wall_and_host(fresh_fused, steps)
    out["fresh_graphs_host_compiler"] = {"ms_per_step": wall, "host_issue_ms_per_step": host,
                                         "what": "new transcripts every step, graphs compiled on the HOST (round 2's path): compile + "
                                                 "asynchronous upload + the fused wf_fb_logsum call"}

    # ---- the same with the DEVICE compiler (numerator_compile_kernel): the host uploads the transcripts and does no graph
    # work.  One batch ahead, as a data loader prefetches: step i's loss is issued after step i + 1's compile kernels went to
    # their own stream, so they run beside the loss kernels; the header the host plans with comes back asynchronously.
    def compile_dev():
        b = batches[it[0] % len(batches)]
        it[0] += 1
        return wf.compile_ctc_k2_graphs(topo, b, device=wl.dev)

    pending = [compile_dev()]

    def fused_on(fsa):
        gh, gd = fsa.blob_host(), fsa.blob_dev(wl.dev)
        lp = wl.lp
        n = C.c_size_t(0)
        _lib.check(lib.wf_fb_workspace_bytes(_np(gh), None, _np(wl.segs), wl.B, wl.B, wl.V, 0, 1, C.byref(n)))
        ws = torch.empty(n.value, dtype=torch.uint8, device=wl.dev)
        _lib.check(lib.wf_fb_logsum(_np(gh), _ptr(gd), None, _ptr(lp), lp.stride(0), lp.stride(1), wl.B, wl.T, wl.V,
                                    _np(wl.segs), wl.B, 0.0, 0, _ptr(wl.gscale), _ptr(wl.tot), _ptr(wl.grad), _ptr(ws),
                                    n.value, C.c_void_p(stream)))

    def fresh_fused_dev():
        nxt = compile_dev()            # step i + 1's graphs: enqueued on the compile stream
        fused_on(pending.pop())        # step i's loss on the graphs compiled during step i - 1
        pending.append(nxt)

    def fresh_dropin_dev():
        nxt = compile_dev()
        dropin(pending.pop())
        pending.append(nxt)

    def fresh_fused_dev_serial():      # no prefetch: compile, wait for its header, loss
        fused_on(compile_dev())

    wall, host = _wall_and_host(fresh_fused_dev, steps)
    out["fresh_graphs"] = {"ms_per_step": wall, "host_issue_ms_per_step": host, "compiled_on": pending[0].compiled_on,
                           "what": "new transcripts every step, graphs compiled ON THE DEVICE one batch ahead "
                                   "(compile_ctc_k2_graphs(device=...): ONE launch of numerator_compile_kernel on its own stream "
                                   "beside the previous step's loss; it reads the transcripts from pinned host memory and writes "
                                   "the words the host plans with back there, no copy is enqueued) + the fused wf_fb_logsum call"}
    wall, host = _wall_and_host(fresh_fused_dev_serial, steps)
    out["fresh_graphs_no_prefetch"] = {"ms_per_step": wall, "host_issue_ms_per_step": host,
                                       "what": "the same without the one-batch prefetch: compile on the device, wait for its header, loss"}
    wall, host = _wall_and_host(fresh_dropin_dev, steps)
    out["fresh_graphs_dropin"] = {"ms_per_step": wall, "host_issue_ms_per_step": host,
                                  "what": "new transcripts every step, compiled on the device one batch ahead + the drop-in loss + backward"}
    # the compile kernels alone
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cs = wf.compile._compile_stream(wl.dev)
    torch.cuda.synchronize()
    ev[0].record(cs)
    for _ in range(10):
        compile_dev()
    ev[1].record(cs)
    torch.cuda.synchronize()
    out["compile_device_ms"] = ev[0].elapsed_time(ev[1]) / 10
    wf.set_output_beam_mode(wf.graph.DEFAULT_BEAM_MODE)
    return out


def viterbi_bench(dev, S, B=8, T=500, V=5001, reps=3):
    """cfg5: tropical best path + back-trace on a synthetic TLG-like graph (A = 3 S) shared by
    the batch; exact search.  Algorithmic bytes per utterance-frame: 4 V + 16 A + 4 S."""
    from waterfall_amd.synth import best_path_checksum, cfg5_workload
    g, fsa, lp, dense = cfg5_workload(dev, S, B, T, V)
    arcs = fsa.arcs
    wf.best_path(fsa, dense)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        wf.best_path(fsa, dense)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    A = arcs.shape[0]
    out = wf.best_path(fsa, dense)
    csum = best_path_checksum(out["scores"].cpu().numpy(), out["arcs"].cpu().numpy())
    return {"ms_per_call": ms, "score_sum": csum[0], "arcs_checksum": csum[1],
            "checked_by": "tests/test_gpu_parity.py::test_cfg5_benched_workload_bit_exact_against_c_oracle (same workload, C oracle)", "us_per_frame": ms * 1e3 / T, "frames_per_s": B * T / (ms * 1e-3),
            # SURVEY 8(d) charges 4V + 16A + 4S per utterance-frame; big_step_shared_kernel reads the arcs ONCE per frame for the 8
            # utterances of a lane group, so the bytes a frame's launch must move are 16A per GROUP plus, per utterance, the
            # back-pointers (4S), alpha in and out (2 x 4S) and the K gathered emissions
            "survey_8d_bytes_GBps": B * T * (4 * V + 16 * A + 4 * S) / (ms * 1e-3) / 1e9,
            "survey_8d_bytes_GBps_is": "SURVEY 8(d) bytes per utterance-frame / time: NOT a bandwidth (arcs are shared by 8 utterances)",
            "launch_model_bytes_per_frame": ((B + 7) // 8) * 16 * A + B * (4 * S + 8 * S),
            "launch_model_GBps": T * (((B + 7) // 8) * 16 * A + B * (4 * S + 8 * S)) / (ms * 1e-3) / 1e9,
            "desc": f"cfg5 Viterbi + back-trace, synthetic TLG-like graph S={S} A={A}, B={B}, T={T}, V={V}, exact search, "
                    "includes the host-side label extraction of wf.best_path"}


def search_bench(dev, S, B=8, T=500, V=5001, beam=20.0, max_active=10000, reps=3, peak=10.0):
    """cfg5 with the beam-pruned search (wf_search; the reference's k2.get_lattice(search_beam=20) at
    utils/lattice_analysis.py:125) beside the exact sweep, on the PEAKED posterior of synth.cfg5_workload (a beam has
    nothing to hold on to in noise); `same_path`: the pruned best path equals the exact one for every utterance."""
    from waterfall_amd.synth import cfg5_workload
    g, fsa, lp, dense = cfg5_workload(dev, S, B, T, V, peaked=True, peak=peak)

    def timed(f):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r
    ms_exact, exact = timed(lambda: wf.best_path(fsa, dense))
    ms, out = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, max_active, log_tot=False))
    ms_log, out2 = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, max_active, log_tot=True, return_active=True))
    # get_lattice's output_beam (lattice_analysis.sh's lattice_beam = 8): the backward launch over the finished lattice
    ms_lat, out3 = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, max_active, log_tot=True, output_beam=8.0))
    # ... and that lattice restricted to the best path's word sequence (compose(lattice, W), lattice_analysis.py:137-139):
    # word labels on ~4 % of the arcs (~16 words per utterance), as tools/time_lattice_words.py
    rng = np.random.default_rng(1)
    lab = fsa.arcs[:, 2]
    fsa.aux_labels = np.where(lab >= 0, np.where(rng.random(lab.size) < 0.04, rng.integers(1, 1000, lab.size), 0), -1).astype(np.int32)
    words = wf.pruned_search(fsa, dense, beam, 30, max_active, log_tot=False)["aux_labels"]
    ms_w, outw = timed(lambda: wf.lattice_words(fsa, dense, words, beam, 8.0, 30, max_active))
    return {"ms_per_call": ms, "ms_per_call_with_log_total": ms_log, "ms_per_call_with_output_beam8": ms_lat,
            "ms_per_call_lattice_restricted_to_words": ms_w,
            "mean_P_words_in_lattice_over_P_lattice": float(torch.exp(outw["tot_words"] - outw["tot_log"]).mean().item()),
            "lattice_arcs_per_frame_output_beam8": float(out3["lattice_arcs"].double().mean().item()) / T,
            "log_total_minus_output_beam8_total": float((out2["tot_log"].double() - out3["tot_log"].double()).mean().item()),
            "exact_sweep_ms_per_call": ms_exact,
            "speedup_vs_exact": ms_exact / ms, "same_path": bool(torch.equal(out["arcs"], exact["arcs"])),
            "same_score": bool(torch.equal(out["scores"], exact["scores"])),
            "mean_active_states": float(out2["n_active"].float().mean().item()),
            "max_active_states_seen": int(out2["n_active"].max().item()),
            "planted_label_logit_boost": peak,
            "checked_by": "tests/test_search.py (infinite beam == exact sweep bit for bit; finite beams == oracle.beam_search; "
                          "output_beam == oracle.lattice_search; every test on search_kernel and on search_mw_kernel)",
            "desc": f"cfg5 beam-pruned search, S={S} A={fsa.arcs.shape[0]}, B={B}, T={T}, V={V}, search_beam={beam}, "
                    f"max_active_states={max_active}, synthetic posterior with the planted path's logit boosted by {peak} "
                    "(10: a confident model, ~30 active states; 2.6 / 2.3 / 2.0: ~740 / 2 000 / 5 000 active); wall time of "
                    "wf.pruned_search incl. host"}


def ctc_matrices(targets, V, dev):
    """The (trans [B,S,S], transform [B,S,V], trans_lengths [B]) a CTC target sequence turns into for the reference's
    manual_ctc losses (blank = 0; S = 2 L + 1: blank, label, blank, ..., blank), built on the device; finfo.min marks
    "not allowed" as deprecated/datapipe_manual_ctc.py:245-281 pads."""
    B, L = targets.shape
    S = 2 * L + 1
    fmin = torch.finfo(torch.float32).min
    trans = torch.full((B, S, S), fmin, dtype=torch.float32, device=dev)
    i = torch.arange(L, device=dev)
    trans[:, 2 * i, 2 * i] = 0.0
    trans[:, 2 * i[1:], 2 * i[1:] - 1] = 0.0
    trans[:, 2 * i + 1, 2 * i + 1] = 0.0
    trans[:, 2 * i + 1, 2 * i] = 0.0
    b, k = torch.nonzero(targets[:, 1:] != targets[:, :-1], as_tuple=True)     # label k + 1 differs from label k: skip the blank
    trans[b, 2 * (k + 1) + 1, 2 * (k + 1) - 1] = 0.0
    trans[:, S - 1, S - 1] = 0.0
    trans[:, S - 1, S - 2] = 0.0
    transform = torch.full((B, S, V), fmin, dtype=torch.float32, device=dev)
    transform[:, 0::2, 0] = 0.0
    transform[torch.arange(B, device=dev)[:, None], (2 * i + 1)[None, :], targets] = 0.0
    return trans, transform, torch.full((B,), S, dtype=torch.int64, device=dev)


def manual_ctc_bench(dev, B=32, T=500, V=59, L=80, reps=20):
    """The reference's manual_ctc loss (waterfall/manual_ctc/ctc.py:94-105: ctc_loss(log_probs, input_lengths, trans,
    trans_lengths, transform)) through its drop-in at cfg2's sizes: matrices -> graph (built on the device, one copy to
    the host) -> forward-backward kernels -> backward().  The reference's own implementation materialises a
    [B, T, S, V] broadcast per call."""
    from waterfall_amd import manual_ctc
    gen = torch.Generator(device=dev).manual_seed(777)
    x = torch.log_softmax(torch.randn(B, T, V, generator=gen, device=dev) * 3.0, -1).requires_grad_(True)
    targets = torch.randint(1, V, (B, L), generator=gen, device=dev)
    trans, transform, lens = ctc_matrices(targets, V, dev)
    il = torch.full((B,), T, dtype=torch.int64)

    def step():
        x.grad = None
        loss = manual_ctc.ctc_loss(x, il, trans, lens, transform, reduction="sum")
        loss.backward()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        l = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        manual_ctc.matrices_to_fsa(trans, lens, transform)
    ms_graph = (time.perf_counter() - t0) / reps * 1e3
    # the same loss from torch's own CTC on the same targets (blank 0): the check of this line
    ref = torch.nn.functional.ctc_loss(x.detach().transpose(0, 1), targets, il, torch.full((B,), L, dtype=torch.int64),
                                       blank=0, reduction="sum")
    return {"ms_per_step": ms, "frames_per_s": B * T / (ms * 1e-3), "graph_build_ms": ms_graph,
            "loss": float(l.item()), "torch_ctc_loss_same_inputs": float(ref.item()),
            "desc": f"manual_ctc.ctc_loss + backward, B={B}, T={T}, V={V}, L={L} (S={2 * L + 1}); graph_build_ms = matrices_to_fsa "
                    "alone (masked nonzero on the device + one copy + table preparation on the host)",
            "checked_by": "tests/test_gpu_parity.py::test_manual_ctc_dropin_against_reference_goldens"}


def num_den_bench(dev, B=32, T=500, P=5000, L=64, reps=10):
    """The numerator - denominator loss of the reference's k2 branch (rnnp.py:64-104) at cfg4 mmictc sizes under
    autograd: graphloss(num) - graphloss(den) (what the reference's lines do, unchanged) beside graphloss_num_den
    (one node, one gradient box: WF_ACCUMULATE).  Wall time per loss + backward step, resident graphs."""
    phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
    topo = wf.Topology("mmictc", phones)
    V = topo.num_nn_output
    rng = np.random.default_rng(777)
    num = wf.compile_ctc_k2_graphs(topo, [[int(x) + 2 for x in rng.integers(0, P, size=L)] for _ in range(B)])
    den = wf.create_fsa_vec([topo.fsa] * B)
    gen = torch.Generator(device=dev).manual_seed(777)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=gen, device=dev) * 3.0, -1).requires_grad_(True)
    seg = torch.tensor([[i, 0, T] for i in range(B)], dtype=torch.int32)
    out = {}
    for name in ("two_call", "num_den", "two_call_auto_beam10", "two_call_pruned_beam10"):
        # (the first two in the "exact" beam mode, as the headline; the third with the drop-in's default: output_beam = 10 honoured)
        # ("auto", the drop-in's default: the numerators' lattice is pruned -- fast path --, the hub-form denominators' is not)
        wf.set_output_beam_mode("pruned" if "pruned" in name else "auto" if "auto" in name else "exact")

        def step():
            lp.grad = None
            d = wf.DenseFsaVec(lp, seg)
            if name == "num_den":
                loss = wf.graphloss_num_den(num, den, d, reduction="sum")
            else:
                loss = wf.graphloss(num, d, reduction="sum") - wf.graphloss(den, d, reduction="sum")
            loss.backward()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            l = step()
        torch.cuda.synchronize()
        out[name + "_ms_per_step"] = (time.perf_counter() - t0) / reps * 1e3
        out[name + "_loss"] = float(l.detach())
        out[name + "_grad_abs_sum"] = float(lp.grad.abs().sum())
    wf.set_output_beam_mode(wf.graph.DEFAULT_BEAM_MODE)
    out["desc"] = f"mmictc numerators (L={L}) - denominators (S={2 * (P + 1) + 2}), V={V}, B={B}, T={T}, reduction=sum, loss + backward under autograd"
    out["checked_by"] = "tests/test_gpu_parity.py::test_graphloss_num_den_equals_the_two_call_form"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3_ctc_bpe5k_num", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = --batch utterances per GPU; strong = --batch utterances in all, dealt over the ranks "
                         "(BASELINE.json configs[2] is B=32 sharded over 8 GPUs)")
    ap.add_argument("--grad-sums", action="store_true",
                    help="N > 1: append the V per-label gradient sums (wf_label_grad_sums) to the fused all-reduce buffer. Off by "
                         "default so that the N > 1 step is the N = 1 step plus the collective (the driver computes scaling "
                         "efficiency from the two); the kernel's cost is reported at N = 1 as label_grad_sums_ms")
    ap.add_argument("--min-region-ms", type=float, default=100.0,
                    help="repeat the timed K-step region until the regions add up to this (median and min are reported)")
    ap.add_argument("--output-beam", type=float, default=0.0,
                    help="> 0: run the step with output_beam pruning (k2.intersect_dense's output_beam, graph.py:61-62,81: WF_PRUNED); "
                         "0 (default): the exact sum over all lattice arcs")
    ap.add_argument("--others", action="store_true", help="(default at N=1) also time the other BASELINE workloads")
    ap.add_argument("--no-others", action="store_true", help="headline only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("WF_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on a box with fewer GPUs than ranks
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
            local = local % max(torch.cuda.device_count(), 1)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} but the job has {dist.get_world_size()} ranks")
        if dist.get_backend() != backend:
            raise SystemExit(f"process group backend is {dist.get_backend()}, expected {backend}")
    elif args.gpus != 1:
        raise SystemExit(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    strong = world > 1 and args.scaling == "strong"
    if strong:
        if args.batch < world:
            raise SystemExit(f"--scaling strong: {args.batch} utterances cannot be dealt over {world} ranks")
        wl = Workload(args.workload, args.batch, args.frames, dev, seed=777, ragged=args.ragged, shard=(rank, world), beam=args.output_beam)
    else:
        wl = Workload(args.workload, args.batch, args.frames, dev, seed=777 + rank, ragged=args.ragged, beam=args.output_beam)
    grad_sums = world > 1 and args.grad_sums
    fused = torch.zeros(3 + (wl.V if grad_sums else 0), dtype=torch.float32, device=dev)
    times = time_steps(wl, args.steps, args.warmup, dist, stream, fused, min_region_s=args.min_region_ms * 1e-3)
    if dist is not None:
        t = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # every region: the slowest rank's time
        times = t.tolist()
    dt = float(np.median(times))
    dt_min = float(np.min(times))
    if dist is not None:
        tf = torch.tensor([float(wl.frames), float(wl.arc_updates)], dtype=torch.float64, device=dev)
        dist.all_reduce(tf)
        total_frames, total_arcs = tf[0].item(), tf[1].item()
    else:
        total_frames, total_arcs = float(wl.frames), float(wl.arc_updates)
    ms_step = dt / args.steps * 1e3
    value = total_frames / (dt / args.steps)

    out = {
        "metric": "DWFST loss+grad utterance-frames/sec",
        "value": value, "unit": "utterance-frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "timed_regions": len(times), "timed_region_ms_total": sum(times) * 1e3,
        "ms_per_step_min": dt_min / args.steps * 1e3, "value_at_min": total_frames / (dt_min / args.steps),
        "statistic": f"median over {len(times)} timed regions of exactly {args.steps} steps each (value, ms_per_step); "
                     "the fastest region alongside (value_at_min, ms_per_step_min)",
        "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]['desc']}, "
                               + (f"B={args.batch} in all, dealt over {world} GPUs, " if strong else f"B={args.batch} per GPU, ")
                               + f"T={args.frames}{' ragged' if args.ragged else ''}, loss+grad, reduction=sum, "
                               + (f"output_beam={args.output_beam:g} (pruned lattice)" if args.output_beam > 0 else "exact (no output_beam pruning)"),
                   "scaling": "strong" if strong else "weak",
                   "batch_per_gpu": wl.B, "batch_total": args.batch if (strong or world == 1) else args.batch * world,
                   "frames": args.frames, "V": wl.V,
                   "states_per_graph": wl.S_mean, "arcs_per_graph": wl.A_mean,
                   "parallelism": (f"batch-sharded x{world}, one fused all-reduce per step of [loss_sum, n_frames, n_infeasible"
                                   + (f", {wl.V} per-label gradient sums]" if grad_sums else "]")) if world > 1 else "single GPU"},
        "arc_updates_per_sec": 2 * total_arcs / (dt / args.steps),
        "fallback_utterances": wl.fallbacks(stream),
    }
    if rank == 0:
        kt = kernel_times(wl, stream)
        fill_in_sweep = kt.pop("fill_in_sweep")
        sweep_span_ms = kt.pop("sweep_device_span")
        kb = wl.kernel_bytes(lin=kt["check"] >= 0)
        dom = max(kt, key=lambda k: kt[k])
        if fill_in_sweep:
            # the gradient zero-fill is done by extra workgroups of the sweep's own launch: that launch's algorithmic
            # bytes are the sweep's plus the fill's (kernels.zero is the fill measured alone, on the side stream)
            kb = dict(kb, sweep=kb["sweep"] + kb["zero"], sweep_alone=kb["sweep"])
        ach = kb[dom] / (kt[dom] * 1e-3) / 1e9
        default_wl = (args.workload == "cfg3_ctc_bpe5k_num" and args.batch == 32 and args.frames == 500
                      and not args.ragged and args.output_beam == 0)
        if default_wl:
            kname = KERNEL_NAMES.get(dom, dom)
        elif wl.S_mean > 4096 and kt["check"] >= 0:      # large denominators (cfg4): DESIGN.md section 4
            from waterfall_amd import csrc_layout as _cl
            hubform = bool(wl.gh[_cl.HDR_WORDS + _cl.INFO["HUB_G"]] > 0)
            kname = ({"sweep": "sweep_hub_kernel", "post": "post_hub_kernel"} if hubform else
                     {"sweep": "sweep_linbig_kernel", "post": "post_big_kernel"}).get(dom, KERNEL_NAMES.get(dom, dom))
        else:
            kname = dom + " (kernel picked by graph size: DESIGN.md section 4)"
        rp_us = rocprof_avg_us(dom, default_wl)
        out["roofline"] = {"bound": "hbm", "kernel": kname,
                           "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, default_wl),
                           "avg_launch_ms": kt[dom], "algorithmic_bytes_per_launch": kb[dom],
                           "convention": "achieved / frac: algorithmic bytes / the HIP-event interval measured in this run "
                                         "(avg_launch_ms); *_rocprof: the same bytes / the kernel's average duration in the "
                                         "committed rocprofv3 summary, null when that summary is of another kernel source",
                           "rocprof_avg_launch_us": rp_us,
                           # the launch's own duration measured INSIDE the kernel (s_memrealtime, latest workgroup end - earliest
                           # start): free of the HIP-event latency in avg_launch_ms and of the profiler's inflation of rocprof_*
                           "in_kernel_launch_us": sweep_span_ms * 1e3 if (dom == "sweep" and sweep_span_ms > 0) else None,
                           "achieved_in_kernel": kb[dom] / (sweep_span_ms * 1e-3) / 1e9 if (dom == "sweep" and sweep_span_ms > 0) else None,
                           "frac_in_kernel": kb[dom] / (sweep_span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if (dom == "sweep" and sweep_span_ms > 0) else None,
                           "achieved_rocprof": kb[dom] / (rp_us * 1e-6) / 1e9 if rp_us else None,
                           "frac_rocprof": kb[dom] / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS if rp_us else None,
                           "profiles_match_kernel_source": profiles_current(),
                           "kernel_source_sha256": kernel_source_hash(),
                           "fill_in_sweep_launch": fill_in_sweep,
                           "note": "dominant kernel by duration. The alpha/beta recursion itself is dependency-latency bound (T "
                                   "sequential frame steps on 2B chain wavefronts; kernels.sweep.us_per_frame_step is its figure "
                                   "of merit, kernels.sweep_alone its bytes); with fill_in_sweep_launch the same launch also "
                                   "carries the bandwidth-bound half of the step -- the zero-fill of the gradient box by extra "
                                   "workgroups on the CUs the 2B sweep workgroups leave idle -- and its algorithmic bytes are the "
                                   "sum of both (rows + emissions + the 4V-per-frame gradient rows of SURVEY.md 8(d)). "
                                   "avg_launch_ms is the HIP-event interval around the launch on its stream (launch / event "
                                   "latency included: a few us more than the kernel's own duration in the committed rocprofv3 "
                                   "summary, rocprof_avg_launch_us). traffic: profiles/" + PROFILE_TAG + "_pmc_traffic.json "
                                   "(FETCH_SIZE doubled + WRITE_SIZE, per launch)"}
        out["kernels"] = {k: {"avg_ms": kt[k], "algorithmic_bytes": kb[k],
                              "GBps": kb[k] / (kt[k] * 1e-3) / 1e9 if kt[k] > 0 else None} for k in kt}
        if fill_in_sweep:
            out["kernels"]["sweep"]["algorithmic_bytes_sweep_alone"] = kb["sweep_alone"]
        out["kernels"]["sweep"]["us_per_frame_step"] = kt["sweep"] * 1e3 / args.frames
        out["kernels"]["sweep"]["in_kernel_span_ms"] = sweep_span_ms if sweep_span_ms > 0 else None
        # SURVEY 8(d)'s ALGORITHMIC bytes (4V log_probs row + 4V gradient row + 8S per frame) over the step time: a
        # figure of merit, NOT moved bytes -- the numerators' gather reads K of the V columns (whole 128-byte lines of them)
        out["path_GBps_survey_8d"] = kb["path_survey_8d"] / (ms_step * 1e-3) / 1e9
        out["path_GBps_survey_8d_is"] = "algorithmic bytes of SURVEY 8(d) / step time; the bytes actually moved are traffic_GBps"
        step_traffic = pmc_step_traffic(default_wl)
        out["traffic_bytes_per_step_pmc"] = step_traffic
        out["traffic_GBps"] = step_traffic / (ms_step * 1e-3) / 1e9 if step_traffic else None
        out["traffic_frac_of_hbm_peak"] = step_traffic / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS if step_traffic else None
        if world == 1:
            try:   # the optional tail of the fused all-reduce buffer: its kernel, timed here
                buf = torch.zeros(3 + wl.V, dtype=torch.float32, device=dev)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                wf.dist.fused_loss_stats(wl.tot, wl.frames, out=buf, grad=wl.grad)
                ev[0].record()
                for _ in range(10):
                    wf.dist.fused_loss_stats(wl.tot, wl.frames, out=buf, grad=wl.grad)
                ev[1].record()
                torch.cuda.synchronize()
                ref = wl.grad.reshape(-1, wl.V).double().sum(0)
                out["label_grad_sums_ms"] = ev[0].elapsed_time(ev[1]) / 10
                out["label_grad_sums_max_abs_err"] = float((buf[3:].double() - ref).abs().max().item())
            except Exception as e:   # noqa: BLE001
                out["label_grad_sums_ms"] = {"error": repr(e)}
            try:
                out.update(dropin_times(wl, max(20, args.steps)))
            except Exception as e:   # noqa: BLE001
                out["dropin"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(wl)
            except Exception as e:   # noqa: BLE001  (the GPU numbers above stand on their own)
                out["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and not args.no_others and (args.others or default_wl):
            others = {}
            # (a failure here must not cost the headline line: it is recorded in place of the numbers)
            for name in WORKLOADS:
                if name == args.workload:
                    continue
                try:
                    B = 32
                    w2 = Workload(name, B, args.frames, dev, seed=777)
                    n2 = max(5, args.steps // 5)
                    d2 = float(np.median(time_steps(w2, n2, 3, None, stream, fused[:3], min_region_s=0.03)))
                    k2 = kernel_times(w2, stream, reps=5)
                    k2["fill_in_sweep"] = bool(k2["fill_in_sweep"])
                    if k2["sweep_device_span"] < 0:
                        k2.pop("sweep_device_span")
                    others[name] = {"frames_per_s": w2.frames / (d2 / n2), "ms_per_step": d2 / n2 * 1e3,
                                    "fallback_utterances": w2.fallbacks(stream),
                                    "arc_updates_per_sec": 2 * w2.arc_updates / (d2 / n2), "kernels_ms": k2,
                                    "desc": WORKLOADS[name]["desc"]}
                    del w2
                except Exception as e:   # noqa: BLE001
                    others[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
            # the reference's own semantics: every caller passes output_beam (default 10, graph.py:61-62,81); the same
            # fused call with WF_PRUNED (tropical pass + keep masks + masked sweeps; tests/test_pruned.py)
            for name, key in (("cfg3_ctc_bpe5k_num", "cfg3_pruned_beam10"), ("cfg2_2state_char_num", "cfg2_num_pruned_beam10"),
                              ("cfg2_2state_char_den", "cfg2_den_pruned_beam10")):
                try:
                    w2 = Workload(name, 32, args.frames, dev, seed=777, beam=10.0)
                    n2 = max(5, args.steps // 5)
                    d2 = float(np.median(time_steps(w2, n2, 3, None, stream, fused[:3], min_region_s=0.03)))
                    exact_ms = ms_step if name == args.workload else others.get(name, {}).get("ms_per_step")
                    others[key] = {"frames_per_s": w2.frames / (d2 / n2), "ms_per_step": d2 / n2 * 1e3,
                                   "exact_ms_per_step": exact_ms, "ratio_to_exact": (d2 / n2 * 1e3 / exact_ms) if exact_ms else None,
                                   "fallback_utterances": w2.fallbacks(stream),
                                   "desc": WORKLOADS[name]["desc"] + ", output_beam=10 (WF_PRUNED): lattice arcs more than 10 below "
                                           "the tropical best path are dropped, as k2.intersect_dense(output_beam=10) states",
                                   "checked_by": "tests/test_pruned.py::test_pruned_at_baseline_size_against_oracle"}
                    del w2
                except Exception as e:   # noqa: BLE001
                    others[key] = {"error": repr(e)}
                torch.cuda.empty_cache()
            # use_double_scores=True (graph.py:83; every lattice_analysis.py call): fp64 totals from the same fast kernels
            try:
                w2 = Workload("cfg3_ctc_bpe5k_num", 32, args.frames, dev, seed=777, double=True)
                n2 = max(5, args.steps // 5)
                d2 = float(np.median(time_steps(w2, n2, 3, None, stream, fused[:3], min_region_s=0.03)))
                exact_ms = ms_step if args.workload == "cfg3_ctc_bpe5k_num" else others.get("cfg3_ctc_bpe5k_num", {}).get("ms_per_step")
                others["cfg3_double"] = {"frames_per_s": w2.frames / (d2 / n2), "ms_per_step": d2 / n2 * 1e3,
                                         "float_ms_per_step": exact_ms, "fallback_utterances": w2.fallbacks(stream),
                                         "desc": WORKLOADS["cfg3_ctc_bpe5k_num"]["desc"] + ", use_double_scores=True (WF_DOUBLE_ACC | "
                                                 "WF_TOT_F64): fp64 totals from sweep_band_kernel, fp32 gradient",
                                         "checked_by": "tests/test_gpu_parity.py::test_double_scores_cfg3_size_rtol_1e9"}
                del w2
            except Exception as e:   # noqa: BLE001
                others["cfg3_double"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:
                others["manual_ctc_cfg2"] = manual_ctc_bench(dev)
            except Exception as e:   # noqa: BLE001
                others["manual_ctc_cfg2"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            for S in (100_000, 1_000_000):
                try:
                    others["cfg5_viterbi_S%.0e" % S] = viterbi_bench(dev, S)
                except Exception as e:   # noqa: BLE001
                    others["cfg5_viterbi_S%.0e" % S] = {"error": repr(e)}
                torch.cuda.empty_cache()
            # the search where real decoding sits: 10^3 - 10^4 active states (the posterior's peakedness sets the count)
            for peak, key in ((2.6, "cfg5_search_S1e+06_beam20_active740"), (2.3, "cfg5_search_S1e+06_beam20_active2000"),
                              (2.0, "cfg5_search_S1e+06_beam20_active5000")):
                try:
                    others[key] = search_bench(dev, 1_000_000, peak=peak)
                except Exception as e:   # noqa: BLE001
                    others[key] = {"error": repr(e)}
                torch.cuda.empty_cache()
            try:
                others["cfg5_search_S1e+06_beam20"] = search_bench(dev, 1_000_000)
            except Exception as e:   # noqa: BLE001
                others["cfg5_search_S1e+06_beam20"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:
                others["cfg4_mmictc_num_minus_den_autograd"] = num_den_bench(dev)
            except Exception as e:   # noqa: BLE001
                others["cfg4_mmictc_num_minus_den_autograd"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            out["other_workloads"] = others
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
