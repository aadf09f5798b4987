/* wf_dwfst.h -- C ABI of libwfdwfst.so: the MI355X (gfx950) DWFST loss / Viterbi hot path.
 *
 * This is the drop-in boundary for the k2 calls made by the reference
 * (ZhaoZeyu1995/Waterfall, paths below are relative to that repository):
 *
 *   waterfall/graph.py:61-66     k2.intersect_dense(graph, dense, output_beam)
 *                                + lattice.get_tot_scores(log_semiring=True, use_double_scores)
 *                                + the autograd backward of both        -> wf_fb_logsum()
 *   waterfall/models.py:256-261  k2.DenseFsaVec(log_probs, supervision_segments)
 *                                -> no copy is made: (log_probs pointer, strides, seg[B,3]) is the
 *                                   dense FSA vector; wf_fb_logsum()/wf_viterbi() take it directly
 *   utils/lattice_analysis.py:129,149,155  get_tot_scores(True, False) (tropical)  -> wf_tot_scores()
 *   utils/lattice_analysis.py:79,133,158   k2.shortest_path / one_best_decoding    -> wf_viterbi()
 *   waterfall/models.py:263-270  linear_fsa -> remove_epsilon_and_add_self_loops -> arc_sort ->
 *                                compose(topo, ., treat_epsilons_specially=False)  -> wf_compose_linear()
 *   waterfall/manual_ctc/ctc.py:94-105  ctc_loss(log_probs, input_lengths, trans, trans_lengths,
 *                                transform): converted to a graph on the host (Python, manual_ctc.py)
 *                                and run through wf_fb_logsum()
 *
 * The reference reaches k2 through pybind; it has no FFI of its own.  The reference-side binding a
 * maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures (hipStream_t is passed as
 *     void*; NULL = the default stream).
 *   - every function returns 0 on success or a negative WF_E* code; wf_last_error() returns a
 *     thread-local message.  No C++ exception crosses this boundary.
 *   - the library never allocates or frees device memory and keeps no pointer after returning:
 *     the caller allocates outputs and the workspace (size from wf_fb_workspace_bytes()).
 *   - device work is stream-ordered and asynchronous.  The per-call tables (a few KB) are staged in a
 *     pinned ring owned by the calling thread's context and copied with one asynchronous copy; the host
 *     waits only if the copy issued 8 calls earlier by the same thread has not finished yet.
 *   - per host thread and device the library keeps a small context (two side streams, their events, the
 *     pinned ring): created lazily on first use or by wf_init(), released by wf_shutdown().  It is the only
 *     mutable state besides the thread-local error string; entry points are re-entrant across threads.
 *   - an utterance whose graph has no path of its length is NOT an error: tot = -inf, grad = 0.
 *
 * Graph layout (host side, k2's Fsa layout): arcs int32[A][4] = {src, dst, label, score_bits}
 * with state ids local to their FSA, arcs of one FSA contiguous; start state 0; the last state
 * of every FSA is the super-final state, entered only by label -1 arcs; labels >= 0 index the
 * last dimension of log_probs.
 */
#ifndef WF_DWFST_H_
#define WF_DWFST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WF_OK 0
#define WF_EINVAL (-1)    /* bad argument / malformed graph          */
#define WF_ENOSPC (-2)    /* caller buffer / workspace too small     */
#define WF_ERANGE (-3)    /* label >= V, state out of range, ...     */
#define WF_EHIP (-4)      /* a HIP runtime call failed               */
#define WF_EUNSUPPORTED (-5)

/* flags for wf_fb_logsum / wf_tot_scores / wf_viterbi */
#define WF_DOUBLE_ACC 1u  /* k2 use_double_scores=True: fp64 from the emissions to the total.  Graphs of the banded wave sweep
                             (every numerator, the small denominators) keep the fast probability-domain kernels, whose
                             recursion is fp64 anyway; the others take the log-semiring sweep with fp64 exp / log        */
#define WF_TROPICAL 2u    /* wf_tot_scores only: max instead of log-sum-exp                       */
#define WF_KEEP_AB 4u     /* wf_fb_logsum with grad == NULL: still run both sweeps and leave alpha/
                             beta in the workspace so that wf_fb_backward() can follow later       */
#define WF_PRUNED 8u      /* wf_fb_logsum is called with 0 < output_beam < inf: pass it to
                             wf_fb_workspace_bytes, wf_fb_logsum and wf_fb_backward alike (the tropical
                             first pass and its thresholds need workspace)                          */
#define WF_PREZERO 16u    /* wf_fb_logsum(flags | WF_KEEP_AB | WF_PREZERO, grad != NULL): the gradient is NOT computed; the
                             grad box is zero-filled on a side stream beside the sweeps, so that the
                             wf_fb_backward(flags | WF_PREZERO, same grad) that follows only scatters the posteriors
                             (the autograd split of loss and gradient without paying for the fill twice over)   */
#define WF_ACCUMULATE 32u /* wf_fb_logsum with grad != NULL (no counterpart in k2: its gradients meet in autograd): the
                             gradient box already holds values and this call's gradient is ADDED to them -- nothing is
                             zero-filled, only the K label columns of each row are touched.  For the numerator of a
                             numerator - denominator loss (rnnp.py:91-104) after the denominator's call wrote the box:
                             one buffer, no fill, no addition pass.  Graphs of the probability-domain path with at most
                             512 states / 1024 arcs (every topo o transcript numerator); others: WF_EUNSUPPORTED (asked
                             with the flag, wf_fb_workspace_bytes says so beforehand).  Also taken by wf_fb_backward after a
                             wf_fb_logsum(WF_KEEP_AB, grad == NULL): the numerator's sweeps may then run on another stream
                             beside the denominator's.                                                                   */

#define WF_TOT_F64 64u    /* wf_fb_logsum with WF_DOUBLE_ACC: `tot` is fp64 [B] (what get_tot_scores(use_double_scores=True)
                             returns: graph.py:66, lattice_analysis.py:128); the gradient stays fp32 like log_probs      */

typedef struct {
    int32_t num_fsas;                 /* G                                                     */
    const int32_t* arcs;              /* host, [num_arcs][4]                                   */
    const int32_t* fsa_arc_offsets;   /* host, [G+1]                                           */
    const int32_t* fsa_num_states;    /* host, [G]                                             */
} wf_graph_host;

int wf_version(void);
const char* wf_last_error(void);

/* ---- set-up / tear-down (no counterpart in the reference: k2 owns its CUDA context implicitly) -----
 * wf_init(): set-up for the calling host thread on the CURRENT device (idempotent): side streams + events, the
 * pinned staging ring, kernel attributes, the device address of the static scratch array.  Optional for every entry
 * point except wf_fb_logsum_static (the others do it lazily).  wf_shutdown(): waits for the side streams of the
 * calling thread and destroys its contexts on all devices.
 * wf_debug_reload_env(): the WF_* environment test hooks are read once per process; tests that change them call this. */
int wf_init(void);
int wf_shutdown(void);
int wf_debug_reload_env(void);
/* Test aid (no counterpart in the reference): the number of utterances of the last wf_fb_logsum call on `workspace` that
 * were redone by the log-semiring fallback (rows of the probability-domain sweep that failed the row check).  Arguments
 * as for wf_fb_workspace_bytes of that call; synchronises `stream`.  Tests use it to tell a kernel that is right from
 * one whose mistakes the fallback repairs.                                                                          */
int wf_debug_fallback_count(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                            int32_t num_seqs, int32_t V, uint32_t flags, int32_t need_grad,
                            const void* workspace, size_t workspace_bytes, void* stream, int32_t* count);

/* ---- graph preparation (host only, no GPU needed) ------------------------------------------
 * Builds the position-independent "prepared graph": per FSA the incoming-arc CSR (forward sweep),
 * outgoing-arc CSR (backward sweep), label-sorted arc lists (posterior reduction), final arcs and
 * the compact label table.  The caller copies the blob to the device once per graph and passes
 * both copies to the compute calls (the host copy is read for launch planning only).
 * Replaces what k2 does inside Fsa construction / arc_sort (datapipe.py:180, models.py:267).   */
int wf_graph_prepare_bytes(const wf_graph_host* g, size_t* bytes);
int wf_graph_prepare(const wf_graph_host* g, void* blob_host, size_t bytes);

/* ---- numerator graph compiler (host) --------------------------------------------------------
 * compose(topo, linear transcript with epsilon self-loops), models.py:263-270.
 * topo: one FSA (arcs + aux = output phone label, 0 = epsilon, -1 on final arcs).
 * Two-call protocol: with out_arcs == NULL only the sizes are written.                          */
int wf_compose_linear(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs,
                      int32_t topo_num_states, const int32_t* phones, int32_t num_phones,
                      int32_t* out_arcs, int32_t* out_aux, int32_t* out_num_arcs,
                      int32_t* out_num_states, int32_t arc_capacity);

/* Batched form (one call per training batch; SURVEY.md 8(f) rank 1: the graph compile runs on every step in
 * front of the loss, models.py:263-270 / datapipe.py:233-256).  The topology is indexed once by (source state,
 * phone) with wf_topo_index_build (index: caller-allocated int32[wf_topo_index_words]); each product state then
 * touches only the topology arcs with phone 0 or phone b[i].  Graphs are identical to wf_compose_linear's.
 * phones / phones_off[B+1]: the transcripts, concatenated.  out_arc_off[B+1], out_num_states[B] are always
 * written; arcs only when out_arcs != NULL and arc_capacity suffices (else WF_ENOSPC, with out_arc_off[B] =
 * the number of arcs needed).
 * ctc_cross != 0: topo is the CTC topology (utils/get_token_fst_ctc.py:25-53) WITHOUT its O(P^2) arcs
 * "state of another phone -> state of phone p, p : p" (:37-40), which are then implied by phone_state[p] /
 * phone_token[p] (int32[num_phones], 0 = the phone has no state): 25 M arcs at P = 5000 are never built.   */
int wf_topo_index_words(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs, int32_t topo_num_states,
                        int64_t* words);
int wf_topo_index_build(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs,
                        int32_t topo_num_states, int32_t* index);
int wf_compose_linear_batch(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs,
                            int32_t topo_num_states, const int32_t* index, int32_t ctc_cross,
                            const int32_t* phone_state, const int32_t* phone_token, int32_t num_phones,
                            const int32_t* phones, const int32_t* phones_off, int32_t B,
                            int32_t* out_arcs, int32_t* out_aux, int64_t arc_capacity,
                            int32_t* out_arc_off, int32_t* out_num_states);

/* wf_compose_linear_batch + wf_graph_prepare in ONE call, the utterances spread over `nthreads` persistent host threads:
 * what a training step needs between the data loader and the loss (models.py:263-270 / datapipe.py:233-256 run on every
 * step).  Writes the FsaVec arrays as wf_compose_linear_batch does and the prepared blob (give it pinned memory and
 * upload with one asynchronous copy).  Every graph sits at the start of an upper-bound region of the blob (gaps in
 * between), so the words to upload are blob[2] (WF_H_WORDS).  WF_ENOSPC: out_arc_off[B] arcs / *blob_bytes_needed
 * bytes are needed.                                                                                                 */
int wf_numerator_compile(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs,
                         int32_t topo_num_states, const int32_t* index, int32_t ctc_cross,
                         const int32_t* phone_state, const int32_t* phone_token, int32_t num_phones,
                         const int32_t* phones, const int32_t* phones_off, int32_t B,
                         int32_t* out_arcs, int32_t* out_aux, int64_t arc_capacity,
                         int32_t* out_arc_off, int32_t* out_num_states,
                         void* blob, size_t blob_bytes, size_t* blob_bytes_needed, int32_t nthreads);

/* ---- the numerator graph compiler ON THE DEVICE (models.py:263-270 / datapipe.py:233-256 run on every
 * training step) -----------------------------------------------------------------------------------------------------
 * Same graphs and the same prepared tables as wf_numerator_compile, built by one wavefront per utterance from the flat
 * transcripts: nothing but the transcripts crosses PCIe and the host does no graph work.
 * wf_topo_device_build: the topology tables the kernel reads (arc arrays, the wf_topo_index_build index, the factored-CTC
 *   tables, per phone the states its arcs enter), once per topology, into a host buffer the caller uploads.  out == NULL:
 *   only *words_needed.
 * wf_numerator_compile_device_bytes: sizes of the caller-allocated DEVICE buffers for a batch with these transcript lengths.
 * wf_numerator_compile_device: two launches on `stream`.  topo_tab: the HOST copy of the table (read for sizes only).
 *   phones_dev / phones_off_dev: the transcripts on the device; phones_off: the same offsets on the host.  Outputs, all on
 *   the device: blob_dev (the prepared graph vector: header, per-FSA info words, tables; every graph at the start of a
 *   capacity-sized region), arcs_dev [.][4] + aux_dev (the FsaVec arrays, compact), meta_dev int32 [3B + 1] = arc offsets
 *   [B + 1] | states [B] | status [B].  status != 0 (transcript beyond 256 positions, more than 257 states / 1024 arcs, a
 *   phone out of range, ...): that utterance was not compiled, the blob's magic word is 0 and the caller compiles the batch
 *   on the host.  To plan the compute calls the caller copies the first 8 + 64 B words (rounded up to 4) of blob_dev (header + info
 *   words) to the host: that prefix is all the host side of wf_fb_logsum / wf_viterbi / wf_search reads of a blob.     */
int wf_topo_device_build(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs, int32_t topo_num_states,
                         const int32_t* index, int64_t index_words, int32_t ctc_cross, const int32_t* phone_state,
                         const int32_t* phone_token, int32_t num_phones, int32_t* out, int64_t out_words, int64_t* words_needed);
int wf_numerator_compile_device_bytes(const int32_t* topo_tab, const int32_t* phones_off, int32_t B, size_t* blob_bytes,
                                      int64_t* arc_capacity, size_t* work_bytes);
int wf_numerator_compile_device(const int32_t* topo_tab, const void* topo_tab_dev, const int32_t* phones_dev,
                                const int32_t* phones_off_dev, const int32_t* phones_off, int32_t B, void* blob_dev,
                                size_t blob_bytes, int32_t* arcs_dev, int32_t* aux_dev, int64_t arc_capacity,
                                int32_t* meta_dev, void* work_dev, size_t work_bytes, void* stream);

/* The same compiler as ONE launch per batch for a training loop (waterfall_amd/compile.py): the caller brings one device
 * buffer of *dev_bytes (any allocator) and two PINNED, device-visible host buffers (hipHostMalloc / torch pin_memory); the
 * kernel reads the transcripts from in_host and writes the words the host plans with into out_host itself, and the
 * workgroup that finishes last closes the batch (arc offsets, WF_I_ARC_BASE, the blob header).  No copy is enqueued,
 * nothing waits on the host.
 * wf_numerator_compile_job_bytes: sizes for these transcript lengths, or (phones_off == NULL) the most any batch of B
 *   transcripts can take; layout[5] = {meta words M, header words H, blob bytes (the blob starts dev_buf), byte offset of
 *   the arc staging rows, arc capacity} (wf_numerator_compile_job returns the same five for the batch it ran).
 * wf_numerator_compile_job: in_host = phones_off [B + 1] (phones_off[0] = 0) followed by the flat transcripts.  out_host
 *   receives M + H words: meta (arc offsets [B + 1] | states [B] | status [B] | staging offsets [B] | the number of graphs
 *   with direct label columns, padded to M) then the
 *   blob's header and per-FSA info words (H) -- valid once wf_compile_job_wait(*job) returns.  The FsaVec arcs stay in
 *   dev_buf as staging rows {src, dst, label, weight bits, aux label}: utterance b's rows start at row `staging offset
 *   b`, arc offsets[b + 1] - arc offsets[b] of them (a loss never reads them; compile.py fetches them on demand).
 *   main_stream != NULL: compile_stream first waits for what main_stream has queued so far (for a dev_buf whose
 *   allocator orders reuse on main_stream); NULL: the caller vouches that compile_stream may write dev_buf at once.
 *   in_host must stay untouched until the job has completed.  Labels < 1: WF_EINVAL.
 *   reuse_of: 0, or the job that used this dev_buf before (a caller that recycles its buffers): compile_stream then also
 *   waits for that job's kernel and for the event of wf_compile_job_release(reuse_of).
 * wf_compile_job_release: the caller is done with the job's graphs; records, on the streams that read them, the event a
 *   later job that reuses the buffer waits for.
 * wf_compile_job_wait: blocks the host until the job has completed.  wf_compile_job_join: makes `stream` wait for the
 *   job (no host wait).  Jobs are numbered per device; at most 64 are tracked at once (an older one is waited for when
 *   its slot is taken).  No counterpart in the reference: its compiler runs on the host inside the training step. */
int wf_numerator_compile_job_bytes(const int32_t* topo_tab, const int32_t* phones_off, int32_t B, size_t* dev_bytes,
                                   size_t* out_words, int64_t* layout);
int wf_numerator_compile_job(const int32_t* topo_tab, const void* topo_tab_dev, const int32_t* in_host, int32_t B,
                             void* dev_buf, size_t dev_bytes, int32_t* out_host, size_t out_words, int32_t device,
                             void* main_stream, void* compile_stream, int64_t reuse_of, int64_t* job, int64_t* layout);
int wf_compile_job_wait(int32_t device, int64_t job);
int wf_compile_job_join(int32_t device, int64_t job, void* stream);
int wf_compile_job_release(int32_t device, int64_t job, void* const* streams, int32_t num_streams);

/* The same for ARBITRARY epsilon-free right-hand graphs (one per utterance), e.g. the phone-level graph of a word
 * sequence under a lexicon with alternative pronunciations and optional silence (datapipe.py:244-254: L_inv o words,
 * inverted, epsilons removed, epsilon self-loops added, then k2.compose(topo, ., treat_epsilons_specially=False)).
 * b_arcs int32 [.,4] = {src, dst, phone >= 1 | -1, fp32 score bits}, sorted by source state, label -1 arcs into the last
 * state; b_aux: word ids (or NULL: the output aux label is the phone); b_arc_off[B+1], b_num_states[B].
 * Output label = token, aux = b's aux, score = fp32 sum.  Same two-call protocol as wf_compose_linear_batch.          */
int wf_compose_fsa_batch(const int32_t* topo_arcs, const int32_t* topo_aux, int32_t topo_num_arcs,
                         int32_t topo_num_states, const int32_t* index, const int32_t* b_arcs, const int32_t* b_aux,
                         const int32_t* b_arc_off, const int32_t* b_num_states, int32_t B,
                         int32_t* out_arcs, int32_t* out_aux, int64_t arc_capacity,
                         int32_t* out_arc_off, int32_t* out_num_states);

/* ---- workspace ------------------------------------------------------------------------------
 * utt2fsa: host [B] FSA index used by utterance i, or NULL for the identity (B == G); a single
 * shared denominator topology (rnnp.py:91-92) is utt2fsa = {0,0,...}.
 * seg: host int32 [B][3] = {sequence index b, start frame, num frames} (models.py:256-258).  A sequence named by
 * several segments of one call: the forward-backward entry points (wf_fb_logsum / wf_fb_backward / wf_tot_scores with
 * WF_DOUBLE_ACC) return WF_EUNSUPPORTED -- their gradient kernels walk one segment per sequence; call them once per
 * round (round r = the r-th segment of every sequence; utt2fsa selects the round's graphs from the same prepared blob,
 * gradients of later rounds go to a second box and are added), as the Python drop-in does.  wf_viterbi / wf_search take
 * such batches as they are.                                                                      */
/* num_seqs: the first dimension of log_probs (the value the compute calls receive): sizes the seq -> utterance map. */
int wf_fb_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg,
                          int32_t B, int32_t num_seqs, int32_t V, uint32_t flags, int32_t need_grad, size_t* bytes);

/* *fast = 1 when wf_fb_logsum(flags | WF_PRUNED) keeps these graphs on the fast kernels (the banded path: every topo o
 * transcript numerator, denominators of <= 64 states), 0 when it takes the masked log-semiring sweeps (hub-form
 * denominators, graphs beyond the LDS, WF_DOUBLE_ACC).  No counterpart in the reference (k2 has one intersect_dense):
 * the drop-in's "auto" output_beam mode decides with it.                                                              */
int wf_fb_pruned_is_fast(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                         int32_t num_seqs, int32_t V, uint32_t flags, int32_t* fast);

/* ---- log-semiring forward-backward: total scores and gradient (graph.py:61-66 + autograd) ---
 * log_probs: device fp32, element (b,t,v) at b*strideB + t*strideT + v.
 * output_beam: <= 0 or +inf disables pruning (exact sum over all lattice arcs); otherwise (with
 *   WF_PRUNED in flags) a lattice arc (t, a) is kept iff alpha^_t[src] + x + beta^_{t+1}[dst] >=
 *   best - output_beam in the tropical semiring (k2.intersect_dense's output_beam, graph.py:61-62;
 *   k2 is not available to pin this arithmetic against: SURVEY 8(0) / 8(c)), at the cost of a
 *   tropical forward-backward pass in front of the log-semiring one.
 * grad_scale: DEVICE fp32 [B] or NULL(=1): grad[b,t,v] = grad_scale[i] * sum of arc posteriors,
 *   i.e. -1 for d(-tot)/d log_probs, -1/(B*target_len) for reduction 'mean' (graph.py:74-76).
 * tot: device fp32 [B].  grad: device fp32, same strides as log_probs, every element of the
 *   B x Tmax x V box named by seg rows is overwritten (zeros outside the segments' frames);
 *   NULL = forward only (e.g. a denominator under torch.no_grad(), conformer.py:413).
 * Tmax: number of frames per sequence in the grad tensor.                                        */
int wf_fb_logsum(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                 const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                 int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, float output_beam,
                 uint32_t flags, const float* grad_scale, float* tot, float* grad,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- the same call for callers that cannot allocate --------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  For callers that must not call hipMalloc (code under HIP graph capture, tight
 * serving loops, harnesses that forbid allocation): wf_fb_logsum with the workspace replaced by a static per-device
 * array inside the library (WF_STATIC_SCRATCH_BYTES, held by every GPU the library runs on until the process ends).
 * Same kernels and results as wf_fb_logsum.  Calls nothing that allocates; requires wf_init() by the calling thread on
 * the current device.  Returns WF_ENOSPC when the call needs more scratch than the array holds (cfg3, B=32 T=500
 * V=5001 L=64 needs 44 MiB; cfg4's 10 002-state denominators do not fit) or its tables exceed a 64 KiB staging slot,
 * WF_EUNSUPPORTED for WF_KEEP_AB / WF_PREZERO (a later wf_fb_backward needs a caller workspace) and for graphs beyond
 * the LDS-resident sweeps.  All calls on one GPU share the array: calls that overlap on one GPU (two streams, two
 * threads) race; stream-ordered calls are safe.                                                                     */
#define WF_STATIC_SCRATCH_BYTES (96u << 20)
int wf_static_scratch_bytes(size_t* bytes);
/* Test aid (no counterpart in the reference): the number of allocating calls the library has made in this process so far
 * (pinned staging ring, side streams, events; it never allocates device memory).  Unchanged across a call = that call
 * allocated nothing. */
int wf_debug_alloc_count(int64_t* count);
int wf_fb_logsum_static(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                        const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                        int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, float output_beam,
                        uint32_t flags, const float* grad_scale, float* tot, float* grad, void* stream);

/* ---- deferred gradient: the autograd backward of graph.py:61-66 --------------------------------
 * Must follow a wf_fb_logsum(..., flags | WF_KEEP_AB, grad = NULL, same graph / seg / workspace) on
 * the same stream: reads alpha/beta/tot left there and writes grad rows scaled by grad_scale (the
 * upstream d loss / d tot_b, a DEVICE array, so no host synchronisation is needed).               */
int wf_fb_backward(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                   const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs,
                   int32_t Tmax, int32_t V, const int32_t* seg, int32_t B, uint32_t flags,
                   const float* grad_scale, const float* tot, float* grad,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- the drop-in's eager gradient (graph.py:66-76 + autograd; waterfall_amd/graph.py) ----------------------------
 * graphloss() under autograd runs the FUSED wf_fb_logsum in its forward (grad_scale[b] = -w_b, the reduction's own
 * factor) and keeps the gradient for backward, which only has to multiply it by the upstream gradient:
 * wf_loss_reduce: out[0] = sum_b w[b] * (-tot[b]) (w NULL: 1) -- 'sum', and 'mean' with w[b] = 1 / (B * target_len[b]).
 * wf_grad_scale: grad[., ., label columns of the utterance's graph] *= gs[u] (per_utt) or gs[0]; wavefronts whose
 * factor is exactly 1 -- the usual loss.backward() -- exit at once.  Must follow the wf_fb_logsum(grad != NULL) call
 * with the same graph / seg / workspace on the same stream (reads the tables it left there).  gs: DEVICE fp32.   */
int wf_loss_reduce(const float* tot, const float* w, int32_t B, float* out, void* stream);
int wf_grad_scale(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const int32_t* seg,
                  int32_t B, int32_t num_seqs, int32_t Tmax, int32_t V, int64_t strideB, int64_t strideT,
                  const float* gs, int32_t per_utt, float* grad, void* workspace, size_t workspace_bytes, void* stream);

/* ---- state posteriors (waterfall/manual_ctc/prior.py:4-55 compute_align) -----------------------
 * Must follow a wf_fb_logsum(..., flags | WF_KEEP_AB, same graph / seg / workspace) on the same stream.
 * out: device fp32, element (i, t, s) at i*out_strideB + t*out_strideT + s for segment i, frame t
 * (relative to the segment's start), state s < Smax:
 *   log P(path of utterance i is in state s after consuming frame t) = alpha_{t+1}[s] + beta_{t+1}[s] - tot;
 * -inf for dead / padding states, frames >= the segment's length and infeasible utterances.      */
int wf_state_posteriors(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                        const int32_t* seg, int32_t B, int32_t V, int32_t num_seqs, uint32_t flags,
                        float* out, int64_t out_strideB, int64_t out_strideT, int32_t Smax, int32_t Tmax,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- total scores only, log or tropical (lattice_analysis.py:128-129) ------------------------
 * tot: device fp32 [B] (or fp64 [B] when WF_DOUBLE_ACC is set).  num_seqs / Tmax: the first two dimensions of
 * log_probs; every compute entry point checks each segment against them (WF_ERANGE) before anything is launched.
 * Workspace: wf_fb_workspace_bytes(flags, need_grad = 0) with the same num_seqs.                 */
int wf_tot_scores(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
                  const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t Tmax,
                  int32_t V, const int32_t* seg, int32_t B, uint32_t flags, void* tot,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- tropical best path (k2.shortest_path / one_best_decoding) -------------------------------
 * best_score device fp32 [B]; best_arcs device int32 [B][Tmax+1]: arc indices into the FsaVec's
 * arc array (T_b emitting arcs then the final arc; -1 padded; all -1 when infeasible).
 * Ties are broken by the lowest arc index at every back-trace step.                             */
int wf_viterbi_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg,
                               int32_t B, int32_t V, uint32_t flags, size_t* bytes);
int wf_viterbi(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa,
               const float* log_probs, int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V,
               const int32_t* seg, int32_t B, int32_t Tmax, uint32_t flags, float* best_score,
               int32_t* best_arcs, void* workspace, size_t workspace_bytes, void* stream);

/* ---- beam-pruned frame-synchronous search (k2.get_lattice(log_prob, log_prob_len, TLG, search_beam, output_beam,
 * min_active_states, max_active_states), utils/lattice_analysis.py:125 with the beams of lattice_analysis.sh:9-11) --------
 * Token passing over the outgoing arcs of a decoding graph of any size; one workgroup per utterance, or several (up to 8
 * while the launch fills at most half of the chip) with a bounded barrier in device memory between them.  A state of frame t
 * survives when its tropical score is within search_beam of the frame's best; when more than max_active_states survive
 * (0: no limit) or fewer than min_active_states, the cut-off moves along a 1024-bin histogram of the frame's scores.
 * search_beam = +inf with max_active_states = 0 is the exact search (scores and arcs equal wf_viterbi's bit for bit).
 * best_score / best_arcs: as wf_viterbi (ties -> lowest arc index).  tot_log (nullable): device fp32 [B], or fp64 [B] with
 * WF_DOUBLE_ACC: log-semiring total score of the pruned lattice (all paths through surviving states).  n_active (nullable):
 * device int32 [B][Tmax+1], surviving states per frame.  output_beam: wf_search_lattice below.  k2's own pruning
 * arithmetic (dynamic beams) is parity unpinned: the reference holds no fixture for it; see INTEGRATION.md.            */
int wf_search_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B, int32_t V,
                              int32_t max_active_states, int32_t want_log_tot, size_t* bytes);
int wf_search(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
              int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B, int32_t Tmax,
              float search_beam, int32_t min_active_states, int32_t max_active_states, uint32_t flags,
              float* best_score, int32_t* best_arcs, void* tot_log, int32_t* n_active, void* workspace,
              size_t workspace_bytes, void* stream);

/* The same search followed by get_lattice's output_beam (utils/lattice_analysis.py:125, lattice_beam of lattice_analysis.sh):
 * one more launch walks the lattice the search left (every arc between a survivor of frame t and a survivor of frame t + 1,
 * and the final arcs) backwards and keeps the arcs with alpha(src) + score + beta(dst) >= best_score - output_beam (tropical
 * alpha / beta within that lattice, fp32); tot_log (required; fp32 [B] or fp64 [B] with WF_DOUBLE_ACC) is the log-semiring
 * total of the lattice that remains -- what lattice.get_tot_scores(True, True) reads at lattice_analysis.py:128 --
 * lattice_arcs (nullable, device int64 [B]) the number of arcs in it.  best_score / best_arcs / n_active as wf_search
 * (the best path never depends on output_beam).  output_beam = +inf keeps every arc that lies on a complete path: the total
 * then equals wf_search's.  The rule is oracle.lattice_search's; k2's own arithmetic is parity unpinned as above.      */
int wf_search_lattice_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B, int32_t V,
                                      int32_t max_active_states, size_t* bytes);
int wf_search_lattice(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
                      int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B, int32_t Tmax,
                      float search_beam, float output_beam, int32_t min_active_states, int32_t max_active_states,
                      uint32_t flags, float* best_score, int32_t* best_arcs, void* tot_log, int64_t* lattice_arcs,
                      int32_t* n_active, void* workspace, size_t workspace_bytes, void* stream);

/* ... and the lattice restricted to a word sequence per utterance: `latticeW = k2.connect(k2.compose(lattice, W))` with
 * W = k2.linear_fst(words, words), and latticeW.get_tot_scores(True, True) / (True, False)
 * (utils/lattice_analysis.py:137-139,161-162: the numerators of the script's three ratios).  One more launch after the two
 * of wf_search_lattice walks the output lattice backwards over cells (record, words consumed): an arc with aux label 0 keeps
 * the count, one whose aux label is the next word advances it, any other aux label leaves the lattice.  aux_dev: device
 * int32, one aux label per arc of the FsaVec (0 = epsilon, -1 on final arcs); words / word_off: host, utterance i's word
 * ids (> 0) are words[word_off[i] .. word_off[i + 1]).  tot_words (fp32 [B], fp64 with WF_DOUBLE_ACC): log-semiring total
 * over the lattice's paths that spell the words; best_words (fp32 [B]): the tropical one; -inf when there is none.
 * Everything else as wf_search_lattice.  Rule: oracle.lattice_words (parity unpinned against k2, as above).            */
int wf_search_lattice_words_workspace_bytes(const void* graph_host, const int32_t* utt2fsa, const int32_t* seg, int32_t B,
                                            int32_t V, int32_t max_active_states, const int32_t* words,
                                            const int32_t* word_off, size_t* bytes);
int wf_search_lattice_words(const void* graph_host, const void* graph_dev, const int32_t* utt2fsa, const float* log_probs,
                            int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t V, const int32_t* seg, int32_t B,
                            int32_t Tmax, float search_beam, float output_beam, int32_t min_active_states,
                            int32_t max_active_states, uint32_t flags, const int32_t* aux_dev, const int32_t* words,
                            const int32_t* word_off, float* best_score, int32_t* best_arcs, void* tot_log, void* tot_words,
                            float* best_words, void* workspace, size_t workspace_bytes, void* stream);

/* ---- multi-GPU: the buffer of the one fused all-reduce per step (the reference gets these numbers
 * from Lightning's DDP reductions around graph.py:66-76).  out[0] = -(sum of the finite tot[i]), out[1] = n_frames,
 * out[2] = number of infeasible utterances (tot = -inf); device pointers, one small launch on `stream`.            */
int wf_loss_stats(const float* tot, int32_t B, float n_frames, float* out, void* stream);
/* The optional tail of that buffer: out[v] = sum over n < num_seqs, t < Tmax of grad[n, t, v], v < V --
 * the per-label gradient sums a training loop logs.  grad: the box wf_fb_logsum filled (element (n, t, v) at
 * n*strideB + t*strideT + v); out: device fp32 [V]; one streaming pass over the box on `stream`.                       */
int wf_label_grad_sums(const float* grad, int64_t strideB, int64_t strideT, int32_t num_seqs, int32_t Tmax, int32_t V,
                       float* out, void* stream);

/* ---- measurement aid (no counterpart in the reference) ------------------------------------------
 * wf_timing_enable(1): the following wf_fb_logsum / wf_fb_backward calls of this host thread record
 * HIP events on the caller's stream around each kernel; wf_timing_read() waits for them and
 * returns ms[6] = {emission gather, alpha/beta sweep, posterior/gradient, gradient zero-fill (runs on
 * a side stream concurrently with the sweep), row check + fallback} kernel times in ms (-1 = not run), and ms[5] = the
 * banded sweep launch's own duration measured INSIDE the kernel (s_memrealtime: latest workgroup end - earliest
 * workgroup start), which carries neither event nor profiler overhead.
 * Used by bench.py for the per-kernel roofline numbers.                                          */
int wf_timing_enable(int on);
int wf_timing_read(float* ms);

#ifdef __cplusplus
}
#endif
#endif /* WF_DWFST_H_ */
