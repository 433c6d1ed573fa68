/*
 * s2vt_hip.h — C ABI of libs2vt_hip.so: the MI355X (gfx950) implementation of the S2VT hot path.
 *
 * The reference (Kamino666/S2VT-video-caption) has no native boundary: the path sits behind the
 * Python class `S2VT(nn.Module)` (S2VTModel.py:10-110) and `MaskCriterion` (utils.py:6-26) and all
 * arithmetic is delegated to torch ops.  Each entry point below therefore names the reference call
 * site(s) whose torch op(s) it replaces; `S2VTModel.py` / `utils.py` at the root of this repository
 * are the reference-side bindings (ctypes) a maintainer would drop in — see INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, a positive hipError_t or -1 (bad argument) on failure;
 *    `s2vt_last_error()` gives the message (thread-local).  No C++ exception crosses the ABI.
 *  - all pointers are DEVICE pointers into memory owned by the caller (torch); fp32 unless noted;
 *    parameter tensors use the reference's state_dict layout.
 *  - every call is asynchronous and ordered on `stream` (a hipStream_t passed as void*); nothing is
 *    allocated: scratch comes from the caller-provided workspace (size from *_workspace_bytes).
 *  - one process drives one GPU (one process per GPU for data parallelism).  The library keeps
 *    process-wide state - GEMM arithmetic mode, pipeline block size, its internal side stream, the gradient-ready
 *    events of the last backward - so entry points may be called from any host thread (torch runs the backward
 *    on its autograd thread) but not from several threads at the same time.
 *  - dims: B batch, L = `length` frames (= padded caption length), F feat_dim, H dim_hid,
 *    E dim_embed, V vocab_size; T = 2L-1 LSTM steps.  Internal activations are time-major
 *    [t][b][:]; the external tensors keep the reference's batch-major layout.
 */
#ifndef S2VT_HIP_H
#define S2VT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2VT_ABI_VERSION 9

/* negative return codes (positive ones are hipError_t values) */
#define S2VT_ERR_ARG (-1)      /* bad argument */
#define S2VT_ERR_INDEX (-2)    /* a target id outside [0, vocab_size): the reference raises IndexError (S2VTModel.py:71) */
#define S2VT_ERR_TIMEOUT (-3)  /* a hand-off wait of a persistent recurrence kernel timed out */

typedef struct s2vt_dims {
    int32_t B, L, F, H, E, V;
} s2vt_dims;

/* The 13 parameter tensors of S2VT.__init__ (S2VTModel.py:19-28), reference names in comments. */
typedef struct s2vt_params {
    const float* vid_w_ih;  /* vid_rnn.weight_ih_l0  [4H, H]   */
    const float* vid_w_hh;  /* vid_rnn.weight_hh_l0  [4H, H]   */
    const float* vid_b_ih;  /* vid_rnn.bias_ih_l0    [4H]      */
    const float* vid_b_hh;  /* vid_rnn.bias_hh_l0    [4H]      */
    const float* word_w_ih; /* word_rnn.weight_ih_l0 [4H, E+H] columns = [embed | vid_out] (S2VTModel.py:75) */
    const float* word_w_hh; /* word_rnn.weight_hh_l0 [4H, H]   */
    const float* word_b_ih; /* word_rnn.bias_ih_l0   [4H]      */
    const float* word_b_hh; /* word_rnn.bias_hh_l0   [4H]      */
    const float* feat_w;    /* feat_linear.weight    [H, F]    */
    const float* feat_b;    /* feat_linear.bias      [H]       */
    const float* out_w;     /* out_linear.weight     [V, H]    */
    const float* out_b;     /* out_linear.bias       [V]       */
    const float* emb_w;     /* embedding.weight      [V, E]    */
} s2vt_params;

/* Gradient outputs, same order and shapes; every tensor is OVERWRITTEN (not accumulated). */
typedef struct s2vt_grads {
    float* vid_w_ih; float* vid_w_hh; float* vid_b_ih; float* vid_b_hh;
    float* word_w_ih; float* word_w_hh; float* word_b_ih; float* word_b_hh;
    float* feat_w; float* feat_b; float* out_w; float* out_b; float* emb_w;
} s2vt_grads;

int s2vt_abi_version(void);
const char* s2vt_last_error(void);

/* ---------------------------------------------------------------- whole-path entry points */

/* The batch size the whole-path drivers RUN at for a caller's batch B in the current arithmetic mode: in the plane modes (gemm mode
 * 3 / 1) a batch that is not a multiple of 64 and has at least `pad_min_batch` rows (option, default 33) is padded to the next
 * multiple inside the workspace (zero features, token 0; the pad rows' logits / ids are never handed out and their gradient
 * contributions are exact zeros), so that it takes the plane GEMMs, the persistent recurrence kernels and the decode cache.
 * Smaller batches - the reference's own defaults are 16 (train.py:27) and 10 (eval.py:27) - and gemm mode 0 run as they are on
 * the launch-per-timestep path (exact-fp32 MFMA tiles, gate GEMVs up to B = 4), which is FASTER at those sizes than 64 padded
 * rows (measured: profiles/round5_ragged_batches.txt). */
int32_t s2vt_padded_batch(int32_t B);

/* Bytes of workspace s2vt_train_forward/backward need (saved activations + scratch). */
size_t s2vt_train_workspace_bytes(const s2vt_dims* d);

/* S2VT.forward(feats, targets, mode='train')  (S2VTModel.py:48-54, 63-81; called at train.py:120,142).
 *   feats   [B, L, F]; targets int64 [B, L-1] with row stride targets_ld (= caption[:, :-1]);
 *   logits  [B, L-1, V] out.  The workspace then holds what s2vt_train_backward needs. */
int s2vt_train_forward(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                       int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* The same with out_dropout > 0 (S2VTModel.py:25,79: `res = self.out_drop(res)` between word_rnn and out_linear in
 * training mode).  out_mask: [(L-1)*B, H] TIME-MAJOR (row j*B + b = caption step j of sample b), entries 0 or 1/(1-p),
 * drawn by the caller (torch's dropout on a ones tensor: the reference's RNG stream); the backward must get the same mask. */
int s2vt_train_forward_dropout(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                               int64_t targets_ld, const float* out_mask, float* logits, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Device-side errors are asynchronous: a target id outside [0, V) (the reference's nn.Embedding raises IndexError,
 * S2VTModel.py:71) or a timed-out hand-off of the persistent recurrence raise a flag that every s2vt_train_forward
 * copies to the host at its end.  The NEXT s2vt_train_forward / s2vt_train_backward that finds the copy complete returns
 * S2VT_ERR_INDEX / S2VT_ERR_TIMEOUT for it; a caller that has just synchronised the stream (loss.item()) calls this with
 * wait = 1 and gets the error of the forward it has just run.  0 = no error (or, with wait = 0, not known yet). */
int s2vt_check_async_error(int32_t wait);

/* One depth of the batched beam search (beam.py; S2VTModel.py:205-223 for every expandable node of every sample at once):
 *   1. one zero-input vid_rnn step for the whole batch: (vid_h_in, vid_c_in) [B,H] -> (vid_h_out, vid_c_out);
 *   2. for the R expandable rows r (sample row_b[r], parent state row row_state[r] of the previous depth's table,
 *      last token tok[r]): word_rnn step on [Emb[tok] | vid_h_out[row_b]] from (word_h_in, word_c_in)[row_state] ->
 *      the new state table (word_h_out, word_c_out) [R,H];
 *   3. out_linear, log_softmax and the 20 most probable tokens of every row in ASCENDING token order (the order the
 *      reference pushes them): top_ix [R,20] int32, top_lp [R,20] fp32 log-probs.
 * d->B = batch size; R may be 0 (only the vid step runs).  Workspace: s2vt_beam_workspace_bytes(d, max R). */
size_t s2vt_beam_workspace_bytes(const s2vt_dims* d, int32_t max_rows);
int s2vt_beam_step(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                   const int32_t* tok, const float* vid_h_in, const float* vid_c_in, float* vid_h_out, float* vid_c_out,
                   const float* word_h_in, const float* word_c_in, float* word_h_out, float* word_c_out, int32_t* top_ix,
                   float* top_lp, void* workspace, size_t workspace_bytes, void* stream);
/* The same depth with the weight-derived images of a greedy decode of the SAME weights (`cache` as filled by
 * s2vt_greedy_decode_cached: plane images of W_v / W_o, the per-token gate table): vid_out's half of the gate input once per sample,
 * the embedded word from the table, out_linear in split precision on the bf16 matrix cores.  Results as s2vt_beam_step to fp32
 * rounding (the fixtures' beam ids are unchanged). */
int s2vt_beam_step_cached(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                          const int32_t* tok, const float* vid_h_in, const float* vid_c_in, float* vid_h_out, float* vid_c_out,
                          const float* word_h_in, const float* word_c_in, float* word_h_out, float* word_c_out, int32_t* top_ix,
                          float* top_lp, void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, void* stream);

/* The reference's per-sample priority queues of the beam search (S2VTModel.py:186-238) ON THE DEVICE, all samples of the batch in one
 * launch per depth: heapq's own sift-down / sift-up replayed per sample (one lane each, heap in LDS), so that the pop order - also
 * among EQUAL scores, where it depends on the binary heap's layout - is the reference's.  `state`: s2vt_beam_queue_bytes() of
 * caller-owned device memory that lives for one search.  Rows of the batched step are fixed: r = b * beam_width + slot.
 *   depth = 1          : initialise, pop the roots; writes row_b / row_state / row_tok for the first s2vt_beam_step
 *   depth = 2..max     : push the children of the depth just stepped (top_ix / top_lp [B*beam_width][20] of s2vt_beam_step), record
 *                        each sample's best entry, freeze samples whose queue holds <= beam_width entries (:227-228), pop the next
 *                        beam (queue cleared, finished entries re-inserted: :190-202) and write the rows of the next step
 *   depth = 0          : the final push (nothing is popped any more)
 * A frozen sample's slots carry token 0 / state row 0 (their step results are ignored).  The int32 at byte 0 of `state` counts the
 * frozen samples (== B: the reference's loop would stop; further calls change nothing).
 * s2vt_beam_queue_result: back-trace of every sample's best entry (:231-236) -> out_tokens[b][0..out_len[b]) = <sos> .. last word. */
size_t s2vt_beam_queue_bytes(int32_t B, int32_t beam_width, int32_t max_depth);
int s2vt_beam_queue_step(int32_t B, int32_t beam_width, int32_t max_depth, int32_t sos_ix, int32_t eos_ix, int32_t depth, void* state,
                         size_t state_bytes, const int32_t* top_ix, const float* top_lp, int32_t* row_b, int32_t* row_state,
                         int32_t* row_tok, void* stream);
int s2vt_beam_queue_result(int32_t B, int32_t beam_width, int32_t max_depth, void* state, size_t state_bytes, int32_t* out_tokens,
                           int32_t out_cap, int32_t* out_len, void* stream);

/* Data-parallel overlap (no reference counterpart: the reference is single-device).  After s2vt_train_backward has
 * RETURNED (all of its work is enqueued), make `stream` wait until a group of that call's parameter gradients is
 * final, so that their all-reduce can run under the rest of the backward:
 *   group 0 = out_linear (weight, bias): final ~1 ms into the backward;
 *   group 1 = word_rnn (4 tensors) + embedding: final before the vid_rnn / feat_linear weight-gradient GEMMs.
 * The remaining gradients (vid_rnn, feat_linear) are final when the backward's own stream is. */
int s2vt_backward_wait_grads(int32_t group, void* stream);
/* Order check of that overlap for the last s2vt_train_backward of the plane drivers: how many persistent BPTT launches it
 * enqueued, and how many of them were already enqueued when gradient group 0's event was recorded for the last time.  The two
 * must be equal: a persistent launch needs all of its workgroups resident, so no collective may be released beside one (the
 * out_linear all-reduce then overlaps the weight-gradient GEMMs behind the recurrence instead). */
int s2vt_backward_order(int32_t* persistent_bptt_launches, int32_t* group0_recorded_after);

/* Autograd of the above (loss.backward(), train.py:124) given dlogits [B, L-1, V] (contiguous).
 * dfeats [B, L, F] may be NULL (nothing reads it in the reference: SURVEY.md §3.1 note). */
int s2vt_train_backward(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                        const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream);
int s2vt_train_backward_dropout(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                                const float* out_mask, const s2vt_grads* g, float* dfeats, void* workspace,
                                size_t workspace_bytes, void* stream);

size_t s2vt_decode_workspace_bytes(const s2vt_dims* d);

/* S2VT.forward(feats, mode='test')  (S2VTModel.py:82-110; called at eval.py:52): greedy decode,
 * ids int64 [B, L-1] out, never stops at <eos>, lowest index wins argmax ties. */
int s2vt_greedy_decode(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The same decode with the WEIGHT-derived images kept across calls (eval.py:48-52 calls model(feats, mode='test') once per batch
 * with fixed weights): plane images of W_f / W_ih1 / W_v / W_o and the per-token gate-input table Emb·W_e^T live in `cache`
 * (s2vt_decode_cache_bytes; caller-owned device memory that outlives the call).  cache_valid == 0: this call fills the cache;
 * != 0: the caller vouches that the parameters, dims and library modes are those of the call that filled it (the Python binding
 * keys on every parameter's data pointer and version counter) and the call reuses it - 0.65 ms less of a 9.2-ms decode at
 * B = 128.  Calls that share a cache must be ordered by the stream.  The per-call workspace is s2vt_decode_workspace_bytes as
 * for s2vt_greedy_decode (whose weight images live behind the per-call part of that workspace and are rebuilt every call). */
size_t s2vt_decode_cache_bytes(const s2vt_dims* d);
/* 1 if s2vt_greedy_decode_cached reads (and, with cache_valid == 0, fills) the cache for this batch size in the current mode;
 * 0 for the batches that decode on the launch-per-timestep path (gemm mode 0, or fewer than 24 clips: measured faster there than
 * 64 padded rows on the plane path) - the caller must then not mark its cache as filled. */
int32_t s2vt_decode_uses_cache(const s2vt_dims* d);
int s2vt_greedy_decode_cached(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, int32_t cache_valid,
                              void* stream);

/* optimizer.step() of the reference's loop (train.py:89-93,126: torch.optim.Adam, default betas / eps, no weight decay, no
 * amsgrad) as ONE launch over flat fp32 buffers of n elements - every parameter, its gradient and its two moments at the same
 * offsets (s2vt-video-caption_amd/optim.py lays a model's parameters out that way).  torch's arithmetic operation for operation:
 * m += (g - m)(1 - beta1); v = beta2 v + (1 - beta2) g g; p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).
 * step counts from 1; the hyper-parameters are doubles because torch forms 1 - beta and the bias corrections from the Python doubles
 * before rounding to fp32 once. */
int s2vt_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                   double eps, int64_t step, void* stream);

/* MaskCriterion's inner nn.CrossEntropyLoss() (utils.py:11,22): mean CE of logits [B, L-1, V] against
 * target[:, 1:] (target int64 [B, L], row stride target_ld).  lse [B*(L-1)] and rowloss [B*(L-1)] are
 * caller-provided scratch (lse is consumed by the backward); loss_out is one device float.
 * A target id outside [0, V) (nn.CrossEntropyLoss raises IndexError; ignore_index is not used by the reference) is
 * detected on the device and reported as S2VT_ERR_INDEX by s2vt_check_async_error / the next call, like the
 * embedding's check in s2vt_train_forward; the row's loss is then computed against the clamped id. */
int s2vt_mean_ce_forward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target,
                         int64_t target_ld, float* lse, float* rowloss, float* loss_out, void* stream);
/* dlogits = (softmax(logits) - onehot(target)) * gout[0] / (B*(L-1)); gout is a device scalar. */
int s2vt_mean_ce_backward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target,
                          int64_t target_ld, const float* lse, const float* gout, float* dlogits, void* stream);

/* MaskCriterion.forward as a whole (utils.py:13-26; called at train.py:122,143): the mean CE above, then
 * `loss = sum(mean_ce * w) / sum(w)` with w = mask[:, 1:] (mask fp32 [B, L], row stride mask_ld), product by product as the
 * reference evaluates it (NaN for an all-zero mask).  out3 = {loss, mean_ce, sum(w)} (three device floats). */
int s2vt_mask_criterion_forward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target, int64_t target_ld,
                                const float* mask, int64_t mask_ld, float* lse, float* rowloss, float* out3, void* stream);
/* autograd of those lines: g_ce[0] = sum_i (gout[0] / out3[2]) * w_i, the `gout` of s2vt_mean_ce_backward / _fused. */
int s2vt_mask_criterion_backward(int32_t B, int32_t Lm1, const float* mask, int64_t mask_ld, const float* out3, const float* gout,
                                 float* g_ce, void* stream);

/* The same gradient handed to s2vt_train_backward WITHOUT an fp32 dlogits tensor (utils.py:22 under loss.backward(), train.py:124):
 * evaluates (softmax(logits) - onehot(target)) * gout[0] / (B*(L-1)) inside the plane-split pass of the TRAIN workspace the
 * logits came from - the operand planes and bias-gradient partial sums the backward's first kernel would otherwise produce from
 * dlogits, bit for bit - and marks the workspace so that the following s2vt_train_backward may be called with dlogits == NULL.
 * Plane-driver workspaces only (B % 64 == 0, gemm mode 1 or 3); lse from s2vt_mean_ce_forward. */
int s2vt_mean_ce_backward_fused(const s2vt_dims* d, const float* logits, const int64_t* target, int64_t target_ld, const float* lse,
                                const float* gout, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- per-op entry points
 * (the pieces the whole-path drivers are built from; exported for tests, profiling and reuse) */

/* C[M,N] (+)= op(A)·op(B) (+ bias[N]) in fp32 on the matrix cores.
 *   a_kmajor: A stored [M][K] (else [K][M]);  b_kmajor: B stored [N][K] (else [K][N]).
 * Stands for torch's addmm/mm under nn.Linear / nn.LSTM input projections (S2VTModel.py:54,67,77,80). */
int s2vt_gemm_f32(int32_t a_kmajor, int32_t b_kmajor, int32_t M, int32_t N, int32_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t accumulate,
                  void* stream);

/* Same contraction with caller-provided scratch (`ws`, ws_floats floats) that lets small grids split K into
 * slices combined in a fixed order (deterministic); this is the form the whole-path drivers use. */
int s2vt_gemm_f32_splitk(int32_t a_kmajor, int32_t b_kmajor, int32_t M, int32_t N, int32_t K, const float* A,
                         int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                         int32_t accumulate, float* ws, size_t ws_floats, void* stream);

/* Split-precision path (split.hip, gemm_x3.hip, gemm_bf16.hip).  An fp32 matrix [rows][cols] is rewritten as `nplanes`
 * (1 or 3) bf16 planes p0+p1+p2 = x (24 mantissa bits) of a k-major GEMM operand, zero-filled in k beyond the data
 * (kpad % 64 == 0, ldo >= nplanes*kpad):
 *   nplanes = 1: plain rows, element (r, k) at r*ldo + k;
 *   nplanes = 3: BLOCKED layout of gemm_x3.hip - per 64-row block and 16-wide k chunk one 6-KB record of six 1-KB
 *     pieces (plane, k half) of 64 rows x 8 k:  (r/64)*(64*ldo) + (k/16)*3072 + (pl*2 + (k%16)/8)*512 + (r%64)*8 + k%8
 *     (bf16 units); the buffer must hold whole 64-row blocks: ceil(operand rows / 64) * 64 * ldo elements.
 * transpose = 0: operand rows = input rows, k = input columns; transpose = 1: operand rows = input columns
 * (out_rows_pad >= cols), k = input rows (kpad >= rows). */
int s2vt_split_planes(int32_t nplanes, int32_t transpose, const float* in, int64_t ld, int32_t rows, int32_t cols,
                      uint16_t* out, int64_t ldo, int32_t kpad, int32_t out_rows_pad, void* stream);
/* C[M,N] (+)= A[M,K]·B[N,K]^T (+bias) from bf16 planes (layouts above) on the bf16 matrix cores, fp32 accumulate;
 * K = kpad.  nplanes = 3 sums the six leading plane products (fp32-equivalent, ~2^-23 relative; LDS-DMA kernel,
 * 256x256 tiles); nplanes = 1 is a plain bf16 GEMM.  ws/ws_floats: optional scratch for deterministic split-K. */
int s2vt_gemm_bf16_nt(int32_t nplanes, int32_t M, int32_t N, int32_t K, const uint16_t* A, int64_t lda, const uint16_t* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t accumulate, float* ws,
                      size_t ws_floats, void* stream);
/* The weight-gradient form, C[M,N] (+)= X_A^T · X_B, from the ROW images of its operands: A = blocked 3-plane image (transpose = 0
 * above) of X_A [K rows][M columns], B = that of X_B [K rows][N columns], both starting at a 64-row block; K (a multiple of 64)
 * image rows are contracted.  The kernel reads its fragments transposed (ds_read_b64_tr_b16), so dG / dlogits / h are split ONCE,
 * as rows, and serve both the data-gradient GEMMs (as A[M,K]) and the weight-gradient GEMMs (autograd of S2VTModel.py:54,67,77,80).
 * nplanes = 1: the same from plain bf16 row images (rows of at least M / N columns). */
int s2vt_gemm_bf16_tt(int32_t nplanes, int32_t M, int32_t N, int32_t K, const uint16_t* A, int64_t lda, const uint16_t* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t accumulate, float* ws, size_t ws_floats,
                      void* stream);

/* Test / calibration hook of s2vt_gemm_bf16_nt: force the workgroup tile height (nplanes = 1: 128, 192, 256 or 320 rows; nplanes = 3:
 * 128, 192 or 256; 0 = the launcher's time model picks the height whose tile count fills whole rounds of the compute units) and
 * the split-K factor (0 = model).  Results do not depend on the tile height beyond the order of the fp32 sums along k, which it does not change
 * either (split-K does). */
int s2vt_gemm_tune(int32_t nplanes, int32_t tile_rows, int32_t nsplit);

/* feat_linear (S2VTModel.py:54): x1[l*B+b, :] = feats[b, l, :]·W^T + bias  (time-major output [L*B, H]). */
int s2vt_feat_proj_fwd(const s2vt_dims* d, const float* feats, const float* w, const float* bias, float* x1,
                       void* stream);
/* its weight/bias gradients from dx1 [L*B, H] (time-major); dfeats optional. colsum_ws: >= s2vt_colsum_ws_floats(L*B, H) floats. */
int s2vt_feat_proj_bwd(const s2vt_dims* d, const float* feats, const float* w, const float* dx1, float* dw,
                       float* dbias, float* dfeats, float* colsum_ws, void* stream);
size_t s2vt_colsum_ws_floats(int64_t rows, int32_t cols);

/* One LSTM timestep, gates i,f,g,o (what nn.LSTM runs per step: S2VTModel.py:67,77,86,93,103):
 *   G = gx (or bias if gx == NULL) + h_prev·W_hh^T ; cell update; writes h_out, c_out and, if stash != NULL,
 *   the activated gates [B,4H] for the backward.  h_prev/c_prev NULL = zero state.  Row stride of every
 *   [B, *] operand = its natural width. */
int s2vt_lstm_step_fwd(int32_t B, int32_t H, const float* gx, const float* bias, const float* w_hh,
                       const float* h_prev, const float* c_prev, float* h_out, float* c_out, float* stash,
                       void* stream);
/* One DECODE step of word_rnn (S2VTModel.py:100-103: embedding of the previous word, concatenated in front of vid_rnn's output,
 * one nn.LSTM step): G = gx + h_prev·W_hh^T + Emb[token(b)]·W_e^T with gx [B,4H] = the vid_out half of the gate input + both
 * biases, emb [V,E], w_e = the first E columns of word_rnn.weight_ih (row stride ldw_e).  token(b) = tok[b] (int32), else the
 * packed argmax word of the previous step (s2vt_decode_step_argmax: 0xFFFFFFFF - low 32 bits), else tok_const for every row.
 * A token id outside [0, V) - nn.Embedding raises IndexError there - is read as token 0 and reported as S2VT_ERR_INDEX by
 * s2vt_check_async_error / the next call: it never addresses memory outside the table. */
int s2vt_lstm_step_fwd_token(int32_t B, int32_t H, int32_t E, int32_t V, const float* gx, const float* w_hh, const float* h_prev,
                             const float* c_prev, const float* emb, const float* w_e, int64_t ldw_e, const int32_t* tok,
                             const unsigned long long* tok_packed, int32_t tok_const, float* h_out, float* c_out, void* stream);
/* BPTT of one step: dh = dh_out + dg_next·W_hh (w_hh_t = W_hh^T [H,4H]); dc (in/out, [B,H]) carries dL/dc;
 * dg [B,4H] out (may alias stash). dg_next NULL at the last step; dc_is_zero = 1 there. */
int s2vt_lstm_step_bwd(int32_t B, int32_t H, const float* dg_next, const float* w_hh_t, const float* dh_out,
                       const float* stash, const float* c, const float* c_prev, float* dc, int32_t dc_is_zero,
                       float* dg, void* stream);

/* A whole layer: T steps from zero state. gx [n_gx*B, 4H] time-major covers steps 0..n_gx-1 (x-part + both
 * biases); later steps see `bias` only (the zero-padded input of S2VTModel.py:64-65).
 * Outputs time-major h_all/c_all [T*B, H], stash [T*B, 4H] (may alias gx). */
int s2vt_lstm_seq_fwd(int32_t T, int32_t B, int32_t H, const float* gx, int32_t n_gx, const float* bias,
                      const float* w_hh, float* h_all, float* c_all, float* stash, void* stream);
/* BPTT over the layer: dh_out [T*B, H] time-major with dh_out rows for steps < dh_first absent (treated as 0);
 * stash is overwritten by dG [T*B,4H]; w_hh_t and dc [B,H] are scratch provided by the caller. */
int s2vt_lstm_seq_bwd(int32_t T, int32_t B, int32_t H, const float* w_hh, const float* dh_out, int32_t dh_first,
                      const float* c_all, float* stash_dg, float* w_hh_t, float* dc, void* stream);

/* Config-3 arithmetic of one LSTM layer forward (nn.LSTM at S2VTModel.py:67/:77 with bf16 operands, fp32 accumulate,
 * fp32 cell state): gx_stash [T*B,4H] gate input in (steps < n_gx; bias for the rest), activated gates out; h_all,
 * c_all [T*B,H] out.  persistent = 0: one launch per timestep; 1: one persistent launch per `block` timesteps
 * (0 = all T) with the W_hh slice of each compute unit resident in registers.  workspace[0] (int32) is set to 1 if a
 * hand-off wait of the persistent kernel timed out. */
size_t s2vt_lstm_seq_bf16_workspace_bytes(int32_t T, int32_t B, int32_t H);
int s2vt_lstm_seq_fwd_bf16(int32_t T, int32_t B, int32_t H, float* gx_stash, int32_t n_gx, const float* bias,
                           const float* w_hh, float* h_all, float* c_all, void* workspace, size_t workspace_bytes,
                           int32_t persistent, int32_t block, void* stream);
/* Two independent layers of one shape with every block of timesteps of BOTH in one persistent launch (the schedule
 * of the whole-path driver: vid_rnn block k+1 next to word_rnn block k); workspace = 2 x the single-layer size. */
int s2vt_lstm_seq_fwd_bf16_pair(int32_t T, int32_t B, int32_t H, float* gx_stash0, float* gx_stash1, int32_t n_gx,
                                const float* bias0, const float* bias1, const float* w_hh0, const float* w_hh1,
                                float* h_all0, float* h_all1, float* c_all0, float* c_all1, void* workspace,
                                size_t workspace_bytes, int32_t block, void* stream);
/* Config-3 arithmetic of one layer's BPTT (autograd of nn.LSTM, train.py:124, with bf16 operands dG / W_hh^T and fp32
 * accumulation, cell-state gradient and outputs): stash_dg [T*B,4H] activated gates in, fp32 dG out (in place); dh_out =
 * gradient arriving from above for steps >= dh_first ([(T-dh_first)*B, H], nullable); c_all [T*B,H] from the forward.
 * persistent / block as in s2vt_lstm_seq_fwd_bf16; the _pair form runs two layers side by side in one launch. */
size_t s2vt_lstm_seq_bwd_bf16_workspace_bytes(int32_t T, int32_t B, int32_t H);
int s2vt_lstm_seq_bwd_bf16(int32_t T, int32_t B, int32_t H, const float* w_hh, const float* dh_out, int32_t dh_first,
                           const float* c_all, float* stash_dg, void* workspace, size_t workspace_bytes,
                           int32_t persistent, int32_t block, void* stream);
int s2vt_lstm_seq_bwd_bf16_pair(int32_t T, int32_t B, int32_t H, const float* w_hh0, const float* w_hh1, const float* dh_out0,
                                const float* dh_out1, int32_t dh_first, const float* c_all0, const float* c_all1,
                                float* stash_dg0, float* stash_dg1, void* workspace, size_t workspace_bytes, int32_t block,
                                void* stream);
/* Recurrence schedule inside the whole-path train drivers (option "persist"): 0 = one launch per timestep everywhere (a card
 * shared with another process); 1 (default) = persistent kernels where the shape allows: forward and BPTT of the bf16
 * configuration (gemm mode 1, lstm_persist.hip), forward of the fp32-equivalent configuration (gemm mode 3, the split-precision
 * kernel of lstm_persist_x3.hip; option "persist_x3_fwd") and, with option "persist_x3_bwd", its BPTT.  Negative: query.
 * Returns the previous mode. */
int s2vt_set_recurrence_mode(int32_t mode);
/* Which recurrence kernels s2vt_train_forward / s2vt_train_backward would run for (B, H) in the current modes:
 * *fwd, *bwd = 0 one launch per timestep, 1 persistent bf16, 3 persistent split precision. */
int s2vt_recurrence_plan(int32_t B, int32_t H, int32_t* fwd, int32_t* bwd);

/* Persistent forward recurrence in split precision (lstm_persist_x3.hip): the same computation as s2vt_lstm_seq_fwd with ONE launch
 * per `block` timesteps (0 = all T) and each compute unit's slice of W_hh resident in registers: both operands of
 * h_{t-1} . W_hh^T as three bf16 planes, six plane products on the bf16 matrix cores, fp32 accumulate - fp32-equivalent like gemm
 * mode 3; one workgroup per compute unit keeps its W_hh planes in 384 registers per lane.  H <= 1024, B % 32 == 0.  Layer 1
 * pointers may all be null (one layer); otherwise both layers share every launch.  gx_stash: gate input in, activated gates
 * out (in place).  workspace from s2vt_lstm_seq_x3_workspace_bytes (0 = shape not supported on this device; word 0 of the
 * workspace: hand-off time-out flag).  Whole-path use: the forward recurrences of s2vt_train_forward in gemm mode 3
 * (S2VTModel.py:67,77; option "persist_x3_fwd"). */
size_t s2vt_lstm_seq_x3_workspace_bytes(int32_t T, int32_t B, int32_t H);
int s2vt_lstm_seq_fwd_x3_persist(int32_t T, int32_t B, int32_t H, float* gx_stash0, float* gx_stash1, int32_t n_gx,
                              const float* bias0, const float* bias1, const float* w_hh0, const float* w_hh1, float* h_all0,
                              float* h_all1, float* c_all0, float* c_all1, int32_t block, void* workspace,
                              size_t workspace_bytes, void* stream);
/* BPTT in split precision (lstm_persist_x3.hip), the counterpart of s2vt_lstm_seq_fwd_x3_persist: the contraction
 * dG_{t+1} . W_hh is split over the gate columns - every workgroup multiplies the dG tile it has just computed with its 64 rows
 * of W_hh (planes resident in registers) for all H outputs, scatters the fp32 partial sums per consumer and gathers the ones
 * addressed to it (fixed summation order).  Same arithmetic contract as s2vt_lstm_seq_bwd (fp32-equivalent); stash_dg: activated
 * gates in, dG out (in place).  Layer 1 pointers may all be null.  block: timesteps per launch (0 = all T).
 * Whole-path use: option "persist_x3_bwd" routes the BPTT of s2vt_train_backward (gemm mode 3) through it (one stream, both
 * layers per launch). */
size_t s2vt_lstm_seq_bwd_x3_workspace_bytes(int32_t T, int32_t B, int32_t H, int32_t block);
int s2vt_lstm_seq_bwd_x3_persist(int32_t T, int32_t B, int32_t H, const float* w_hh0, const float* w_hh1, const float* dh_out0,
                                 const float* dh_out1, int32_t dh_first, const float* c_all0, const float* c_all1, float* stash_dg0,
                                 float* stash_dg1, int32_t block, void* workspace, size_t workspace_bytes, void* stream);
/* One greedy decode step's out_linear + argmax (S2VTModel.py:95-96,105-106): packed[b] (zeroed by the caller)
 * receives max over v of (ordered(logit) << 32 | (0xFFFFFFFF - v)); token = 0xFFFFFFFF - low 32 bits. */
int s2vt_decode_step_argmax(int32_t B, int32_t H, int32_t V, const float* h, const float* w_out, const float* b_out,
                            unsigned long long* packed, void* stream);

/* The same decode step on the bf16 matrix cores with fp32-equivalent arithmetic (three bf16 planes per operand, six plane
 * products: csrc/argmax_x3.hip) - what s2vt_greedy_decode runs per step when the batch is a multiple of 64.  h and w_out are
 * split into plane images in `workspace` (s2vt_decode_step_argmax_x3_workspace_bytes) by this call. */
size_t s2vt_decode_step_argmax_x3_workspace_bytes(int32_t B, int32_t H, int32_t V);
int s2vt_decode_step_argmax_x3(int32_t B, int32_t H, int32_t V, const float* h, const float* w_out, const float* b_out,
                               unsigned long long* packed, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- run-time options
 * ONE table holds every switch of the library (csrc/options.hip).  The first read of an option takes S2VT_<NAME> (upper case)
 * from the environment when it is set; s2vt_set_option changes it afterwards and returns the previous value (a negative value
 * only queries; INT32_MIN: unknown name).  The typed setters below are views of the same table.
 *   gemm_mode       3 | 1 | 0   arithmetic of the batched GEMMs (s2vt_set_gemm_mode)
 *   persist         1 | 0       persistent recurrence kernels (s2vt_set_recurrence_mode)
 *   persist_x3_fwd  1 | 0       split-precision persistent forward of gemm mode 3
 *   persist_x3_bwd  2 | 1 | 0   split-precision persistent BPTT of gemm mode 3 (layers per launch: option bptt_solo): 1 wherever the
 *                               shape is supported, 2 only where a workgroup carries one 32-row chain (B = 64 at H = 1000), 0 never
 *   pipe_block      32          timesteps per pipeline block (s2vt_set_pipeline_block)
 *   graph           0 | 1       hipGraph replay of the train launch sequences (s2vt_set_graph_mode)
 *   decode_fused    1 | 0       schedule of the greedy decode's token steps (s2vt_set_decode_schedule)
 *   cu_reserve      0..128      the persistent GEMMs of s2vt_train_backward plan their grids for this many compute units fewer FROM
 *                               THE RELEASE OF GRADIENT GROUP 0 (s2vt_backward_wait_grads) to the end of the backward - the span in
 *                               which the communication kernels of a data-parallel run hold compute units; everything before
 *                               it (forward, BPTT and the GEMMs beside it) plans for the whole device
 *   bptt_units      0 | 16 | 32 hidden units per workgroup of the persistent bf16 BPTT (0: 32 where two layers fit the device)
 *   corun           3 (0..5)    persistent split-precision schedule (B = 64): this many TENTHS of a GEMM that nothing waits for (dW_o's
 *                               k range) run beside EACH one-layer first / last stage of the BPTT, planned for the compute units
 *                               that stage leaves idle (0: every GEMM alone on the device)
 *   bptt_solo       1 | 0       (with corun > 0, B = 64) every persistent BPTT launch carries ONE layer - word_rnn's blocks, then vid_rnn's -
 *                               on half of the compute units, with dW_o, the dh1 GEMMs and word_rnn's weight gradients on the other half;
 *                               0: two layers per launch, GEMM parts only beside the first / last (one-layer) launch
 *   pad_min_batch   33 (1..64)  ragged batches (B % 64 != 0) of at least this many rows - a greedy decode: three quarters of it - are
 *                               padded to a multiple of 64 inside the workspace (s2vt_padded_batch); smaller ones run as they
 *                               are, launches per timestep (faster there: profiles/round5_ragged_batches.txt)
 *   gemv            1 | 2 | 0   the launch-per-timestep forward step (s2vt_lstm_step_fwd and the drivers built on it) as gate
 *                               GEMVs - h staged in LDS, weight rows streamed to registers, wavefront shuffle reductions
 *                               (csrc/lstm_gemv.hip) - instead of the 16-row fp32-MFMA tile kernel: 1 = at B <= 4 (where it
 *                               measured faster), 2 = at every B <= 8, 0 = never
 * Do not change gemm_mode / persist / pipe_block between a forward and its backward (the backward refuses). */
int32_t s2vt_set_option(const char* name, int32_t value);
int32_t s2vt_option_count(void);
const char* s2vt_option_name(int32_t index);

/* Arithmetic of the batched GEMMs inside the whole-path train drivers: 0 = fp32-input MFMA (exact fp32 products),
 * 3 = split precision (3 bf16 planes per operand, six plane products on the bf16 matrix cores: fp32-equivalent to
 * ~2^-23 relative; default when B % 64 == 0, env S2VT_GEMM_MODE).  Returns the previous mode; a negative argument only queries.  Call it between a
 * forward and its backward only if you also re-query the workspace size. */
int s2vt_set_gemm_mode(int32_t mode);

/* Layer pipelining of the whole-path drivers: the two LSTM layers run as a software pipeline on two streams in
 * blocks of `steps` timesteps (default 32, or env S2VT_PIPE_BLOCK; the persistent bf16 schedule evens the default out over
 * the L frames - 27 at L = 80 - because its launches pair a block of one layer with a block of the other); 0 runs everything
 * on the caller's stream (kernels then never overlap: used to time kernels in isolation).  Returns the previous value. */
int s2vt_set_pipeline_block(int32_t steps);
/* Schedule of the token-dependent decode steps of s2vt_greedy_decode[_cached] on the plane path (the loop of
 * /root/reference/S2VTModel.py:98-107): 1 (default; env S2VT_DECODE_FUSED) = per step ONE launch that computes out_linear +
 * argmax of step t and, as extra row blocks, h_t W_hh^T of step t+1 (which does not depend on the token), then a one-thread-
 * per-cell launch that adds the token's gate-table row and updates the cell; 0 = a step kernel and an argmax kernel per step,
 * the two batch halves as independent chains on two streams.  Same ids either way.  Returns the previous value; any other
 * argument only queries. */
int s2vt_set_decode_schedule(int32_t schedule);
/* The encode phase of a decode alone - /root/reference/S2VTModel.py:56-60 (mode='beam_search': vid_rnn over the L frames, word_rnn
 * over the padded vid_out) - on the plane path, with the weight-derived images in the caller's cache as in
 * s2vt_greedy_decode_cached (cache_valid == 0: filled here, every image a decode or a beam search of these weights reads).
 * workspace: s2vt_decode_workspace_bytes(d).  Out: vid_h, vid_c, word_h, word_c [B, H], the states after frame L-1 that
 * S2VT.beam_search (:149) starts from.  Shapes the persistent split-precision recurrence does not take (B % 64, H > 1024,
 * fp32-MFMA mode) return S2VT_ERR_ARG. */
int s2vt_decode_encode_cached(const s2vt_dims* d, const s2vt_params* p, const float* feats, void* workspace, size_t workspace_bytes,
                              void* cache, size_t cache_bytes, int32_t cache_valid, float* vid_h, float* vid_c, float* word_h,
                              float* word_c, float* gx_dec, int32_t depth, void* stream);
/* gx_dec (nullable) [depth][B][4H], depth <= L-1: vid_rnn's decode-phase steps take no input and see no token
 * (/root/reference/S2VTModel.py:208-210 runs one per depth inside the search loop), so the first `depth` of them run here in one
 * persistent launch and their half of word_rnn's gate input (h1_t W_v^T + both biases) comes out of one GEMM.
 * s2vt_beam_step_gx is s2vt_beam_step_cached for a depth whose slice gx_dec[depth - 1] replaces the vid_rnn state arguments. */
int s2vt_beam_step_gx(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                      const int32_t* tok, const float* gx_vid, const float* word_h_in, const float* word_c_in, float* word_h_out,
                      float* word_c_out, int32_t* top_ix, float* top_lp, void* workspace, size_t workspace_bytes, void* cache,
                      size_t cache_bytes, void* stream);
/* 1 if the internal side stream was verified to execute concurrently with the caller's stream (it is chosen by a
 * one-time calibration at the first pipelined call: HIP may map two streams onto one hardware queue), 0 if no
 * candidate overlapped (the drivers still run, serially), -1 before the first pipelined call. */
int s2vt_pipeline_overlaps(void);

/* Launch-sequence capture: with on = 1 (or env S2VT_GRAPH=1) the ~350 launches of an s2vt_train_forward / s2vt_train_backward
 * call of the plane drivers (B % 64 == 0) are captured into a hipGraph the second time the same argument set - every pointer,
 * dims, modes, stream - is seen, and replayed with one hipGraphLaunch from then on (at most 8 executables are cached).  Results
 * are those of the eager sequence bit for bit; s2vt_backward_wait_grads(0 / 1) then completes with the whole backward (the
 * gradient all-reduce follows it instead of overlapping it).  Off by default; not active while s2vt_prof_enable(1).  Returns the previous
 * setting; a negative argument only queries.  s2vt_graph_stats: graphs captured / launches replayed so far. */
int s2vt_set_graph_mode(int32_t on);
int s2vt_graph_stats(int64_t* captures, int64_t* replays);

/* Test support (tests/test_gpu_kernels.py: co-residency of the persistent recurrence): launches `workgroups` one-wave
 * workgroups that each hold `lds_bytes` (<= 160 KB) of LDS and spin for `microseconds` (<= 5 s) - a stand-in for a foreign
 * kernel (an RCCL all-reduce on a communication stream, another tenant of the GPU) sitting on the compute units. */
int s2vt_test_occupy_cus(int32_t workgroups, int32_t lds_bytes, int64_t microseconds, void* stream);

/* ---------------------------------------------------------------- live kernel timing (bench.py)
 * When enabled, launch sites bracket kernels of one kind with hipEvents on the launch stream.
 * kinds: 0 gemm, 1 lstm_step_fwd (whole sequence loop), 2 lstm_step_bwd (whole sequence loop),
 *        3 ce, 4 logits_argmax, 5 gemm launches planned for PART of the compute units (option corun: beside a one-layer
 *        persistent launch; kind 0 counts only the GEMMs that have the device to themselves). */
int s2vt_prof_enable(int32_t on);
/* Synchronises the recorded events; returns summed milliseconds and launch count for `kind`. */
int s2vt_prof_read(int32_t kind, double* total_ms, int64_t* launches);
/* Milliseconds during which at least one bracket of `kind` was open (union of the intervals over both lanes): brackets of
 * one kind overlap where the two lanes run the same kind of kernel side by side, and the sum above counts that time twice. */
int s2vt_prof_read_busy(int32_t kind, double* busy_ms);
int s2vt_prof_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* S2VT_HIP_H */
