// Internal launch interface between the HIP kernels and the C-ABI drivers (api.hip).
#pragma once
#include "common.h"

namespace s2vt {

// ---- gemm.hip
int gemm_f32(hipStream_t stream, bool a_kmajor, bool b_kmajor, int M, int N, int K,
             const float* A, int64_t lda, RowMap amap, const float* B, int64_t ldb, RowMap bmap,
             float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate,
             float* splitk_ws = nullptr, size_t splitk_ws_floats = 0);

// ---- gemm_bf16.hip / split.hip (split-precision and bf16 GEMM on the bf16 matrix cores)
int gemm_bf16_nt(hipStream_t stream, int nplanes, int M, int N, int K, const unsigned short* A, int64_t lda,
                 const unsigned short* B, int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias,
                 bool accumulate, float* splitk_ws, size_t splitk_ws_floats);
// split-precision GEMM on the blocked 3-plane layout (gemm_x3.hip); gemm_bf16_nt(nplanes = 3) forwards here
int gemm_x3(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
            int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
            size_t splitk_ws_floats);
// C[M,N] (+)= X_A^T X_B from the blocked ROW-plane images of X_A [K][M], X_B [K][N] (transposed fragment reads, gemm_x3.hip)
int gemm_x3_tt(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
               int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
               size_t splitk_ws_floats);
// plain bf16 GEMM on bf16 rows with LDS-DMA staging (gemm_b1.hip); gemm_bf16_nt(nplanes = 1) forwards here
int gemm_b1(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
            int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
            size_t splitk_ws_floats);
int gemm_b1_tt(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
               int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
               size_t splitk_ws_floats);
void gemm_b1_tune(int tile_rows, int nsplit);       // 0 / 0: the launcher's time model decides (default)
void gemm_x3_tune(int tile_rows, int nsplit);
int split_planes(hipStream_t s, int nplanes, bool transpose, const float* in, int64_t ld, RowMap imap, int rows, int cols,
                 unsigned short* out, int64_t ldo, int kpad, int out_rows_pad);

// optional input transform of split_planes_dual: the input is the logits [B*(L-1)][V] (batch-major rows) and what is split is
// the mean-CE gradient (exp(logit - lse[r]) - onehot(target)) * gout / rows (ce.hip: ce_bwd_kernel's expression)
struct CeGradArgs {
    const float* lse;            // [rows] from mean_ce_fwd
    const int64_t* target;       // [B][ldt]: row r = b*Lm1 + j reads target[b*ldt + j + 1]
    const float* gout;           // device scalar
    int Lm1; int64_t ldt;
    // optional (bf16 operands): the scale gout / rows is applied as its POWER OF TWO only - the one-hot entries (p - 1) * scale are
    // the same number for every row, and bf16's rounding of that constant (1 / 20224 -> x 0.99808) would scale the whole gradient;
    // the mantissa f = scale / 2^e in [1, 2) is written here for the fp32 consumers of the planes' products to multiply by (the
    // partial column sums of this pass already carry it)
    float* alpha_out;
};
int split_planes_dual(hipStream_t s, int nplanes, const float* in, int64_t ld, RowMap imap, int rows, int cols,
                      unsigned short* out_r, int64_t ldo_r, int kpad_r, unsigned short* out_t, int64_t ldo_t, int kpad_t,
                      float* colpart, const CeGradArgs* ce = nullptr);
int colsum_finish(hipStream_t s, const float* partial, int nchunks, int cols, float* out, bool accumulate);
int scale_by_device_scalar(hipStream_t s, float* x, int64_t n, const float* alpha);      // x *= *alpha (misc.hip)

// ---- lstm.hip
struct StepFwdArgs {
    int B, H;
    // segment 1: recurrent contraction h_prev[B,H] · W_hh[4H,H]^T  (h_prev == nullptr: h = 0)
    const float* h_prev; int64_t ldh;
    const float* w_hh; int64_t ldw;
    // segment 2 (optional): x2[row(b), 0:K2] · W2[4H, 0:K2]^T, row(b) = token id (embedding gather)
    const float* x2; int64_t ldx2; int K2;
    const float* w2; int64_t ldw2;
    const int32_t* tok_idx;                  // int32 token per batch row, or
    const unsigned long long* tok_packed;    // packed argmax word of the previous decode step, or
    int tok_const;                           // one token for every row (<sos>); used when both null
    // pre-computed gate input (x-part + both biases) per batch row, or bias only
    const float* gx; int64_t ldgx;
    const int32_t* gx_idx;                   // optional: batch row b reads gx row gx_idx[b] (beam search: every beam slot of a sample
                                             // shares the sample's vid_rnn half of the gate input)
    const float* bias;                       // [4H], used when gx == nullptr
    const float* c_prev; int64_t ldc;        // nullptr: c = 0
    float* h_out; int64_t ldho;
    float* h_out2; int64_t ldho2;            // optional second copy (batch-major view)
    float* c_out; int64_t ldco;
    float* stash; int64_t ldst;              // [B,4H] activated gates i,f,g,o (train only)
    // optional: h_t also as three bf16 planes in the blocked operand layout of gemm_x3.hip / argmax_x3.hip (rows = batch,
    // k = hidden unit, ld = 3 * kpad elements): the decode step hands its h_t to the plane-path argmax kernel without a
    // split launch in between.  Columns H..kpad of the image must have been zeroed by the caller.
    unsigned short* h_planes; int64_t ldhp;
    // optional: a per-TOKEN gate-input table added to gx in the epilogue, row = the token the x2 segment would gather
    // (tok_idx / tok_packed / tok_const): gtab[tok][4H] = Emb[tok]·W_e^T computed once per decode call replaces the
    // E-wide second K segment of every decode step (x2 must then be null)
    const float* gx_tab; int64_t ldtab;
    // guard of the token path (tok_idx / tok_packed / tok_const): an id outside [0, tok_limit) is read as token 0 and raises
    // *tok_err (the S2VT_ERR_INDEX flag word of the caller's workspace) instead of addressing memory outside the table - a
    // producer bug (e.g. a packed argmax word that no workgroup wrote) then surfaces as an error code, not as a GPU fault.
    // tok_limit == 0: no token segment in use
    int tok_limit; int* tok_err;
    // optional: CONTRACTION ONLY - z_out[b][g*H + u] = sum_k h_prev[b][k] W_hh[g*H + u][k] (the reduced partial sums, nothing
    // added) and no cell update: the recurrent half of a decode step that does not depend on the previous step's token, so it
    // runs beside that step's out_linear + argmax; lstm_cell_pointwise() finishes the step (same additions in the same order)
    float* z_out; int64_t ldz;
};
int lstm_step_fwd(hipStream_t stream, const StepFwdArgs& a);
// the cell update of a step whose contraction ran as its own launch (StepFwdArgs::z_out): the token segment (gx_tab + token
// source + guard), gx, c_prev and every output of `a` are used; h_prev / w_hh are not read, a.z_out is the INPUT
int lstm_cell_pointwise(hipStream_t stream, const StepFwdArgs& a);
int lstm_step_fwd2(hipStream_t stream, const StepFwdArgs& a, const StepFwdArgs* b);   // two independent steps, one launch
// lstm_gemv.hip: the same step for B <= 8 as gate GEMVs (h staged in LDS, weight rows streamed to registers, wavefront shuffle
// reductions); lstm_step_fwd routes there when option "gemv" is on and lstm_step_fwd_gemv_ok(a)
bool lstm_step_fwd_gemv_ok(const StepFwdArgs& a);
int lstm_step_fwd_gemv(hipStream_t stream, const StepFwdArgs& a);

struct StepBwdArgs {
    int B, H;
    // dh_rec = dG_next[B,4H] · W_hh[4H,H]  (computed against the transposed copy W_hh^T [H,4H])
    const float* dg_next; int64_t lddg;      // nullptr at the last timestep
    const float* w_hh_t; int64_t ldwt;
    const float* dh_out; int64_t lddho;      // gradient arriving from above at this step (nullable)
    const float* stash; int64_t ldst;        // activated gates of this step
    const float* c; int64_t ldc;             // c_t
    const float* c_prev; int64_t ldcp;       // c_{t-1} (nullptr: 0)
    float* dc; int64_t lddc;                 // in: dL/dc_t carried from t+1 (ignored if first), out: dL/dc_{t-1}
    int dc_is_zero;                          // 1 at the last timestep (no incoming dc)
    float* dg; int64_t lddg_out;             // out: pre-activation gate grads [B,4H]
};
int lstm_step_bwd(hipStream_t stream, const StepBwdArgs& a);
int lstm_step_bwd2(hipStream_t stream, const StepBwdArgs& a, const StepBwdArgs* b);

// ---- lstm_bf16.hip: bf16-operand timestep kernels (config 3)
struct StepFwdBf16Args {
    int B, H, Kp;                                   // Kp: k extent of the bf16 operands (H zero-padded to 64)
    const unsigned short* hb_prev; int64_t ldhb;    // bf16 h_{t-1} rows (nullptr: h = 0)
    const unsigned short* wb; int64_t ldwb;         // bf16 W_hh rows [4H][Kp]
    const float* gx; int64_t ldgx;
    const float* bias;
    const float* c_prev; int64_t ldc;
    float* h_out; int64_t ldho;                     // optional fp32 copy of h_t
    unsigned short* hb_out; int64_t ldhbo;          // bf16 row image of h_t (next operand + GEMM plane)
    float* c_out; int64_t ldco;
    float* stash; int64_t ldst;
};
int lstm_step_fwd_bf16(hipStream_t stream, const StepFwdBf16Args& a);

struct StepBwdBf16Args {
    int B, H, Kp;                                   // Kp: 4H zero-padded to 64
    const unsigned short* dgb_next; int64_t lddgb;  // bf16 dG_{t+1} rows (nullptr at the last step)
    const unsigned short* wtb; int64_t ldwtb;       // bf16 W_hh^T rows [H][Kp]
    const float* dh_out; int64_t lddho;
    const float* stash; int64_t ldst;
    const float* c; int64_t ldc;
    const float* c_prev; int64_t ldcp;
    float* dc; int64_t lddc;
    int dc_is_zero;
    float* dg; int64_t lddg;                        // fp32 dG_t (may alias stash)
    unsigned short* dgb; int64_t lddgbo;            // bf16 row image of dG_t
};
int lstm_step_bwd_bf16(hipStream_t stream, const StepBwdBf16Args& a);

// ---- lstm_persist.hip: persistent bf16 recurrence (one launch per block of timesteps, W_hh slice resident per CU)
struct SeqFwdBf16Args {
    int B, H, Kp;                                   // Kp: H zero-padded to a multiple of 64 (<= 1024)
    int t0, t1, n_gx;                               // steps [t0, t1); gx present for t < n_gx, bias otherwise
    const unsigned short* wb; int64_t ldwb;         // bf16 W_hh rows [4H][Kp]
    unsigned short* hb; int64_t ldhb;               // bf16 h rows, time-major [T*B][ldhb]: h_{t-1} in, h_t out
    float* gx_stash;                                // [T*B][4H]: gate input in, activated gates i,f,g,o out (in place)
    const float* bias;                              // [4H]
    float* h_all;                                   // [T*B][H] fp32 h_t (optional)
    float* c_all;                                   // [T*B][H] fp32 c_t
    unsigned int* sync;                             // lstm_persist_sync_bytes() of counters (zeroed by the launcher)
    int* err;                                       // set to 1 if a hand-off wait timed out
    int RB, NS;                                     // set by the launcher: rows per workgroup, 32-row chains in it
    unsigned long long* stamps; int stamp_block;    // timing experiments only (experiment.h); null in the product
};
int lstm_seq_fwd_bf16_persist_supported(int B, int H, int Kp);
size_t lstm_persist_sync_bytes();
int lstm_seq_fwd_bf16_persist(hipStream_t stream, SeqFwdBf16Args a);
// two layers (e.g. vid_rnn block k+1 and word_rnn block k) side by side in ONE launch; b == nullptr: one layer
int lstm_seq_fwd_bf16_persist2(hipStream_t stream, SeqFwdBf16Args a, const SeqFwdBf16Args* b);

struct SeqBwdBf16Args {
    int B, H, Kp;                                   // Kp: 4H zero-padded to a multiple of 64 (<= 4096)
    int T, t0, t1;                                  // BPTT over steps t1-1 .. t0 of a T-step layer
    const unsigned short* wtb; int64_t ldwtb;       // bf16 W_hh^T rows [H][Kp]
    unsigned short* dgb; int64_t lddgb;             // bf16 dG rows, time-major [T*B][lddgb]: dG_{t+1} in, dG_t out
    const float* dh_out; int dh_first;              // gradient from above for steps >= dh_first, [(T-dh_first)*B][H] (nullable)
    float* stash_dg;                                // [T*B][4H]: activated gates in, fp32 dG out (in place)
    const float* c_all;                             // [T*B][H]
    float* dc;                                      // [B][H] dL/dc carried between launches (ignored when t1 == T)
    unsigned int* sync; int* err;
    int RB, NS;                                     // set by the launcher
    unsigned long long* stamps; int stamp_block;    // timing experiments only (experiment.h); null in the product
};
int lstm_seq_bwd_bf16_persist_supported(int B, int H, int Kp4);
int lstm_seq_bwd_bf16_persist2(hipStream_t stream, SeqBwdBf16Args a, const SeqBwdBf16Args* b);

// split-precision (three bf16 planes per operand, six plane products: fp32-equivalent) persistent forward, lstm_persist_x3.hip
struct SeqFwdX3Args {
    int B, H, Kp;                                   // Kp = H rounded up to 64 (<= 1024): row length of the plane images
    int t0, t1, n_gx;
    const unsigned short* wp; int64_t wplane, ldw;  // W_hh as planes [3][4H][ldw] (split3_rows; columns >= H zero)
    unsigned short* hp; int64_t hplane, ldh;        // h_t as planes [3][T*B][ldh], time-major: the hand-off payload
    float* h_all;                                   // [T*B][H] fp32 h_t (output only)
    float* gx_stash; const float* bias; float* c_all;
    int no_stash;                                   // 1: the activated gates are not written back (inference: greedy decode)
    unsigned int* sync; int* err;
    // optional (whole-path drivers, B % 64 == 0): h_t ALSO as the blocked 3-plane row image the batched GEMMs read (gemm_x3.hip
    // layout: rows = t * B + b, k = unit, ld = 3 * pad64(H) elements) - the 16-byte slots of that image are the hand-off payload's
    // own (plane, row, 8 units) pieces, so no split pass reads h back; the caller keeps the k16 records past 16 * ceil(H / 16) at zero
    unsigned short* hblk; int64_t ldhblk;
    int RB, NS;                                     // set by the launcher
    unsigned long long* stamps; int stamp_block;    // timing experiments only (experiment.h); null in the product
};
int lstm_seq_fwd_x3_persist_supported(int B, int H);
int lstm_seq_fwd_x3_persist_single_workgroups(int B, int H);      // workgroups of a launch with ONE layer (0: unsupported)
int lstm_seq_fwd_x3_persist2(hipStream_t stream, SeqFwdX3Args a, const SeqFwdX3Args* b);
int split3_rows(hipStream_t stream, const float* src, int64_t ld, int R, int C, int Cp, unsigned short* dst, int64_t plane);
struct SeqBwdX3Args {
    int B, H, Kp, Hp;                               // Kp = H rounded up to 64 (output columns j), Hp = H rounded up to 16
    int T, t0, t1;                                  // BPTT over steps t1-1 .. t0 of a T-step layer
    const unsigned short* wtp; int64_t wplane, ldw; // W_hh^T as planes [3][Kp][4 Hp] (split3_wt)
    const float* dh_out; int dh_first;              // gradient from above for steps >= dh_first, [(T-dh_first)*B][H] (nullable)
    float* stash_dg;                                // [T*B][4H]: activated gates in, fp32 dG out (in place)
    const float* c_all;                             // [T*B][H]
    float* dc;                                      // [B][H] dL/dc carried between launches (ignored when t1 == T)
    float* part; int64_t part_slot; int nslots;     // partial-sum ring [nslots][chains][nC][nC][32][16] fp32; nslots > t1 - t0
    unsigned int* sync; int* err;
    // optional (whole-path driver; needs H % 8 == 0): dG_t ALSO leaves as the blocked 3-plane ROW image the batched GEMMs read
    // (gemm_x3.hip layout: rows = t * B + b, k = g * H + u, ld = 3 * pad64(4H) elements; the caller keeps the k padding at
    // zero) together with its column sums over each 32-row chain (the bias gradient's partial sums, colsum_finish over T * B / 32
    // chunks): the tile is in LDS as planes anyway, so no split pass reads dG back.  skip_dg: the fp32 dG is then not stored.
    unsigned short* dgp; int64_t lddgp;
    float* colpart;                                 // [T * B / 32][4H]
    int skip_dg;
    int RB, NS;                                     // set by the launcher
    unsigned long long* stamps; int stamp_block;    // timing experiments only (experiment.h); null in the product
};
int lstm_seq_bwd_x3_persist_supported(int B, int H);
size_t lstm_seq_bwd_x3_part_slot_floats(int B, int H);
int lstm_seq_bwd_x3_persist2(hipStream_t stream, SeqBwdX3Args a, const SeqBwdX3Args* b);
int split3_wt(hipStream_t stream, const float* wt, int H, int Kp, int Hp, unsigned short* dst, int64_t plane);

struct LogitsArgmaxArgs {
    int B, H, V;
    const float* h; int64_t ldh;
    const float* w_out; int64_t ldw;         // [V,H]
    const float* b_out;
    unsigned long long* packed;              // [B] zero-initialised; atomicMax of (ordered logit << 32 | ~index)
    unsigned long long* stamps;              // timing experiments only (experiment.h); null in the product
};
int logits_argmax(hipStream_t stream, const LogitsArgmaxArgs& a);

// ---- argmax_x3.hip: the same decode step on the bf16 matrix cores (operands as blocked 3-plane images, split.hip)
struct ArgmaxX3Args {
    int B, V, K;                                    // K: H zero-padded to a multiple of 64
    const unsigned short* W; int64_t ldw;           // planes of W_o [rows64(V)][3K] (written once per decode call)
    const unsigned short* Hp; int64_t ldh;          // planes of h_t [rows64(B)][3K] (re-split every step)
    const float* bias;                              // b_o [V] (nullable)
    unsigned long long* packed;                     // [B] zero-initialised; atomicMax of (ordered logit << 32 | ~index)
    // optional second role of the same launch: row blocks past the vocabulary's multiply the SAME h_t planes with a second
    // weight image and write the products instead of reducing them - z[b][m] = h_t[b]·W2[m], m < M2 (the decode loop: W2 =
    // W_hh of word_rnn, z = the recurrent half of step t+1's gates, which does not depend on step t's token).  M2 == 0: none.
    // v_off = number of leading vocabulary row blocks the launch leaves out (cdiv(V, 64): the second role alone).
    const unsigned short* W2; int64_t ldw2; int M2;
    float* z; int64_t ldz;
    int v_off;
    int dbg;                                        // timing experiments only (S2VT_AX_DBG): 0 in the product
    unsigned long long* stamps;                     // timing experiments only (experiment.h); null in the product
};
int logits_argmax_x3(hipStream_t stream, const ArgmaxX3Args& a);

// ---- misc.hip
int add_vectors(hipStream_t s, const float* a, const float* b, float* out, int n);
int adam_flat(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
              int64_t step);
int mul_vectors(hipStream_t s, const float* a, const float* b, float* out, int64_t n);
int transpose_f32(hipStream_t s, const float* in, int rows, int cols, float* out);   // out[cols][rows]
int colsum_f32(hipStream_t s, const float* x, int64_t rows, int cols, int64_t ld, float* partial, float* out,
               bool accumulate);
size_t colsum_partial_floats(int64_t rows, int cols);
int targets_to_time_major(hipStream_t s, const int64_t* targets, int B, int Lm1, int64_t ld, int V, int32_t* out,
                          int* err_flag);
// beam-search fan-out (ce.hip): log_softmax + the 20 most probable tokens per row in ascending token order
int top20_logprob(hipStream_t s, const float* logits, int64_t ld, int64_t rows, int V, int32_t* top_ix, float* top_lp);
int gather_rows_f32(hipStream_t s, const float* src, int64_t ld, const int32_t* idx, int64_t rows, int cols, float* out);
size_t embedding_grad_ws_ints(int64_t rows, int V);
int embedding_grad(hipStream_t s, const float* d_rows, int64_t rows, int E, const int32_t* tok, int V, float* d_emb, int* ws);
int unpack_tokens(hipStream_t s, const unsigned long long* packed, int steps, int B, int64_t* out_ids);
int fill_zero(hipStream_t s, void* p, size_t bytes);
int occupy_cus(hipStream_t s, int workgroups, int lds_bytes, long long microseconds);   // test support (co-residency tests)
int zero_pad_cols_u16(hipStream_t s, unsigned short* p, int64_t rows, int64_t ld, int col0, int col1);

// ---- ce.hip
int mean_ce_fwd(hipStream_t s, const float* logits, int64_t rows, int V, const int64_t* target, int Lm1, int64_t ldt,
                float* lse, float* rowloss, float* loss_out, int* err_flag);
int mean_ce_bwd(hipStream_t s, const float* logits, int64_t rows, int V, const int64_t* target, int Lm1, int64_t ldt,
                const float* lse, const float* gout, float* dlogits);
// MaskCriterion as a whole (utils.py:13-26): per-row CE, then mean / mask weighting / division in one workgroup; out3 = {loss, mean_ce, sum(w)}
int mask_criterion_fwd(hipStream_t s, const float* logits, int64_t rows, int V, const int64_t* target, int Lm1, int64_t ldt,
                       const float* mask, int64_t ldm, float* lse, float* rowloss, float* out3, int* err_flag);
int mask_criterion_bwd(hipStream_t s, const float* mask, int64_t ldm, int64_t rows, int Lm1, const float* fwd_out, const float* gout,
                       float* g_ce);

}  // namespace s2vt
