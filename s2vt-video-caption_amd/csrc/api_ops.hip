// Per-op entry points of include/s2vt_hip.h: the pieces the whole-path drivers are built from (criterion, GEMMs, plane split,
// feature projection, timestep and sequence kernels), exported for tests, profiling and reuse (Att_Baseline).
#include "api_internal.h"

using namespace s2vt;

extern "C" {

// ------------------------------------------------------------------ loss
// torch.optim.Adam.step() of train.py:126 over flat buffers (see adam_flat in misc.hip)
int s2vt_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                   double eps, int64_t step, void* stream) {
    return adam_flat((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step);
}

int s2vt_mean_ce_forward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target,
                         int64_t target_ld, float* lse, float* rowloss, float* loss_out, void* stream) {
    S2VT_REQUIRE(B > 0 && Lm1 > 0 && V > 0, "s2vt_mean_ce_forward: bad dims");
    hipStream_t st = (hipStream_t)stream;
    // target ids outside [0, V): flagged on the device, reported like the embedding's (s2vt_check_async_error)
    int* flags = nullptr;
    int rc;
    if ((rc = device_flags(&flags))) return rc;
    const int rc0 = poll_async_error(false);
    if ((rc = fill_zero(st, flags, 4 * sizeof(int)))) return rc;
    {
        ProfScope ps(st, K_CE, 1);
        if ((rc = mean_ce_fwd(st, logits, (int64_t)B * Lm1, V, target, Lm1, target_ld, lse, rowloss, loss_out, flags)))
            return rc;
    }
    return rc0 ? rc0 : post_async_error(st, flags, 2);
}
int s2vt_mean_ce_backward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target,
                          int64_t target_ld, const float* lse, const float* gout, float* dlogits, void* stream) {
    S2VT_REQUIRE(B > 0 && Lm1 > 0 && V > 0, "s2vt_mean_ce_backward: bad dims");
    ProfScope ps((hipStream_t)stream, K_CE, 1);
    return mean_ce_bwd((hipStream_t)stream, logits, (int64_t)B * Lm1, V, target, Lm1, target_ld, lse, gout, dlogits);
}

// MaskCriterion.forward as ONE call (utils.py:13-26): the per-row CE kernel, then a single workgroup that forms the mean, weights it
// with mask[:, 1:] product by product and divides by the mask's sum - the reference's arithmetic (NaN for an all-zero mask) without
// the dozen elementwise / reduction launches the host framework spends on those three lines.  out3 = {loss, mean_ce, sum(mask[:, 1:])}.
int s2vt_mask_criterion_forward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target, int64_t target_ld,
                                const float* mask, int64_t mask_ld, float* lse, float* rowloss, float* out3, void* stream) {
    S2VT_REQUIRE(B > 0 && Lm1 > 0 && V > 0 && mask_ld >= Lm1 + 1, "s2vt_mask_criterion_forward: bad dims");
    hipStream_t st = (hipStream_t)stream;
    int* flags = nullptr;
    int rc;
    if ((rc = device_flags(&flags))) return rc;
    const int rc0 = poll_async_error(false);
    if ((rc = fill_zero(st, flags, 4 * sizeof(int)))) return rc;
    {
        ProfScope ps(st, K_CE, 1);
        if ((rc = mask_criterion_fwd(st, logits, (int64_t)B * Lm1, V, target, Lm1, target_ld, mask, mask_ld, lse, rowloss, out3, flags)))
            return rc;
    }
    return rc0 ? rc0 : post_async_error(st, flags, 2);
}
// Its autograd down to the mean CE: g_ce[0] = sum_i (gout[0] / sum(w)) * w_i - the `gout` of s2vt_mean_ce_backward[_fused].
int s2vt_mask_criterion_backward(int32_t B, int32_t Lm1, const float* mask, int64_t mask_ld, const float* out3, const float* gout,
                                 float* g_ce, void* stream) {
    S2VT_REQUIRE(B > 0 && Lm1 > 0 && mask_ld >= Lm1 + 1, "s2vt_mask_criterion_backward: bad dims");
    return mask_criterion_bwd((hipStream_t)stream, mask, mask_ld, (int64_t)B * Lm1, Lm1, out3, gout, g_ce);
}

// ------------------------------------------------------------------ per-op entry points
int s2vt_gemm_f32(int32_t a_kmajor, int32_t b_kmajor, int32_t M, int32_t N, int32_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t accumulate,
                  void* stream) {
    return gemm((hipStream_t)stream, a_kmajor != 0, b_kmajor != 0, M, N, K, A, lda, ID, B, ldb, ID, C, ldc, ID, bias,
                accumulate != 0);
}

int s2vt_gemm_f32_splitk(int32_t a_kmajor, int32_t b_kmajor, int32_t M, int32_t N, int32_t K, const float* A,
                         int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                         int32_t accumulate, float* ws, size_t ws_floats, void* stream) {
    GemmWsScope gscope(ws, ws_floats);
    return gemm((hipStream_t)stream, a_kmajor != 0, b_kmajor != 0, M, N, K, A, lda, ID, B, ldb, ID, C, ldc, ID, bias,
                accumulate != 0);
}

int s2vt_split_planes(int32_t nplanes, int32_t transpose, const float* in, int64_t ld, int32_t rows, int32_t cols,
                      uint16_t* out, int64_t ldo, int32_t kpad, int32_t out_rows_pad, void* stream) {
    return split_planes((hipStream_t)stream, nplanes, transpose != 0, in, ld, ID, rows, cols, out, ldo, kpad, out_rows_pad);
}

int s2vt_gemm_bf16_nt(int32_t nplanes, int32_t M, int32_t N, int32_t K, const uint16_t* A, int64_t lda, const uint16_t* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t accumulate, float* ws,
                      size_t ws_floats, void* stream) {
    ProfScope ps((hipStream_t)stream, K_GEMM, 1);
    return gemm_bf16_nt((hipStream_t)stream, nplanes, M, N, K, A, lda, B, ldb, C, ldc, ID, bias, accumulate != 0, ws,
                        ws_floats);
}

int s2vt_gemm_bf16_tt(int32_t nplanes, int32_t M, int32_t N, int32_t K, const uint16_t* A, int64_t lda, const uint16_t* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t accumulate, float* ws, size_t ws_floats,
                      void* stream) {
    S2VT_REQUIRE(nplanes == 3 || nplanes == 1, "s2vt_gemm_bf16_tt: planes must be 1 or 3");
    ProfScope ps((hipStream_t)stream, K_GEMM, 1);
    if (nplanes == 1) return gemm_b1_tt((hipStream_t)stream, M, N, K, A, lda, B, ldb, C, ldc, ID, bias, accumulate != 0, ws, ws_floats);
    return gemm_x3_tt((hipStream_t)stream, M, N, K, A, lda, B, ldb, C, ldc, ID, bias, accumulate != 0, ws, ws_floats);
}

int s2vt_feat_proj_fwd(const s2vt_dims* d, const float* feats, const float* w, const float* bias, float* x1,
                       void* stream) {
    S2VT_REQUIRE(dims_ok(d) && feats && w && x1, "s2vt_feat_proj_fwd: null/invalid argument");
    return gemm((hipStream_t)stream, true, true, d->B * d->L, d->H, d->F, feats, d->F, ID, w, d->F, ID, x1, d->H,
                perm(d->L, d->B), bias, false);
}

size_t s2vt_colsum_ws_floats(int64_t rows, int32_t cols) { return colsum_partial_floats(rows, cols); }

int s2vt_feat_proj_bwd(const s2vt_dims* d, const float* feats, const float* w, const float* dx1, float* dw,
                       float* dbias, float* dfeats, float* colsum_ws, void* stream) {
    S2VT_REQUIRE(dims_ok(d) && feats && w && dx1 && dw && dbias && colsum_ws, "s2vt_feat_proj_bwd: null/invalid argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = gemm(st, false, false, d->H, d->F, d->L * d->B, dx1, d->H, ID, feats, d->F, perm(d->B, d->L), dw, d->F, ID,
                   nullptr, false)))
        return rc;
    if ((rc = colsum_f32(st, dx1, (int64_t)d->L * d->B, d->H, d->H, colsum_ws, dbias, false))) return rc;
    if (dfeats)
        return gemm(st, true, false, d->L * d->B, d->F, d->H, dx1, d->H, ID, w, d->F, ID, dfeats, d->F, perm(d->B, d->L),
                    nullptr, false);
    return 0;
}

int s2vt_lstm_step_fwd(int32_t B, int32_t H, const float* gx, const float* bias, const float* w_hh,
                       const float* h_prev, const float* c_prev, float* h_out, float* c_out, float* stash,
                       void* stream) {
    StepFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H;
    a.h_prev = h_prev; a.ldh = H; a.w_hh = w_hh; a.ldw = H;
    a.gx = gx; a.ldgx = 4 * (int64_t)H; a.bias = bias;
    a.c_prev = c_prev; a.ldc = H;
    a.h_out = h_out; a.ldho = H; a.c_out = c_out; a.ldco = H;
    a.stash = stash; a.ldst = 4 * (int64_t)H;
    ProfScope ps((hipStream_t)stream, K_STEP_FWD, 1);
    return lstm_step_fwd((hipStream_t)stream, a);
}

int s2vt_lstm_step_fwd_token(int32_t B, int32_t H, int32_t E, int32_t V, const float* gx, const float* w_hh, const float* h_prev,
                             const float* c_prev, const float* emb, const float* w_e, int64_t ldw_e, const int32_t* tok,
                             const unsigned long long* tok_packed, int32_t tok_const, float* h_out, float* c_out, void* stream) {
    S2VT_REQUIRE(B > 0 && H > 0 && E > 0 && V > 0 && gx && w_hh && emb && w_e && h_out && c_out && ldw_e >= E,
                 "s2vt_lstm_step_fwd_token: null/invalid argument");
    hipStream_t st = (hipStream_t)stream;
    int* flags = nullptr;
    int rc;
    if ((rc = device_flags(&flags))) return rc;
    const int rc0 = poll_async_error(false);
    if ((rc = fill_zero(st, flags, 4 * sizeof(int)))) return rc;
    StepFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H;
    a.h_prev = h_prev; a.ldh = H; a.w_hh = w_hh; a.ldw = H;
    a.x2 = emb; a.ldx2 = E; a.K2 = E; a.w2 = w_e; a.ldw2 = ldw_e;
    a.tok_idx = tok; a.tok_packed = tok_packed; a.tok_const = tok_const;
    a.tok_limit = V; a.tok_err = flags;
    a.gx = gx; a.ldgx = 4 * (int64_t)H;
    a.c_prev = c_prev; a.ldc = H;
    a.h_out = h_out; a.ldho = H; a.c_out = c_out; a.ldco = H;
    {
        ProfScope ps(st, K_STEP_FWD, 1);
        if ((rc = lstm_step_fwd(st, a))) return rc;
    }
    return rc0 ? rc0 : post_async_error(st, flags, 2);
}

int s2vt_lstm_step_bwd(int32_t B, int32_t H, const float* dg_next, const float* w_hh_t, const float* dh_out,
                       const float* stash, const float* c, const float* c_prev, float* dc, int32_t dc_is_zero,
                       float* dg, void* stream) {
    StepBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H;
    a.dg_next = dg_next; a.lddg = 4 * (int64_t)H; a.w_hh_t = w_hh_t; a.ldwt = 4 * (int64_t)H;
    a.dh_out = dh_out; a.lddho = H;
    a.stash = stash; a.ldst = 4 * (int64_t)H;
    a.c = c; a.ldc = H; a.c_prev = c_prev; a.ldcp = H;
    a.dc = dc; a.lddc = H; a.dc_is_zero = dc_is_zero;
    a.dg = dg; a.lddg_out = 4 * (int64_t)H;
    ProfScope ps((hipStream_t)stream, K_STEP_BWD, 1);
    return lstm_step_bwd((hipStream_t)stream, a);
}

int s2vt_lstm_seq_fwd(int32_t T, int32_t B, int32_t H, const float* gx, int32_t n_gx, const float* bias,
                      const float* w_hh, float* h_all, float* c_all, float* stash, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh && h_all && c_all && n_gx >= 0 && n_gx <= T,
                 "s2vt_lstm_seq_fwd: bad arguments");
    S2VT_REQUIRE(n_gx == 0 || gx, "s2vt_lstm_seq_fwd: gx missing");
    S2VT_REQUIRE(n_gx == T || bias, "s2vt_lstm_seq_fwd: bias needed for steps without gx");
    S2VT_REQUIRE(stash == nullptr || stash == gx || n_gx == 0,
                 "s2vt_lstm_seq_fwd: stash must alias gx (in-place) or gx must be absent");
    hipStream_t st = (hipStream_t)stream;
    if (stash) return seq_fwd(st, 0, T, B, H, stash, n_gx, bias, w_hh, h_all, c_all, true);
    return seq_fwd(st, 0, T, B, H, const_cast<float*>(gx), n_gx, bias, w_hh, h_all, c_all, false);
}

int s2vt_lstm_seq_bwd(int32_t T, int32_t B, int32_t H, const float* w_hh, const float* dh_out, int32_t dh_first,
                      const float* c_all, float* stash_dg, float* w_hh_t, float* dc, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh && c_all && stash_dg && w_hh_t && dc, "s2vt_lstm_seq_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = transpose_f32(st, w_hh, 4 * H, H, w_hh_t))) return rc;
    return seq_bwd(st, T, 0, T, B, H, w_hh_t, dh_out, dh_first, c_all, stash_dg, dc);
}

// bf16-operand layer forward (config 3 arithmetic) as its own entry point: kernel-level parity tests and benchmarks.
// workspace: [err int x64][sync][W_hh bf16 rows][h bf16 rows]
struct SeqBf16WS { int* err; unsigned int* sync; PB wb, hb; size_t bytes; };
static SeqBf16WS carve_seq_bf16(int T, int B, int H, void* base) {
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    SeqBf16WS w;
    w.err = c.take<int>(64);
    w.sync = c.take<unsigned int>(lstm_persist_sync_bytes() / sizeof(unsigned int));
    auto mk = [&](size_t rows, size_t k) {
        PB b;
        b.kpad = pad64((int)k);
        b.ld = b.kpad;
        b.p = c.take<unsigned short>(rows64(rows) * (size_t)b.ld);
        return b;
    };
    w.wb = mk((size_t)4 * H, H);
    w.hb = mk((size_t)T * B, H);
    w.bytes = align_up(c.off, 256);
    return w;
}
size_t s2vt_lstm_seq_bf16_workspace_bytes(int32_t T, int32_t B, int32_t H) {
    if (T <= 0 || B <= 0 || H <= 0) return 0;
    return carve_seq_bf16(T, B, H, nullptr).bytes;
}
static int seq_bf16_prepare(hipStream_t st, const SeqBf16WS& w, int T, int B, int H, const float* w_hh) {
    int rc;
    if ((rc = fill_zero(st, w.err, 64 * sizeof(int)))) return rc;
    if ((rc = split_planes(st, 1, false, w_hh, H, ID, 4 * H, H, w.wb.p, w.wb.ld, w.wb.kpad, (int)rows64((size_t)4 * H)))) return rc;
    return zero_pad_cols_u16(st, w.hb.p, (int64_t)T * B, w.hb.ld, H, w.hb.kpad);
}
static SeqFwdBf16Args seq_bf16_args(const SeqBf16WS& w, int B, int H, int t0, int t1, float* gx_stash, int n_gx,
                                    const float* bias, float* h_all, float* c_all) {
    SeqFwdBf16Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.Kp = w.hb.kpad;
    a.t0 = t0; a.t1 = t1; a.n_gx = n_gx;
    a.wb = w.wb.p; a.ldwb = w.wb.ld;
    a.hb = w.hb.p; a.ldhb = w.hb.ld;
    a.gx_stash = gx_stash; a.bias = bias;
    a.h_all = h_all; a.c_all = c_all;
    a.sync = w.sync; a.err = w.err;
#ifdef S2VT_EXPERIMENT_STAMPS
    a.stamps = g_xstamps; a.stamp_block = g_xstamp_block;
#endif
    return a;
}

int s2vt_lstm_seq_fwd_bf16(int32_t T, int32_t B, int32_t H, float* gx_stash, int32_t n_gx, const float* bias,
                           const float* w_hh, float* h_all, float* c_all, void* workspace, size_t workspace_bytes,
                           int32_t persistent, int32_t block, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && gx_stash && w_hh && h_all && c_all && workspace && n_gx >= 0 && n_gx <= T,
                 "s2vt_lstm_seq_fwd_bf16: bad arguments");
    S2VT_REQUIRE(n_gx == T || bias, "s2vt_lstm_seq_fwd_bf16: bias needed for steps without gx");
    const SeqBf16WS w = carve_seq_bf16(T, B, H, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_lstm_seq_fwd_bf16: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = seq_bf16_prepare(st, w, T, B, H, w_hh))) return rc;
    if (!persistent) return seq_fwd_bf16(st, 0, T, B, H, gx_stash, n_gx, bias, w.wb, w.hb, h_all, c_all);
    S2VT_REQUIRE(lstm_seq_fwd_bf16_persist_supported(B, H, w.hb.kpad), "s2vt_lstm_seq_fwd_bf16: shape not supported by the persistent kernel");
    const int blk = block > 0 ? block : T;
    for (int t0 = 0; t0 < T; t0 += blk) {
        const int t1 = (t0 + blk < T) ? t0 + blk : T;
        ProfScope ps(st, K_STEP_FWD, t1 - t0);
        if ((rc = lstm_seq_fwd_bf16_persist(st, seq_bf16_args(w, B, H, t0, t1, gx_stash, n_gx, bias, h_all, c_all)))) return rc;
    }
    return 0;
}

// Two independent layers of the same shape, every block of timesteps of both in ONE persistent launch (the schedule the
// whole-path driver uses for vid_rnn block k+1 next to word_rnn block k); workspace = 2 x the single-layer size.
int s2vt_lstm_seq_fwd_bf16_pair(int32_t T, int32_t B, int32_t H, float* gx_stash0, float* gx_stash1, int32_t n_gx,
                                const float* bias0, const float* bias1, const float* w_hh0, const float* w_hh1,
                                float* h_all0, float* h_all1, float* c_all0, float* c_all1, void* workspace,
                                size_t workspace_bytes, int32_t block, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && gx_stash0 && gx_stash1 && w_hh0 && w_hh1 && h_all0 && h_all1 && c_all0 && c_all1 &&
                     workspace && n_gx >= 0 && n_gx <= T && (n_gx == T || (bias0 && bias1)),
                 "s2vt_lstm_seq_fwd_bf16_pair: bad arguments");
    const size_t one = carve_seq_bf16(T, B, H, nullptr).bytes;
    S2VT_REQUIRE(workspace_bytes >= 2 * one, "s2vt_lstm_seq_fwd_bf16_pair: workspace %zu < %zu bytes", workspace_bytes, 2 * one);
    const SeqBf16WS w0 = carve_seq_bf16(T, B, H, workspace);
    const SeqBf16WS w1 = carve_seq_bf16(T, B, H, reinterpret_cast<char*>(workspace) + one);
    S2VT_REQUIRE(lstm_seq_fwd_bf16_persist_supported(B, H, w0.hb.kpad), "s2vt_lstm_seq_fwd_bf16_pair: shape not supported by the persistent kernel");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = seq_bf16_prepare(st, w0, T, B, H, w_hh0))) return rc;
    if ((rc = seq_bf16_prepare(st, w1, T, B, H, w_hh1))) return rc;
    const int blk = block > 0 ? block : T;
    for (int t0 = 0; t0 < T; t0 += blk) {
        const int t1 = (t0 + blk < T) ? t0 + blk : T;
        ProfScope ps(st, K_STEP_FWD, 2 * (t1 - t0));
        SeqFwdBf16Args a1 = seq_bf16_args(w1, B, H, t0, t1, gx_stash1, n_gx, bias1, h_all1, c_all1);
        a1.err = w0.err;
        if ((rc = lstm_seq_fwd_bf16_persist2(st, seq_bf16_args(w0, B, H, t0, t1, gx_stash0, n_gx, bias0, h_all0, c_all0), &a1)))
            return rc;
    }
    return 0;
}

// bf16-operand BPTT of one layer as its own entry point (kernel-level parity tests and benchmarks).
// workspace: [err int x64][sync][W_hh^T bf16 rows][dG bf16 rows][dc]
struct SeqBwdBf16WS { int* err; unsigned int* sync; PB wt, dgb; float* dc; size_t bytes; };
static SeqBwdBf16WS carve_seq_bwd_bf16(int T, int B, int H, void* base) {
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    SeqBwdBf16WS w;
    w.err = c.take<int>(64);
    w.sync = c.take<unsigned int>(lstm_persist_sync_bytes() / sizeof(unsigned int));
    auto mk = [&](size_t rows, size_t k) {
        PB b;
        b.kpad = pad64((int)k);
        b.ld = b.kpad;
        b.p = c.take<unsigned short>(rows64(rows) * (size_t)b.ld);
        return b;
    };
    w.wt = mk((size_t)H, (size_t)4 * H);
    w.dgb = mk((size_t)T * B, (size_t)4 * H);
    w.dc = c.take<float>((size_t)B * H);
    w.bytes = align_up(c.off, 256);
    return w;
}
size_t s2vt_lstm_seq_bwd_bf16_workspace_bytes(int32_t T, int32_t B, int32_t H) {
    if (T <= 0 || B <= 0 || H <= 0) return 0;
    return carve_seq_bwd_bf16(T, B, H, nullptr).bytes;
}
static int seq_bwd_bf16_prepare(hipStream_t st, const SeqBwdBf16WS& w, int T, int B, int H, const float* w_hh) {
    int rc;
    if ((rc = fill_zero(st, w.err, 64 * sizeof(int)))) return rc;
    if ((rc = split_planes(st, 1, true, w_hh, H, ID, 4 * H, H, w.wt.p, w.wt.ld, w.wt.kpad, (int)rows64((size_t)H)))) return rc;
    return zero_pad_cols_u16(st, w.dgb.p, (int64_t)T * B, w.dgb.ld, 4 * H, w.dgb.kpad);
}
// persistent: 0 = one launch per timestep, 1 = one persistent launch per `block` steps (0 = all T).
// stash_dg [T*B,4H]: activated gates in, fp32 dG out; dh_out rows for steps >= dh_first (nullable).
int s2vt_lstm_seq_bwd_bf16(int32_t T, int32_t B, int32_t H, const float* w_hh, const float* dh_out, int32_t dh_first,
                           const float* c_all, float* stash_dg, void* workspace, size_t workspace_bytes,
                           int32_t persistent, int32_t block, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh && c_all && stash_dg && workspace && dh_first >= 0, "s2vt_lstm_seq_bwd_bf16: bad arguments");
    const SeqBwdBf16WS w = carve_seq_bwd_bf16(T, B, H, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_lstm_seq_bwd_bf16: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = seq_bwd_bf16_prepare(st, w, T, B, H, w_hh))) return rc;
    if (!persistent) return seq_bwd_bf16(st, T, 0, T, B, H, w.wt, dh_out, dh_first, c_all, stash_dg, w.dgb, w.dc);
    S2VT_REQUIRE(lstm_seq_bwd_bf16_persist_supported(B, H, w.dgb.kpad), "s2vt_lstm_seq_bwd_bf16: shape not supported by the persistent kernel");
    const int blk = block > 0 ? block : T;
    for (int t1 = T; t1 > 0; t1 -= blk) {
        const int t0 = (t1 - blk > 0) ? t1 - blk : 0;
        ProfScope ps(st, K_STEP_BWD, t1 - t0);
        if ((rc = lstm_seq_bwd_bf16_persist2(st, seq_bwd_bf16_args(T, t0, t1, B, H, w.wt, w.dgb, dh_out, dh_first, c_all, stash_dg,
                                                                  w.dc, w.sync, w.err), nullptr)))
            return rc;
    }
    return 0;
}
// two independent layers of one shape, every block of both in ONE persistent launch; workspace = 2 x the single-layer size
int s2vt_lstm_seq_bwd_bf16_pair(int32_t T, int32_t B, int32_t H, const float* w_hh0, const float* w_hh1, const float* dh_out0,
                                const float* dh_out1, int32_t dh_first, const float* c_all0, const float* c_all1,
                                float* stash_dg0, float* stash_dg1, void* workspace, size_t workspace_bytes, int32_t block,
                                void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh0 && w_hh1 && c_all0 && c_all1 && stash_dg0 && stash_dg1 && workspace && dh_first >= 0,
                 "s2vt_lstm_seq_bwd_bf16_pair: bad arguments");
    const size_t one = carve_seq_bwd_bf16(T, B, H, nullptr).bytes;
    S2VT_REQUIRE(workspace_bytes >= 2 * one, "s2vt_lstm_seq_bwd_bf16_pair: workspace %zu < %zu bytes", workspace_bytes, 2 * one);
    const SeqBwdBf16WS w0 = carve_seq_bwd_bf16(T, B, H, workspace);
    const SeqBwdBf16WS w1 = carve_seq_bwd_bf16(T, B, H, reinterpret_cast<char*>(workspace) + one);
    S2VT_REQUIRE(lstm_seq_bwd_bf16_persist_supported(B, H, w0.dgb.kpad), "s2vt_lstm_seq_bwd_bf16_pair: shape not supported by the persistent kernel");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = seq_bwd_bf16_prepare(st, w0, T, B, H, w_hh0))) return rc;
    if ((rc = seq_bwd_bf16_prepare(st, w1, T, B, H, w_hh1))) return rc;
    const int blk = block > 0 ? block : T;
    for (int t1 = T; t1 > 0; t1 -= blk) {
        const int t0 = (t1 - blk > 0) ? t1 - blk : 0;
        ProfScope ps(st, K_STEP_BWD, 2 * (t1 - t0));
        const SeqBwdBf16Args a1 = seq_bwd_bf16_args(T, t0, t1, B, H, w1.wt, w1.dgb, dh_out1, dh_first, c_all1, stash_dg1, w1.dc, w1.sync, w0.err);
        if ((rc = lstm_seq_bwd_bf16_persist2(st, seq_bwd_bf16_args(T, t0, t1, B, H, w0.wt, w0.dgb, dh_out0, dh_first, c_all0, stash_dg0,
                                                                  w0.dc, w0.sync, w0.err), &a1)))
            return rc;
    }
    return 0;
}

#ifdef S2VT_EXPERIMENT_STAMPS
extern "C" int s2vt_experiment_set_stamps(unsigned long long* buf, int block) { g_xstamps = buf; g_xstamp_block = block; return 0; }
#endif

// Which recurrence kernels the whole-path train drivers would run for (B, H) in the current modes:
// 0 = one launch per timestep, 1 = persistent bf16, 2 = persistent exact-fp32 MFMA, 3 = persistent split precision.
int s2vt_recurrence_plan(int32_t B, int32_t H, int32_t* fwd, int32_t* bwd) {
    S2VT_REQUIRE(B > 0 && H > 0 && fwd && bwd, "s2vt_recurrence_plan: bad arguments");
    *fwd = *bwd = 0;
    const int gm = gemm_mode();
    if (pipe_block() <= 0 || gm == 0 || B % 64 != 0) return 0;     // (the plane drivers run at B % 64 == 0 in gemm modes 1 and 3)
    if (gm == 1) {
        if (persist_on() && lstm_seq_fwd_bf16_persist_supported(B, H, pad64(H))) *fwd = 1;
        if (persist_on() && lstm_seq_bwd_bf16_persist_supported(B, H, pad64(4 * H))) *bwd = 1;
        return 0;
    }
    if (H <= 1024 && persist_x3_fwd_on() && lstm_seq_fwd_x3_persist_supported(B, H)) *fwd = 3;
    if (H <= 1024 && persist_x3_bwd_on(B, H) && lstm_seq_bwd_x3_persist_supported(B, H)) *bwd = 3;
    return 0;
}

int s2vt_set_recurrence_mode(int32_t mode) {
    return option_set(O_PERSIST, mode < 0 ? -1 : (mode ? 1 : 0));
}

// split-precision persistent forward (lstm_persist_x3.hip) as its own entry point.
// workspace: [err int x64][sync A][sync B][W planes 0][W planes 1][h planes 0][h planes 1]
size_t s2vt_lstm_seq_x3_workspace_bytes(int32_t T, int32_t B, int32_t H) {
    if (T <= 0 || B <= 0 || H <= 0 || H > 1024 || !lstm_seq_fwd_x3_persist_supported(B, H)) return 0;      // 0: shape not supported
    const size_t Kp = (size_t)(H + 63) / 64 * 64;
    return 256 + 2 * lstm_persist_sync_bytes() + 2 * align_up(3 * 4 * (size_t)H * Kp * 2, 256) + 2 * align_up(3 * (size_t)T * B * Kp * 2, 256);
}
int s2vt_lstm_seq_fwd_x3_persist(int32_t T, int32_t B, int32_t H, float* gx_stash0, float* gx_stash1, int32_t n_gx,
                                 const float* bias0, const float* bias1, const float* w_hh0, const float* w_hh1, float* h_all0,
                                 float* h_all1, float* c_all0, float* c_all1, int32_t block, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && gx_stash0 && w_hh0 && h_all0 && c_all0 && workspace && n_gx >= 0 && n_gx <= T &&
                     (n_gx == T || bias0), "s2vt_lstm_seq_fwd_x3_persist: bad arguments");
    S2VT_REQUIRE(H <= 1024 && lstm_seq_fwd_x3_persist_supported(B, H), "s2vt_lstm_seq_fwd_x3_persist: shape not supported");
    S2VT_REQUIRE(workspace_bytes >= s2vt_lstm_seq_x3_workspace_bytes(T, B, H), "s2vt_lstm_seq_fwd_x3_persist: workspace too small");
    const bool two = gx_stash1 != nullptr;
    S2VT_REQUIRE(!two || (w_hh1 && h_all1 && c_all1 && (n_gx == T || bias1)), "s2vt_lstm_seq_fwd_x3_persist: second layer incomplete");
    const int64_t Kp = (H + 63) / 64 * 64;
    const size_t wbytes = align_up(3 * 4 * (size_t)H * Kp * 2, 256), hbytes = align_up(3 * (size_t)T * B * Kp * 2, 256);
    char* base = reinterpret_cast<char*>(workspace);
    int* err = reinterpret_cast<int*>(base);
    unsigned int* sa = reinterpret_cast<unsigned int*>(base + 256);
    unsigned int* sb = reinterpret_cast<unsigned int*>(base + 256 + lstm_persist_sync_bytes());
    char* q = base + 256 + 2 * lstm_persist_sync_bytes();
    unsigned short* wp[2] = {reinterpret_cast<unsigned short*>(q), reinterpret_cast<unsigned short*>(q + wbytes)};
    unsigned short* hp[2] = {reinterpret_cast<unsigned short*>(q + 2 * wbytes), reinterpret_cast<unsigned short*>(q + 2 * wbytes + hbytes)};
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = fill_zero(st, err, 256))) return rc;
    if ((rc = split3_rows(st, w_hh0, H, 4 * H, H, (int)Kp, wp[0], 4 * (int64_t)H * Kp))) return rc;
    if (two && (rc = split3_rows(st, w_hh1, H, 4 * H, H, (int)Kp, wp[1], 4 * (int64_t)H * Kp))) return rc;
    const int blk = block > 0 ? block : T;
    for (int t0 = 0; t0 < T; t0 += blk) {
        const int t1 = (t0 + blk < T) ? t0 + blk : T;
        ProfScope ps(st, K_STEP_FWD, (two ? 2 : 1) * (t1 - t0));
        const SeqFwdX3Args a0 = persist_fwd_x3_args(t0, t1, B, H, T, Kp, gx_stash0, n_gx, bias0, wp[0], hp[0], h_all0, c_all0, sa, err);
        SeqFwdX3Args a1;
        if (two) a1 = persist_fwd_x3_args(t0, t1, B, H, T, Kp, gx_stash1, n_gx, bias1, wp[1], hp[1], h_all1, c_all1, sb, err);
        if ((rc = lstm_seq_fwd_x3_persist2(st, a0, two ? &a1 : nullptr))) return rc;
    }
    return 0;
}
// split-precision persistent BPTT as its own entry point.
// workspace: [err int x64][sync A][sync B] then per layer [W_hh^T fp32][W_hh^T planes][dc][partial-sum ring]
static size_t bwd_x3_ws_layer_bytes(int T, int B, int H, int nslots) {
    const size_t Kp = (size_t)(H + 63) / 64 * 64, Hp = (size_t)(H + 15) / 16 * 16;
    return align_up((size_t)H * 4 * H * 4, 256) + align_up(3 * Kp * 4 * Hp * 2, 256) + align_up((size_t)B * H * 4, 256) +
           align_up((size_t)nslots * lstm_seq_bwd_x3_part_slot_floats(B, H) * 4, 256);
}
size_t s2vt_lstm_seq_bwd_x3_workspace_bytes(int32_t T, int32_t B, int32_t H, int32_t block) {
    if (T <= 0 || B <= 0 || H <= 0 || H > 1024 || B % 32) return 0;
    const int blk = (block > 0 && block < T) ? block : T;
    return 256 + 2 * lstm_persist_sync_bytes() + 2 * bwd_x3_ws_layer_bytes(T, B, H, blk + 1);
}
int s2vt_lstm_seq_bwd_x3_persist(int32_t T, int32_t B, int32_t H, const float* w_hh0, const float* w_hh1, const float* dh_out0,
                                 const float* dh_out1, int32_t dh_first, const float* c_all0, const float* c_all1, float* stash_dg0,
                                 float* stash_dg1, int32_t block, void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh0 && c_all0 && stash_dg0 && workspace && dh_first >= 0,
                 "s2vt_lstm_seq_bwd_x3_persist: bad arguments");
    S2VT_REQUIRE(H <= 1024 && lstm_seq_bwd_x3_persist_supported(B, H), "s2vt_lstm_seq_bwd_x3_persist: shape not supported");
    S2VT_REQUIRE(workspace_bytes >= s2vt_lstm_seq_bwd_x3_workspace_bytes(T, B, H, block), "s2vt_lstm_seq_bwd_x3_persist: workspace too small");
    const bool two = stash_dg1 != nullptr;
    S2VT_REQUIRE(!two || (w_hh1 && c_all1), "s2vt_lstm_seq_bwd_x3_persist: second layer incomplete");
    const int blk = (block > 0 && block < T) ? block : T;
    const int64_t Kp = (H + 63) / 64 * 64, Hp = (H + 15) / 16 * 16;
    const int64_t pslot = (int64_t)lstm_seq_bwd_x3_part_slot_floats(B, H);
    char* base = reinterpret_cast<char*>(workspace);
    int* err = reinterpret_cast<int*>(base);
    unsigned int* sy[2] = {reinterpret_cast<unsigned int*>(base + 256), reinterpret_cast<unsigned int*>(base + 256 + lstm_persist_sync_bytes())};
    char* q = base + 256 + 2 * lstm_persist_sync_bytes();
    float* wt[2]; unsigned short* wtp[2]; float* dc[2]; float* part[2];
    for (int l = 0; l < 2; ++l) {
        wt[l] = reinterpret_cast<float*>(q); q += align_up((size_t)H * 4 * H * 4, 256);
        wtp[l] = reinterpret_cast<unsigned short*>(q); q += align_up(3 * (size_t)Kp * 4 * Hp * 2, 256);
        dc[l] = reinterpret_cast<float*>(q); q += align_up((size_t)B * H * 4, 256);
        part[l] = reinterpret_cast<float*>(q); q += align_up((size_t)(blk + 1) * pslot * 4, 256);
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = fill_zero(st, err, 256))) return rc;
    const float* whh[2] = {w_hh0, w_hh1};
    for (int l = 0; l < (two ? 2 : 1); ++l) {
        if ((rc = transpose_f32(st, whh[l], 4 * H, H, wt[l]))) return rc;
        if ((rc = split3_wt(st, wt[l], H, (int)Kp, (int)Hp, wtp[l], Kp * 4 * Hp))) return rc;
    }
    for (int t1 = T; t1 > 0; t1 -= blk) {
        const int t0 = (t1 - blk > 0) ? t1 - blk : 0;
        ProfScope ps(st, K_STEP_BWD, (two ? 2 : 1) * (t1 - t0));
        const SeqBwdX3Args a0 = persist_bwd_x3_args(T, t0, t1, B, H, Kp, Hp, wtp[0], dh_out0, dh_first, c_all0, stash_dg0, dc[0],
                                                    part[0], pslot, blk + 1, sy[0], err);
        SeqBwdX3Args a1;
        if (two) a1 = persist_bwd_x3_args(T, t0, t1, B, H, Kp, Hp, wtp[1], dh_out1, dh_first, c_all1, stash_dg1, dc[1],
                                          part[1], pslot, blk + 1, sy[1], err);
        if ((rc = lstm_seq_bwd_x3_persist2(st, a0, two ? &a1 : nullptr))) return rc;
    }
    return 0;
}

}
