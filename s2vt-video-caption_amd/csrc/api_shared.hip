// Helpers the whole-path drivers and the per-op entry points share: the launch-per-timestep recurrence loops, pipeline block
// boundaries, plane-operand wrappers (split kernels, plane GEMMs) and the argument builders of the persistent recurrence kernels.
#include "api_internal.h"

namespace s2vt {

// LSTM layer forward over steps [t0, t1) (time-major buffers, zero initial state at t = 0).
int seq_fwd(hipStream_t st, int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
                   const float* w_hh, float* h_all, float* c_all, bool write_stash) {
    if (t1 <= t0) return 0;
    ProfScope ps(st, K_STEP_FWD, t1 - t0);
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    for (int t = t0; t < t1; ++t) {
        StepFwdArgs a;
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = H;
        a.h_prev = t ? h_all + (t - 1) * BH : nullptr; a.ldh = H;
        a.w_hh = w_hh; a.ldw = H;
        a.gx = (t < n_gx) ? gx_stash + t * B4H : nullptr; a.ldgx = 4 * (int64_t)H;
        a.bias = bias;
        a.c_prev = t ? c_all + (t - 1) * BH : nullptr; a.ldc = H;
        a.h_out = h_all + t * BH; a.ldho = H;
        a.c_out = c_all + t * BH; a.ldco = H;
        a.stash = write_stash ? gx_stash + t * B4H : nullptr; a.ldst = 4 * (int64_t)H;
        a.tok_const = 0;
        int rc = lstm_step_fwd(st, a);
        if (rc) return rc;
    }
    return 0;
}

// BPTT over steps t1-1 .. t0 of a T-step layer; stash_dg [T*B,4H] holds activated gates on entry, dG on exit.
int seq_bwd(hipStream_t st, int T, int t0, int t1, int B, int H, const float* w_hh_t, const float* dh_out,
                   int dh_first, const float* c_all, float* stash_dg, float* dc) {
    if (t1 <= t0) return 0;
    ProfScope ps(st, K_STEP_BWD, t1 - t0);
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    for (int t = t1 - 1; t >= t0; --t) {
        StepBwdArgs a;
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = H;
        a.dg_next = (t < T - 1) ? stash_dg + (t + 1) * B4H : nullptr; a.lddg = 4 * (int64_t)H;
        a.w_hh_t = w_hh_t; a.ldwt = 4 * (int64_t)H;
        a.dh_out = (dh_out && t >= dh_first) ? dh_out + (int64_t)(t - dh_first) * BH : nullptr; a.lddho = H;
        a.stash = stash_dg + t * B4H; a.ldst = 4 * (int64_t)H;
        a.c = c_all + t * BH; a.ldc = H;
        a.c_prev = t ? c_all + (t - 1) * BH : nullptr; a.ldcp = H;
        a.dc = dc; a.lddc = H;
        a.dc_is_zero = (t == T - 1) ? 1 : 0;
        a.dg = stash_dg + t * B4H; a.lddg_out = 4 * (int64_t)H;
        int rc = lstm_step_bwd(st, a);
        if (rc) return rc;
    }
    return 0;
}

// Block length of the ONE-stream persistent bf16 schedule: a launch of stage k runs block k of one layer next to block k-1 of the
// other, and lasts as long as the LONGER of the two - with 32-step blocks over L = 80 frames the blocks are 32, 32, 16 | 32, 32, 15
// and the seven launches of a pass cover 207 timestep slots for 159 timesteps; equal blocks (27, 27, 26 | 27, 27, 25) cover 187
// (measured: BPTT 2.93 -> 2.65 ms, forward 1.89 -> 1.76 ms per config-3 step).  Only the default block (32) is rebalanced; an
// explicit S2VT_PIPE_BLOCK / s2vt_set_pipeline_block value is taken as given.
int balanced_block(int L, int blk) {
    if (blk != 32 || L <= 0) return blk;
    const int n = (L + 31) / 32;
    return (L + n - 1) / n;
}
// Block boundaries over [0, T) with L (first caption step) forced to be a boundary.
std::vector<int> pipe_bounds(int T, int L, int blk) {
    std::vector<int> b;
    if (blk <= 0) { b.push_back(0); b.push_back(L); b.push_back(T); return b; }
    for (int t = 0; t < L; t += blk) b.push_back(t);
    for (int t = L; t < T; t += blk) b.push_back(t);
    b.push_back(T);
    return b;
}

int XP = 3;        // planes per operand of the running plane driver (3 or 1); set by the entry points

// rows [r0, r0+rows) of the operand <- planes of in[rows][cols]
int psplit(const Lane& ln, const PB& dst, int r0, const float* in, int64_t ld, RowMap imap, int rows, int cols) {
    return split_planes(ln.s, XP, false, in, ld, imap, rows, cols, dst.p + (int64_t)r0 * dst.ld, dst.ld, dst.kpad, rows);
}
// one pass over in[rows][cols]: row planes into r (operand rows r0..), transposed planes into t (k range k0..),
// 64-row partial column sums into colpart (each may be null)
int pdual(const Lane& ln, const float* in, int64_t ld, RowMap imap, int rows, int cols, const PB* r, int r0,
                 const PB* t, int k0, float* colpart) {
    if (!r && !t && !colpart) return 0;          // (bf16 mode with transposed-read GEMMs: the recurrence kernels wrote the rows already)
    return split_planes_dual(ln.s, XP, in, ld, imap, rows, cols, r ? r->p + (int64_t)r0 * r->ld : nullptr, r ? r->ld : 0,
                             r ? r->kpad : 0, t ? t->p + koff(k0) : nullptr, t ? t->ld : 0, t ? pad64(rows) : 0,
                             colpart);
}
// C[M,N] (+)= A[rows a0.., k ka..ka+K) · B[rows b0.., k kb..kb+K)^T
int pgemm(const Lane& ln, int M, int N, int K, const PB& A, int a0, int ka, const PB& B, int b0, int kb, float* C,
                 int64_t ldc, RowMap cm, const float* bias, bool acc) {
    ProfScope ps(ln.s, cu_plan_cap_current() ? K_GEMM_CORUN : K_GEMM, 1);
    return gemm_bf16_nt(ln.s, XP, M, N, pad64(K), A.p + (int64_t)a0 * A.ld + koff(ka), A.ld,
                        B.p + (int64_t)b0 * B.ld + koff(kb), B.ld, C, ldc, cm, bias, acc, ln.gws, ln.gws_floats);
}

// The weight-gradient GEMMs (dW = dG^T h, dW_o = dlogits^T h2) read BOTH operands transposed from the row planes the forward / the
// BPTT hand-over already wrote (gemm_x3_kernel<MI, true>, gemm_b1_kernel<4, true>): transposed twins of dG, dlogits, h, x1 and the
// embedded words are never written.
// C[M,N] = A_img[a_row0 .., :M]^T . B_img[b_row0 .., :N] over K image rows (row offsets: multiples of 64)
int pgemm_tt(const Lane& ln, int M, int N, int K, const PB& A, int a_row0, const PB& B, int b_row0, float* C, int64_t ldc,
                    RowMap cm, const float* bias, bool acc) {
    ProfScope ps(ln.s, cu_plan_cap_current() ? K_GEMM_CORUN : K_GEMM, 1);
    if (XP == 1)
        return gemm_b1_tt(ln.s, M, N, K, A.p + (int64_t)a_row0 * A.ld, A.ld, B.p + (int64_t)b_row0 * B.ld, B.ld, C, ldc, cm, bias, acc,
                          ln.gws, ln.gws_floats);
    return gemm_x3_tt(ln.s, M, N, K, A.p + (int64_t)a_row0 * A.ld, A.ld, B.p + (int64_t)b_row0 * B.ld, B.ld, C, ldc, cm, bias, acc,
                      ln.gws, ln.gws_floats);
}

// bf16-operand layer forward over steps [t0, t1): hb = bf16 row images of h (time-major, ld = hb.ld), the k-major
// plane the batched GEMMs read as well
int seq_fwd_bf16(hipStream_t st, int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
                        const PB& wb, const PB& hb, float* h_all, float* c_all) {
    if (t1 <= t0) return 0;
    ProfScope ps(st, K_STEP_FWD, t1 - t0);
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    for (int t = t0; t < t1; ++t) {
        StepFwdBf16Args a;
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = H; a.Kp = hb.kpad;
        a.hb_prev = t ? hb.p + (int64_t)(t - 1) * B * hb.ld : nullptr; a.ldhb = hb.ld;
        a.wb = wb.p; a.ldwb = wb.ld;
        a.gx = (t < n_gx) ? gx_stash + t * B4H : nullptr; a.ldgx = 4 * (int64_t)H;
        a.bias = bias;
        a.c_prev = t ? c_all + (t - 1) * BH : nullptr; a.ldc = H;
        a.h_out = h_all + t * BH; a.ldho = H;
        a.hb_out = hb.p + (int64_t)t * B * hb.ld; a.ldhbo = hb.ld;
        a.c_out = c_all + t * BH; a.ldco = H;
        a.stash = gx_stash + t * B4H; a.ldst = 4 * (int64_t)H;
        int rc = lstm_step_fwd_bf16(st, a);
        if (rc) return rc;
    }
    return 0;
}
// bf16-operand BPTT over steps t1-1 .. t0: dgb = bf16 row images of dG (time-major), wt = W_hh^T bf16 rows
int seq_bwd_bf16(hipStream_t st, int T, int t0, int t1, int B, int H, const PB& wt, const float* dh_out,
                        int dh_first, const float* c_all, float* stash_dg, const PB& dgb, float* dc) {
    if (t1 <= t0) return 0;
    ProfScope ps(st, K_STEP_BWD, t1 - t0);
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    for (int t = t1 - 1; t >= t0; --t) {
        StepBwdBf16Args a;
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = H; a.Kp = dgb.kpad;
        a.dgb_next = (t < T - 1) ? dgb.p + (int64_t)(t + 1) * B * dgb.ld : nullptr; a.lddgb = dgb.ld;
        a.wtb = wt.p; a.ldwtb = wt.ld;
        a.dh_out = (dh_out && t >= dh_first) ? dh_out + (int64_t)(t - dh_first) * BH : nullptr; a.lddho = H;
        a.stash = stash_dg + t * B4H; a.ldst = 4 * (int64_t)H;
        a.c = c_all + t * BH; a.ldc = H;
        a.c_prev = t ? c_all + (t - 1) * BH : nullptr; a.ldcp = H;
        a.dc = dc; a.lddc = H;
        a.dc_is_zero = (t == T - 1) ? 1 : 0;
        a.dg = stash_dg + t * B4H; a.lddg = 4 * (int64_t)H;
        a.dgb = dgb.p + (int64_t)t * B * dgb.ld; a.lddgbo = dgb.ld;
        int rc = lstm_step_bwd_bf16(st, a);
        if (rc) return rc;
    }
    return 0;
}

SeqBwdX3Args persist_bwd_x3_args(int T, int t0, int t1, int B, int H, int64_t Kp, int64_t Hp, const unsigned short* wtp,
                                        const float* dh_out, int dh_first, const float* c_all, float* stash_dg, float* dc,
                                        float* part, int64_t part_slot, int nslots, unsigned int* sync, int* err) {
    SeqBwdX3Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.Kp = (int)Kp; a.Hp = (int)Hp; a.T = T; a.t0 = t0; a.t1 = t1;
    a.wtp = wtp; a.wplane = Kp * 4 * Hp; a.ldw = 4 * Hp;
    a.dh_out = dh_out; a.dh_first = dh_first;
    a.stash_dg = stash_dg; a.c_all = c_all; a.dc = dc;
    a.part = part; a.part_slot = part_slot; a.nslots = nslots;
    a.sync = sync; a.err = err;
#ifdef S2VT_EXPERIMENT_STAMPS
    a.stamps = g_xstamps; a.stamp_block = g_xstamp_block;
#endif
    return a;
}
SeqFwdX3Args persist_fwd_x3_args(int t0, int t1, int B, int H, int T, int64_t Kp, float* gx_stash, int n_gx, const float* bias,
                                        const unsigned short* wp, unsigned short* hp, float* h_all, float* c_all,
                                        unsigned int* sync, int* err) {
    SeqFwdX3Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.Kp = (int)Kp; a.t0 = t0; a.t1 = t1; a.n_gx = n_gx;
    a.wp = wp; a.wplane = 4 * (int64_t)H * Kp; a.ldw = Kp;
    a.hp = hp; a.hplane = (int64_t)T * B * Kp; a.ldh = Kp;
    a.h_all = h_all; a.gx_stash = gx_stash; a.bias = bias; a.c_all = c_all;
    a.sync = sync; a.err = err;
#ifdef S2VT_EXPERIMENT_STAMPS
    a.stamps = g_xstamps; a.stamp_block = g_xstamp_block;
#endif
    return a;
}
bool persist_fwd_ok(int B, int H, const PB& wb, const PB& hb) {
    return persist_on() && lstm_seq_fwd_bf16_persist_supported(B, H, hb.kpad) && hb.kpad == wb.kpad;
}
SeqFwdBf16Args persist_fwd_args(int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
                                       const PB& wb, const PB& hb, float* h_all, float* c_all, unsigned int* sync, int* err) {
    SeqFwdBf16Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.Kp = hb.kpad;
    a.t0 = t0; a.t1 = t1; a.n_gx = n_gx;
    a.wb = wb.p; a.ldwb = wb.ld;
    a.hb = hb.p; a.ldhb = hb.ld;
    a.gx_stash = gx_stash; a.bias = bias;
    a.h_all = h_all; a.c_all = c_all;
    a.sync = sync; a.err = err;
    return a;
}

SeqBwdBf16Args seq_bwd_bf16_args(int T, int t0, int t1, int B, int H, const PB& wt, const PB& dgb, const float* dh_out,
                                        int dh_first, const float* c_all, float* stash_dg, float* dc, unsigned int* sync, int* err) {
    SeqBwdBf16Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.Kp = dgb.kpad;
    a.T = T; a.t0 = t0; a.t1 = t1;
    a.wtb = wt.p; a.ldwtb = wt.ld;
    a.dgb = dgb.p; a.lddgb = dgb.ld;
    a.dh_out = dh_out; a.dh_first = dh_first;
    a.stash_dg = stash_dg; a.c_all = c_all; a.dc = dc;
    a.sync = sync; a.err = err;
#ifdef S2VT_EXPERIMENT_STAMPS
    a.stamps = g_xstamps; a.stamp_block = g_xstamp_block;
#endif
    return a;
}
}  // namespace s2vt
