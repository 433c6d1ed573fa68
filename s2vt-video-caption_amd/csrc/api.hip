// C-ABI drivers of libs2vt_hip.so (declared in include/s2vt_hip.h): argument checking, workspace
// carving and the stream-ordered launch sequences of the S2VT train forward/backward and greedy
// decode.  Host code only; every kernel lives in gemm/lstm/ce/misc.hip.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "../../include/s2vt_hip.h"
#include "common.h"
#include "kernels.h"

namespace s2vt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

// ------------------------------------------------------------------ live kernel timing
enum { K_GEMM = 0, K_STEP_FWD = 1, K_STEP_BWD = 2, K_CE = 3, K_ARGMAX = 4, K_NKINDS = 5 };
struct ProfRec { hipEvent_t a, b; int kind; int64_t launches; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;

struct ProfScope {
    hipStream_t s; bool on; ProfRec r;
    ProfScope(hipStream_t stream, int kind, int64_t launches) : s(stream), on(g_prof) {
        if (!on) return;
        if (!g_free.empty()) {
            r.a = g_free.back().first; r.b = g_free.back().second; g_free.pop_back();
        } else {
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
        }
        r.kind = kind; r.launches = launches;
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        g_recs.push_back(r);
    }
};

static const RowMap ID = {nullptr, 0, 0};
static inline RowMap perm(int inner, int outer) { return RowMap{nullptr, inner, outer}; }
static inline RowMap gather(const int32_t* idx) { return RowMap{idx, 0, 0}; }

// split-K scratch of the driver that is running (set by the whole-path entry points from their workspace)
static thread_local float* g_gws = nullptr;
static thread_local size_t g_gws_floats = 0;
struct GemmWsScope {
    GemmWsScope(float* p, size_t n) { g_gws = p; g_gws_floats = n; }
    ~GemmWsScope() { g_gws = nullptr; g_gws_floats = 0; }
};

static int gemm(hipStream_t st, bool ak, bool bk, int M, int N, int K, const float* A, int64_t lda, RowMap am,
                const float* B, int64_t ldb, RowMap bm, float* C, int64_t ldc, RowMap cm, const float* bias,
                bool acc) {
    ProfScope ps(st, K_GEMM, 1);
    return gemm_f32(st, ak, bk, M, N, K, A, lda, am, B, ldb, bm, C, ldc, cm, bias, acc, g_gws, g_gws_floats);
}

// scratch floats for split-K slabs: up to 4 slices of the largest small-grid GEMM output of the path
static size_t gemm_ws_floats(const s2vt_dims& d) {
    const size_t B = d.B, L = d.L, F = d.F, H = d.H, E = d.E, T = 2 * L - 1;
    size_t m = T * B * H;                       // dh1 / x1-like activations
    if (4 * H * (E + H) > m) m = 4 * H * (E + H);
    if (H * F > m) m = H * F;
    if (L * B * F / 4 > m) m = L * B * F / 4;   // dfeats (rarely split)
    return 4 * m;
}

// ------------------------------------------------------------------ workspace carving
struct Carver {
    char* base; size_t off; size_t cap;
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += n * sizeof(T);
        return p;
    }
};

struct TrainWS {
    float *bsum1, *bsum2, *x1, *s1, *h1, *c1, *s2, *h2, *c2;
    float *wt, *dh1, *dh2dec, *dx1, *de, *dc, *colsum, *gws;
    size_t gws_floats;
    int32_t* tok;
    int* err;
    size_t bytes;
};

static bool dims_ok(const s2vt_dims* d) {
    return d && d->B > 0 && d->L > 1 && d->F > 0 && d->H > 0 && d->E > 0 && d->V > 0;
}

static TrainWS carve_train(const s2vt_dims& d, void* base) {
    const size_t B = d.B, L = d.L, H = d.H, E = d.E, V = d.V, T = 2 * L - 1;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    TrainWS w;
    w.bsum1 = c.take<float>(4 * H);
    w.bsum2 = c.take<float>(4 * H);
    w.x1 = c.take<float>(L * B * H);
    w.s1 = c.take<float>(T * B * 4 * H);
    w.h1 = c.take<float>(T * B * H);
    w.c1 = c.take<float>(T * B * H);
    w.s2 = c.take<float>(T * B * 4 * H);
    w.h2 = c.take<float>(T * B * H);
    w.c2 = c.take<float>(T * B * H);
    w.tok = c.take<int32_t>((L - 1) * B);
    w.err = c.take<int>(4);
    // backward-only scratch
    w.wt = c.take<float>(H * 4 * H);
    w.dh1 = c.take<float>(T * B * H);
    w.dh2dec = c.take<float>((L - 1) * B * H);
    w.dx1 = c.take<float>(L * B * H);
    w.de = c.take<float>((L - 1) * B * E);
    w.dc = c.take<float>(B * H);
    size_t cs = colsum_partial_floats((int64_t)T * B, (int)(4 * H));
    size_t cs2 = colsum_partial_floats((int64_t)(L - 1) * B, (int)V);
    size_t cs3 = colsum_partial_floats((int64_t)L * B, (int)H);
    w.colsum = c.take<float>(cs > cs2 ? (cs > cs3 ? cs : cs3) : (cs2 > cs3 ? cs2 : cs3));
    w.gws_floats = gemm_ws_floats(d);
    w.gws = c.take<float>(w.gws_floats);
    w.bytes = align_up(c.off, 256);
    return w;
}

// One LSTM layer forward over T steps (time-major buffers).
static int seq_fwd(hipStream_t st, int T, int B, int H, float* gx_stash, int n_gx, const float* bias,
                   const float* w_hh, float* h_all, float* c_all, bool write_stash) {
    ProfScope ps(st, K_STEP_FWD, T);
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    for (int t = 0; t < T; ++t) {
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

// BPTT over one layer; stash_dg [T*B,4H] holds the activated gates on entry and dG on exit.
static int seq_bwd(hipStream_t st, int T, int B, int H, const float* w_hh_t, const float* dh_out, int dh_first,
                   const float* c_all, float* stash_dg, float* dc) {
    ProfScope ps(st, K_STEP_BWD, T);
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    for (int t = T - 1; t >= 0; --t) {
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

}  // namespace s2vt

using namespace s2vt;

extern "C" {

int s2vt_abi_version(void) { return S2VT_ABI_VERSION; }
const char* s2vt_last_error(void) { return g_err; }

size_t s2vt_train_workspace_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    return carve_train(*d, nullptr).bytes;
}

int s2vt_train_forward(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                       int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && targets && logits && workspace, "s2vt_train_forward: null/invalid argument");
    const TrainWS w = carve_train(*d, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_train_forward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    GemmWsScope gscope(w.gws, w.gws_floats);
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1;
    const int64_t BH = (int64_t)B * H;
    int rc;
    if ((rc = add_vectors(st, p->vid_b_ih, p->vid_b_hh, w.bsum1, 4 * H))) return rc;
    if ((rc = add_vectors(st, p->word_b_ih, p->word_b_hh, w.bsum2, 4 * H))) return rc;
    if ((rc = fill_zero(st, w.err, 4 * sizeof(int)))) return rc;
    if ((rc = targets_to_time_major(st, targets, B, L - 1, targets_ld, V, w.tok, w.err))) return rc;
    // x1 (time-major) = feats·W_f^T + b_f                                     S2VTModel.py:54
    if ((rc = gemm(st, true, true, B * L, H, F, feats, F, ID, p->feat_w, F, ID, w.x1, H, perm(L, B), p->feat_b, false)))
        return rc;
    // gx1 = x1·W_ih1^T + (b_ih1 + b_hh1) for the L real frames                 S2VTModel.py:64-67
    if ((rc = gemm(st, true, true, L * B, 4 * H, H, w.x1, H, ID, p->vid_w_ih, H, ID, w.s1, 4 * H, ID, w.bsum1, false)))
        return rc;
    if ((rc = seq_fwd(st, T, B, H, w.s1, L, w.bsum1, p->vid_w_hh, w.h1, w.c1, true))) return rc;
    // gx2 = [embed | h1]·W_ih2^T + biases: vid_out half for all T steps, embed half for steps >= L   :71-77
    if ((rc = gemm(st, true, true, T * B, 4 * H, H, w.h1, H, ID, p->word_w_ih + E, E + H, ID, w.s2, 4 * H, ID, w.bsum2,
                   false)))
        return rc;
    if ((rc = gemm(st, true, true, (L - 1) * B, 4 * H, E, p->emb_w, E, gather(w.tok), p->word_w_ih, E + H, ID,
                   w.s2 + (int64_t)L * B * 4 * H, 4 * H, ID, nullptr, true)))
        return rc;
    if ((rc = seq_fwd(st, T, B, H, w.s2, T, w.bsum2, p->word_w_hh, w.h2, w.c2, true))) return rc;
    // logits[b, j, :] = h2[L + j]·W_o^T + b_o                                   S2VTModel.py:78-80
    if ((rc = gemm(st, true, true, (L - 1) * B, V, H, w.h2 + L * BH, H, ID, p->out_w, H, ID, logits, V, perm(B, L - 1),
                   p->out_b, false)))
        return rc;
    return 0;
}

int s2vt_train_backward(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                        const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && dlogits && g && workspace, "s2vt_train_backward: null/invalid argument");
    const TrainWS w = carve_train(*d, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_train_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    GemmWsScope gscope(w.gws, w.gws_floats);
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    const int R = (L - 1) * B;
    int rc;
    // ---- out_linear                                                        (autograd of S2VTModel.py:80)
    if ((rc = gemm(st, true, false, R, H, V, dlogits, V, ID, p->out_w, H, ID, w.dh2dec, H, perm(L - 1, B), nullptr, false)))
        return rc;
    if ((rc = gemm(st, false, false, V, H, R, dlogits, V, ID, w.h2 + L * BH, H, perm(L - 1, B), g->out_w, H, ID, nullptr,
                   false)))
        return rc;
    if ((rc = colsum_f32(st, dlogits, R, V, V, w.colsum, g->out_b, false))) return rc;
    // ---- word_rnn BPTT                                                     (autograd of :77)
    if ((rc = transpose_f32(st, p->word_w_hh, 4 * H, H, w.wt))) return rc;
    if ((rc = seq_bwd(st, T, B, H, w.wt, w.dh2dec, L, w.c2, w.s2, w.dc))) return rc;
    if ((rc = gemm(st, false, false, 4 * H, H, (T - 1) * B, w.s2 + B4H, 4 * H, ID, w.h2, H, ID, g->word_w_hh, H, ID,
                   nullptr, false)))
        return rc;
    if ((rc = gemm(st, false, false, 4 * H, H, T * B, w.s2, 4 * H, ID, w.h1, H, ID, g->word_w_ih + E, E + H, ID, nullptr,
                   false)))
        return rc;
    // dW_ih2[:, :E] = dG2[L..]^T · Emb[tok]: the embedded rows are gathered once (time-major) into w.de, which
    // is free until the d(embedded words) GEMM below overwrites it
    if ((rc = gather_rows_f32(st, p->emb_w, E, w.tok, R, E, w.de))) return rc;
    if ((rc = gemm(st, false, false, 4 * H, E, R, w.s2 + (int64_t)L * B4H, 4 * H, ID, w.de, E, ID, g->word_w_ih, E + H,
                   ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(st, w.s2, (int64_t)T * B, 4 * H, 4 * H, w.colsum, g->word_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->word_b_hh, g->word_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, st));
    // gradient into vid_out (dh1) and into the embedded words                (autograd of :71-75)
    if ((rc = gemm(st, true, false, T * B, H, 4 * H, w.s2, 4 * H, ID, p->word_w_ih + E, E + H, ID, w.dh1, H, ID, nullptr,
                   false)))
        return rc;
    if ((rc = gemm(st, true, false, R, E, 4 * H, w.s2 + (int64_t)L * B4H, 4 * H, ID, p->word_w_ih, E + H, ID, w.de, E, ID,
                   nullptr, false)))
        return rc;
    if ((rc = fill_zero(st, g->emb_w, sizeof(float) * (size_t)V * E))) return rc;
    if ((rc = embedding_scatter_add(st, w.de, R, E, w.tok, g->emb_w))) return rc;
    // ---- vid_rnn BPTT                                                      (autograd of :67)
    if ((rc = transpose_f32(st, p->vid_w_hh, 4 * H, H, w.wt))) return rc;
    if ((rc = seq_bwd(st, T, B, H, w.wt, w.dh1, 0, w.c1, w.s1, w.dc))) return rc;
    if ((rc = gemm(st, false, false, 4 * H, H, (T - 1) * B, w.s1 + B4H, 4 * H, ID, w.h1, H, ID, g->vid_w_hh, H, ID,
                   nullptr, false)))
        return rc;
    if ((rc = gemm(st, false, false, 4 * H, H, L * B, w.s1, 4 * H, ID, w.x1, H, ID, g->vid_w_ih, H, ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(st, w.s1, (int64_t)T * B, 4 * H, 4 * H, w.colsum, g->vid_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->vid_b_hh, g->vid_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, st));
    // ---- feat_linear                                                       (autograd of :54)
    if ((rc = gemm(st, true, false, L * B, H, 4 * H, w.s1, 4 * H, ID, p->vid_w_ih, H, ID, w.dx1, H, ID, nullptr, false)))
        return rc;
    if ((rc = gemm(st, false, false, H, F, L * B, w.dx1, H, ID, feats, F, perm(B, L), g->feat_w, F, ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(st, w.dx1, (int64_t)L * B, H, H, w.colsum, g->feat_b, false))) return rc;
    if (dfeats) {
        if ((rc = gemm(st, true, false, L * B, F, H, w.dx1, H, ID, p->feat_w, F, ID, dfeats, F, perm(B, L), nullptr, false)))
            return rc;
    }
    return 0;
}

// ------------------------------------------------------------------ greedy decode
struct DecodeWS {
    float *bsum1, *bsum2, *x1, *gx1, *h1, *c1, *gx2, *h2, *c2, *gws;
    size_t gws_floats;
    unsigned long long* packed;
    size_t bytes;
};
static DecodeWS carve_decode(const s2vt_dims& d, void* base) {
    const size_t B = d.B, L = d.L, H = d.H, T = 2 * L - 1;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    DecodeWS w;
    w.bsum1 = c.take<float>(4 * H);
    w.bsum2 = c.take<float>(4 * H);
    w.x1 = c.take<float>(L * B * H);
    w.gx1 = c.take<float>(L * B * 4 * H);
    w.h1 = c.take<float>(T * B * H);
    w.c1 = c.take<float>(B * H);
    w.gx2 = c.take<float>(T * B * 4 * H);
    w.h2 = c.take<float>(2 * B * H);
    w.c2 = c.take<float>(B * H);
    w.packed = c.take<unsigned long long>((L - 1) * B);
    w.gws_floats = gemm_ws_floats(d);
    w.gws = c.take<float>(w.gws_floats);
    w.bytes = align_up(c.off, 256);
    return w;
}

size_t s2vt_decode_workspace_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    return carve_decode(*d, nullptr).bytes;
}

int s2vt_greedy_decode(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                       void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && ids && workspace, "s2vt_greedy_decode: null/invalid argument");
    S2VT_REQUIRE(sos_ix >= 0 && sos_ix < d->V, "s2vt_greedy_decode: sos_ix %d outside vocabulary %d", sos_ix, d->V);
    const DecodeWS w = carve_decode(*d, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_greedy_decode: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    GemmWsScope gscope(w.gws, w.gws_floats);
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    int rc;
    if ((rc = add_vectors(st, p->vid_b_ih, p->vid_b_hh, w.bsum1, 4 * H))) return rc;
    if ((rc = add_vectors(st, p->word_b_ih, p->word_b_hh, w.bsum2, 4 * H))) return rc;
    if ((rc = fill_zero(st, w.packed, sizeof(unsigned long long) * (size_t)(L - 1) * B))) return rc;
    if ((rc = gemm(st, true, true, B * L, H, F, feats, F, ID, p->feat_w, F, ID, w.x1, H, perm(L, B), p->feat_b, false)))
        return rc;
    if ((rc = gemm(st, true, true, L * B, 4 * H, H, w.x1, H, ID, p->vid_w_ih, H, ID, w.gx1, 4 * H, ID, w.bsum1, false)))
        return rc;
    {   // vid_rnn over all T steps (S2VTModel.py:64-67); c updated in place, h kept for the word layer
        ProfScope ps(st, K_STEP_FWD, T);
        for (int t = 0; t < T; ++t) {
            StepFwdArgs a;
            memset(&a, 0, sizeof(a));
            a.B = B; a.H = H;
            a.h_prev = t ? w.h1 + (t - 1) * BH : nullptr; a.ldh = H;
            a.w_hh = p->vid_w_hh; a.ldw = H;
            a.gx = (t < L) ? w.gx1 + t * B4H : nullptr; a.ldgx = 4 * (int64_t)H;
            a.bias = w.bsum1;
            a.c_prev = t ? w.c1 : nullptr; a.ldc = H;
            a.h_out = w.h1 + t * BH; a.ldho = H;
            a.c_out = w.c1; a.ldco = H;
            if ((rc = lstm_step_fwd(st, a))) return rc;
        }
    }
    if ((rc = gemm(st, true, true, T * B, 4 * H, H, w.h1, H, ID, p->word_w_ih + E, E + H, ID, w.gx2, 4 * H, ID, w.bsum2,
                   false)))
        return rc;
    for (int t = 0; t < T; ++t) {
        {   // word_rnn: encode steps see a zero embedding (:84-86), decode steps Emb[prev token] (:89-103)
            ProfScope ps(st, K_STEP_FWD, 1);
            StepFwdArgs a;
            memset(&a, 0, sizeof(a));
            a.B = B; a.H = H;
            a.h_prev = t ? w.h2 + ((t - 1) & 1) * BH : nullptr; a.ldh = H;
            a.w_hh = p->word_w_hh; a.ldw = H;
            if (t >= L) {
                a.x2 = p->emb_w; a.ldx2 = E; a.K2 = E;
                a.w2 = p->word_w_ih; a.ldw2 = E + H;
                a.tok_packed = (t > L) ? w.packed + (int64_t)(t - L - 1) * B : nullptr;
                a.tok_const = sos_ix;
            }
            a.gx = w.gx2 + t * B4H; a.ldgx = 4 * (int64_t)H;
            a.c_prev = t ? w.c2 : nullptr; a.ldc = H;
            a.h_out = w.h2 + (t & 1) * BH; a.ldho = H;
            a.c_out = w.c2; a.ldco = H;
            if ((rc = lstm_step_fwd(st, a))) return rc;
        }
        if (t >= L) {  // out_linear + argmax (:95-96, :105-106)
            ProfScope ps(st, K_ARGMAX, 1);
            LogitsArgmaxArgs la;
            la.B = B; la.H = H; la.V = V;
            la.h = w.h2 + (t & 1) * BH; la.ldh = H;
            la.w_out = p->out_w; la.ldw = H; la.b_out = p->out_b;
            la.packed = w.packed + (int64_t)(t - L) * B;
            if ((rc = logits_argmax(st, la))) return rc;
        }
    }
    return unpack_tokens(st, w.packed, L - 1, B, ids);
}

// ------------------------------------------------------------------ loss
int s2vt_mean_ce_forward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target,
                         int64_t target_ld, float* lse, float* rowloss, float* loss_out, void* stream) {
    S2VT_REQUIRE(B > 0 && Lm1 > 0 && V > 0, "s2vt_mean_ce_forward: bad dims");
    ProfScope ps((hipStream_t)stream, K_CE, 1);
    return mean_ce_fwd((hipStream_t)stream, logits, (int64_t)B * Lm1, V, target, Lm1, target_ld, lse, rowloss, loss_out,
                       nullptr);
}
int s2vt_mean_ce_backward(int32_t B, int32_t Lm1, int32_t V, const float* logits, const int64_t* target,
                          int64_t target_ld, const float* lse, const float* gout, float* dlogits, void* stream) {
    S2VT_REQUIRE(B > 0 && Lm1 > 0 && V > 0, "s2vt_mean_ce_backward: bad dims");
    ProfScope ps((hipStream_t)stream, K_CE, 1);
    return mean_ce_bwd((hipStream_t)stream, logits, (int64_t)B * Lm1, V, target, Lm1, target_ld, lse, gout, dlogits);
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
    if (stash) return seq_fwd(st, T, B, H, stash, n_gx, bias, w_hh, h_all, c_all, true);
    return seq_fwd(st, T, B, H, const_cast<float*>(gx), n_gx, bias, w_hh, h_all, c_all, false);
}

int s2vt_lstm_seq_bwd(int32_t T, int32_t B, int32_t H, const float* w_hh, const float* dh_out, int32_t dh_first,
                      const float* c_all, float* stash_dg, float* w_hh_t, float* dc, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh && c_all && stash_dg && w_hh_t && dc, "s2vt_lstm_seq_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = transpose_f32(st, w_hh, 4 * H, H, w_hh_t))) return rc;
    return seq_bwd(st, T, B, H, w_hh_t, dh_out, dh_first, c_all, stash_dg, dc);
}

int s2vt_decode_step_argmax(int32_t B, int32_t H, int32_t V, const float* h, const float* w_out, const float* b_out,
                            unsigned long long* packed, void* stream) {
    LogitsArgmaxArgs la;
    la.B = B; la.H = H; la.V = V; la.h = h; la.ldh = H; la.w_out = w_out; la.ldw = H; la.b_out = b_out;
    la.packed = packed;
    ProfScope ps((hipStream_t)stream, K_ARGMAX, 1);
    return logits_argmax((hipStream_t)stream, la);
}

// ------------------------------------------------------------------ live timing
int s2vt_prof_enable(int32_t on) { g_prof = on != 0; return 0; }

int s2vt_prof_reset(void) {
    for (auto& r : g_recs) g_free.emplace_back(r.a, r.b);
    g_recs.clear();
    return 0;
}

int s2vt_prof_read(int32_t kind, double* total_ms, int64_t* launches) {
    S2VT_REQUIRE(kind >= 0 && kind < K_NKINDS && total_ms && launches, "s2vt_prof_read: bad arguments");
    double ms = 0.0;
    int64_t n = 0;
    for (auto& r : g_recs) {
        if (r.kind != kind) continue;
        S2VT_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        S2VT_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms += t;
        n += r.launches;
    }
    *total_ms = ms;
    *launches = n;
    return 0;
}

}  // extern "C"
