// The batched beam-search depth step (S2VTModel.py:186-228 for all (sample, beam slot) rows of a depth at once).
#include "api_internal.h"

using namespace s2vt;

extern "C" {

// ---------------------------------------------------------------------------------- batched beam-search depth
struct BeamWS {
    float *bsum1, *bsum2, *ph, *pc, *gx, *logits, *gws;
    int* err;                // [0]: a token id outside [0, V) reached the word step (reported as S2VT_ERR_INDEX)
    PB pvid, pword;          // plane path (s2vt_beam_step_cached): planes of vid_rnn's h [B rows] and of the word step's h_t [R rows]
    size_t gws_floats, bytes;
};
static BeamWS carve_beam(const s2vt_dims& d, int max_rows, void* base) {
    const size_t H = d.H, V = d.V, R = (size_t)max_rows;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    BeamWS w;
    w.bsum1 = c.take<float>(4 * H);
    w.bsum2 = c.take<float>(4 * H);
    w.ph = c.take<float>(R * H);
    w.pc = c.take<float>(R * H);
    w.gx = c.take<float>(R * 4 * H);
    w.logits = c.take<float>(R * V);
    w.gws_floats = 4 * R * (V > 4 * H ? V : 4 * H);
    w.gws = c.take<float>(w.gws_floats);
    w.err = c.take<int>(4);
    {
        const int kp = pad64((int)H);
        w.pvid.kpad = w.pword.kpad = kp;
        w.pvid.ld = w.pword.ld = 3 * (int64_t)kp;
        w.pvid.p = c.take<unsigned short>(rows64((size_t)d.B) * (size_t)w.pvid.ld);
        w.pword.p = c.take<unsigned short>(rows64(R) * (size_t)w.pword.ld);
    }
    w.bytes = align_up(c.off, 256);
    return w;
}

size_t s2vt_beam_workspace_bytes(const s2vt_dims* d, int32_t max_rows) {
    if (!d || max_rows <= 0) return 0;
    return carve_beam(*d, max_rows, nullptr).bytes;
}

static int beam_step_impl(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                          const int32_t* tok, const float* vid_h_in, const float* vid_c_in, float* vid_h_out, float* vid_c_out,
                          const float* word_h_in, const float* word_c_in, float* word_h_out, float* word_c_out, int32_t* top_ix,
                          float* top_lp, void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, void* stream,
                          const float* gx_vid = nullptr);
// the depth step with vid_rnn's part precomputed (s2vt_decode_encode_cached, gx_dec[depth - 1]): word step, out_linear, fan-out
int s2vt_beam_step_gx(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                      const int32_t* tok, const float* gx_vid, const float* word_h_in, const float* word_c_in, float* word_h_out,
                      float* word_c_out, int32_t* top_ix, float* top_lp, void* workspace, size_t workspace_bytes, void* cache,
                      size_t cache_bytes, void* stream) {
    S2VT_REQUIRE(cache && gx_vid, "s2vt_beam_step_gx: null cache / gx_vid");
    return beam_step_impl(d, p, R, row_b, row_state, tok, nullptr, nullptr, nullptr, nullptr, word_h_in, word_c_in, word_h_out,
                          word_c_out, top_ix, top_lp, workspace, workspace_bytes, cache, cache_bytes, stream, gx_vid);
}
int s2vt_beam_step(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                   const int32_t* tok, const float* vid_h_in, const float* vid_c_in, float* vid_h_out, float* vid_c_out,
                   const float* word_h_in, const float* word_c_in, float* word_h_out, float* word_c_out, int32_t* top_ix,
                   float* top_lp, void* workspace, size_t workspace_bytes, void* stream) {
    return beam_step_impl(d, p, R, row_b, row_state, tok, vid_h_in, vid_c_in, vid_h_out, vid_c_out, word_h_in, word_c_in, word_h_out,
                          word_c_out, top_ix, top_lp, workspace, workspace_bytes, nullptr, 0, stream);
}
int s2vt_beam_step_cached(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                          const int32_t* tok, const float* vid_h_in, const float* vid_c_in, float* vid_h_out, float* vid_c_out,
                          const float* word_h_in, const float* word_c_in, float* word_h_out, float* word_c_out, int32_t* top_ix,
                          float* top_lp, void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, void* stream) {
    S2VT_REQUIRE(cache, "s2vt_beam_step_cached: null cache");
    return beam_step_impl(d, p, R, row_b, row_state, tok, vid_h_in, vid_c_in, vid_h_out, vid_c_out, word_h_in, word_c_in, word_h_out,
                          word_c_out, top_ix, top_lp, workspace, workspace_bytes, cache, cache_bytes, stream);
}
static int beam_step_impl(const s2vt_dims* d, const s2vt_params* p, int32_t R, const int32_t* row_b, const int32_t* row_state,
                          const int32_t* tok, const float* vid_h_in, const float* vid_c_in, float* vid_h_out, float* vid_c_out,
                          const float* word_h_in, const float* word_c_in, float* word_h_out, float* word_c_out, int32_t* top_ix,
                          float* top_lp, void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, void* stream,
                          const float* gx_vid) {
    S2VT_REQUIRE(d && p && workspace && (gx_vid || (vid_h_in && vid_c_in && vid_h_out && vid_c_out)), "s2vt_beam_step: null argument");
    S2VT_REQUIRE(R >= 0 && (R == 0 || (row_b && row_state && tok && word_h_in && word_c_in && word_h_out && word_c_out &&
                                        top_ix && top_lp)),
                 "s2vt_beam_step: null row argument");
    const int B = d->B, H = d->H, E = d->E, V = d->V;
    S2VT_REQUIRE(workspace_bytes >= carve_beam(*d, R > 0 ? R : 1, nullptr).bytes, "s2vt_beam_step: workspace too small");
    const BeamWS w = carve_beam(*d, R > 0 ? R : 1, workspace);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!gx_vid && (rc = add_vectors(st, p->vid_b_ih, p->vid_b_hh, w.bsum1, 4 * H))) return rc;
    if (!gx_vid && (rc = add_vectors(st, p->word_b_ih, p->word_b_hh, w.bsum2, 4 * H))) return rc;
    if (!gx_vid) {   // one zero-input vid_rnn step for the whole batch (S2VTModel.py:208-210)
        StepFwdArgs a = {};
        a.B = B; a.H = H;
        a.h_prev = vid_h_in; a.ldh = H; a.w_hh = p->vid_w_hh; a.ldw = H;
        a.bias = w.bsum1;
        a.c_prev = vid_c_in; a.ldc = H;
        a.h_out = vid_h_out; a.ldho = H; a.c_out = vid_c_out; a.ldco = H;
        if ((rc = lstm_step_fwd(st, a))) return rc;
    }
    if (R == 0) return 0;
    const Lane ln{st, w.gws, w.gws_floats, nullptr};
    // parents' word_rnn states, the vid_out half of the gate input (A rows gathered by sample), then the word step
    // with the embedding rows gathered inside the kernel's second K segment (:211-212)
    if ((rc = gather_rows_f32(st, word_h_in, H, row_state, R, H, w.ph))) return rc;
    if ((rc = gather_rows_f32(st, word_c_in, H, row_state, R, H, w.pc))) return rc;
    if (cache && gemm_mode() != 0 && H <= 1024) {     // (any batch: the row counts of its GEMMs are free, the images are the cache's)
        // PLANE PATH (the weight-derived images of a greedy decode of the same weights, s2vt_greedy_decode_cached: W_v and W_o
        // as blocked 3-plane operands, the per-token gate table): the vid_out half of the gate input once per SAMPLE (every beam
        // slot reads its sample's row), the embedded word from the table, out_linear on the bf16 matrix cores in split
        // precision (fp32-equivalent) from the h_t planes the step kernel writes itself
        XP = 3;
        S2VT_REQUIRE(cache_bytes >= carve_decode_const(*d, nullptr).bytes, "s2vt_beam_step_cached: cache too small");
        const DecodeConst kc = carve_decode_const(*d, cache);
        if (!gx_vid) {
            if ((rc = psplit(ln, w.pvid, 0, vid_h_out, H, ID, B, H))) return rc;
            if ((rc = pgemm(ln, B, 4 * H, H, w.pvid, 0, 0, kc.wv, 0, 0, w.gx, 4 * H, ID, w.bsum2, false))) return rc;
        }
        if ((rc = fill_zero(st, w.pword.p, rows64((size_t)R) * (size_t)w.pword.ld * sizeof(unsigned short)))) return rc;
        {
            StepFwdArgs a = {};
            a.B = R; a.H = H;
            a.h_prev = w.ph; a.ldh = H; a.w_hh = p->word_w_hh; a.ldw = H;
            a.gx_tab = kc.gtab; a.ldtab = 4 * (int64_t)H; a.tok_idx = tok;
            a.tok_limit = V; a.tok_err = w.err;
            a.gx = gx_vid ? gx_vid : w.gx; a.ldgx = 4 * H; a.gx_idx = row_b;
            a.c_prev = w.pc; a.ldc = H;
            a.h_out = word_h_out; a.ldho = H; a.c_out = word_c_out; a.ldco = H;
            a.h_planes = w.pword.p; a.ldhp = w.pword.ld;
            if ((rc = fill_zero(st, w.err, 4 * sizeof(int)))) return rc;
            if ((rc = lstm_step_fwd(st, a))) return rc;
        }
        if ((rc = pgemm(ln, R, V, H, w.pword, 0, 0, kc.wo, 0, 0, w.logits, V, ID, p->out_b, false))) return rc;
        if ((rc = top20_logprob(st, w.logits, V, R, V, top_ix, top_lp))) return rc;
        return post_async_error(st, w.err, 3);       // (ring slot: no wait for the previous depth step)
    }
    S2VT_REQUIRE(!gx_vid, "s2vt_beam_step_gx: the precomputed vid_rnn half needs the plane path (gemm mode 1 / 3, H <= 1024)");
    if ((rc = lgemm(ln, true, true, R, 4 * H, H, vid_h_out, H, gather(row_b), p->word_w_ih + E, E + H, ID, w.gx, 4 * H, ID,
                    w.bsum2, false)))
        return rc;
    {
        StepFwdArgs a = {};
        a.B = R; a.H = H;
        a.h_prev = w.ph; a.ldh = H; a.w_hh = p->word_w_hh; a.ldw = H;
        a.x2 = p->emb_w; a.ldx2 = E; a.K2 = E; a.w2 = p->word_w_ih; a.ldw2 = E + H; a.tok_idx = tok;
        a.tok_limit = V; a.tok_err = w.err;            // nn.Embedding raises IndexError for such an id (S2VTModel.py:211)
        a.gx = w.gx; a.ldgx = 4 * H;
        a.c_prev = w.pc; a.ldc = H;
        a.h_out = word_h_out; a.ldho = H; a.c_out = word_c_out; a.ldco = H;
        if ((rc = fill_zero(st, w.err, 4 * sizeof(int)))) return rc;
        if ((rc = lstm_step_fwd(st, a))) return rc;
    }
    if ((rc = lgemm(ln, true, true, R, V, H, word_h_out, H, ID, p->out_w, H, ID, w.logits, V, ID, p->out_b, false))) return rc;   // (:213)
    if ((rc = top20_logprob(st, w.logits, V, R, V, top_ix, top_lp))) return rc;                                                    // (:214-219)
    return post_async_error(st, w.err, 3);       // (ring slot: no wait for the previous depth step)
}


}
