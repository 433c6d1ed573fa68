// Greedy decode (S2VTModel.py:82-110, mode='test'), its encode phase handed out for the beam search, and the decode step's
// out_linear + argmax entry points.
#include "api_internal.h"

using namespace s2vt;

extern "C" {

// ------------------------------------------------------------------ greedy decode
struct DecodeWS {
    float *bsum1, *bsum2, *x1, *gx1, *h1, *c1, *gx2, *h2, *c2, *gws_a, *gws_b;
    float* zbuf;                           // [B][4H]: h_t·W_hh^T, the recurrent half of the next decode step's gates
    size_t gws_floats;
    unsigned long long* packed;
    PB feats, px1, ph1;                    // packed planes of per-call activations (split-precision mode only)
    PB ph2;                                // the decode step's h_t planes
    PB embp, wep;                          // planes of the embedding table and of W_e (scratch of the per-token table's GEMM)
    // persistent split-precision recurrence of the ENCODE phase (lstm_persist_x3.hip): per-step cell states, word_rnn's encode
    // outputs, the h_t plane images of both layers, hand-off counters, error flags (xkp == 0: not provided)
    int64_t xkp;
    float *c1_all, *c2_all, *h2_all;
    unsigned short *xh1, *xh2;
    unsigned int *psync_a, *psync_b;
    int* err;
    size_t bytes;
};
static DecodeWS carve_decode(const s2vt_dims& d, void* base) {
    const size_t B = d.B, L = d.L, F = d.F, H = d.H, T = 2 * L - 1;
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
    w.zbuf = c.take<float>(B * 4 * H);
    w.packed = c.take<unsigned long long>((L - 1) * B);
    w.gws_floats = gemm_ws_floats(d);
    w.gws_a = c.take<float>(w.gws_floats);
    w.gws_b = c.take<float>(w.gws_floats);
    if (planes_ok(d)) {
        XP = 3;     // greedy decode must stay fp32-equivalent (bit-exact ids): split precision in every plane mode
        auto mk = [&](size_t rows, size_t k) {
            PB b;
            b.kpad = pad64((int)k);
            b.ld = (int64_t)XP * b.kpad;
            b.p = c.take<unsigned short>(rows64(rows) * (size_t)b.ld);
            return b;
        };
        w.feats = mk(B * L, F); w.px1 = mk(L * B, H);
        w.ph1 = mk(T * B, H);   w.ph2 = mk(B, H);
        w.embp = mk(d.V, d.E);  w.wep = mk(4 * H, d.E);
    }
    w.xkp = (planes_ok(d) && H <= 1024 && pipe_block() > 0 && persist_x3_fwd_on()) ? (int64_t)pad64((int)H) : 0;   // (0: the persistent encode phase is not selectable)
    w.c1_all = c.take<float>(w.xkp ? T * B * H : 0);
    w.c2_all = c.take<float>(w.xkp ? L * B * H : 0);
    w.h2_all = c.take<float>(w.xkp ? L * B * H : 0);
    w.xh1 = c.take<unsigned short>(w.xkp ? 3 * T * B * (size_t)w.xkp : 0);
    w.xh2 = c.take<unsigned short>(w.xkp ? 3 * L * B * (size_t)w.xkp : 0);
    w.psync_a = c.take<unsigned int>(lstm_persist_sync_bytes() / sizeof(unsigned int));
    w.psync_b = c.take<unsigned int>(lstm_persist_sync_bytes() / sizeof(unsigned int));
    w.err = c.take<int>(4);
    w.bytes = align_up(c.off, 256);
    return w;
}
// What a decode derives from the WEIGHTS alone (plane images of W_f, W_ih1, W_v, W_o and the per-token gate-input table):
// carved from the tail of the call's workspace, or from a caller-kept cache that outlives the call (s2vt_greedy_decode_cached)
}  // (C linkage ends: shared with api_beam.hip)
namespace s2vt {
DecodeConst carve_decode_const(const s2vt_dims& d, void* base) {
    const size_t F = d.F, H = d.H;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    DecodeConst k;
    k.gtab = nullptr;
    k.xw1 = k.xw2 = nullptr;
    k.wf = k.wih1 = k.wv = k.wo = k.whh = PB{nullptr, 0, 0};
    if (gemm_mode() != 0) {     // (the images depend on the weights' dims only: one cache serves every batch size)
        XP = 3;
        auto mk = [&](size_t rows, size_t kk) {
            PB b;
            b.kpad = pad64((int)kk);
            b.ld = (int64_t)XP * b.kpad;
            b.p = c.take<unsigned short>(rows64(rows) * (size_t)b.ld);
            return b;
        };
        k.wf = mk(H, F); k.wih1 = mk(4 * H, H); k.wv = mk(4 * H, H); k.wo = mk(d.V, H);
        k.gtab = c.take<float>((size_t)d.V * 4 * H);
        const size_t xkp = (H <= 1024) ? (size_t)pad64((int)H) : 0;
        k.xw1 = c.take<unsigned short>(3 * 4 * H * xkp);
        k.xw2 = c.take<unsigned short>(3 * 4 * H * xkp);
        k.whh = mk(4 * H, H);       // (last: the images in front keep their offsets)
    }
    k.bytes = align_up(c.off, 256);
    return k;
}
}  // namespace s2vt
extern "C" {

// Batches that are not multiples of 64 are padded inside the workspace, like the train drivers' (api_train.hip): zero features for
// the pad samples, whose ids / states are never handed out.  A plain greedy decode of fewer than 24 clips (eval.py:27 decodes 10 at
// a time) is the exception (batch_pads): there the launch-per-timestep path - 16-row fp32-MFMA tiles, gate GEMVs up to B = 4 - is
// faster than 64 padded rows on the plane path; the encode phase for the beam search always takes the plane path.
static inline bool batch_padded(const s2vt_dims& d, bool encode_only = false) {
    return encode_only ? (gemm_mode() != 0 && d.B % 64 != 0) : batch_pads(d.B, true);
}
static inline s2vt_dims padded_dims(const s2vt_dims& d) { s2vt_dims q = d; q.B = (d.B + 63) / 64 * 64; return q; }
struct DecodePad { float* feats; int64_t* ids; float* states; float* gx_dec; size_t bytes; };
static DecodePad carve_decode_pad(const s2vt_dims& d, const s2vt_dims& dp, void* base) {
    const size_t Bp = dp.B, L = d.L, F = d.F, H = d.H;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    DecodePad w;
    w.feats = c.take<float>(Bp * L * F);
    w.ids = c.take<int64_t>(Bp * (L - 1));
    w.states = c.take<float>(4 * Bp * H);                  // vid_h, vid_c, word_h, word_c of s2vt_decode_encode_cached
    w.gx_dec = c.take<float>((L - 1) * Bp * 4 * H);
    w.bytes = align_up(c.off, 256);
    return w;
}
static size_t decode_core_bytes(const s2vt_dims& d) { return align_up(carve_decode(d, nullptr).bytes + carve_decode_const(d, nullptr).bytes, 256); }
size_t s2vt_decode_workspace_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    if (batch_padded(*d, true)) {       // (sized for either use of the workspace: s2vt_decode_encode_cached pads every ragged batch)
        const s2vt_dims dp = padded_dims(*d);
        return decode_core_bytes(dp) + carve_decode_pad(*d, dp, nullptr).bytes;
    }
    return carve_decode(*d, nullptr).bytes + carve_decode_const(*d, nullptr).bytes;
}
size_t s2vt_decode_cache_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    return carve_decode_const(*d, nullptr).bytes;
}
int32_t s2vt_decode_uses_cache(const s2vt_dims* d) {
    if (!dims_ok(d) || gemm_mode() == 0) return 0;
    return (d->B % 64 == 0 || batch_padded(*d)) ? 1 : 0;
}

struct EncodeOut { float *vid_h, *vid_c, *word_h, *word_c; float* gx_dec; int depth; };      // states [B, H] after the L encode steps;
                                     // optional: word_rnn's vid_out gate input (+ biases) of the first `depth` decode steps [depth][B][4H]
static int greedy_decode_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, bool cache_valid,
                              void* stream, const EncodeOut* enc = nullptr);
// schedule of the 79 token-dependent decode steps on the plane path: 1 = fused (the next step's recurrent GEMM inside the
// argmax launch + a cell-update launch), 0 = a step kernel and an argmax kernel per step, batch halves as two chains
static int decode_schedule() { return option(O_DECODE_FUSED); }
int s2vt_set_decode_schedule(int32_t schedule) { return option_set(O_DECODE_FUSED, (schedule == 0 || schedule == 1) ? schedule : -1); }
int s2vt_greedy_decode(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return greedy_decode_impl(d, p, feats, sos_ix, ids, workspace, workspace_bytes, nullptr, 0, false, stream);
}
int s2vt_greedy_decode_cached(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, int32_t cache_valid,
                              void* stream) {
    S2VT_REQUIRE(cache, "s2vt_greedy_decode_cached: null cache");
    return greedy_decode_impl(d, p, feats, sos_ix, ids, workspace, workspace_bytes, cache, cache_bytes, cache_valid != 0, stream);
}
// The ENCODE phase of the decode alone (S2VTModel.py:56-60 for mode='beam_search', the same computation as :64-86 of mode='test'):
// feature projection, both layers over the L frames on the plane path, the weight images in the caller's cache (filled here when
// cache_valid == 0 - every image a decode or a beam search of these weights reads).  Out: the four [B, H] states a beam search
// starts from.  Shapes the persistent split-precision recurrence does not take return S2VT_ERR_ARG (the caller keeps its own encoder).
int s2vt_decode_encode_cached(const s2vt_dims* d, const s2vt_params* p, const float* feats, void* workspace, size_t workspace_bytes,
                              void* cache, size_t cache_bytes, int32_t cache_valid, float* vid_h, float* vid_c, float* word_h,
                              float* word_c, float* gx_dec, int32_t depth, void* stream) {
    S2VT_REQUIRE(cache && vid_h && vid_c && word_h && word_c, "s2vt_decode_encode_cached: null argument");
    S2VT_REQUIRE(!gx_dec || (d && depth > 0 && depth <= d->L - 1), "s2vt_decode_encode_cached: depth must be in [1, L-1]");
    const EncodeOut enc{vid_h, vid_c, word_h, word_c, gx_dec, gx_dec ? depth : 0};
    return greedy_decode_impl(d, p, feats, 0, nullptr, workspace, workspace_bytes, cache, cache_bytes, cache_valid != 0, stream, &enc);
}
static int greedy_decode_core(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, bool cache_valid,
                              void* stream, const EncodeOut* enc);
static int greedy_decode_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, bool cache_valid,
                              void* stream, const EncodeOut* enc) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && (ids || enc) && workspace, "s2vt_greedy_decode: null/invalid argument");
    if (!batch_padded(*d, enc != nullptr))
        return greedy_decode_core(d, p, feats, sos_ix, ids, workspace, workspace_bytes, cache, cache_bytes, cache_valid, stream, enc);
    const s2vt_dims dp = padded_dims(*d);
    const size_t core = decode_core_bytes(dp);
    const DecodePad s = carve_decode_pad(*d, dp, reinterpret_cast<char*>(workspace) + core);
    S2VT_REQUIRE(workspace_bytes >= core + s.bytes, "s2vt_greedy_decode: workspace %zu < %zu bytes", workspace_bytes, core + s.bytes);
    hipStream_t st = (hipStream_t)stream;
    const size_t B = d->B, Bp = dp.B, L = d->L, F = d->F, H = d->H;
    int rc;
    S2VT_HIP(hipMemcpyAsync(s.feats, feats, B * L * F * sizeof(float), hipMemcpyDeviceToDevice, st));
    if ((rc = fill_zero(st, s.feats + B * L * F, (Bp - B) * L * F * sizeof(float)))) return rc;
    if (!enc) {
        if ((rc = greedy_decode_core(&dp, p, s.feats, sos_ix, s.ids, workspace, core, cache, cache_bytes, cache_valid, stream, nullptr))) return rc;
        S2VT_HIP(hipMemcpyAsync(ids, s.ids, B * (L - 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    const EncodeOut pe{s.states, s.states + Bp * H, s.states + 2 * Bp * H, s.states + 3 * Bp * H, enc->depth > 0 ? s.gx_dec : nullptr, enc->depth};
    if ((rc = greedy_decode_core(&dp, p, s.feats, sos_ix, nullptr, workspace, core, cache, cache_bytes, cache_valid, stream, &pe))) return rc;
    float* outs[4] = {enc->vid_h, enc->vid_c, enc->word_h, enc->word_c};
    for (int k = 0; k < 4; ++k)
        S2VT_HIP(hipMemcpyAsync(outs[k], s.states + (size_t)k * Bp * H, B * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (enc->depth > 0)      // [depth][Bp][4H] -> [depth][B][4H]
        S2VT_HIP(hipMemcpy2DAsync(enc->gx_dec, B * 4 * H * sizeof(float), s.gx_dec, Bp * 4 * H * sizeof(float), B * 4 * H * sizeof(float),
                                  (size_t)enc->depth, hipMemcpyDeviceToDevice, st));
    return 0;
}
static int greedy_decode_core(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, bool cache_valid,
                              void* stream, const EncodeOut* enc) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && (ids || enc) && workspace, "s2vt_greedy_decode: null/invalid argument");
    S2VT_REQUIRE(sos_ix >= 0 && sos_ix < d->V, "s2vt_greedy_decode: sos_ix %d outside vocabulary %d", sos_ix, d->V);
    const DecodeWS w = carve_decode(*d, workspace);
    const size_t kbytes = carve_decode_const(*d, nullptr).bytes;
    S2VT_REQUIRE(workspace_bytes >= w.bytes + (cache ? 0 : kbytes), "s2vt_greedy_decode: workspace %zu < %zu bytes", workspace_bytes,
                 w.bytes + (cache ? 0 : kbytes));
    S2VT_REQUIRE(!cache || cache_bytes >= kbytes, "s2vt_greedy_decode_cached: cache %zu < %zu bytes", cache_bytes, kbytes);
    // weight-derived images: in the caller's cache (filled by a call with cache_valid == 0, reused while the weights stand) or
    // behind the per-call part of the workspace (rebuilt by every call)
    const DecodeConst kc = carve_decode_const(*d, cache ? cache : reinterpret_cast<char*>(workspace) + w.bytes);
    const bool fill = !(cache && cache_valid);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    const bool x3 = planes_ok(*d);
    if (x3) XP = 3;
    const int blk = pipe_block();
    // encode phase (both layers, L steps) and vid_rnn's input-free decode steps as persistent split-precision launches; only the
    // 79 token-dependent word_rnn steps stay one launch (+ argmax) per step
    const bool use_px = x3 && blk > 0 && w.xkp > 0 && persist_x3_fwd_on() && lstm_seq_fwd_x3_persist_supported(B, H);
    // (refused BEFORE anything is enqueued: the caller frees the workspace on this error)
    S2VT_REQUIRE(!enc || use_px, "s2vt_decode_encode_cached: this shape / mode does not take the persistent split-precision encode phase");
    hipStream_t sx = st;
    int rc;
    if (blk > 0 && (rc = side_stream(st, &sx))) return rc;
    const Lane la{st, w.gws_a, w.gws_floats, nullptr};     // vid_rnn lane (caller's stream)
    const Lane lb{sx, w.gws_b, w.gws_floats, nullptr};     // word_rnn lane: encode, then the 79 decode steps
    size_t ev = 0;
    if ((rc = add_vectors(st, p->vid_b_ih, p->vid_b_hh, w.bsum1, 4 * H))) return rc;
    if ((rc = add_vectors(st, p->word_b_ih, p->word_b_hh, w.bsum2, 4 * H))) return rc;
    if ((rc = fill_zero(st, w.packed, sizeof(unsigned long long) * (size_t)(L - 1) * B))) return rc;
    if ((rc = fill_zero(st, w.err, 4 * sizeof(int)))) return rc;
    if ((rc = handoff(st, sx, ev++))) return rc;
    // feature projection + vid_rnn input GEMM                                  S2VTModel.py:54, 64-67
    const bool ax3 = x3;          // out_linear + argmax on the plane path (argmax_x3.hip); fp32-input MFMA otherwise (lstm.hip)
    // per-token gate-input table instead of the embedding K segment of the 79 decode steps: one V x 4H x E GEMM (0.5 ms at
    // V = 12000) against B x 4H x E of MFMA work and E/(E+H) of the operand traffic in EVERY decode step - pays from B ~ 64
    const bool use_tab = x3;
    // A caller-kept cache outlives this call's choices (batch size, recurrence mode, pipeline block, experiment switches): a call
    // that fills it builds EVERY weight-derived image it holds, not only the ones this call reads - a later call on the same
    // weights with another batch or mode then finds its images whatever it selects (cache_valid says "the weights stand",
    // nothing about who filled it).
    const bool fill_all = fill && cache != nullptr;
    const int64_t ckp = (x3 && H <= 1024) ? (int64_t)pad64(H) : 0;      // row length of the W_hh plane images (carve_decode_const)
    if (ckp > 0 && fill && (use_px || fill_all)) {
        if ((rc = split3_rows(sx, p->vid_w_hh, H, 4 * H, H, (int)ckp, kc.xw1, 4 * (int64_t)H * ckp))) return rc;
        if ((rc = split3_rows(sx, p->word_w_hh, H, 4 * H, H, (int)ckp, kc.xw2, 4 * (int64_t)H * ckp))) return rc;
    }
    if (x3) {
        // W_o planes: constant over the 79 decode steps; h_t planes: written by the decode steps themselves, k padding zeroed here
        if (fill && (ax3 || fill_all) && (rc = psplit(lb, kc.wo, 0, p->out_w, H, ID, V, H))) return rc;
        if (ax3 && (rc = fill_zero(sx, w.ph2.p, rows64((size_t)B) * (size_t)w.ph2.ld * sizeof(unsigned short)))) return rc;
        if (fill && (use_tab || fill_all)) {  // gtab[v] = Emb[v]·W_e^T for every token (S2VTModel.py:90-93,100-103: embedding + the embed columns of word_rnn's W_ih)
            if ((rc = psplit(lb, w.embp, 0, p->emb_w, E, ID, V, E))) return rc;
            if ((rc = psplit(lb, w.wep, 0, p->word_w_ih, E + H, ID, 4 * H, E))) return rc;
            if ((rc = pgemm(lb, V, 4 * H, E, w.embp, 0, 0, w.wep, 0, 0, kc.gtab, 4 * H, ID, nullptr, false))) return rc;
        }
        if (fill && (rc = psplit(lb, kc.wv, 0, p->word_w_ih + E, E + H, ID, 4 * H, H))) return rc;
        if (fill && (rc = psplit(lb, kc.whh, 0, p->word_w_hh, H, ID, 4 * H, H))) return rc;
        if ((rc = psplit(la, w.feats, 0, feats, F, ID, B * L, F))) return rc;
        if (fill && (rc = psplit(la, kc.wf, 0, p->feat_w, F, ID, H, F))) return rc;
        if (fill && (rc = psplit(la, kc.wih1, 0, p->vid_w_ih, H, ID, 4 * H, H))) return rc;
        if ((rc = pgemm(la, B * L, H, F, w.feats, 0, 0, kc.wf, 0, 0, w.x1, H, perm(L, B), p->feat_b, false))) return rc;
        if ((rc = psplit(la, w.px1, 0, w.x1, H, ID, L * B, H))) return rc;
        if ((rc = pgemm(la, L * B, 4 * H, H, w.px1, 0, 0, kc.wih1, 0, 0, w.gx1, 4 * H, ID, w.bsum1, false))) return rc;
    } else {
        if ((rc = lgemm(la, true, true, B * L, H, F, feats, F, ID, p->feat_w, F, ID, w.x1, H, perm(L, B), p->feat_b, false)))
            return rc;
        if ((rc = lgemm(la, true, true, L * B, 4 * H, H, w.x1, H, ID, p->vid_w_ih, H, ID, w.gx1, 4 * H, ID, w.bsum1, false)))
            return rc;
    }
    // one word_rnn step on stream s (+ out_linear / argmax for a decode step): encode steps see a zero embedding (:84-86), decode
    // steps Emb[prev token] (:89-103)
    // (b0, nb): the batch rows [b0, b0 + nb) of the step - the whole batch, or one half of it when the decode runs as two
    // independent chains on two streams (b0 a multiple of 64: the plane images are blocked by 64 rows)
    auto word_args = [&](int t, const float* hprev, const float* cprev, int b0, int nb) -> StepFwdArgs {
        const int64_t o1 = (int64_t)b0 * H, o4 = 4 * o1;
        StepFwdArgs a;
        memset(&a, 0, sizeof(a));
        a.B = nb; a.H = H;
        a.h_prev = hprev ? hprev + o1 : nullptr; a.ldh = H;
        a.w_hh = p->word_w_hh; a.ldw = H;
        if (t >= L) {
            if (use_tab) {
                a.gx_tab = kc.gtab; a.ldtab = 4 * (int64_t)H;
            } else {
                a.x2 = p->emb_w; a.ldx2 = E; a.K2 = E;
                a.w2 = p->word_w_ih; a.ldw2 = E + H;
            }
            a.tok_packed = (t > L) ? w.packed + (int64_t)(t - L - 1) * B + b0 : nullptr;
            a.tok_const = sos_ix;
            // the packed word is the previous step's argmax: a producer that left it unwritten would decode as token
            // 0xFFFFFFFF - clamped and flagged (w.err[0], S2VT_ERR_INDEX) instead of read from beyond the table
            a.tok_limit = V; a.tok_err = w.err;
        }
        a.gx = w.gx2 + t * B4H + o4; a.ldgx = 4 * (int64_t)H;
        a.c_prev = cprev ? cprev + o1 : nullptr; a.ldc = H;
        a.h_out = w.h2 + (t & 1) * BH + o1; a.ldho = H;
        a.c_out = w.c2 + o1; a.ldco = H;
        if (t >= L && ax3) { a.h_planes = w.ph2.p + (int64_t)b0 * w.ph2.ld; a.ldhp = w.ph2.ld; }
        return a;
    };
    auto word_step = [&](hipStream_t s, int t, const float* hprev, const float* cprev, int b0 = 0, int nb = -1) -> int {
        int r;
        if (nb < 0) nb = B;
        const int64_t o1 = (int64_t)b0 * H;
        {
            ProfScope ps(s, K_STEP_FWD, 1);
            const StepFwdArgs a = word_args(t, hprev, cprev, b0, nb);
            if ((r = lstm_step_fwd(s, a))) return r;
        }
        if (t >= L && ax3) {  // out_linear + argmax (:95-96, :105-106) on the bf16 matrix cores (argmax_x3.hip); the
            ProfScope ps(s, K_ARGMAX, 1);       // step kernel above wrote h_t as planes (StepFwdArgs::h_planes)
            ArgmaxX3Args ax;
            memset(&ax, 0, sizeof(ax));
            ax.B = nb; ax.V = V; ax.K = kc.wo.kpad;
            ax.W = kc.wo.p; ax.ldw = kc.wo.ld;
            ax.Hp = w.ph2.p + (int64_t)b0 * w.ph2.ld; ax.ldh = w.ph2.ld;
            ax.bias = p->out_b;
            ax.packed = w.packed + (int64_t)(t - L) * B + b0;
            ax.dbg = 0; ax.stamps = nullptr;
            if ((r = logits_argmax_x3(s, ax))) return r;
        } else if (t >= L) {  // the same on the fp32-input MFMA (lstm.hip), for batches the plane path does not take
            ProfScope ps(s, K_ARGMAX, 1);
            LogitsArgmaxArgs la2;
            la2.B = nb; la2.H = H; la2.V = V;
            la2.h = w.h2 + (t & 1) * BH + o1; la2.ldh = H;
            la2.w_out = p->out_w; la2.ldw = H; la2.b_out = p->out_b;
            la2.packed = w.packed + (int64_t)(t - L) * B + b0;
            la2.stamps = nullptr;
            if ((r = logits_argmax(s, la2))) return r;
        }
        return 0;
    };
    if (use_px) {
        if ((rc = handoff(sx, st, ev++))) return rc;            // lane B's weight images before their first use on this stream
        // (vid_rnn's h_t rows arrive in the GEMM's row image w.ph1 from the persistent kernel itself - SeqFwdX3Args::hblk; the k16
        // records past the last column slice are zeroed here)
        {
            const size_t kc0 = (size_t)cdiv(H, 16), kc1 = (size_t)(w.ph1.kpad / 16);
            if (kc1 > kc0) S2VT_HIP(hipMemset2DAsync(w.ph1.p + kc0 * 3072, (size_t)64 * w.ph1.ld * 2, 0, (kc1 - kc0) * 6144, (size_t)(T * B / 64), st));
        }
        auto gx2_block = [&](int t0, int t1) -> int {           // vid_out half of word_rnn's gate input for steps [t0, t1) (+ biases)
            return pgemm(la, (t1 - t0) * B, 4 * H, H, w.ph1, t0 * B, 0, kc.wv, 0, 0, w.gx2 + t0 * B4H, 4 * H, ID, w.bsum2, false);
        };
        const std::vector<int> be = pipe_bounds(L, L, balanced_block(L, blk));     // blocks over the L encode steps
        const int nb = (int)be.size() - 1;
        // vid_rnn's blocks: the encode blocks and ONE block of its decode-phase steps (no input, no token) - the partner of word_rnn's
        // last encode block, which used to run alone on half of the device
        const int Tend = enc ? L + enc->depth : T;
        std::vector<int> bv(be);
        if (Tend > L) bv.push_back(L + (be[nb] - be[nb - 1]) < Tend ? L + (be[nb] - be[nb - 1]) : Tend);
        const int nbv = (int)bv.size() - 1;
        for (int k = 0; k <= nb; ++k) {          // stage k: vid_rnn block k next to word_rnn block k-1 (as in s2vt_train_forward)
            const bool hv = k < nbv, hw = k >= 1;
            SeqFwdX3Args av, aw;
            if (hv) {
                av = persist_fwd_x3_args(bv[k], bv[k + 1], B, H, T, w.xkp, w.gx1, L, w.bsum1, kc.xw1, w.xh1, w.h1, w.c1_all, w.psync_a, w.err + 1);
                av.no_stash = 1;
                av.hblk = w.ph1.p; av.ldhblk = w.ph1.ld;
            }
            if (hw) {
                aw = persist_fwd_x3_args(be[k - 1], be[k], B, H, L, w.xkp, w.gx2, L, w.bsum2, kc.xw2, w.xh2, w.h2_all, w.c2_all, w.psync_b, w.err + 1);
                aw.no_stash = 1;
            }
            {
                ProfScope ps(st, K_STEP_FWD, (hv ? bv[k + 1] - bv[k] : 0) + (hw ? be[k] - be[k - 1] : 0));
                if (hv && hw) rc = lstm_seq_fwd_x3_persist2(st, av, &aw);
                else rc = lstm_seq_fwd_x3_persist2(st, hv ? av : aw, nullptr);
                if (rc) return rc;
            }
            if (hv && (rc = gx2_block(bv[k], bv[k + 1]))) return rc;
        }
        const int tv = bv.back();                // vid_rnn steps done so far (>= L)
        if (enc) {       // the encode phase was what was asked for: the states after step L - 1
            const size_t nb_ = (size_t)BH * sizeof(float);
            S2VT_HIP(hipMemcpyAsync(enc->vid_h, w.h1 + (int64_t)(L - 1) * BH, nb_, hipMemcpyDeviceToDevice, st));
            S2VT_HIP(hipMemcpyAsync(enc->vid_c, w.c1_all + (int64_t)(L - 1) * BH, nb_, hipMemcpyDeviceToDevice, st));
            S2VT_HIP(hipMemcpyAsync(enc->word_h, w.h2_all + (int64_t)(L - 1) * BH, nb_, hipMemcpyDeviceToDevice, st));
            S2VT_HIP(hipMemcpyAsync(enc->word_c, w.c2_all + (int64_t)(L - 1) * BH, nb_, hipMemcpyDeviceToDevice, st));
            if (enc->depth > 0) {
                // vid_rnn's decode-phase steps take no input and see no token (S2VTModel.py:208-210 inside the depth loop): the
                // first `depth` of them in one launch, their half of word_rnn's gate input in one GEMM
                const int Td = L + enc->depth;
                if (tv < Td) {
                    SeqFwdX3Args av = persist_fwd_x3_args(tv, Td, B, H, T, w.xkp, w.gx1, L, w.bsum1, kc.xw1, w.xh1, w.h1, w.c1_all, w.psync_a, w.err + 1);
                    av.no_stash = 1;
                    av.hblk = w.ph1.p; av.ldhblk = w.ph1.ld;
                    {
                        ProfScope ps(st, K_STEP_FWD, Td - tv);
                        if ((rc = lstm_seq_fwd_x3_persist2(st, av, nullptr))) return rc;
                    }
                    if ((rc = gx2_block(tv, Td))) return rc;
                }
                S2VT_HIP(hipMemcpyAsync(enc->gx_dec, w.gx2 + (int64_t)L * B4H, (size_t)enc->depth * B4H * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            return post_async_error(st, w.err);
        }
        if (tv < T) {   // the rest of vid_rnn's decode steps (no input: bias only): one launch that may use the whole device
            SeqFwdX3Args av = persist_fwd_x3_args(tv, T, B, H, T, w.xkp, w.gx1, L, w.bsum1, kc.xw1, w.xh1, w.h1, w.c1_all, w.psync_a, w.err + 1);
            av.no_stash = 1;
            av.hblk = w.ph1.p; av.ldhblk = w.ph1.ld;
            {
                ProfScope ps(st, K_STEP_FWD, T - tv);
                if ((rc = lstm_seq_fwd_x3_persist2(st, av, nullptr))) return rc;
            }
            if ((rc = gx2_block(tv, T))) return rc;
        }
        // The 79 token-dependent steps.  A decode step is two dependent launches (word_rnn step, out_linear + argmax) that
        // each leave part of the chip idle (188 of 256 compute units in the argmax; launch gaps and tails between the two) and
        // batch rows never interact: at B % 128 == 0 the two halves of the batch run as two INDEPENDENT chains on the two
        // streams, so one half's step kernel fills the other half's gaps
        // Fused schedule (s2vt_set_decode_schedule(1), the default; option "decode_fused" = 0 selects the two-chain schedule below): h_t·W_hh^T of step t+1 does not depend on step t's token - only the
        // per-token rows of the gate table do - so it is computed BESIDE step t's out_linear + argmax, by the same launch: W_hh's
        // 4H rows are 63 more row blocks of the plane-path argmax kernel (188 + 63 workgroups: one wave of the 256 compute
        // units), which write their products to w.zbuf instead of reducing them.  A one-thread-per-cell launch then finishes
        // step t+1 (gates = z + gx + table row of the token, in the fused step's order).  Two launches per step on ONE stream,
        // and the chain is argmax + cell update instead of argmax + recurrent GEMM + cell update.
        if (decode_schedule() == 1 && ax3 && use_tab && kc.whh.p && kc.whh.kpad == kc.wo.kpad) {
            auto pair = [&](int t, bool with_logits, bool with_z) -> int {     // logits + argmax of step t (h_t planes) | z of step t+1
                ProfScope ps(st, K_ARGMAX, 1);
                ArgmaxX3Args ax;
                memset(&ax, 0, sizeof(ax));
                ax.B = B; ax.V = V; ax.K = kc.wo.kpad;
                ax.W = kc.wo.p; ax.ldw = kc.wo.ld;
                ax.Hp = w.ph2.p; ax.ldh = w.ph2.ld;
                ax.bias = p->out_b;
                ax.packed = w.packed + (int64_t)(with_logits ? t - L : 0) * B;
                if (with_z) { ax.W2 = kc.whh.p; ax.ldw2 = kc.whh.ld; ax.M2 = 4 * H; ax.z = w.zbuf; ax.ldz = 4 * (int64_t)H; }
                ax.v_off = with_logits ? 0 : cdiv(V, 64);
                return logits_argmax_x3(st, ax);
            };
            // h_{L-1} of the encode phase as blocked planes, then z(L) alone
            if ((rc = handoff(sx, st, ev++))) return rc;
            if ((rc = psplit(la, w.ph2, 0, w.h2_all + (int64_t)(L - 1) * BH, H, ID, B, H))) return rc;
            if ((rc = pair(L, false, true))) return rc;
            for (int t = L; t < T; ++t) {
                StepFwdArgs a = word_args(t, nullptr, t == L ? w.c2_all + (int64_t)(L - 1) * BH : w.c2, 0, B);
                a.z_out = w.zbuf; a.ldz = 4 * (int64_t)H;
                {
                    ProfScope ps(st, K_STEP_FWD, 1);
                    if ((rc = lstm_cell_pointwise(st, a))) return rc;
                }
                if ((rc = pair(t, true, t + 1 < T))) return rc;
            }
            if ((rc = unpack_tokens(st, w.packed, L - 1, B, ids))) return rc;
            return post_async_error(st, w.err);
        }
        const int nh = (ax3 && B % 128 == 0 && sx != st) ? 2 : 1;
        if (nh == 2 && (rc = handoff(st, sx, ev++))) return rc;
        for (int t = L; t < T; ++t)
            for (int hf = 0; hf < nh; ++hf)
                if ((rc = word_step(hf ? sx : st, t, t == L ? w.h2_all + (int64_t)(L - 1) * BH : w.h2 + ((t - 1) & 1) * BH,
                                    t == L ? w.c2_all + (int64_t)(L - 1) * BH : w.c2, hf * (B / nh), B / nh)))
                    return rc;
        if (nh == 2 && (rc = handoff(sx, st, ev++))) return rc;
        if ((rc = unpack_tokens(st, w.packed, L - 1, B, ids))) return rc;
        return post_async_error(st, w.err);                   // (a timed-out hand-off surfaces like the train path's)
    }
    const std::vector<int> bd = pipe_bounds(T, L, blk);
    for (size_t k = 0; k + 1 < bd.size(); ++k) {
        const int t0 = bd[k], t1 = bd[k + 1];
        {   // lane A: vid_rnn over all T steps (S2VTModel.py:64-67); c updated in place, h kept for the word layer
            ProfScope ps(st, K_STEP_FWD, t1 - t0);
            for (int t = t0; t < t1; ++t) {
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
        if ((rc = handoff(st, sx, ev++))) return rc;
        // lane B: vid_out half of the word_rnn gate input for this block (+ biases)
        if (x3) {
            if ((rc = psplit(lb, w.ph1, t0 * B, w.h1 + t0 * BH, H, ID, (t1 - t0) * B, H))) return rc;
            if ((rc = pgemm(lb, (t1 - t0) * B, 4 * H, H, w.ph1, t0 * B, 0, kc.wv, 0, 0, w.gx2 + t0 * B4H, 4 * H, ID, w.bsum2,
                            false)))
                return rc;
        } else {
            if ((rc = lgemm(lb, true, true, (t1 - t0) * B, 4 * H, H, w.h1 + t0 * BH, H, ID, p->word_w_ih + E, E + H, ID,
                            w.gx2 + t0 * B4H, 4 * H, ID, w.bsum2, false)))
                return rc;
        }
        for (int t = t0; t < t1; ++t)
            if ((rc = word_step(sx, t, t ? w.h2 + ((t - 1) & 1) * BH : nullptr, t ? w.c2 : nullptr))) return rc;
    }
    if ((rc = handoff(sx, st, ev++))) return rc;
    if ((rc = unpack_tokens(st, w.packed, L - 1, B, ids))) return rc;
    return post_async_error(st, w.err);
}

int s2vt_decode_step_argmax(int32_t B, int32_t H, int32_t V, const float* h, const float* w_out, const float* b_out,
                            unsigned long long* packed, void* stream) {
    LogitsArgmaxArgs la;
    la.B = B; la.H = H; la.V = V; la.h = h; la.ldh = H; la.w_out = w_out; la.ldw = H; la.b_out = b_out;
    la.packed = packed;
    la.stamps = nullptr;
#ifdef S2VT_EXPERIMENT_STAMPS
    la.stamps = g_xstamps;
#endif
    ProfScope ps((hipStream_t)stream, K_ARGMAX, 1);
    return logits_argmax((hipStream_t)stream, la);
}

// The same decode step on the bf16 matrix cores (argmax_x3.hip): both operands are split into blocked 3-plane images in the
// caller's workspace first (inside s2vt_greedy_decode W_o is split once per call, h_t once per step).
static size_t argmax_x3_ws_bytes(int B, int H, int V) {
    const size_t kp = (size_t)pad64(H);
    return (rows64((size_t)V) + rows64((size_t)B)) * 3 * kp * sizeof(unsigned short) + 512;
}
size_t s2vt_decode_step_argmax_x3_workspace_bytes(int32_t B, int32_t H, int32_t V) {
    return (B > 0 && H > 0 && V > 0) ? argmax_x3_ws_bytes(B, H, V) : 0;
}
int s2vt_decode_step_argmax_x3(int32_t B, int32_t H, int32_t V, const float* h, const float* w_out, const float* b_out,
                               unsigned long long* packed, void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(B > 0 && H > 0 && V > 0 && h && w_out && packed && workspace, "s2vt_decode_step_argmax_x3: bad arguments");
    S2VT_REQUIRE(workspace_bytes >= argmax_x3_ws_bytes(B, H, V), "s2vt_decode_step_argmax_x3: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int kp = pad64(H);
    Carver c{reinterpret_cast<char*>(workspace), 0, 0};
    unsigned short* wp = c.take<unsigned short>(rows64((size_t)V) * 3 * kp);
    unsigned short* hp = c.take<unsigned short>(rows64((size_t)B) * 3 * kp);
    int rc;
    if ((rc = split_planes(st, 3, false, w_out, H, ID, V, H, wp, 3 * (int64_t)kp, kp, (int)rows64((size_t)V)))) return rc;
    if ((rc = split_planes(st, 3, false, h, H, ID, B, H, hp, 3 * (int64_t)kp, kp, (int)rows64((size_t)B)))) return rc;
    ArgmaxX3Args ax;
    memset(&ax, 0, sizeof(ax));
    ax.B = B; ax.V = V; ax.K = kp;
    ax.W = wp; ax.ldw = 3 * (int64_t)kp;
    ax.Hp = hp; ax.ldh = 3 * (int64_t)kp;
    ax.bias = b_out;
    ax.packed = packed;
    ax.dbg = 0;
    ax.stamps = nullptr;
#ifdef S2VT_EXPERIMENT_STAMPS
    ax.stamps = g_xstamps;
#endif
    ProfScope ps(st, K_ARGMAX, 1);
    return logits_argmax_x3(st, ax);
}


}
