// s2vt_train_forward / s2vt_train_backward (S2VTModel.py:48-81 and its autograd; train.py:116-127): workspace carving, the
// two-lane / persistent launch sequences of the plane drivers and of the fp32-MFMA driver, gradient-group events for the
// data-parallel overlap, MaskCriterion's backward fused into the hand-over.
#include "api_internal.h"

namespace s2vt {

struct TrainWS {
    float *bsum1, *bsum2, *x1, *s1, *h1, *c1, *s2, *h2, *c2;
    float *wt1, *wt2, *dh1, *dh2dec, *dx1, *de, *dc1, *dc2, *colsum_a, *colsum_b, *colsum_c, *gws_a, *gws_b;
    float* ce_alpha;         // [1] mantissa of the mean-CE scale (bf16 mode, fused criterion backward: CeGradArgs::alpha_out)
    size_t gws_floats;
    int32_t* tok;
    int* embws;              // embedding_grad scratch (heavy-token list)
    int* err;                // [0] target id out of range, [1] persistent-recurrence hand-off timed out
    unsigned int *psync_a, *psync_b;     // hand-off counters of the persistent recurrence kernels (one block per lane)
    unsigned short *xw1, *xw2, *xh1, *xh2;   // split-precision persistent forward (lstm_persist_x3.hip): W_hh planes [3][4H][Kp],
    int64_t xkp;                             // h_t planes [3][T*B][Kp] per layer; Kp = H rounded up to 64 (0: H > 1024, no images)
    bool xfwd;                               // the four images above are provided (the persistent x3 forward is selectable)
    unsigned short *xwt1, *xwt2;             // split-precision persistent BPTT: W_hh^T planes [3][Kp][4 Hp] per layer and the
    float *xpart1, *xpart2;                  // partial-sum rings [xnslots][B/32][nC][nC][32][16] (xnslots = 0: not provided)
    int xnslots; int64_t xpslot, xhp;
    size_t bytes;
};

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
    w.embws = c.take<int>(embedding_grad_ws_ints((int64_t)(L - 1) * B, (int)d.V));
    w.err = c.take<int>(4);
    w.psync_a = c.take<unsigned int>(lstm_persist_sync_bytes() / sizeof(unsigned int));
    w.psync_b = c.take<unsigned int>(lstm_persist_sync_bytes() / sizeof(unsigned int));
    w.xkp = (H <= 1024) ? (int64_t)((H + 63) / 64 * 64) : 0;
    // images of the split-precision persistent forward: only where that kernel can be selected (the shape / mode part of
    // train_forward_x3's predicate - no device query here: s2vt_train_workspace_bytes has no side effect and works without a
    // GPU); the bf16 configuration (gemm mode 1) and the fp32-MFMA mode never read them (0.5 GB at B = 256)
    w.xfwd = w.xkp > 0 && gemm_mode() == 3 && B % 64 == 0 && pipe_block() > 0 && persist_x3_fwd_on();
    w.xw1 = c.take<unsigned short>(w.xfwd ? 3 * 4 * H * w.xkp : 0);
    w.xw2 = c.take<unsigned short>(w.xfwd ? 3 * 4 * H * w.xkp : 0);
    w.xh1 = c.take<unsigned short>(w.xfwd ? 3 * T * B * w.xkp : 0);
    w.xh2 = c.take<unsigned short>(w.xfwd ? 3 * T * B * w.xkp : 0);
    {   // ring slots: one more than the longest block of the backward's pipeline (a slot is written once per launch)
        // (no device query in a workspace-size function: the rings are provided wherever the option allows the kernel at all)
        const bool can = w.xkp > 0 && B % 32 == 0 && pipe_block() > 0 && option(O_PERSIST_X3_BWD) != 0 && persist_on() &&
                         (option(O_PERSIST_X3_BWD) == 1 || B * cdiv((int)H, 16) / 32 * 2 <= 256);
        const size_t maxblk = (size_t)pipe_block() < T ? (size_t)pipe_block() : T;
        w.xnslots = can ? (int)maxblk + 1 : 0;
        w.xhp = (int64_t)((H + 15) / 16 * 16);
        w.xpslot = can ? (int64_t)lstm_seq_bwd_x3_part_slot_floats((int)B, (int)H) : 0;
        w.xwt1 = c.take<unsigned short>(can ? 3 * (size_t)w.xkp * 4 * w.xhp : 0);
        w.xwt2 = c.take<unsigned short>(can ? 3 * (size_t)w.xkp * 4 * w.xhp : 0);
        w.xpart1 = c.take<float>((size_t)w.xnslots * w.xpslot);
        w.xpart2 = c.take<float>((size_t)w.xnslots * w.xpslot);
    }
    // backward-only scratch (two of everything that the two concurrently running layers touch)
    w.wt1 = c.take<float>(H * 4 * H);
    w.wt2 = c.take<float>(H * 4 * H);
    w.dh1 = c.take<float>(T * B * H);
    w.dh2dec = c.take<float>((L - 1) * B * H);
    w.dx1 = c.take<float>(L * B * H);
    w.de = c.take<float>((L - 1) * B * E);
    w.dc1 = c.take<float>(B * H);
    w.dc2 = c.take<float>(B * H);
    w.ce_alpha = c.take<float>(64);
    size_t cs = 2 * colsum_partial_floats((int64_t)T * B, (int)(4 * H));      // (chunks of 32 rows when the persistent BPTT writes them)
    size_t cs2 = colsum_partial_floats((int64_t)(L - 1) * B, (int)V);
    size_t cs3 = colsum_partial_floats((int64_t)L * B, (int)H);
    const size_t csm = cs > cs2 ? (cs > cs3 ? cs : cs3) : (cs2 > cs3 ? cs2 : cs3);
    w.colsum_a = c.take<float>(csm);
    w.colsum_b = c.take<float>(csm);
    w.colsum_c = c.take<float>(csm);
    w.gws_floats = gemm_ws_floats(d);
    w.gws_a = c.take<float>(w.gws_floats);
    w.gws_b = c.take<float>(w.gws_floats);
    w.bytes = align_up(c.off, 256);
    return w;
}

// "Gradient group is final" events of the last s2vt_train_backward on this thread's device (data-parallel overlap):
// group 0 = out_linear (weight, bias), group 1 = word_rnn (4 tensors) + embedding; the rest is final with the call's stream.
static hipEvent_t g_grad_ev[2] = {nullptr, nullptr};
static bool g_grad_ev_set[2] = {false, false};
// Order check of the data-parallel overlap (s2vt_backward_order): a persistent BPTT launch needs every one of its workgroups
// resident, so the event that releases the out_linear all-reduce (group 0) must have been recorded LAST behind the last such
// launch of the backward - g_bwd_group0_after = persistent BPTT launches enqueued when group 0's event was last recorded
static int g_bwd_persist_launches = 0, g_bwd_group0_after = 0;
static int grads_ready(int group, hipStream_t s) {
    if (group == 0) {
        g_bwd_group0_after = g_bwd_persist_launches;
        cu_reserve_window(true);         // from here on a collective's kernels may hold compute units beside this backward's GEMMs
    }
    if (!g_grad_ev[group]) S2VT_HIP(hipEventCreateWithFlags(&g_grad_ev[group], hipEventDisableTiming));
    // inside a capture nothing is recorded (an event recorded on a capturing stream cannot be waited for from outside, and
    // external event-record nodes are refused by this runtime): the backward driver records both groups behind the graph
    // launch instead, so under s2vt_set_graph_mode(1) the gradient all-reduce follows the backward rather than overlapping it
    if (!graph_capturing()) S2VT_HIP(hipEventRecord(g_grad_ev[group], s));
    g_grad_ev_set[group] = true;
    return 0;
}

struct PlaneWS {
    // forward
    PB feats, wf, x1, wih1, h1, we, wv, emb, h2r, wo, whh1, whh2;
    // backward
    PB dlog, woT, wvT, weT, wih1T, dg2, dg1, whh1T, whh2T;
    PB h2decB;               // decode-step hidden states as ROW planes in batch-major order (the k order of dlogits' rows): dW_o's B operand
    size_t bytes;
};

static PlaneWS carve_planes(const s2vt_dims& d, void* base) {
    const size_t B = d.B, L = d.L, F = d.F, H = d.H, E = d.E, V = d.V, T = 2 * L - 1, R = (L - 1) * B;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    auto mk = [&](size_t rows, size_t k) {
        PB b;
        b.kpad = pad64((int)k);
        b.ld = (int64_t)XP * b.kpad;
        b.p = c.take<unsigned short>(rows64(rows) * (size_t)b.ld);
        return b;
    };
    PlaneWS w;
    w.feats = mk(B * L, F);   w.wf = mk(H, F);       w.x1 = mk(L * B, H);    w.wih1 = mk(4 * H, H);
    w.h1 = mk(T * B, H);      w.we = mk(4 * H, E);   w.wv = mk(4 * H, H);    w.emb = mk(R, E);
    w.h2r = mk(T * B, H);     w.wo = mk(V, H);
    if (XP == 1) {            // bf16 mode: recurrent weights as bf16 rows (forward) and transposed rows (BPTT)
        w.whh1 = mk(4 * H, H); w.whh2 = mk(4 * H, H); w.whh1T = mk(H, 4 * H); w.whh2T = mk(H, 4 * H);
    } else {
        w.whh1 = w.whh2 = w.whh1T = w.whh2T = PB{nullptr, 0, 0};
    }
    w.dlog = mk(R, V);        w.woT = mk(H, V);
    w.wvT = mk(H, 4 * H);     w.weT = mk(E, 4 * H);  w.wih1T = mk(H, 4 * H);
    w.dg2 = mk(T * B, 4 * H); w.dg1 = mk(T * B, 4 * H);
    w.h2decB = mk(R, H);
    w.bytes = align_up(c.off, 256);
    return w;
}

// What a forward was run with, keyed by its workspace: s2vt_train_backward must find the same arithmetic mode and
// recurrence schedule (they decide how the workspace is carved and which images the forward left in it), otherwise it
// refuses instead of reading a differently carved workspace.  Host-side only.
struct FwdRecord { s2vt_dims d; int gemm_mode, planes, persist, blk; unsigned long long seq; bool dlog_ready; };
static std::map<const void*, FwdRecord> g_fwd_records;
static std::mutex g_fwd_mutex;          // autograd runs the backward on its own thread
static unsigned long long g_fwd_seq = 0;
static void record_forward(const void* ws, const s2vt_dims& d, bool planes) {
    std::lock_guard<std::mutex> lock(g_fwd_mutex);
    if (g_fwd_records.size() >= 64 && !g_fwd_records.count(ws)) {     // forwards that never ran a backward (validation,
        auto oldest = g_fwd_records.begin();                          // forward-only tools): the OLDEST record goes, never
        for (auto it = g_fwd_records.begin(); it != g_fwd_records.end(); ++it)      // one of a forward still awaiting its backward
            if (it->second.seq < oldest->second.seq) oldest = it;
        g_fwd_records.erase(oldest);
    }
    g_fwd_records[ws] = FwdRecord{d, gemm_mode(), planes ? ((gemm_mode() == 1) ? 1 : 3) : 0, persist_bits(), pipe_block(), ++g_fwd_seq, false};
}
static int check_forward_record(const void* ws, const s2vt_dims& d, bool planes, bool* dlog_ready = nullptr) {
    FwdRecord r;
    {
        std::lock_guard<std::mutex> lock(g_fwd_mutex);
        auto it = g_fwd_records.find(ws);
        S2VT_REQUIRE(it != g_fwd_records.end(), "s2vt_train_backward: no s2vt_train_forward has run on this workspace");
        r = it->second;
        g_fwd_records.erase(it);
    }
    if (dlog_ready) *dlog_ready = r.dlog_ready;
    S2VT_REQUIRE(memcmp(&r.d, &d, sizeof(d)) == 0, "s2vt_train_backward: dims differ from the forward that filled this workspace");
    const int planes_now = planes ? ((gemm_mode() == 1) ? 1 : 3) : 0;
    S2VT_REQUIRE(r.gemm_mode == gemm_mode() && r.planes == planes_now && r.persist == persist_bits() && r.blk == pipe_block(),
                 "s2vt_train_backward: the forward ran with gemm mode %d / recurrence options %d / pipeline block %d, now %d / %d / %d: the "
                 "workspace layout differs (do not change gemm_mode / persist / persist_x3_fwd / persist_x3_bwd / pipe_block "
                 "between a forward and its backward)", r.gemm_mode, r.persist, r.blk, gemm_mode(), persist_bits(), pipe_block());
    return 0;
}

// out_mask: optional out_drop mask (S2VTModel.py:79), time-major [(L-1)*B, H], entries 0 or 1/(1-p); nullptr = no dropout.
// The masked decode-step hidden states replace the row planes of the logits GEMM (the recurrence is done with them by then).
// (the weight-gradient GEMMs read the UNMASKED rows of q.h2r transposed in the backward, so the masked rows go to the scratch
// image q.h2decB - which the backward fills itself before it reads it - and *a_img / *a_row0 name the logits GEMM's operand)
static int masked_logits_planes(const Lane& ln, const TrainWS& w, const PlaneWS& q, const float* out_mask, int B, int L, int H,
                                const PB** a_img, int* a_row0) {
    *a_img = &q.h2r; *a_row0 = L * B;
    if (!out_mask) return 0;
    const int R = (L - 1) * B;
    int rc;
    if ((rc = mul_vectors(ln.s, w.h2 + (int64_t)L * B * H, out_mask, w.dh2dec, (int64_t)R * H))) return rc;   // dh2dec: free in the forward
    *a_img = &q.h2decB; *a_row0 = 0;
    return psplit(ln, **a_img, *a_row0, w.dh2dec, H, ID, R, H);
}

static int train_forward_x3(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                            int64_t targets_ld, float* logits, const TrainWS& w, const PlaneWS& q, hipStream_t st,
                            const float* out_mask) {
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1, R = (L - 1) * B;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    const int blk = pipe_block();
    hipStream_t sx = st;
    int rc;
    if (blk > 0 && (rc = side_stream(st, &sx))) return rc;
    const Lane la{st, w.gws_a, w.gws_floats, w.colsum_a};
    const Lane lb{sx, w.gws_b, w.gws_floats, w.colsum_b};
    size_t ev = 0;
    if ((rc = add_vectors(st, p->vid_b_ih, p->vid_b_hh, w.bsum1, 4 * H))) return rc;
    if ((rc = add_vectors(st, p->word_b_ih, p->word_b_hh, w.bsum2, 4 * H))) return rc;
    if ((rc = fill_zero(st, w.err, 4 * sizeof(int)))) return rc;
    if ((rc = targets_to_time_major(st, targets, B, L - 1, targets_ld, V, w.tok, w.err))) return rc;
    const bool bf = (XP == 1);        // bf16 mode: bf16 timestep kernels write the h planes themselves
    if (bf) {   // zero the k padding of the bf16 h row images (valid columns are written by the step kernels)
        if ((rc = zero_pad_cols_u16(st, q.h1.p, (int64_t)T * B, q.h1.ld, H, q.h1.kpad))) return rc;
        if ((rc = zero_pad_cols_u16(st, q.h2r.p, (int64_t)T * B, q.h2r.ld, H, q.h2r.kpad))) return rc;
        if ((rc = pdual(la, p->vid_w_hh, H, ID, 4 * H, H, &q.whh1, 0, &q.whh1T, 0, nullptr))) return rc;
    }
    if ((rc = handoff(st, sx, ev++))) return rc;
    if (bf && (rc = pdual(lb, p->word_w_hh, H, ID, 4 * H, H, &q.whh2, 0, &q.whh2T, 0, nullptr))) return rc;
    const bool px3_fwd = !bf && XP == 3 && blk > 0 && w.xfwd && persist_x3_fwd_on() && lstm_seq_fwd_x3_persist_supported(B, H);
    if (px3_fwd) {   // W_hh of both layers as row-major planes for the persistent split-precision recurrence
        if ((rc = split3_rows(sx, p->vid_w_hh, H, 4 * H, H, (int)w.xkp, w.xw1, 4 * (int64_t)H * w.xkp))) return rc;
        if ((rc = split3_rows(sx, p->word_w_hh, H, 4 * H, H, (int)w.xkp, w.xw2, 4 * (int64_t)H * w.xkp))) return rc;
    }
    // lane B: word_rnn / out_linear weights and the embedded caption words as planes; embedded-word half of gx2
    // (row planes for this forward, transposed planes for the coming backward: one read of each tensor)
    if ((rc = pdual(lb, p->word_w_ih, E + H, ID, 4 * H, E, &q.we, 0, &q.weT, 0, nullptr))) return rc;
    if ((rc = pdual(lb, p->word_w_ih + E, E + H, ID, 4 * H, H, &q.wv, 0, &q.wvT, 0, nullptr))) return rc;
    if ((rc = pdual(lb, p->out_w, H, ID, V, H, &q.wo, 0, &q.woT, 0, nullptr))) return rc;
    if ((rc = pdual(lb, p->emb_w, E, gather(w.tok), R, E, &q.emb, 0, nullptr, 0, nullptr))) return rc;
    // (co-run: the last gxe_c rows of this GEMM and the first lg_c decode steps of the logits GEMM run beside the one-layer first /
    // last stage of the persistent schedule below, on the compute units those leave idle - as dW_o does in the backward)
    const int f_wgs = px3_fwd ? lstm_seq_fwd_x3_persist_single_workgroups(B, H) : 0;
    const int f_cus = f_wgs > 0 ? (planned_compute_units() - f_wgs - 2) / 8 * 8 : 0;
    const bool stages3 = px3_fwd && pipe_bounds(T, L, balanced_block(L, blk)).size() >= 3;      // at least two blocks: one-layer first / last stages
    const int tenths = (stages3 && sx != st && f_cus >= 64 && R >= 2560 && B % 64 == 0) ? option(O_CORUN) : 0;
    const int gxe_c = (R / 64 * (2 * tenths < 8 ? 2 * tenths : 8) / 10) * 64;
    if ((rc = pgemm(lb, R - gxe_c, 4 * H, E, q.emb, 0, 0, q.we, 0, 0, w.s2 + (int64_t)L * B4H, 4 * H, ID, w.bsum2, false))) return rc;
    // lane A: feature projection and vid_rnn input GEMM                       S2VTModel.py:54, 64-67
    if ((rc = psplit(la, q.feats, 0, feats, F, ID, B * L, F))) return rc;
    if ((rc = psplit(la, q.wf, 0, p->feat_w, F, ID, H, F))) return rc;
    if ((rc = pdual(la, p->vid_w_ih, H, ID, 4 * H, H, &q.wih1, 0, &q.wih1T, 0, nullptr))) return rc;
    if ((rc = pgemm(la, B * L, H, F, q.feats, 0, 0, q.wf, 0, 0, w.x1, H, perm(L, B), p->feat_b, false))) return rc;
    if ((rc = pdual(la, w.x1, H, ID, L * B, H, &q.x1, 0, nullptr, 0, nullptr))) return rc;
    if ((rc = pgemm(la, L * B, 4 * H, H, q.x1, 0, 0, q.wih1, 0, 0, w.s1, 4 * H, ID, w.bsum1, false))) return rc;
    const bool pbf_fwd = bf && blk > 0 && persist_fwd_ok(B, H, q.whh1, q.h1);
    const std::vector<int> bd = pipe_bounds(T, L, (pbf_fwd || px3_fwd) ? balanced_block(L, blk) : blk);
    if (px3_fwd) {
        // fp32-equivalent persistent schedule (lstm_persist_x3.hip: split precision on the bf16 matrix cores), ONE stream:
        // stage k = vid_rnn block k next to word_rnn block k-1
        if ((rc = handoff(sx, st, ev++))) return rc;
        const int nb = (int)bd.size() - 1;
        // the kernel writes h_t into the GEMMs' row images itself (the hand-off payload's own 16-byte pieces); the k16 records past
        // the last column slice are zeroed here (H = 1000: units 1008..1023)
        for (const PB* img : {&q.h1, &q.h2r}) {
            const size_t kc0 = (size_t)cdiv(H, 16), kc1 = (size_t)(img->kpad / 16);
            if (kc1 > kc0) S2VT_HIP(hipMemset2DAsync(img->p + kc0 * 3072, (size_t)64 * img->ld * 2, 0, (kc1 - kc0) * 6144, (size_t)(T * B / 64), st));
        }
        // decode steps whose logits run beside the last (word_rnn-only) stage: their h2 rows are final before it starts
        int lg_c = (nb >= 2 && !out_mask) ? (L - 1) * ((2 * tenths + 1) / 3) / 10 : 0;
        if (lg_c > bd[nb - 1] - L) lg_c = bd[nb - 1] - L > 0 ? bd[nb - 1] - L : 0;
        for (int k = 0; k <= nb; ++k) {
            const bool hv = k < nb, hw = k >= 1;
            const bool co_first = k == 0 && nb >= 2 && gxe_c > 0, co_last = k == nb && lg_c > 0;
            if ((co_first || co_last) && (rc = handoff(st, sx, ev++))) return rc;       // (the part starts with the stage, not before it)
            {
            ProfScope ps(st, K_STEP_FWD, (hv ? bd[k + 1] - bd[k] : 0) + (hw ? bd[k] - bd[k - 1] : 0));
            SeqFwdX3Args av, aw;
            if (hv) av = persist_fwd_x3_args(bd[k], bd[k + 1], B, H, T, w.xkp, w.s1, L, w.bsum1, w.xw1, w.xh1, w.h1, w.c1, w.psync_a, w.err + 1);
            if (hw) aw = persist_fwd_x3_args(bd[k - 1], bd[k], B, H, T, w.xkp, w.s2, T, w.bsum2, w.xw2, w.xh2, w.h2, w.c2, w.psync_b, w.err + 1);
            av.hblk = q.h1.p; av.ldhblk = q.h1.ld;
            aw.hblk = q.h2r.p; aw.ldhblk = q.h2r.ld;
            if (hv && hw) rc = lstm_seq_fwd_x3_persist2(st, av, &aw);
            else rc = lstm_seq_fwd_x3_persist2(st, hv ? av : aw, nullptr);
            }
            if (rc) return rc;
            if (co_first || co_last) {
                {
                    CuPlanCap cap(f_cus);
                    if (co_first)
                        rc = pgemm(lb, gxe_c, 4 * H, E, q.emb, R - gxe_c, 0, q.we, 0, 0, w.s2 + (int64_t)L * B4H + (int64_t)(R - gxe_c) * 4 * H, 4 * H,
                                   ID, w.bsum2, false);
                    else
                        rc = pgemm(lb, lg_c * B, V, H, q.h2r, L * B, 0, q.wo, 0, 0, logits, V, perm(B, L - 1), p->out_b, false);
                }
                if (rc || (rc = handoff(sx, st, ev++))) return rc;
            }
            if (hv) {
                const int t0 = bd[k], t1 = bd[k + 1];
                const bool cap = t0 >= L;
                if ((rc = pgemm(la, (t1 - t0) * B, 4 * H, H, q.h1, t0 * B, 0, q.wv, 0, 0, w.s2 + t0 * B4H, 4 * H, ID,
                                cap ? nullptr : w.bsum2, cap)))
                    return rc;
            }
        }
        const PB* lg; int lg0;
        if ((rc = masked_logits_planes(la, w, q, out_mask, B, L, H, &lg, &lg0))) return rc;
        // (decode step t' of row (t', b) lands in logits row b (L-1) + t': a range of steps from t'0 on = the same row map, t'0 rows further)
        return pgemm(la, R - lg_c * B, V, H, *lg, lg0 + lg_c * B, 0, q.wo, 0, 0, logits + (int64_t)lg_c * V, V, perm(B, L - 1), p->out_b, false);
    }
    if (pbf_fwd) {
        // Persistent schedule, ONE stream: the launch of pipeline stage k runs vid_rnn block k next to word_rnn block k-1
        // (lstm_persist.hip: two workgroups per CU, each layer's W_hh slices resident in registers); between two
        // launches the plane split + input GEMM of the vid block just finished run alone on the chip.
        if ((rc = handoff(sx, st, ev++))) return rc;              // weight planes / embedded-word half from lane B
        const int nb = (int)bd.size() - 1;
        for (int k = 0; k <= nb; ++k) {
            const bool hv = k < nb, hw = k >= 1;
            SeqFwdBf16Args av, aw;
            if (hv) av = persist_fwd_args(bd[k], bd[k + 1], B, H, w.s1, L, w.bsum1, q.whh1, q.h1, w.h1, w.c1, w.psync_a, w.err + 1);
            if (hw) aw = persist_fwd_args(bd[k - 1], bd[k], B, H, w.s2, T, w.bsum2, q.whh2, q.h2r, w.h2, w.c2, w.psync_b, w.err + 1);
            {
                ProfScope ps(st, K_STEP_FWD, (hv ? bd[k + 1] - bd[k] : 0) + (hw ? bd[k] - bd[k - 1] : 0));
                if (hv && hw) rc = lstm_seq_fwd_bf16_persist2(st, av, &aw);
                else rc = lstm_seq_fwd_bf16_persist2(st, hv ? av : aw, nullptr);
                if (rc) return rc;
            }
            if (hv) {   // vid_out half of the word_rnn gate input for block k
                const int t0 = bd[k], t1 = bd[k + 1];
                const bool cap = t0 >= L;
                if ((rc = pgemm(la, (t1 - t0) * B, 4 * H, H, q.h1, t0 * B, 0, q.wv, 0, 0, w.s2 + t0 * B4H, 4 * H, ID,
                                cap ? nullptr : w.bsum2, cap)))
                    return rc;
            }
        }
        const PB* lg; int lg0;
        if ((rc = masked_logits_planes(la, w, q, out_mask, B, L, H, &lg, &lg0))) return rc;
        return pgemm(la, R, V, H, *lg, lg0, 0, q.wo, 0, 0, logits, V, perm(B, L - 1), p->out_b, false);
    }
    for (size_t k = 0; k + 1 < bd.size(); ++k) {
        const int t0 = bd[k], t1 = bd[k + 1];
        if (bf) {
            if ((rc = seq_fwd_bf16(st, t0, t1, B, H, w.s1, L, w.bsum1, q.whh1, q.h1, w.h1, w.c1))) return rc;
        } else {
            if ((rc = seq_fwd(st, t0, t1, B, H, w.s1, L, w.bsum1, p->vid_w_hh, w.h1, w.c1, true))) return rc;
        }
        if ((rc = handoff(st, sx, ev++))) return rc;
        const bool cap = t0 >= L;
        if ((rc = pdual(lb, w.h1 + t0 * BH, H, ID, (t1 - t0) * B, H, bf ? nullptr : &q.h1, t0 * B, nullptr, t0 * B, nullptr)))
            return rc;
        if ((rc = pgemm(lb, (t1 - t0) * B, 4 * H, H, q.h1, t0 * B, 0, q.wv, 0, 0, w.s2 + t0 * B4H, 4 * H, ID,
                        cap ? nullptr : w.bsum2, cap)))
            return rc;
        if (bf) {
            if ((rc = seq_fwd_bf16(sx, t0, t1, B, H, w.s2, T, w.bsum2, q.whh2, q.h2r, w.h2, w.c2))) return rc;
        } else {
            if ((rc = seq_fwd(sx, t0, t1, B, H, w.s2, T, w.bsum2, p->word_w_hh, w.h2, w.c2, true))) return rc;
        }
        // h2 planes: transposed (k = time-major row) for dW_hh2; row planes of the decode steps for the logits GEMM
        if ((rc = pdual(lb, w.h2 + t0 * BH, H, ID, (t1 - t0) * B, H, bf ? nullptr : &q.h2r, t0 * B, nullptr,
                        t0 * B, nullptr)))
            return rc;
    }
    const PB* lg; int lg0;
    if ((rc = masked_logits_planes(lb, w, q, out_mask, B, L, H, &lg, &lg0))) return rc;
    if ((rc = pgemm(lb, R, V, H, *lg, lg0, 0, q.wo, 0, 0, logits, V, perm(B, L - 1), p->out_b, false))) return rc;
    return handoff(sx, st, ev++);
}

static int train_backward_x3(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                             const s2vt_grads* g, float* dfeats, const TrainWS& w, const PlaneWS& q, hipStream_t st,
                             const float* out_mask, bool dlog_ready) {
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1, R = (L - 1) * B;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    const int blk = pipe_block();
    hipStream_t sx = st;
    int rc;
    if (blk > 0 && (rc = side_stream(st, &sx))) return rc;
    const Lane la{st, w.gws_a, w.gws_floats, w.colsum_a};     // word_rnn lane
    const Lane lb{sx, w.gws_b, w.gws_floats, w.colsum_b};     // vid_rnn lane
    size_t ev = 0;
    const bool bf = (XP == 1);
    if (bf) {   // zero the k padding of the bf16 dG row images
        if ((rc = zero_pad_cols_u16(st, q.dg2.p, (int64_t)T * B, q.dg2.ld, 4 * H, q.dg2.kpad))) return rc;
        if ((rc = zero_pad_cols_u16(st, q.dg1.p, (int64_t)T * B, q.dg1.ld, 4 * H, q.dg1.kpad))) return rc;
    }
    if ((rc = handoff(st, sx, ev++))) return rc;
    // lane A: dlogits planes in both orientations + its column sums (one read), gradient into the decode-step
    // hidden states (k = V), then word_rnn BPTT.  (W^T planes were written by the forward.)
    // (dlog_ready: s2vt_mean_ce_backward_fused wrote these planes and partial sums straight from the logits)
    if (!dlog_ready && (rc = pdual(la, dlogits, V, ID, R, V, &q.dlog, 0, nullptr, 0, w.colsum_c))) return rc;
    if ((rc = handoff(st, sx, ev++))) return rc;
    if ((rc = pgemm(la, R, H, V, q.dlog, 0, 0, q.woT, 0, 0, w.dh2dec, H, perm(L - 1, B), nullptr, false))) return rc;
    // bf16 mode with the fused criterion backward: the dlogits planes carry the power of two of gout / rows only (split.hip); the
    // mantissa multiplies the two fp32 products of those planes - here and dW_o below (the bias gradient already has it)
    const bool ce_pow2 = dlog_ready && bf;
    if (ce_pow2 && (rc = scale_by_device_scalar(st, w.dh2dec, (int64_t)R * H, w.ce_alpha))) return rc;
    if (out_mask && (rc = mul_vectors(st, w.dh2dec, out_mask, w.dh2dec, (int64_t)R * H))) return rc;      // autograd of out_drop
    if (!bf && (rc = transpose_f32(st, p->word_w_hh, 4 * H, H, w.wt2))) return rc;       // (the bf16 BPTT reads the W_hh^T planes instead)
    const bool pbf_bwd = bf && blk > 0 && persist_on() && lstm_seq_bwd_bf16_persist_supported(B, H, q.dg2.kpad) && q.dg2.kpad == q.whh2T.kpad;
    const bool px3_bwd = !bf && XP == 3 && blk > 0 && w.xnslots > 0 && persist_x3_bwd_on(B, H) && lstm_seq_bwd_x3_persist_supported(B, H) &&
                         w.xnslots > (blk < T ? blk : T);
    // rows of dW_o's k range that run beside EACH of the two one-layer BPTT stages (0: no co-run): 3/10 of the rows each - a stage
    // (option corun, tenths) - a stage lasts about as long as that part takes on the idle half of the device (profiles/round5_corun.txt)
    const int x3_ns = px3_bwd ? lstm_seq_bwd_x3_persist_supported(B, H) : 0;                 // chains per workgroup
    const int one_layer_wgs = x3_ns > 0 ? (B / (32 * x3_ns)) * cdiv(H, 16) : 1 << 20;
    const int corun_cus = (planned_compute_units() - one_layer_wgs - 2) / 8 * 8;
    const int corun_k = (px3_bwd && option(O_CORUN) && sx != st && corun_cus >= 64 && R >= 2560) ? (R / 64 * option(O_CORUN) / 10) * 64 : 0;
    // lane B meanwhile: out_linear weight/bias gradients (k = batch-major row index) and W_hh1^T
    {
        const float* h2dec = w.h2 + L * BH;
        if (out_mask) {      // dW_o sees the masked hidden states (dx1 is free until the vid_rnn input gradient)
            if ((rc = mul_vectors(sx, w.h2 + L * BH, out_mask, w.dx1, (int64_t)R * H))) return rc;
            h2dec = w.dx1;
        }
        // rows in dlogits' (batch-major) order, read transposed by the GEMM
        if ((rc = psplit(lb, q.h2decB, 0, h2dec, H, perm(L - 1, B), R, H))) return rc;
        // (co-run: dW_o is not needed before the optimizer - its GEMM is cut over k = rows into three parts, two of which run beside the
        // one-layer stages of the persistent BPTT below, on the compute units those leave idle)
        if (!corun_k && (rc = pgemm_tt(lb, V, H, R, q.dlog, 0, q.h2decB, 0, g->out_w, H, ID, nullptr, false))) return rc;
        if (!corun_k && ce_pow2 && (rc = scale_by_device_scalar(sx, g->out_w, (int64_t)V * H, w.ce_alpha))) return rc;
    }
    if ((rc = colsum_finish(sx, w.colsum_c, cdiv(R, 64), V, g->out_b, false))) return rc;
    // (a persistent BPTT re-records "group 0 is final" behind its last launch - no collective may start beside one - so it is not
    // recorded here for those schedules: option cu_reserve counts from the release that holds)
    if (!corun_k && !pbf_bwd && !px3_bwd && (rc = grads_ready(0, sx))) return rc;
    if (!bf && (rc = transpose_f32(sx, p->vid_w_hh, 4 * H, H, w.wt1))) return rc;
    const std::vector<int> bd = pipe_bounds(T, L, (pbf_bwd || px3_bwd) ? balanced_block(L, blk) : blk);
    int bias_chunk = 64;      // rows per partial column sum of dG (32: written by the persistent split-precision BPTT itself)
    bool word_gemms_done = false, demb_done = false;                // (the one-layer BPTT schedule ran word_rnn's weight-gradient GEMMs and the
    size_t word_grads_ev = 0, demb_ev = 0;                          //  embedded-word gradient GEMM already, on lane B: their events;
    int hh1_t0 = 0;                                                 //  dW_hh1's timesteps >= hh1_t0 are done as well)
    const bool solo_sched = corun_k > 0 && bd.size() >= 3 && option(O_BPTT_SOLO) != 0;
    if (px3_bwd) {   // W_hh^T of both layers as planes (each on the lane that transposed it)
        if ((rc = split3_wt(st, w.wt2, H, (int)w.xkp, (int)w.xhp, w.xwt2, w.xkp * 4 * w.xhp))) return rc;
        if ((rc = split3_wt(sx, w.wt1, H, (int)w.xkp, (int)w.xhp, w.xwt1, w.xkp * 4 * w.xhp))) return rc;
    }
    if (px3_bwd) {
        // fp32-equivalent persistent schedule (split precision: lstm_persist_x3.hip), ONE stream: stage k = word_rnn BPTT of
        // block k next to vid_rnn BPTT of block k+1
        if ((rc = handoff(sx, st, ev++))) return rc;               // W_hh1^T and the out_linear gradients of lane B
        const int nb = (int)bd.size() - 1;
        // The kernel hands dG over as the GEMMs' row-plane image itself (+ its 32-row column sums for the bias gradients): its dG tile
        // is in LDS as planes anyway - no split pass reads dG back, the fp32 dG is never stored.  (H % 8: a 16-byte slot of the image
        // holds 8 consecutive units of one gate.)  The k padding [4H, pad64(4H)) of both images is zeroed here once per backward.
        const bool emit = H % 8 == 0;
        if (emit && q.dg2.kpad > 4 * H) {
            const size_t kc0 = (size_t)(4 * H / 16), kc1 = (size_t)(q.dg2.kpad / 16);       // k16 records [kc0, kc1) of every 64-row block
            const size_t first = (4 * H % 16) ? kc0 + 1 : kc0;                              // (4H % 16 == 8: the straddling record's upper half
            for (const PB* img : {&q.dg2, &q.dg1}) {                                        //  is zeroed piece by piece below)
                if (kc1 > first)
                    S2VT_HIP(hipMemset2DAsync(img->p + first * 3072, (size_t)64 * img->ld * 2, 0, (kc1 - first) * 6144, (size_t)(T * B / 64), st));
                if (4 * H % 16)
                    for (int pl = 0; pl < 3; ++pl)
                        S2VT_HIP(hipMemset2DAsync(img->p + kc0 * 3072 + (pl * 2 + 1) * 512, (size_t)64 * img->ld * 2, 0, 1024, (size_t)(T * B / 64), st));
            }
        }
        if (solo_sched) {
            // ONE layer per launch - word_rnn's blocks, then vid_rnn's (the word_rnn BPTT does not depend on vid_rnn's) - on half of the
            // compute units, and the backward's GEMMs on the other half (lane B, planned for the idle units): beside word_rnn's stages
            // dW_o (cut over k = rows, accumulated in a fixed order) and every finished block's dh1 GEMM; beside vid_rnn's stages
            // word_rnn's weight gradients.  A two-layer stage does two blocks in ~440 us with nothing beside it; two one-layer stages take
            // ~2 x 345 us and give half of the device to GEMMs for that long (profiles/round5_corun.txt).
            auto event_at = [&](hipStream_t s_, size_t* idx) -> int {
                hipEvent_t e;
                *idx = ev++;
                const int r = get_event(*idx, &e);
                if (r) return r;
                S2VT_HIP(hipEventRecord(e, s_));
                return 0;
            };
            auto wait_for = [&](hipStream_t s_, size_t idx) -> int {
                hipEvent_t e;
                const int r = get_event(idx, &e);
                if (r) return r;
                S2VT_HIP(hipStreamWaitEvent(s_, e, 0));
                return 0;
            };
            auto emit_args = [&](SeqBwdX3Args& a, bool word) {
                if (!emit) return;
                a.dgp = word ? q.dg2.p : q.dg1.p; a.lddgp = word ? q.dg2.ld : q.dg1.ld;
                a.colpart = word ? w.colsum_a : w.colsum_b; a.skip_dg = 1;
            };
            if ((rc = handoff(st, sx, ev++))) return rc;            // the parts start with the first stage, not beside the dh2 GEMM
            int wo_r = 0;                                           // next row of dW_o's k range
            auto wo_part = [&]() -> int {
                if (wo_r >= R) return 0;
                int kk = corun_k < R - wo_r ? corun_k : R - wo_r;
                if (R - wo_r - kk < 512) kk = R - wo_r;
                const int r = pgemm_tt(lb, V, H, kk, q.dlog, wo_r, q.h2decB, wo_r, g->out_w, H, ID, nullptr, wo_r > 0);
                wo_r += kk;
                return r;
            };
            std::vector<size_t> dh1_done((size_t)nb);
            size_t word_grads_done = 0;
            {
                CuPlanCap cap(corun_cus);
                for (int k = nb - 1; k >= 0; --k) {
                    const int t0 = bd[k], t1 = bd[k + 1];
                    {
                        ProfScope ps(st, K_STEP_BWD, t1 - t0);
                        SeqBwdX3Args aw = persist_bwd_x3_args(T, t0, t1, B, H, w.xkp, w.xhp, w.xwt2, w.dh2dec, L, w.c2, w.s2, w.dc2, w.xpart2, w.xpslot,
                                                              w.xnslots, w.psync_a, w.err + 1);
                        emit_args(aw, true);
                        if ((rc = lstm_seq_bwd_x3_persist2(st, aw, nullptr))) return rc;
                        ++g_bwd_persist_launches;
                    }
                    if ((rc = wo_part())) return rc;
                    if ((rc = handoff(st, sx, ev++))) return rc;    // dG2 rows of block k
                    if (!emit && (rc = pdual(lb, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, &q.dg2, t0 * B, nullptr, t0 * B,
                                             w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
                        return rc;
                    if ((rc = pgemm(lb, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false))) return rc;
                    if ((rc = event_at(sx, &dh1_done[(size_t)k]))) return rc;
                }
                while (wo_r < R)
                    if ((rc = wo_part())) return rc;
                // word_rnn's weight gradients (dG2 is complete): beside vid_rnn's stages
                if ((rc = pgemm_tt(lb, 4 * H, H, (T - 1) * B, q.dg2, B, q.h2r, 0, g->word_w_hh, H, ID, nullptr, false))) return rc;
                if ((rc = pgemm_tt(lb, 4 * H, H, T * B, q.dg2, 0, q.h1, 0, g->word_w_ih + E, E + H, ID, nullptr, false))) return rc;
                if ((rc = pgemm_tt(lb, 4 * H, E, R, q.dg2, L * B, q.emb, 0, g->word_w_ih, E + H, ID, nullptr, false))) return rc;
                if ((rc = event_at(sx, &word_grads_done))) return rc;
                // ... and the gradient into the embedded words (the embedding gradient's input)
                if ((rc = pgemm(lb, R, E, 4 * H, q.dg2, L * B, 0, q.weT, 0, 0, w.de, E, ID, nullptr, false))) return rc;
                if ((rc = event_at(sx, &demb_ev))) return rc;
                demb_done = true;
            }
            // dW_hh1 = dG1[t]^T h1[t-1] over t >= hh1_t0 (the blocks vid_rnn's BPTT finishes first) beside its later stages
            const int kc = nb / 2;
            if (nb >= 4) hh1_t0 = bd[kc];
            for (int k = nb - 1; k >= 0; --k) {
                const int t0 = bd[k], t1 = bd[k + 1];
                if ((rc = wait_for(st, dh1_done[(size_t)k]))) return rc;
                ProfScope ps(st, K_STEP_BWD, t1 - t0);
                SeqBwdX3Args av = persist_bwd_x3_args(T, t0, t1, B, H, w.xkp, w.xhp, w.xwt1, w.dh1, 0, w.c1, w.s1, w.dc1, w.xpart1, w.xpslot,
                                                      w.xnslots, w.psync_b, w.err + 1);
                emit_args(av, false);
                if ((rc = lstm_seq_bwd_x3_persist2(st, av, nullptr))) return rc;
                ++g_bwd_persist_launches;
                if (!emit && (rc = pdual(la, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, &q.dg1, t0 * B, nullptr, t0 * B,
                                         w.colsum_b + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
                if (hh1_t0 > 0 && k == kc) {
                    if ((rc = handoff(st, sx, ev++))) return rc;   // dG1 rows of the blocks [kc, nb)
                    CuPlanCap cap(corun_cus);
                    if ((rc = pgemm_tt(lb, 4 * H, H, (T - hh1_t0) * B, q.dg1, hh1_t0 * B, q.h1, (hh1_t0 - 1) * B, g->vid_w_hh, H, ID, nullptr, false)))
                        return rc;
                }
            }
            bias_chunk = emit ? 32 : 64;
            if ((rc = grads_ready(0, st))) return rc;              // (out_linear's gradients: released behind the last persistent launch)
            if ((rc = handoff(st, sx, ev++))) return rc;           // lane B's vid_rnn gradients need dG1
            word_gemms_done = true;
            word_grads_ev = word_grads_done;                       // lane A's tail ends with "word_rnn + embedding gradients final": after these
        } else
        for (int k = nb - 1; k >= -1; --k) {
            const bool hw = k >= 0, hv = k + 1 <= nb - 1;
            const bool solo = corun_k && nb >= 2 && (k == nb - 1 || k == -1);      // a one-layer stage: half of the compute units idle
            if (solo && (rc = handoff(st, sx, ev++))) return rc;                     // (the part starts with the stage, not before it)
            {
                ProfScope ps(st, K_STEP_BWD, (hw ? bd[k + 1] - bd[k] : 0) + (hv ? bd[k + 2] - bd[k + 1] : 0));
                SeqBwdX3Args aw, av;
                if (hw) aw = persist_bwd_x3_args(T, bd[k], bd[k + 1], B, H, w.xkp, w.xhp, w.xwt2, w.dh2dec, L, w.c2, w.s2, w.dc2,
                                                 w.xpart2, w.xpslot, w.xnslots, w.psync_a, w.err + 1);
                if (hv) av = persist_bwd_x3_args(T, bd[k + 1], bd[k + 2], B, H, w.xkp, w.xhp, w.xwt1, w.dh1, 0, w.c1, w.s1, w.dc1,
                                                 w.xpart1, w.xpslot, w.xnslots, w.psync_b, w.err + 1);
                if (emit) {
                    aw.dgp = q.dg2.p; aw.lddgp = q.dg2.ld; aw.colpart = w.colsum_a; aw.skip_dg = 1;
                    av.dgp = q.dg1.p; av.lddgp = q.dg1.ld; av.colpart = w.colsum_b; av.skip_dg = 1;
                }
                if (hw && hv) rc = lstm_seq_bwd_x3_persist2(st, aw, &av);
                else rc = lstm_seq_bwd_x3_persist2(st, hw ? aw : av, nullptr);
                ++g_bwd_persist_launches;
                if (rc) return rc;
            }
            if (solo) {      // dW_o rows [r0, r0 + corun_k) on lane B, planned for the units the stage leaves idle; the caller's stream
                const int r0 = (k == nb - 1) ? 0 : corun_k;              // goes on when both are done
                {
                    CuPlanCap cap(corun_cus);
                    if ((rc = pgemm_tt(lb, V, H, corun_k, q.dlog, r0, q.h2decB, r0, g->out_w, H, ID, nullptr, r0 > 0))) return rc;
                }
                if ((rc = handoff(sx, st, ev++))) return rc;
            }
            if (hw) {
                const int t0 = bd[k], t1 = bd[k + 1];
                if (!emit && (rc = pdual(la, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, &q.dg2, t0 * B, nullptr, t0 * B,
                                         w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
                if ((rc = pgemm(la, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false)))
                    return rc;
            }
            if (hv && !emit) {
                const int t0 = bd[k + 1], t1 = bd[k + 2];
                if ((rc = pdual(la, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, &q.dg1, t0 * B,
                                nullptr, t0 * B, w.colsum_b + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
            }
        }
        if (!solo_sched) {
        bias_chunk = emit ? 32 : 64;
        if (!corun_k && (rc = grads_ready(0, st))) return rc;      // (see the bf16 branch below)
        if ((rc = handoff(st, sx, ev++))) return rc;
        }
        if (corun_k && !solo_sched) {      // the rest of dW_o's rows: first thing of lane B's tail, then the out_linear gradients are final
            const bool two = nb >= 2;
            const int r0 = two ? 2 * corun_k : 0;
            if ((rc = pgemm_tt(lb, V, H, R - r0, q.dlog, r0, q.h2decB, r0, g->out_w, H, ID, nullptr, two))) return rc;
            if ((rc = grads_ready(0, sx))) return rc;
        }
    } else if (pbf_bwd) {
        // Persistent schedule, ONE stream (mirror of the forward): the launch of stage k runs the word_rnn BPTT of block k
        // next to the vid_rnn BPTT of block k+1 (lstm_persist.hip); between two launches the dG planes / partial column sums
        // of the blocks just finished and the dh1 GEMM of the word block run alone on the chip.
        if ((rc = handoff(sx, st, ev++))) return rc;               // out_linear gradients of lane B first: no GEMM beside
        const int nb = (int)bd.size() - 1;                         // a persistent launch
        for (int k = nb - 1; k >= -1; --k) {
            const bool hw = k >= 0, hv = k + 1 <= nb - 1;
            SeqBwdBf16Args aw, av;
            if (hw) aw = seq_bwd_bf16_args(T, bd[k], bd[k + 1], B, H, q.whh2T, q.dg2, w.dh2dec, L, w.c2, w.s2, w.dc2, w.psync_a, w.err + 1);
            if (hv) av = seq_bwd_bf16_args(T, bd[k + 1], bd[k + 2], B, H, q.whh1T, q.dg1, w.dh1, 0, w.c1, w.s1, w.dc1, w.psync_b, w.err + 1);
            {
                ProfScope ps(st, K_STEP_BWD, (hw ? bd[k + 1] - bd[k] : 0) + (hv ? bd[k + 2] - bd[k + 1] : 0));
                if (hw && hv) rc = lstm_seq_bwd_bf16_persist2(st, aw, &av);
                else rc = lstm_seq_bwd_bf16_persist2(st, hw ? aw : av, nullptr);
                ++g_bwd_persist_launches;
                if (rc) return rc;
            }
            if (hw) {
                const int t0 = bd[k], t1 = bd[k + 1];
                if ((rc = pdual(la, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, nullptr, t0 * B, nullptr, t0 * B,
                                w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
                if ((rc = pgemm(la, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false)))
                    return rc;
            }
            if (hv) {
                const int t0 = bd[k + 1], t1 = bd[k + 2];
                if ((rc = pdual(la, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, nullptr, t0 * B, nullptr, t0 * B,
                                w.colsum_b + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
            }
        }
        // Data-parallel overlap: a persistent launch needs every one of its workgroups resident, so no foreign kernel (the
        // RCCL all-reduce of gradient group 0 on the caller's communication stream) may start beside one and hold LDS /
        // wave slots on a compute unit.  "Group 0 is final" is therefore re-recorded HERE, behind the last persistent
        // launch: s2vt_backward_wait_grads(0) then releases the out_linear all-reduce when the recurrence has left the
        // chip, and it overlaps the weight-gradient GEMMs below instead of the BPTT.
        if ((rc = grads_ready(0, st))) return rc;
        if ((rc = handoff(st, sx, ev++))) return rc;               // lane B's parameter-gradient GEMMs need dG1
    } else {
    for (size_t k = bd.size() - 1; k >= 1; --k) {
        const int t0 = bd[k - 1], t1 = bd[k];
        if (bf) {
            if ((rc = seq_bwd_bf16(st, T, t0, t1, B, H, q.whh2T, w.dh2dec, L, w.c2, w.s2, q.dg2, w.dc2))) return rc;
        } else {
            if ((rc = seq_bwd(st, T, t0, t1, B, H, w.wt2, w.dh2dec, L, w.c2, w.s2, w.dc2))) return rc;
        }
        // dG2 of this block: row planes (dh1, d-embedding GEMMs), transposed planes (weight gradients) and the
        // bias-gradient partial sums, all from one read
        if ((rc = pdual(la, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, bf ? nullptr : &q.dg2, t0 * B, nullptr, t0 * B,
                        w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
            return rc;
        if ((rc = pgemm(la, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false)))
            return rc;
        if ((rc = handoff(st, sx, ev++))) return rc;
        if (bf) {
            if ((rc = seq_bwd_bf16(sx, T, t0, t1, B, H, q.whh1T, w.dh1, 0, w.c1, w.s1, q.dg1, w.dc1))) return rc;
        } else {
            if ((rc = seq_bwd(sx, T, t0, t1, B, H, w.wt1, w.dh1, 0, w.c1, w.s1, w.dc1))) return rc;
        }
        if ((rc = pdual(lb, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, bf ? nullptr : &q.dg1, t0 * B,
                        nullptr, t0 * B, w.colsum_b + (int64_t)(t0 * B / 64) * 4 * H)))
            return rc;
    }
    }
    // lane A: word_rnn parameter gradients + embedding gradient
    // dW = dG^T . (h | emb): row planes of both, read transposed
    if (!word_gemms_done) {
    if ((rc = pgemm_tt(la, 4 * H, H, (T - 1) * B, q.dg2, B, q.h2r, 0, g->word_w_hh, H, ID, nullptr, false))) return rc;
    if ((rc = pgemm_tt(la, 4 * H, H, T * B, q.dg2, 0, q.h1, 0, g->word_w_ih + E, E + H, ID, nullptr, false))) return rc;
    if ((rc = pgemm_tt(la, 4 * H, E, R, q.dg2, L * B, q.emb, 0, g->word_w_ih, E + H, ID, nullptr, false))) return rc;
    }
    if ((rc = colsum_finish(st, w.colsum_a, T * B / bias_chunk, 4 * H, g->word_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->word_b_hh, g->word_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, st));
    if (demb_done) {
        hipEvent_t e;
        if ((rc = get_event(demb_ev, &e))) return rc;
        S2VT_HIP(hipStreamWaitEvent(st, e, 0));
    } else if ((rc = pgemm(la, R, E, 4 * H, q.dg2, L * B, 0, q.weT, 0, 0, w.de, E, ID, nullptr, false))) return rc;
    if ((rc = embedding_grad(st, w.de, R, E, w.tok, V, g->emb_w, w.embws))) return rc;
    if (word_gemms_done) {
        hipEvent_t e;
        if ((rc = get_event(word_grads_ev, &e))) return rc;
        S2VT_HIP(hipStreamWaitEvent(st, e, 0));
    }
    if ((rc = grads_ready(1, st))) return rc;
    // lane B: vid_rnn and feat_linear parameter gradients
    const Lane lt = lb;
    // (one-layer BPTT schedule: the timesteps >= hh1_t0 of this sum ran beside vid_rnn's later stages - the rest is accumulated)
    if ((rc = pgemm_tt(lt, 4 * H, H, ((hh1_t0 > 0 ? hh1_t0 : T) - 1) * B, q.dg1, B, q.h1, 0, g->vid_w_hh, H, ID, nullptr, hh1_t0 > 0))) return rc;
    if ((rc = pgemm_tt(lt, 4 * H, H, L * B, q.dg1, 0, q.x1, 0, g->vid_w_ih, H, ID, nullptr, false))) return rc;
    if ((rc = colsum_finish(lt.s, w.colsum_b, T * B / bias_chunk, 4 * H, g->vid_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->vid_b_hh, g->vid_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, lt.s));
    // dx1 comes out in BATCH-major row order (the order of feats' rows, whose row planes the forward wrote): dW_f = dx1^T feats
    // reads both transposed - no time-major transposed copy of the features, no transposed dx1 (q.x1 is free: dW_ih1 is done)
    if ((rc = pgemm(lt, L * B, H, 4 * H, q.dg1, 0, 0, q.wih1T, 0, 0, w.dx1, H, perm(B, L), nullptr, false))) return rc;
    if ((rc = pdual(lt, w.dx1, H, ID, L * B, H, &q.x1, 0, nullptr, 0, w.colsum_b))) return rc;
    if ((rc = pgemm_tt(lt, H, F, L * B, q.x1, 0, q.feats, 0, g->feat_w, F, ID, nullptr, false))) return rc;
    if ((rc = colsum_finish(lt.s, w.colsum_b, L * B / 64, H, g->feat_b, false))) return rc;
    if (dfeats) {   // rarely requested (nothing reads it in the reference): fp32-MFMA GEMM
        if ((rc = lgemm(lt, true, false, L * B, F, H, w.dx1, H, ID, p->feat_w, F, ID, dfeats, F, ID, nullptr, false)))
            return rc;
    }
    return handoff(sx, st, ev++);
}

}  // namespace s2vt

using namespace s2vt;

extern "C" {

// ------------------------------------------------------------------ batches that are not multiples of 64
// The plane drivers (split-precision / bf16 GEMMs on blocked row images, the persistent recurrence kernels, transposed-read
// weight-gradient GEMMs whose k index is time * B + b) need B % 64 == 0.  Any other batch - the reference's own defaults are
// batch_size = 16 (train.py:27) and 10 (eval.py:27) - is PADDED to the next multiple of 64 inside the workspace instead of
// being sent to the launch-per-timestep fp32-MFMA driver: the pad samples see zero features and token 0, their logits are
// never handed out, and their dlogits rows are zero, so every gradient they contribute is an exact zero (dG = 0 for a row
// whose dh and dc are 0) - the sums the real rows form are unchanged up to the order of fp32 additions.  Staging copies:
// features, targets, logits / dlogits, the dropout mask (a few tens of MB at these batch sizes).
static inline bool batch_padded(const s2vt_dims& d) { return batch_pads(d.B); }
static inline s2vt_dims padded_dims(const s2vt_dims& d) { s2vt_dims q = d; q.B = (d.B + 63) / 64 * 64; return q; }
struct PadWS { float* feats; int64_t* targets; float* logits; float* mask; float* dfeats; size_t bytes; };
static PadWS carve_pad(const s2vt_dims& d, const s2vt_dims& dp, void* base) {
    const size_t Bp = dp.B, L = d.L, F = d.F, H = d.H, V = d.V;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    PadWS w;
    w.feats = c.take<float>(Bp * L * F);
    w.targets = c.take<int64_t>(Bp * (L - 1));
    w.logits = c.take<float>(Bp * (L - 1) * V);          // logits of the forward, dlogits of the backward
    w.mask = c.take<float>((L - 1) * Bp * H);            // out_drop mask, time-major (dropout entry points only)
    w.dfeats = c.take<float>(Bp * L * F);
    w.bytes = align_up(c.off, 256);
    return w;
}
static size_t train_core_bytes(const s2vt_dims& d);

int32_t s2vt_padded_batch(int32_t B) {
    if (B <= 0) return 0;
    s2vt_dims d = {B, 2, 1, 1, 1, 1};
    return batch_padded(d) ? padded_dims(d).B : B;
}

size_t s2vt_train_workspace_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    if (batch_padded(*d)) {
        const s2vt_dims dp = padded_dims(*d);
        return align_up(train_core_bytes(dp), 256) + carve_pad(*d, dp, nullptr).bytes;
    }
    return train_core_bytes(*d);
}
static size_t train_core_bytes(const s2vt_dims& dd) {
    const s2vt_dims* d = &dd;
    size_t n = carve_train(*d, nullptr).bytes;
    if (planes_ok(*d)) {
        const int keep = XP;                       // a size query must not change the state of a running path
        XP = (gemm_mode() == 1) ? 1 : 3;
        n += carve_planes(*d, nullptr).bytes;
        XP = keep;
    }
    return n;
}

static int train_forward_core(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                              int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream,
                              const float* out_mask);
static int train_forward_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                              int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream,
                              const float* out_mask) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && targets && logits && workspace, "s2vt_train_forward: null/invalid argument");
    if (!batch_padded(*d)) return train_forward_core(d, p, feats, targets, targets_ld, logits, workspace, workspace_bytes, stream, out_mask);
    // padded batch: stage [features | zeros], [targets | token 0], run the plane drivers at the padded size, hand out the real rows
    const s2vt_dims dp = padded_dims(*d);
    const size_t core = align_up(train_core_bytes(dp), 256);
    const PadWS s = carve_pad(*d, dp, reinterpret_cast<char*>(workspace) + core);
    S2VT_REQUIRE(workspace_bytes >= core + s.bytes, "s2vt_train_forward: workspace %zu < %zu bytes", workspace_bytes, core + s.bytes);
    hipStream_t st = (hipStream_t)stream;
    const size_t B = d->B, Bp = dp.B, L = d->L, F = d->F, H = d->H, V = d->V;
    int rc;
    S2VT_HIP(hipMemcpyAsync(s.feats, feats, B * L * F * sizeof(float), hipMemcpyDeviceToDevice, st));
    if ((rc = fill_zero(st, s.feats + B * L * F, (Bp - B) * L * F * sizeof(float)))) return rc;
    if ((rc = fill_zero(st, s.targets, Bp * (L - 1) * sizeof(int64_t)))) return rc;
    S2VT_HIP(hipMemcpy2DAsync(s.targets, (L - 1) * sizeof(int64_t), targets, (size_t)targets_ld * sizeof(int64_t), (L - 1) * sizeof(int64_t), B,
                              hipMemcpyDeviceToDevice, st));
    if (out_mask) {      // time-major rows t * B + b -> t * Bp + b; the pad rows' mask is zero
        if ((rc = fill_zero(st, s.mask, (L - 1) * Bp * H * sizeof(float)))) return rc;
        S2VT_HIP(hipMemcpy2DAsync(s.mask, Bp * H * sizeof(float), out_mask, B * H * sizeof(float), B * H * sizeof(float), L - 1,
                                  hipMemcpyDeviceToDevice, st));
    }
    if ((rc = train_forward_core(&dp, p, s.feats, s.targets, (int64_t)(L - 1), s.logits, workspace, core, stream, out_mask ? s.mask : nullptr)))
        return rc;
    S2VT_HIP(hipMemcpyAsync(logits, s.logits, B * (L - 1) * V * sizeof(float), hipMemcpyDeviceToDevice, st));      // batch-major: the first B rows
    return 0;
}
static int train_forward_core(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                              int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream,
                              const float* out_mask) {
    const TrainWS w = carve_train(*d, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_train_forward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    {   // a device-side error of the previous forward, if its flags have arrived
        int rc0 = poll_async_error(false);
        if (rc0) return rc0;
    }
    record_forward(workspace, *d, planes_ok(*d));
    if (planes_ok(*d)) {
        XP = (gemm_mode() == 1) ? 1 : 3;
        const PlaneWS q = carve_planes(*d, reinterpret_cast<char*>(workspace) + w.bytes);
        S2VT_REQUIRE(workspace_bytes >= w.bytes + q.bytes, "s2vt_train_forward: workspace %zu < %zu bytes",
                     workspace_bytes, w.bytes + q.bytes);
        std::vector<uint64_t> key;
        if (graph_on()) {
            key.reserve(32);
            key.push_back(0xF0);
            for (int v : {d->B, d->L, d->F, d->H, d->E, d->V, gemm_mode(), persist_bits(), pipe_block(), option(O_CU_RESERVE)}) key.push_back((uint64_t)v);
            const float* const* pp = reinterpret_cast<const float* const*>(p);
            for (size_t i = 0; i < sizeof(s2vt_params) / sizeof(void*); ++i) key_ptr(key, pp[i]);
            key_ptr(key, feats); key_ptr(key, targets); key.push_back((uint64_t)targets_ld); key_ptr(key, logits);
            key_ptr(key, workspace); key_ptr(key, out_mask); key_ptr(key, st);
        }
        int rc0 = run_graphed(st, key, [&](hipStream_t s_) { return train_forward_x3(d, p, feats, targets, targets_ld, logits, w, q, s_, out_mask); });
        return rc0 ? rc0 : post_async_error(st, w.err);
    }
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    const int blk = pipe_block();
    hipStream_t sx = st;
    int rc;
    if (blk > 0 && (rc = side_stream(st, &sx))) return rc;
    const Lane la{st, w.gws_a, w.gws_floats, w.colsum_a};     // vid_rnn lane (caller's stream)
    const Lane lb{sx, w.gws_b, w.gws_floats, w.colsum_b};     // word_rnn lane
    size_t ev = 0;
    if ((rc = add_vectors(st, p->vid_b_ih, p->vid_b_hh, w.bsum1, 4 * H))) return rc;
    if ((rc = add_vectors(st, p->word_b_ih, p->word_b_hh, w.bsum2, 4 * H))) return rc;
    if ((rc = fill_zero(st, w.err, 4 * sizeof(int)))) return rc;
    if ((rc = targets_to_time_major(st, targets, B, L - 1, targets_ld, V, w.tok, w.err))) return rc;
    if ((rc = handoff(st, sx, ev++))) return rc;
    // lane B, independent of vid_rnn: embedded-word half of the word_rnn gate input (+ both biases) for the
    // L-1 caption steps                                                             S2VTModel.py:71-75
    if ((rc = lgemm(lb, true, true, (L - 1) * B, 4 * H, E, p->emb_w, E, gather(w.tok), p->word_w_ih, E + H, ID,
                    w.s2 + (int64_t)L * B4H, 4 * H, ID, w.bsum2, false)))
        return rc;
    // lane A: x1 (time-major) = feats·W_f^T + b_f ; gx1 = x1·W_ih1^T + biases       S2VTModel.py:54, 64-67
    if ((rc = lgemm(la, true, true, B * L, H, F, feats, F, ID, p->feat_w, F, ID, w.x1, H, perm(L, B), p->feat_b, false)))
        return rc;
    if ((rc = lgemm(la, true, true, L * B, 4 * H, H, w.x1, H, ID, p->vid_w_ih, H, ID, w.s1, 4 * H, ID, w.bsum1, false)))
        return rc;
    const std::vector<int> bd = pipe_bounds(T, L, blk);
    for (size_t k = 0; k + 1 < bd.size(); ++k) {
        const int t0 = bd[k], t1 = bd[k + 1];
        if ((rc = seq_fwd(st, t0, t1, B, H, w.s1, L, w.bsum1, p->vid_w_hh, w.h1, w.c1, true))) return rc;
        if ((rc = handoff(st, sx, ev++))) return rc;
        // vid_out half of the word_rnn gate input for this block: rows < L get the biases here, rows >= L
        // accumulate onto the embedded-word half                                    S2VTModel.py:75-77
        const bool cap = t0 >= L;
        if ((rc = lgemm(lb, true, true, (t1 - t0) * B, 4 * H, H, w.h1 + t0 * BH, H, ID, p->word_w_ih + E, E + H, ID,
                        w.s2 + t0 * B4H, 4 * H, ID, cap ? nullptr : w.bsum2, cap)))
            return rc;
        if ((rc = seq_fwd(sx, t0, t1, B, H, w.s2, T, w.bsum2, p->word_w_hh, w.h2, w.c2, true))) return rc;
    }
    // logits[b, j, :] = (out_drop mask (.)) h2[L + j]·W_o^T + b_o                    S2VTModel.py:78-80
    const float* hdec = w.h2 + L * BH;
    if (out_mask) {
        if ((rc = mul_vectors(sx, hdec, out_mask, w.dh2dec, (int64_t)(L - 1) * B * H))) return rc;     // dh2dec: free in the forward
        hdec = w.dh2dec;
    }
    if ((rc = lgemm(lb, true, true, (L - 1) * B, V, H, hdec, H, ID, p->out_w, H, ID, logits, V, perm(B, L - 1),
                    p->out_b, false)))
        return rc;
    if ((rc = handoff(sx, st, ev++))) return rc;
    return post_async_error(st, w.err);
}

int s2vt_train_forward(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                       int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    return train_forward_impl(d, p, feats, targets, targets_ld, logits, workspace, workspace_bytes, stream, nullptr);
}
int s2vt_train_forward_dropout(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                               int64_t targets_ld, const float* out_mask, float* logits, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return train_forward_impl(d, p, feats, targets, targets_ld, logits, workspace, workspace_bytes, stream, out_mask);
}

static int train_backward_core(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                               const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream,
                               const float* out_mask);
static int train_backward_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                               const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream,
                               const float* out_mask) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && g && workspace, "s2vt_train_backward: null/invalid argument");
    if (!batch_padded(*d)) return train_backward_core(d, p, feats, dlogits, g, dfeats, workspace, workspace_bytes, stream, out_mask);
    // padded batch (see batch_padded): dlogits rows of the pad samples are zero, the staged features / mask are the forward's
    S2VT_REQUIRE(dlogits, "s2vt_train_backward: dlogits is null (the fused criterion backward needs a batch that is a multiple of 64)");
    const s2vt_dims dp = padded_dims(*d);
    const size_t core = align_up(train_core_bytes(dp), 256);
    const PadWS s = carve_pad(*d, dp, reinterpret_cast<char*>(workspace) + core);
    S2VT_REQUIRE(workspace_bytes >= core + s.bytes, "s2vt_train_backward: workspace %zu < %zu bytes", workspace_bytes, core + s.bytes);
    hipStream_t st = (hipStream_t)stream;
    const size_t B = d->B, Bp = dp.B, L = d->L, F = d->F, V = d->V;
    int rc;
    S2VT_HIP(hipMemcpyAsync(s.logits, dlogits, B * (L - 1) * V * sizeof(float), hipMemcpyDeviceToDevice, st));
    if ((rc = fill_zero(st, s.logits + B * (L - 1) * V, (Bp - B) * (L - 1) * V * sizeof(float)))) return rc;
    if ((rc = train_backward_core(&dp, p, s.feats, s.logits, g, dfeats ? s.dfeats : nullptr, workspace, core, stream, out_mask ? s.mask : nullptr)))
        return rc;
    if (dfeats) S2VT_HIP(hipMemcpyAsync(dfeats, s.dfeats, B * L * F * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}
static int train_backward_core(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                               const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream,
                               const float* out_mask) {
    g_bwd_persist_launches = 0;          // (s2vt_backward_order describes THIS backward, whichever driver it takes)
    g_bwd_group0_after = 0;
    struct ReserveWindow {               // option cu_reserve: off until gradient group 0 is released, off again when the backward is enqueued
        ReserveWindow() { cu_reserve_window(false); }
        ~ReserveWindow() { cu_reserve_window(false); }
    } reserve_window;
    const TrainWS w = carve_train(*d, workspace);
    S2VT_REQUIRE(workspace_bytes >= w.bytes, "s2vt_train_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    bool dlog_ready = false;
    {
        int rc0 = poll_async_error(false);          // flags of the forward, if they have arrived already
        if (rc0) return rc0;
        if ((rc0 = check_forward_record(workspace, *d, planes_ok(*d), &dlog_ready))) return rc0;
    }
    S2VT_REQUIRE(dlogits || dlog_ready, "s2vt_train_backward: dlogits is null and s2vt_mean_ce_backward_fused has not run on this workspace");
    if (planes_ok(*d)) {
        XP = (gemm_mode() == 1) ? 1 : 3;
        const PlaneWS q = carve_planes(*d, reinterpret_cast<char*>(workspace) + w.bytes);
        S2VT_REQUIRE(workspace_bytes >= w.bytes + q.bytes, "s2vt_train_backward: workspace %zu < %zu bytes",
                     workspace_bytes, w.bytes + q.bytes);
        std::vector<uint64_t> key;
        if (graph_on()) {
            key.reserve(48);
            key.push_back(0xB0 + (dlog_ready ? 1 : 0));
            for (int v : {d->B, d->L, d->F, d->H, d->E, d->V, gemm_mode(), persist_bits(), pipe_block(), option(O_CU_RESERVE)}) key.push_back((uint64_t)v);
            const float* const* pp = reinterpret_cast<const float* const*>(p);
            for (size_t i = 0; i < sizeof(s2vt_params) / sizeof(void*); ++i) key_ptr(key, pp[i]);
            float* const* gp = reinterpret_cast<float* const*>(g);
            for (size_t i = 0; i < sizeof(s2vt_grads) / sizeof(void*); ++i) key_ptr(key, gp[i]);
            key_ptr(key, feats); key_ptr(key, dlogits); key_ptr(key, dfeats); key_ptr(key, workspace); key_ptr(key, out_mask);
            key_ptr(key, st);
        }
        bool graphed = false;
        int rc0 = run_graphed(st, key, [&](hipStream_t s_) { return train_backward_x3(d, p, feats, dlogits, g, dfeats, w, q, s_, out_mask, dlog_ready); },
                              &graphed);
        if (!rc0 && graphed) {       // (see grads_ready) every gradient group is final behind the graph
            if ((rc0 = grads_ready(0, st))) return rc0;
            if ((rc0 = grads_ready(1, st))) return rc0;
        }
        return rc0 ? rc0 : post_async_error(st, w.err, 1);
    }
    const int B = d->B, L = d->L, F = d->F, H = d->H, E = d->E, V = d->V, T = 2 * L - 1;
    const int64_t BH = (int64_t)B * H, B4H = 4 * BH;
    const int R = (L - 1) * B;
    const int blk = pipe_block();
    hipStream_t sx = st;
    int rc;
    if (blk > 0 && (rc = side_stream(st, &sx))) return rc;
    const Lane la{st, w.gws_a, w.gws_floats, w.colsum_a};     // word_rnn lane (caller's stream)
    const Lane lb{sx, w.gws_b, w.gws_floats, w.colsum_b};     // vid_rnn lane
    size_t ev = 0;
    if ((rc = handoff(st, sx, ev++))) return rc;
    // lane A: gradient into the decode-step hidden states, then word_rnn BPTT       (autograd of S2VTModel.py:80, :77)
    if ((rc = lgemm(la, true, false, R, H, V, dlogits, V, ID, p->out_w, H, ID, w.dh2dec, H, perm(L - 1, B), nullptr, false)))
        return rc;
    if (out_mask && (rc = mul_vectors(st, w.dh2dec, out_mask, w.dh2dec, (int64_t)R * H))) return rc;      // autograd of out_drop
    if ((rc = transpose_f32(st, p->word_w_hh, 4 * H, H, w.wt2))) return rc;
    // lane B meanwhile: out_linear weight/bias gradients (need only dlogits and the (masked) h2) and W_hh1^T
    const float* hdec = w.h2 + L * BH;
    if (out_mask) {
        if ((rc = mul_vectors(sx, hdec, out_mask, w.dx1, (int64_t)R * H))) return rc;      // dx1: free until the vid_rnn input gradient
        hdec = w.dx1;
    }
    if ((rc = lgemm(lb, false, false, V, H, R, dlogits, V, ID, hdec, H, perm(L - 1, B), g->out_w, H, ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(sx, dlogits, R, V, V, lb.colsum, g->out_b, false))) return rc;
    if ((rc = grads_ready(0, sx))) return rc;
    if ((rc = transpose_f32(sx, p->vid_w_hh, 4 * H, H, w.wt1))) return rc;
    const std::vector<int> bd = pipe_bounds(T, L, blk);
    for (size_t k = bd.size() - 1; k >= 1; --k) {
        const int t0 = bd[k - 1], t1 = bd[k];
        if ((rc = seq_bwd(st, T, t0, t1, B, H, w.wt2, w.dh2dec, L, w.c2, w.s2, w.dc2))) return rc;
        // gradient into vid_out for this block: dh1 = dG2·W_v                        (autograd of :75)
        if ((rc = lgemm(la, true, false, (t1 - t0) * B, H, 4 * H, w.s2 + t0 * B4H, 4 * H, ID, p->word_w_ih + E, E + H, ID,
                        w.dh1 + t0 * BH, H, ID, nullptr, false)))
            return rc;
        if ((rc = handoff(st, sx, ev++))) return rc;
        if ((rc = seq_bwd(sx, T, t0, t1, B, H, w.wt1, w.dh1, 0, w.c1, w.s1, w.dc1))) return rc;   // (autograd of :67)
    }
    // lane A: word_rnn parameter gradients + embedding gradient (run while lane B finishes the vid_rnn BPTT)
    if ((rc = lgemm(la, false, false, 4 * H, H, (T - 1) * B, w.s2 + B4H, 4 * H, ID, w.h2, H, ID, g->word_w_hh, H, ID,
                    nullptr, false)))
        return rc;
    if ((rc = lgemm(la, false, false, 4 * H, H, T * B, w.s2, 4 * H, ID, w.h1, H, ID, g->word_w_ih + E, E + H, ID, nullptr,
                    false)))
        return rc;
    // dW_ih2[:, :E] = dG2[L..]^T · Emb[tok]: the embedded rows are gathered once (time-major) into w.de, which
    // is free until the d(embedded words) GEMM below overwrites it
    if ((rc = gather_rows_f32(st, p->emb_w, E, w.tok, R, E, w.de))) return rc;
    if ((rc = lgemm(la, false, false, 4 * H, E, R, w.s2 + (int64_t)L * B4H, 4 * H, ID, w.de, E, ID, g->word_w_ih, E + H,
                    ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(st, w.s2, (int64_t)T * B, 4 * H, 4 * H, la.colsum, g->word_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->word_b_hh, g->word_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, st));
    if ((rc = lgemm(la, true, false, R, E, 4 * H, w.s2 + (int64_t)L * B4H, 4 * H, ID, p->word_w_ih, E + H, ID, w.de, E, ID,
                    nullptr, false)))
        return rc;
    if ((rc = embedding_grad(st, w.de, R, E, w.tok, V, g->emb_w, w.embws))) return rc;
    if ((rc = grads_ready(1, st))) return rc;
    // lane B: vid_rnn and feat_linear parameter gradients                           (autograd of :67, :54)
    if ((rc = lgemm(lb, false, false, 4 * H, H, (T - 1) * B, w.s1 + B4H, 4 * H, ID, w.h1, H, ID, g->vid_w_hh, H, ID,
                    nullptr, false)))
        return rc;
    if ((rc = lgemm(lb, false, false, 4 * H, H, L * B, w.s1, 4 * H, ID, w.x1, H, ID, g->vid_w_ih, H, ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(sx, w.s1, (int64_t)T * B, 4 * H, 4 * H, lb.colsum, g->vid_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->vid_b_hh, g->vid_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, sx));
    if ((rc = lgemm(lb, true, false, L * B, H, 4 * H, w.s1, 4 * H, ID, p->vid_w_ih, H, ID, w.dx1, H, ID, nullptr, false)))
        return rc;
    if ((rc = lgemm(lb, false, false, H, F, L * B, w.dx1, H, ID, feats, F, perm(B, L), g->feat_w, F, ID, nullptr, false)))
        return rc;
    if ((rc = colsum_f32(sx, w.dx1, (int64_t)L * B, H, H, lb.colsum, g->feat_b, false))) return rc;
    if (dfeats) {
        if ((rc = lgemm(lb, true, false, L * B, F, H, w.dx1, H, ID, p->feat_w, F, ID, dfeats, F, perm(B, L), nullptr, false)))
            return rc;
    }
    return handoff(sx, st, ev++);
}

int s2vt_train_backward(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                        const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream) {
    return train_backward_impl(d, p, feats, dlogits, g, dfeats, workspace, workspace_bytes, stream, nullptr);
}
int s2vt_train_backward_dropout(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                                const float* out_mask, const s2vt_grads* g, float* dfeats, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return train_backward_impl(d, p, feats, dlogits, g, dfeats, workspace, workspace_bytes, stream, out_mask);
}

int s2vt_backward_order(int32_t* persistent_bptt_launches, int32_t* group0_recorded_after) {
    if (persistent_bptt_launches) *persistent_bptt_launches = g_bwd_persist_launches;
    if (group0_recorded_after) *group0_recorded_after = g_bwd_group0_after;
    return 0;
}

int s2vt_backward_wait_grads(int32_t group, void* stream) {
    S2VT_REQUIRE(group == 0 || group == 1, "s2vt_backward_wait_grads: group must be 0 (out_linear) or 1 (word_rnn + embedding)");
    S2VT_REQUIRE(g_grad_ev_set[group], "s2vt_backward_wait_grads: no s2vt_train_backward has run yet");
    S2VT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_grad_ev[group], 0));
    return 0;
}


// MaskCriterion's backward FUSED into the hand-over to s2vt_train_backward (utils.py:22 under train.py:124): the mean-CE gradient
// is evaluated from the logits inside the plane-split pass of the train workspace - row planes, transposed planes and the
// out_linear bias-gradient partial sums of dlogits, exactly what the backward's first kernel would write from an fp32
// dlogits tensor - so that tensor is never materialised.  Only for workspaces of the plane drivers (B % 64 == 0, gemm mode 1 / 3).
int s2vt_mean_ce_backward_fused(const s2vt_dims* d, const float* logits, const int64_t* target, int64_t target_ld, const float* lse,
                                const float* gout, void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(dims_ok(d) && logits && target && lse && gout && workspace, "s2vt_mean_ce_backward_fused: null/invalid argument");
    S2VT_REQUIRE(planes_ok(*d), "s2vt_mean_ce_backward_fused: the workspace is not a plane-driver workspace (B %% 64, gemm mode 1 or 3)");
    const TrainWS w = carve_train(*d, workspace);
    {
        std::lock_guard<std::mutex> lock(g_fwd_mutex);
        auto it = g_fwd_records.find(workspace);
        S2VT_REQUIRE(it != g_fwd_records.end() && memcmp(&it->second.d, d, sizeof(*d)) == 0 && it->second.gemm_mode == gemm_mode() &&
                         !it->second.dlog_ready,
                     "s2vt_mean_ce_backward_fused: no matching s2vt_train_forward has run on this workspace");
    }
    XP = (gemm_mode() == 1) ? 1 : 3;
    const PlaneWS q = carve_planes(*d, reinterpret_cast<char*>(workspace) + w.bytes);
    S2VT_REQUIRE(workspace_bytes >= w.bytes + q.bytes, "s2vt_mean_ce_backward_fused: workspace %zu < %zu bytes", workspace_bytes, w.bytes + q.bytes);
    const int R = (d->L - 1) * d->B, V = d->V;
    CeGradArgs ce;
    ce.lse = lse; ce.target = target; ce.gout = gout; ce.Lm1 = d->L - 1; ce.ldt = target_ld;
    ce.alpha_out = (XP == 1) ? w.ce_alpha : nullptr;       // bf16 operands: power-of-two scale in the planes, mantissa downstream
    hipStream_t st = (hipStream_t)stream;
    int rc;
    {
        ProfScope ps(st, K_CE, 1);
        if ((rc = split_planes_dual(st, XP, logits, V, ID, R, V, q.dlog.p, q.dlog.ld, q.dlog.kpad, nullptr, 0, 0, w.colsum_c, &ce)))
            return rc;
    }
    std::lock_guard<std::mutex> lock(g_fwd_mutex);
    g_fwd_records[workspace].dlog_ready = true;
    return 0;
}


}
