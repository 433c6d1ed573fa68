// Host-side internals shared by the C-ABI units of libs2vt_hip.so:
//   api_runtime.hip  errors, asynchronous device-side errors, live timing, launch-sequence capture, the side stream, setters
//   api_shared.hip   plane-operand helpers (split / GEMM wrappers), recurrence loops and argument builders
//   api_train.hip    s2vt_train_forward / s2vt_train_backward (+ dropout, fused criterion backward, gradient-group events)
//   api_decode.hip   greedy decode, the encode phase for the beam search, the decode step's argmax entry points
//   api_beam.hip     the batched beam-search depth step
//   api_ops.hip      per-op entry points (GEMM, split, timestep / sequence kernels, criterion)
// Host code only; every kernel lives in gemm* / lstm* / ce / misc / split / argmax_x3 / beam_queue.hip.
#pragma once
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/s2vt_hip.h"
#include "common.h"
#include "kernels.h"

namespace s2vt {

// ------------------------------------------------------------------ api_runtime.hip
#ifdef S2VT_EXPERIMENT_STAMPS
extern unsigned long long* g_xstamps;     // timing experiments only (experiment.h)
extern int g_xstamp_block;
#endif
const char* last_error_text();

// asynchronous device-side errors (kind: 0 forward / decode, 1 backward, 2 loss, 3 a ring of eight for the beam depth steps)
int poll_async_error(bool wait);
int post_async_error(hipStream_t st, const int* dev_flags, int kind = 0);
int device_flags(int** out);

// live kernel timing: launch sites bracket kernels of one kind with hipEvents on the launch stream while s2vt_prof_enable(1)
enum { K_GEMM = 0, K_STEP_FWD = 1, K_STEP_BWD = 2, K_CE = 3, K_ARGMAX = 4, K_GEMM_CORUN = 5, K_NKINDS = 6 };   // (5: a GEMM planned for PART of the
                                                                                                               //  compute units, beside a one-layer persistent launch)
struct ProfRec { hipEvent_t a, b; int kind; int64_t launches; };
struct ProfScope {
    hipStream_t s; bool on; ProfRec r;
    ProfScope(hipStream_t stream, int kind, int64_t launches);
    ~ProfScope();
};
bool prof_on();

// launch-sequence capture (option "graph"): see api_runtime.hip
bool graph_on();
bool graph_capturing();                   // this thread is inside a capture (nothing may record an external event)
int run_graphed(hipStream_t st, const std::vector<uint64_t>& key, const std::function<int(hipStream_t)>& enqueue, bool* graphed = nullptr);
static inline void key_ptr(std::vector<uint64_t>& k, const void* p) { k.push_back((uint64_t)(uintptr_t)p); }

static const RowMap ID = {nullptr, 0, 0};
static inline RowMap perm(int inner, int outer) { return RowMap{nullptr, inner, outer}; }
static inline RowMap gather(const int32_t* idx) { return RowMap{idx, 0, 0}; }

// split-K scratch of the driver that is running (set by the whole-path entry points from their workspace)
extern thread_local float* g_gws;
extern thread_local size_t g_gws_floats;
struct GemmWsScope {
    GemmWsScope(float* p, size_t n) { g_gws = p; g_gws_floats = n; }
    ~GemmWsScope() { g_gws = nullptr; g_gws_floats = 0; }
};
// fp32-MFMA GEMM with the split-K scratch of the running driver
int gemm(hipStream_t st, bool ak, bool bk, int M, int N, int K, const float* A, int64_t lda, RowMap am, const float* B, int64_t ldb,
         RowMap bm, float* C, int64_t ldc, RowMap cm, const float* bias, bool acc);
size_t gemm_ws_floats(const s2vt_dims& d);      // scratch floats for split-K slabs of a whole-path driver

struct Carver {
    char* base; size_t off; size_t cap;
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += n * sizeof(T);
        return p;
    }
};
// Batches that are not multiples of 64 in a plane mode: from option pad_min_batch rows on (default 33; a greedy decode: three
// quarters of it, 24) the whole-path drivers pad them to the next multiple of 64 inside their workspace; smaller batches run as
// they are on the launch-per-timestep fp32-MFMA path, which is faster there (train step at B = 16, H = 512: 4.4 vs 5.8 ms, at
// B = 48: 6.8 vs 6.1; decode at B = 16, H = 1000: 4.2 vs 4.5 ms, at B = 32: 5.5 vs 4.6; profiles/round5_ragged_batches.txt)
static inline bool batch_pads(int B, bool decode = false) {
    const int from = decode ? option(O_PAD_MIN_BATCH) * 3 / 4 : option(O_PAD_MIN_BATCH);
    return option(O_GEMM_MODE) != 0 && B % 64 != 0 && B >= (from > 0 ? from : 1);
}
static inline bool dims_ok(const s2vt_dims* d) { return d && d->B > 0 && d->L > 1 && d->F > 0 && d->H > 0 && d->E > 0 && d->V > 0; }

// ---- options as the drivers read them (csrc/options.hip)
static inline int pipe_block() { return option(O_PIPE_BLOCK); }      // timesteps per pipeline block; 0 = both layers on the caller's stream
// 0: fp32-input MFMA GEMMs; 3: split precision, 3 bf16 planes (fp32-equivalent); 1: plain bf16 operands (config 3: bf16
// storage, fp32 accumulate) for the batched GEMMs AND the timestep kernels (lstm_bf16.hip)
static inline int gemm_mode() { return option(O_GEMM_MODE); }
// plane modes need every k-offset inside a packed operand to be a multiple of 64 (k = time*B + b): B % 64 == 0
static inline bool planes_ok(const s2vt_dims& d) { return gemm_mode() != 0 && d.B % 64 == 0; }
// Recurrence schedule of the train drivers (option "persist"): 1 (default) = one persistent launch per block of timesteps with
// the W_hh slices resident per CU - lstm_persist.hip in the bf16 configuration (both directions), lstm_persist_x3.hip in the
// fp32-equivalent one (forward: option "persist_x3_fwd", default on; BPTT: option "persist_x3_bwd"); 0 = launches per timestep
// everywhere (lstm.hip / lstm_bf16.hip: a card shared with another process, whose kernels could keep a persistent launch from
// becoming resident).  Round 2's exact-fp32 persistent pair (MFMA-bound, slower end to end: docs/history) is gone.
static inline int persist_mode() { return option(O_PERSIST); }
static inline bool persist_on() { return persist_mode() >= 1; }
static inline bool persist_x3_fwd_on() { return option(O_PERSIST_X3_FWD) != 0 && persist_on(); }
// split-precision persistent BPTT (reduce-scatter over the gate columns, lstm_persist_x3.hip): option persist_x3_bwd = 1 wherever
// the shape is supported, 2 (default) only where a workgroup carries ONE 32-row chain (B = 64 at H = 1000).  There a one-layer
// launch holds half of the compute units and the backward's GEMMs run beside it (options corun / bptt_solo, api_train.hip:
// 9.9-10.2 ms per config-2 step; two layers per launch with nothing beside them 10.7-11.0, the launch-per-timestep BPTT on two
// lanes 11.1); with two or more chains per workgroup (B = 128, 256) no unit is left idle and the two-lane launch-per-timestep
// schedule is 6-8 % faster (20.4 vs 19.2 ms, 39.4 vs 36.5)
static inline bool persist_x3_bwd_on(int B, int H) {
    const int m = option(O_PERSIST_X3_BWD);
    if (m == 0 || !persist_on()) return false;
    return m == 1 || lstm_seq_bwd_x3_persist_supported(B, H) == 1;
}
// the recurrence options that decide how a train workspace is carved and which images a forward leaves in it
static inline int persist_bits() { return persist_mode() | (persist_x3_fwd_on() ? 2 : 0) | (option(O_PERSIST_X3_BWD) << 2); }

// ---- two-lane execution (api_runtime.hip): the two LSTM layers as a software pipeline on two streams
struct Lane {
    hipStream_t s;
    float* gws;
    size_t gws_floats;
    float* colsum;
};
int side_stream(hipStream_t caller, hipStream_t* out);
int side_stream_overlaps();               // 1 verified concurrent, 0 no candidate overlapped, -1 before the first pipelined call
int get_event(size_t i, hipEvent_t* out);
int handoff(hipStream_t from, hipStream_t to, size_t ev_index);       // `to` waits for everything enqueued so far on `from`
int lgemm(const Lane& ln, bool ak, bool bk, int M, int N, int K, const float* A, int64_t lda, RowMap am, const float* B, int64_t ldb,
          RowMap bm, float* C, int64_t ldc, RowMap cm, const float* bias, bool acc);

// ------------------------------------------------------------------ api_shared.hip
// LSTM layer forward over steps [t0, t1) / BPTT over steps t1-1 .. t0 as launches per timestep (time-major buffers)
int seq_fwd(hipStream_t st, int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias, const float* w_hh, float* h_all,
            float* c_all, bool write_stash);
int seq_bwd(hipStream_t st, int T, int t0, int t1, int B, int H, const float* w_hh_t, const float* dh_out, int dh_first,
            const float* c_all, float* stash_dg, float* dc);
int balanced_block(int L, int blk);
std::vector<int> pipe_bounds(int T, int L, int blk);

extern int XP;            // planes per operand of the running plane driver (3 or 1); set by the entry points
static inline int pad64(int x) { return (x + 63) / 64 * 64; }
struct PB { unsigned short* p; int64_t ld; int kpad; };       // packed planes of a k-major operand [rows][k]
// element offset of k index k0 (a multiple of 64) inside an operand: row layout (1 plane) k0; blocked 3-plane layout
// (gemm_x3.hip) k0/16 records of 3072 elements
static inline int64_t koff(int k0) { return XP == 3 ? (int64_t)k0 * 192 : (int64_t)k0 * XP; }
static inline size_t rows64(size_t r) { return (r + 63) / 64 * 64; }
int psplit(const Lane& ln, const PB& dst, int r0, const float* in, int64_t ld, RowMap imap, int rows, int cols);
int pdual(const Lane& ln, const float* in, int64_t ld, RowMap imap, int rows, int cols, const PB* r, int r0, const PB* t, int k0,
          float* colpart);
int pgemm(const Lane& ln, int M, int N, int K, const PB& A, int a0, int ka, const PB& B, int b0, int kb, float* C, int64_t ldc,
          RowMap cm, const float* bias, bool acc);
int pgemm_tt(const Lane& ln, int M, int N, int K, const PB& A, int a_row0, const PB& B, int b_row0, float* C, int64_t ldc, RowMap cm,
             const float* bias, bool acc);

int seq_fwd_bf16(hipStream_t st, int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias, const PB& wb, const PB& hb,
                 float* h_all, float* c_all);
int seq_bwd_bf16(hipStream_t st, int T, int t0, int t1, int B, int H, const PB& wt, const float* dh_out, int dh_first,
                 const float* c_all, float* stash_dg, const PB& dgb, float* dc);
SeqBwdX3Args persist_bwd_x3_args(int T, int t0, int t1, int B, int H, int64_t Kp, int64_t Hp, const unsigned short* wtp,
                                 const float* dh_out, int dh_first, const float* c_all, float* stash_dg, float* dc, float* part,
                                 int64_t part_slot, int nslots, unsigned int* sync, int* err);
SeqFwdX3Args persist_fwd_x3_args(int t0, int t1, int B, int H, int T, int64_t Kp, float* gx_stash, int n_gx, const float* bias,
                                 const unsigned short* wp, unsigned short* hp, float* h_all, float* c_all, unsigned int* sync, int* err);
bool persist_fwd_ok(int B, int H, const PB& wb, const PB& hb);
SeqFwdBf16Args persist_fwd_args(int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias, const PB& wb, const PB& hb,
                                float* h_all, float* c_all, unsigned int* sync, int* err);
SeqBwdBf16Args seq_bwd_bf16_args(int T, int t0, int t1, int B, int H, const PB& wt, const PB& dgb, const float* dh_out, int dh_first,
                                 const float* c_all, float* stash_dg, float* dc, unsigned int* sync, int* err);

// ------------------------------------------------------------------ api_decode.hip (shared with the beam step)
// What a decode derives from the WEIGHTS alone (plane images of W_f, W_ih1, W_v, W_o and the per-token gate-input table):
// carved from the tail of the call's workspace, or from a caller-kept cache that outlives the call (s2vt_greedy_decode_cached)
struct DecodeConst { PB wf, wih1, wv, wo; float* gtab; unsigned short *xw1, *xw2; PB whh; size_t bytes; };   // xw: W_hh planes [3][4H][Kp]; whh: word_rnn's W_hh, blocked
DecodeConst carve_decode_const(const s2vt_dims& d, void* base);

}  // namespace s2vt
