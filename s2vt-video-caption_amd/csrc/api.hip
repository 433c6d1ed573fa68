// C-ABI drivers of libs2vt_hip.so (declared in include/s2vt_hip.h): argument checking, workspace
// carving and the stream-ordered launch sequences of the S2VT train forward/backward and greedy
// decode.  Host code only; every kernel lives in gemm/lstm/ce/misc.hip.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/s2vt_hip.h"
#include "common.h"
#include "kernels.h"

namespace s2vt {

static thread_local char g_err[512] = "";
#ifdef S2VT_EXPERIMENT_STAMPS
static unsigned long long* g_xstamps = nullptr;     // timing experiments only (experiment.h)
static int g_xstamp_block = 0;
#endif

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

// ------------------------------------------------------------------ asynchronous device-side errors
// Kernels cannot return an error code: they raise a flag in the workspace (err[0]: a target id outside [0, V), which
// nn.Embedding / CrossEntropyLoss reject with IndexError in the reference, S2VTModel.py:71; err[1]: a hand-off wait of
// the persistent recurrence timed out).  Every s2vt_train_forward ends with a 16-byte copy of the flags into a pinned
// host word block + an event; the NEXT entry on this process (forward, backward or s2vt_check_async_error) that finds the
// event complete reports the error.  One step late by construction, never silent; callers that synchronise anyway
// (loss.item()) call s2vt_check_async_error(1) right there and get it immediately.
struct ErrRecord { int* host; hipEvent_t ev; bool pending; };
static ErrRecord g_async[3] = {{nullptr, nullptr, false}, {nullptr, nullptr, false}, {nullptr, nullptr, false}};   // forward, backward, loss
static int read_record(ErrRecord& r, bool wait) {
    if (!r.pending) return 0;
    if (wait) {
        S2VT_HIP(hipEventSynchronize(r.ev));
    } else {
        const hipError_t q = hipEventQuery(r.ev);
        if (q == hipErrorNotReady) return 0;
        S2VT_HIP(q);
    }
    r.pending = false;
    const int bad_target = r.host[0], timed_out = r.host[1];
    r.host[0] = r.host[1] = 0;
    if (bad_target) {
        set_error("index out of range: a token id of the previous call (targets of s2vt_train_forward / s2vt_mean_ce_forward, a "
                  "decode or beam step's input token) lies outside [0, vocab_size) (the reference raises IndexError in "
                  "nn.Embedding, S2VTModel.py:71,90,100,211, and in nn.CrossEntropyLoss, utils.py:22)");
        return S2VT_ERR_INDEX;
    }
    if (timed_out) {
        set_error("persistent recurrence kernel: a hand-off wait timed out (its workgroups were not co-resident)");
        return S2VT_ERR_TIMEOUT;
    }
    return 0;
}
static std::mutex g_async_mutex;        // forward (caller's thread) and backward (autograd's thread) both post and poll
static int poll_async_error(bool wait) {
    std::lock_guard<std::mutex> lock(g_async_mutex);
    int first = 0;
    for (int k = 0; k < 3 && !first; ++k) first = read_record(g_async[k], wait);
    if (first) {               // one bad batch flags the forward's AND the loss's record: it is reported once - the records still
        char keep[512];        // pending are awaited and dropped with it (the caller is about to raise; the wait costs nothing then)
        snprintf(keep, sizeof(keep), "%s", s2vt_last_error());
        for (int k = 0; k < 3; ++k) (void)read_record(g_async[k], true);
        set_error("%s", keep);
    }
    return first;
}
static int post_async_error(hipStream_t st, const int* dev_flags, int kind = 0) {
    std::lock_guard<std::mutex> lock(g_async_mutex);
    ErrRecord& r = g_async[kind];
    if (!r.host) {
        S2VT_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.host), 4 * sizeof(int), hipHostMallocDefault));
        r.host[0] = r.host[1] = r.host[2] = r.host[3] = 0;
        S2VT_HIP(hipEventCreateWithFlags(&r.ev, hipEventDisableTiming));
    }
    if (r.pending) {           // an unread record of the same kind: one whole step old, its copy has long completed
        int rc = read_record(r, true);
        if (rc) return rc;
    }
    S2VT_HIP(hipMemcpyAsync(r.host, dev_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    S2VT_HIP(hipEventRecord(r.ev, st));
    r.pending = true;
    return 0;
}

// The four flag words of a per-op entry point that has no workspace of its own (s2vt_mean_ce_forward, s2vt_lstm_step_fwd_token):
// the only device memory the library owns, 16 bytes per device, allocated on first use.
static int device_flags(int** out) {
    static int* flags_of[64] = {};
    int dev = 0;
    S2VT_HIP(hipGetDevice(&dev));
    S2VT_REQUIRE(dev >= 0 && dev < 64, "device index %d", dev);
    if (!flags_of[dev]) S2VT_HIP(hipMalloc(reinterpret_cast<void**>(&flags_of[dev]), 4 * sizeof(int)));
    *out = flags_of[dev];
    return 0;
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

// ------------------------------------------------------------------ launch-sequence capture (hipGraph)
// s2vt_set_graph_mode(1) / S2VT_GRAPH=1: the launch sequence of a whole-path call of the plane drivers (s2vt_train_forward /
// s2vt_train_backward at B % 64 == 0: ~350 launches on two streams each) is captured ONCE per distinct argument set - every
// pointer, the dims, the modes and the stream are the key - and replayed with one hipGraphLaunch afterwards.  The first
// sighting of a key runs eagerly (lazy initialisations: side-stream calibration, occupancy queries, event pool), the second
// is captured, later ones replay.  A training loop presents the same pointers every step once torch's caching allocator
// has settled (parameters, the flat gradient buffer and the batch ring are fixed; workspace and logits come back at the same
// addresses); a key that never repeats simply stays eager.  Not used while live timing is on (the event brackets are not
// capturable).  At most 8 executables are kept (least recently used goes).
static int g_graph_mode = -1;
static thread_local bool g_capturing = false;      // (the enqueue callback runs on the capturing thread)
struct GraphEntry { hipGraphExec_t exec; unsigned long long last_use; int seen; };
static std::map<std::vector<uint64_t>, GraphEntry> g_graphs;
static std::mutex g_graph_mutex;
static unsigned long long g_graph_tick = 0, g_graph_replays = 0, g_graph_captures = 0;
static bool graph_on() {
    if (g_graph_mode < 0) { const char* e = getenv("S2VT_GRAPH"); g_graph_mode = (e && atoi(e) != 0) ? 1 : 0; }
    return g_graph_mode == 1 && !g_prof;
}
template <typename F>
static int run_graphed(hipStream_t st, const std::vector<uint64_t>& key, F&& enqueue, bool* graphed = nullptr) {
    if (graphed) *graphed = false;
    if (!graph_on()) return enqueue(st);
    std::lock_guard<std::mutex> lock(g_graph_mutex);
    // bound the table for keys that never repeat as well (a fresh pointer every step: no capture ever happens and the
    // eviction below would never run): the least recently used entry goes before a ninth is inserted
    auto evict_lru = [&](const GraphEntry* keep) {
        while (g_graphs.size() > 8) {
            auto oldest = g_graphs.end();
            for (auto it = g_graphs.begin(); it != g_graphs.end(); ++it)
                if (&it->second != keep && (oldest == g_graphs.end() || it->second.last_use < oldest->second.last_use)) oldest = it;
            if (oldest == g_graphs.end()) break;
            if (oldest->second.exec) (void)hipGraphExecDestroy(oldest->second.exec);
            g_graphs.erase(oldest);
        }
    };
    GraphEntry& e = g_graphs[key];                  // (a new key: exec = nullptr, seen = 0)
    e.last_use = ++g_graph_tick;
    evict_lru(&e);
    if (e.exec) {
        ++g_graph_replays;
        S2VT_HIP(hipGraphLaunch(e.exec, st));
        if (graphed) *graphed = true;
        return 0;
    }
    if (++e.seen < 2) return enqueue(st);
    if (graphed) *graphed = true;
    // captured on a stream of the library's own: the caller's may be the legacy default stream (torch's current stream
    // unless told otherwise), which cannot be captured; graph nodes carry no stream identity
    static hipStream_t cap = nullptr;
    if (!cap) S2VT_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    S2VT_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
    g_capturing = true;
    const int rc = enqueue(cap);
    g_capturing = false;
    const hipError_t ce = hipStreamEndCapture(cap, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    S2VT_HIP(ce);
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    S2VT_HIP(ie);
    e.exec = exec;
    ++g_graph_captures;
    S2VT_HIP(hipGraphLaunch(exec, st));
    return 0;
}
static void key_ptr(std::vector<uint64_t>& k, const void* p) { k.push_back((uint64_t)(uintptr_t)p); }

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

static bool dims_ok(const s2vt_dims* d) {
    return d && d->B > 0 && d->L > 1 && d->F > 0 && d->H > 0 && d->E > 0 && d->V > 0;
}

static int pipe_block();
static int gemm_mode();
static bool persist_x3_fwd_on();
static bool persist_x3_bwd_on();
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
        const bool can = w.xkp > 0 && B % 32 == 0 && pipe_block() > 0 && persist_x3_bwd_on();    // (opt-in kernel: no rings otherwise)
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
    size_t cs = colsum_partial_floats((int64_t)T * B, (int)(4 * H));
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

// ------------------------------------------------------------------ two-lane execution
// The two LSTM layers are independent except through h1: word_rnn step t needs vid_rnn step t only.  A single
// timestep kernel cannot fill the chip's latency (launch + prologue + epilogue ~4 us of a ~12 us step), so the
// layers run as a software pipeline on TWO streams: while lane A (the caller's stream) runs vid_rnn block k+1,
// lane B runs the batched input GEMM and the word_rnn steps of block k (backward: mirrored).  The step kernels
// are sized (69.6 KB LDS) so that one workgroup of each lane fits a CU.  Events order the hand-offs; nothing is
// allocated per call (stream + events are created once per process).
struct Lane {
    hipStream_t s;
    float* gws;
    size_t gws_floats;
    float* colsum;
};

static hipStream_t g_side = nullptr;
static std::vector<hipEvent_t> g_events;
static int g_pipe_block = -1;      // timesteps per pipeline block; 0 = both layers on the caller's stream
static int pipe_block() {
    if (g_pipe_block < 0) {
        const char* e = getenv("S2VT_PIPE_BLOCK");
        g_pipe_block = e ? atoi(e) : 32;
        if (g_pipe_block < 0) g_pipe_block = 0;
    }
    return g_pipe_block;
}
// HIP multiplexes streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, default 4); two streams that
// land on the same queue execute in order and the layer pipeline silently degenerates to serial execution (seen
// as soon as RCCL has created its own streams).  So the side stream is CHOSEN: candidates are created until one
// demonstrably runs concurrently with the caller's stream (two ~40 us spin kernels finish in about the time of one).
__global__ void spin_kernel(unsigned long long ticks_100mhz) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) {}
}
static bool streams_overlap(hipStream_t a, hipStream_t b) {
    hipEvent_t e0, e1, e2;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess)
        return true;   // cannot test: accept
    float both = 0.f, single = 0.f;
    for (int rep = 0; rep < 2; ++rep) {   // first repetition warms the code object up
        (void)hipEventRecord(e0, a);
        (void)hipStreamWaitEvent(b, e0, 0);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 4000ull);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 4000ull);
        (void)hipEventRecord(e1, b);
        (void)hipStreamWaitEvent(a, e1, 0);
        (void)hipEventRecord(e2, a);
        (void)hipEventSynchronize(e2);
        (void)hipEventElapsedTime(&both, e0, e2);
        (void)hipEventRecord(e0, a);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 4000ull);
        (void)hipEventRecord(e2, a);
        (void)hipEventSynchronize(e2);
        (void)hipEventElapsedTime(&single, e0, e2);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return both < 1.6f * single;
}
static int g_side_overlaps = -1;    // 1: verified concurrent with the first caller stream, 0: no candidate overlapped
static int side_stream(hipStream_t caller, hipStream_t* out) {
    if (!g_side) {
        hipStream_t cand = nullptr;
        g_side_overlaps = 0;
        for (int attempt = 0; attempt < 8 && !g_side_overlaps; ++attempt) {
            S2VT_HIP(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));   // rejected candidates stay alive so
            if (streams_overlap(caller, cand)) g_side_overlaps = 1;            // the next one gets another queue
        }
        g_side = cand;
    }
    *out = g_side;
    return 0;
}
static int get_event(size_t i, hipEvent_t* out) {
    while (g_events.size() <= i) {
        hipEvent_t e;
        S2VT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g_events.push_back(e);
    }
    *out = g_events[i];
    return 0;
}
// `to` waits for everything enqueued so far on `from`
static int handoff(hipStream_t from, hipStream_t to, size_t ev_index) {
    if (from == to) return 0;
    hipEvent_t e;
    int rc = get_event(ev_index, &e);
    if (rc) return rc;
    S2VT_HIP(hipEventRecord(e, from));
    S2VT_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
}

static int lgemm(const Lane& ln, bool ak, bool bk, int M, int N, int K, const float* A, int64_t lda, RowMap am,
                 const float* B, int64_t ldb, RowMap bm, float* C, int64_t ldc, RowMap cm, const float* bias, bool acc) {
    ProfScope ps(ln.s, K_GEMM, 1);
    return gemm_f32(ln.s, ak, bk, M, N, K, A, lda, am, B, ldb, bm, C, ldc, cm, bias, acc, ln.gws, ln.gws_floats);
}

// LSTM layer forward over steps [t0, t1) (time-major buffers, zero initial state at t = 0).
static int seq_fwd(hipStream_t st, int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
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
static int seq_bwd(hipStream_t st, int T, int t0, int t1, int B, int H, const float* w_hh_t, const float* dh_out,
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
static int balanced_block(int L, int blk) {
    if (blk != 32 || L <= 0) return blk;
    const int n = (L + 31) / 32;
    return (L + n - 1) / n;
}
// Block boundaries over [0, T) with L (first caption step) forced to be a boundary.
static std::vector<int> pipe_bounds(int T, int L, int blk) {
    std::vector<int> b;
    if (blk <= 0) { b.push_back(0); b.push_back(L); b.push_back(T); return b; }
    for (int t = 0; t < L; t += blk) b.push_back(t);
    for (int t = L; t < T; t += blk) b.push_back(t);
    b.push_back(T);
    return b;
}

// ------------------------------------------------------------------ split-precision (bf16 x 3) drivers
// Same launch structure as the fp32-MFMA drivers below, but every batched GEMM runs on the bf16 matrix cores
// from 3-plane operands (gemm_bf16.hip: fp32-equivalent products).  Each fp32 tensor that feeds a GEMM is
// rewritten ONCE per consumer orientation into packed planes by the memory-bound split kernels (split.hip),
// which also perform every transpose / gather / batch-major<->time-major permutation, so the MFMA kernel only
// sees k-contiguous operands.  The timestep kernels stay fp32 (they are latency/bandwidth bound, not MFMA bound).
// 0: fp32-input MFMA GEMMs; 3: split precision, 3 bf16 planes (fp32-equivalent); 1: plain bf16 operands (config 3:
// bf16 storage, fp32 accumulate) for the batched GEMMs AND the timestep kernels (lstm_bf16.hip)
static int g_gemm_mode = -1;
static int norm_mode(int m) { return (m == 3 || m == 1) ? m : 0; }
static int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("S2VT_GEMM_MODE");
        g_gemm_mode = norm_mode(e ? atoi(e) : 3);
    }
    return g_gemm_mode;
}
// plane modes need every k-offset inside a packed operand to be a multiple of 64 (k = time*B + b): B % 64 == 0
static bool planes_ok(const s2vt_dims& d) { return gemm_mode() != 0 && d.B % 64 == 0; }

// "Gradient group is final" events of the last s2vt_train_backward on this thread's device (data-parallel overlap):
// group 0 = out_linear (weight, bias), group 1 = word_rnn (4 tensors) + embedding; the rest is final with the call's stream.
static hipEvent_t g_grad_ev[2] = {nullptr, nullptr};
static bool g_grad_ev_set[2] = {false, false};
static int grads_ready(int group, hipStream_t s) {
    if (!g_grad_ev[group]) S2VT_HIP(hipEventCreateWithFlags(&g_grad_ev[group], hipEventDisableTiming));
    // inside a capture nothing is recorded (an event recorded on a capturing stream cannot be waited for from outside, and
    // external event-record nodes are refused by this runtime): the backward driver records both groups behind the graph
    // launch instead, so under s2vt_set_graph_mode(1) the gradient all-reduce follows the backward rather than overlapping it
    if (!g_capturing) S2VT_HIP(hipEventRecord(g_grad_ev[group], s));
    g_grad_ev_set[group] = true;
    return 0;
}

static int XP = 3;        // planes per operand of the running plane driver (3 or 1); set by the entry points
static inline int pad64(int x) { return (x + 63) / 64 * 64; }
struct PB { unsigned short* p; int64_t ld; int kpad; };       // packed planes of a k-major operand [rows][k]
// element offset of k index k0 (a multiple of 64) inside an operand: row layout (1 plane) k0; blocked 3-plane layout
// (gemm_x3.hip) k0/16 records of 3072 elements
static inline int64_t koff(int k0) { return XP == 3 ? (int64_t)k0 * 192 : (int64_t)k0 * XP; }
static inline size_t rows64(size_t r) { return (r + 63) / 64 * 64; }

struct PlaneWS {
    // forward
    PB feats, wf, x1, wih1, h1, we, wv, emb, h2r, wo, whh1, whh2;
    // backward
    PB dlog, woT, dlogT, h2decT, wvT, weT, wih1T, dg2, dg2T, h2T, h1T, embT, dg1, dg1T, x1T, dx1T, featsT, whh1T, whh2T;
    PB h2decB;               // decode-step hidden states as ROW planes in batch-major order (the k order of dlogits' rows): dW_o's B operand
                             // when the weight-gradient GEMMs read their operands transposed (tt_on())
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
    w.dlog = mk(R, V);        w.woT = mk(H, V);      w.dlogT = mk(V, R);     w.h2decT = mk(H, R);
    w.wvT = mk(H, 4 * H);     w.weT = mk(E, 4 * H);  w.wih1T = mk(H, 4 * H); w.dg2 = mk(T * B, 4 * H);
    w.dg2T = mk(4 * H, T * B); w.h2T = mk(H, T * B); w.h1T = mk(H, T * B);   w.embT = mk(E, R);
    w.dg1 = mk(T * B, 4 * H); w.dg1T = mk(4 * H, T * B); w.x1T = mk(H, L * B); w.dx1T = mk(H, L * B);
    w.featsT = mk(F, L * B);
    w.h2decB = mk(R, H);
    w.bytes = align_up(c.off, 256);
    return w;
}

// rows [r0, r0+rows) of the operand <- planes of in[rows][cols]
static int psplit(const Lane& ln, const PB& dst, int r0, const float* in, int64_t ld, RowMap imap, int rows, int cols) {
    return split_planes(ln.s, XP, false, in, ld, imap, rows, cols, dst.p + (int64_t)r0 * dst.ld, dst.ld, dst.kpad, rows);
}
// operand rows = input columns (all `cols` of them), k range [k0, k0+rows) <- planes of in[rows][cols]^T
static int psplitT(const Lane& ln, const PB& dst, int k0, const float* in, int64_t ld, RowMap imap, int rows, int cols) {
    return split_planes(ln.s, XP, true, in, ld, imap, rows, cols, dst.p + koff(k0), dst.ld, pad64(rows), cols);
}
// one pass over in[rows][cols]: row planes into r (operand rows r0..), transposed planes into t (k range k0..),
// 64-row partial column sums into colpart (each may be null)
static int pdual(const Lane& ln, const float* in, int64_t ld, RowMap imap, int rows, int cols, const PB* r, int r0,
                 const PB* t, int k0, float* colpart) {
    if (!r && !t && !colpart) return 0;          // (bf16 mode with transposed-read GEMMs: the recurrence kernels wrote the rows already)
    return split_planes_dual(ln.s, XP, in, ld, imap, rows, cols, r ? r->p + (int64_t)r0 * r->ld : nullptr, r ? r->ld : 0,
                             r ? r->kpad : 0, t ? t->p + koff(k0) : nullptr, t ? t->ld : 0, t ? pad64(rows) : 0,
                             colpart);
}
// C[M,N] (+)= A[rows a0.., k ka..ka+K) · B[rows b0.., k kb..kb+K)^T
static int pgemm(const Lane& ln, int M, int N, int K, const PB& A, int a0, int ka, const PB& B, int b0, int kb, float* C,
                 int64_t ldc, RowMap cm, const float* bias, bool acc) {
    ProfScope ps(ln.s, K_GEMM, 1);
    return gemm_bf16_nt(ln.s, XP, M, N, pad64(K), A.p + (int64_t)a0 * A.ld + koff(ka), A.ld,
                        B.p + (int64_t)b0 * B.ld + koff(kb), B.ld, C, ldc, cm, bias, acc, ln.gws, ln.gws_floats);
}

// Split-precision mode: the weight-gradient GEMMs (dW = dG^T h, dW_o = dlogits^T h2) read BOTH operands transposed from the row
// planes the forward / the BPTT hand-over already wrote (gemm_x3_kernel<4, true>), so the transposed twins of dG, dlogits, h, x1
// and the embedded words are never written (S2VT_TT=0: round 3's transposed planes, for A/B timing)
static bool tt_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("S2VT_TT"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
// C[M,N] = A_img[a_row0 .., :M]^T . B_img[b_row0 .., :N] over K image rows (row offsets: multiples of 64)
static int pgemm_tt(const Lane& ln, int M, int N, int K, const PB& A, int a_row0, const PB& B, int b_row0, float* C, int64_t ldc,
                    RowMap cm, const float* bias, bool acc) {
    ProfScope ps(ln.s, K_GEMM, 1);
    if (XP == 1)
        return gemm_b1_tt(ln.s, M, N, K, A.p + (int64_t)a_row0 * A.ld, A.ld, B.p + (int64_t)b_row0 * B.ld, B.ld, C, ldc, cm, bias, acc,
                          ln.gws, ln.gws_floats);
    return gemm_x3_tt(ln.s, M, N, K, A.p + (int64_t)a_row0 * A.ld, A.ld, B.p + (int64_t)b_row0 * B.ld, B.ld, C, ldc, cm, bias, acc,
                      ln.gws, ln.gws_floats);
}

// bf16-operand layer forward over steps [t0, t1): hb = bf16 row images of h (time-major, ld = hb.ld), the k-major
// plane the batched GEMMs read as well
static int seq_fwd_bf16(hipStream_t st, int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
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
static int seq_bwd_bf16(hipStream_t st, int T, int t0, int t1, int B, int H, const PB& wt, const float* dh_out,
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

// Recurrence schedule of the train drivers: 0 = one launch per timestep (lstm.hip / lstm_bf16.hip); 1 (default) = one
// persistent launch per block of timesteps with the W_hh slices resident per CU (lstm_persist.hip) for the bf16
// configuration, launch per timestep for fp32; 2 = persistent kernels for fp32 as well (lstm_persist_f32.hip).
// Why fp32 is not the default: its contraction runs on the exact-fp32 MFMA (1/16 of the bf16 rate) and is MFMA-bound
// inside the persistent kernel (4.6 us of a 8.6-us timestep at B = 64), two co-resident layers share that pipe, and the
// one-stream persistent schedule gives up the overlap of the batched GEMMs with the recurrence: measured at config 2
// 13.4 ms per step against 12.8 with launches per timestep (22.0 against 22.7 at B = 128).
static int g_persist = -1;
static int persist_mode() {
    if (g_persist < 0) { const char* e = getenv("S2VT_PERSIST"); g_persist = e ? atoi(e) : 1; if (g_persist < 0 || g_persist > 2) g_persist = 1; }
    return g_persist;
}
static bool persist_on() { return persist_mode() >= 1; }
static bool persist_f32_on() { return persist_mode() >= 2; }
// the fp32 persistent kernels per direction (experiments: S2VT_PERSIST_F32_FWD / S2VT_PERSIST_F32_BWD = 0 | 1 override mode 2's "both")
static bool persist_f32_dir_on(int dir) {
    static int ov[2] = {-2, -2};
    if (ov[dir] == -2) { const char* e = getenv(dir ? "S2VT_PERSIST_F32_BWD" : "S2VT_PERSIST_F32_FWD"); ov[dir] = e ? atoi(e) : -1; }
    return ov[dir] >= 0 ? ov[dir] > 0 : persist_f32_on();
}
// split-precision persistent forward for the fp32-equivalent arithmetic (gemm mode 3): S2VT_PERSIST_X3_FWD = 0 | 1
static bool persist_x3_fwd_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("S2VT_PERSIST_X3_FWD"); on = e ? (atoi(e) != 0) : 1; }
    return on && persist_on();
}
// split-precision persistent BPTT (reduce-scatter over the gate columns, lstm_persist_x3.hip): S2VT_PERSIST_X3_BWD = 0 | 1
// (= 2: experiment - one-LAYER persistent launches on the two lanes of the launch-per-timestep schedule: each holds half of the
// compute units, so both fit beside each other and the lanes' GEMMs keep their overlap)
static int persist_x3_bwd_mode() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("S2VT_PERSIST_X3_BWD"); on = e ? atoi(e) : 0; if (on < 0 || on > 2) on = 0; }
    return persist_on() ? on : 0;
}
static bool persist_x3_bwd_on() { return persist_x3_bwd_mode() != 0; }
static SeqBwdX3Args persist_bwd_x3_args(int T, int t0, int t1, int B, int H, int64_t Kp, int64_t Hp, const unsigned short* wtp,
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
static SeqFwdX3Args persist_fwd_x3_args(int t0, int t1, int B, int H, int T, int64_t Kp, float* gx_stash, int n_gx, const float* bias,
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
static bool persist_fwd_ok(int B, int H, const PB& wb, const PB& hb) {
    return persist_on() && lstm_seq_fwd_bf16_persist_supported(B, H, hb.kpad) && hb.kpad == wb.kpad;
}
static SeqFwdBf16Args persist_fwd_args(int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
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

static SeqBwdBf16Args seq_bwd_bf16_args(int T, int t0, int t1, int B, int H, const PB& wt, const PB& dgb, const float* dh_out,
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
static SeqFwdF32Args persist_fwd_f32_args(int t0, int t1, int B, int H, float* gx_stash, int n_gx, const float* bias,
                                          const float* w_hh, float* h_all, float* c_all, unsigned int* sync, int* err) {
    SeqFwdF32Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.t0 = t0; a.t1 = t1; a.n_gx = n_gx;
    a.w_hh = w_hh; a.ldw = H;
    a.h_all = h_all; a.gx_stash = gx_stash; a.bias = bias; a.c_all = c_all;
    a.sync = sync; a.err = err;
#ifdef S2VT_EXPERIMENT_STAMPS
    a.stamps = g_xstamps; a.stamp_block = g_xstamp_block;
#endif
    return a;
}
static SeqBwdF32Args persist_bwd_f32_args(int T, int t0, int t1, int B, int H, const float* w_hh_t, const float* dh_out,
                                          int dh_first, const float* c_all, float* stash_dg, float* dc, unsigned int* sync, int* err) {
    SeqBwdF32Args a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.T = T; a.t0 = t0; a.t1 = t1;
    a.w_hh_t = w_hh_t; a.ldwt = 4 * (int64_t)H;
    a.dh_out = dh_out; a.dh_first = dh_first;
    a.stash_dg = stash_dg; a.c_all = c_all; a.dc = dc;
    a.sync = sync; a.err = err;
    return a;
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
    g_fwd_records[ws] = FwdRecord{d, gemm_mode(), planes ? ((gemm_mode() == 1) ? 1 : 3) : 0, persist_mode(), pipe_block(), ++g_fwd_seq, false};
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
    S2VT_REQUIRE(r.gemm_mode == gemm_mode() && r.planes == planes_now && r.persist == persist_mode() && r.blk == pipe_block(),
                 "s2vt_train_backward: the forward ran with gemm mode %d / recurrence mode %d / pipeline block %d, now %d / %d / %d: the "
                 "workspace layout differs (do not change s2vt_set_gemm_mode / s2vt_set_recurrence_mode / s2vt_set_pipeline_block "
                 "between a forward and its backward)", r.gemm_mode, r.persist, r.blk, gemm_mode(), persist_mode(), pipe_block());
    return 0;
}

// out_mask: optional out_drop mask (S2VTModel.py:79), time-major [(L-1)*B, H], entries 0 or 1/(1-p); nullptr = no dropout.
// The masked decode-step hidden states replace the row planes of the logits GEMM (the recurrence is done with them by then).
// (tt_on(): the weight-gradient GEMMs read the UNMASKED rows of q.h2r transposed in the backward, so the masked rows go to the
// scratch image q.h2decB - which the backward fills itself before it reads it - and *a_img / *a_row0 name the logits GEMM's operand)
static bool tt_on();
static int masked_logits_planes(const Lane& ln, const TrainWS& w, const PlaneWS& q, const float* out_mask, int B, int L, int H,
                                const PB** a_img, int* a_row0) {
    *a_img = &q.h2r; *a_row0 = L * B;
    if (!out_mask) return 0;
    const int R = (L - 1) * B;
    int rc;
    if ((rc = mul_vectors(ln.s, w.h2 + (int64_t)L * B * H, out_mask, w.dh2dec, (int64_t)R * H))) return rc;   // dh2dec: free in the forward
    if (tt_on()) { *a_img = &q.h2decB; *a_row0 = 0; }
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
    const bool tt = tt_on();
    if ((rc = pdual(lb, p->emb_w, E, gather(w.tok), R, E, &q.emb, 0, tt ? nullptr : &q.embT, 0, nullptr))) return rc;
    if ((rc = pgemm(lb, R, 4 * H, E, q.emb, 0, 0, q.we, 0, 0, w.s2 + (int64_t)L * B4H, 4 * H, ID, w.bsum2, false))) return rc;
    // lane A: feature projection and vid_rnn input GEMM                       S2VTModel.py:54, 64-67
    if ((rc = psplit(la, q.feats, 0, feats, F, ID, B * L, F))) return rc;
    if ((rc = psplit(la, q.wf, 0, p->feat_w, F, ID, H, F))) return rc;
    if ((rc = pdual(la, p->vid_w_ih, H, ID, 4 * H, H, &q.wih1, 0, &q.wih1T, 0, nullptr))) return rc;
    if ((rc = pgemm(la, B * L, H, F, q.feats, 0, 0, q.wf, 0, 0, w.x1, H, perm(L, B), p->feat_b, false))) return rc;
    if ((rc = pdual(la, w.x1, H, ID, L * B, H, &q.x1, 0, tt ? nullptr : &q.x1T, 0, nullptr))) return rc;
    if ((rc = pgemm(la, L * B, 4 * H, H, q.x1, 0, 0, q.wih1, 0, 0, w.s1, 4 * H, ID, w.bsum1, false))) return rc;
    const bool pbf_fwd = bf && blk > 0 && persist_fwd_ok(B, H, q.whh1, q.h1);
    const std::vector<int> bd = pipe_bounds(T, L, (pbf_fwd || px3_fwd) ? balanced_block(L, blk) : blk);
    if (px3_fwd || (!bf && blk > 0 && persist_f32_dir_on(0) && lstm_seq_fwd_f32_persist_supported(B, H))) {
        // fp32-equivalent persistent schedule (lstm_persist_x3.hip: split precision on the bf16 matrix cores; lstm_persist_f32.hip:
        // exact-fp32 MFMA), ONE stream: stage k = vid_rnn block k next to word_rnn block k-1
        if ((rc = handoff(sx, st, ev++))) return rc;
        const int nb = (int)bd.size() - 1;
        for (int k = 0; k <= nb; ++k) {
            const bool hv = k < nb, hw = k >= 1;
            {
            ProfScope ps(st, K_STEP_FWD, (hv ? bd[k + 1] - bd[k] : 0) + (hw ? bd[k] - bd[k - 1] : 0));
            if (px3_fwd) {
                SeqFwdX3Args av, aw;
                if (hv) av = persist_fwd_x3_args(bd[k], bd[k + 1], B, H, T, w.xkp, w.s1, L, w.bsum1, w.xw1, w.xh1, w.h1, w.c1, w.psync_a, w.err + 1);
                if (hw) aw = persist_fwd_x3_args(bd[k - 1], bd[k], B, H, T, w.xkp, w.s2, T, w.bsum2, w.xw2, w.xh2, w.h2, w.c2, w.psync_b, w.err + 1);
                if (hv && hw) rc = lstm_seq_fwd_x3_persist2(st, av, &aw);
                else rc = lstm_seq_fwd_x3_persist2(st, hv ? av : aw, nullptr);
            } else {
                SeqFwdF32Args av, aw;
                if (hv) av = persist_fwd_f32_args(bd[k], bd[k + 1], B, H, w.s1, L, w.bsum1, p->vid_w_hh, w.h1, w.c1, w.psync_a, w.err + 1);
                if (hw) aw = persist_fwd_f32_args(bd[k - 1], bd[k], B, H, w.s2, T, w.bsum2, p->word_w_hh, w.h2, w.c2, w.psync_b, w.err + 1);
                if (hv && hw) rc = lstm_seq_fwd_f32_persist2(st, av, &aw);
                else rc = lstm_seq_fwd_f32_persist2(st, hv ? av : aw, nullptr);
            }
            }
            if (rc) return rc;
            if (hw) {
                const int t0 = bd[k - 1], t1 = bd[k];
                const bool cap = t0 >= L;
                if ((rc = pdual(la, w.h2 + t0 * BH, H, ID, (t1 - t0) * B, H, (cap || tt) ? &q.h2r : nullptr, t0 * B, tt ? nullptr : &q.h2T,
                                t0 * B, nullptr)))
                    return rc;
            }
            if (hv) {
                const int t0 = bd[k], t1 = bd[k + 1];
                const bool cap = t0 >= L;
                if ((rc = pdual(la, w.h1 + t0 * BH, H, ID, (t1 - t0) * B, H, &q.h1, t0 * B, tt ? nullptr : &q.h1T, t0 * B, nullptr))) return rc;
                if ((rc = pgemm(la, (t1 - t0) * B, 4 * H, H, q.h1, t0 * B, 0, q.wv, 0, 0, w.s2 + t0 * B4H, 4 * H, ID,
                                cap ? nullptr : w.bsum2, cap)))
                    return rc;
            }
        }
        const PB* lg; int lg0;
        if ((rc = masked_logits_planes(la, w, q, out_mask, B, L, H, &lg, &lg0))) return rc;
        return pgemm(la, R, V, H, *lg, lg0, 0, q.wo, 0, 0, logits, V, perm(B, L - 1), p->out_b, false);
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
            if (hw) {   // h2 of word block k-1: transposed planes for dW_hh2 (the row planes were written by the kernel)
                const int t0 = bd[k - 1], t1 = bd[k];
                if ((rc = pdual(la, w.h2 + t0 * BH, H, ID, (t1 - t0) * B, H, nullptr, t0 * B, tt ? nullptr : &q.h2T, t0 * B, nullptr))) return rc;
            }
            if (hv) {   // vid_out half of the word_rnn gate input for block k
                const int t0 = bd[k], t1 = bd[k + 1];
                const bool cap = t0 >= L;
                if ((rc = pdual(la, w.h1 + t0 * BH, H, ID, (t1 - t0) * B, H, nullptr, t0 * B, tt ? nullptr : &q.h1T, t0 * B, nullptr))) return rc;
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
        if ((rc = pdual(lb, w.h1 + t0 * BH, H, ID, (t1 - t0) * B, H, bf ? nullptr : &q.h1, t0 * B, tt ? nullptr : &q.h1T, t0 * B, nullptr)))
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
        if ((rc = pdual(lb, w.h2 + t0 * BH, H, ID, (t1 - t0) * B, H, ((cap || tt) && !bf) ? &q.h2r : nullptr, t0 * B, tt ? nullptr : &q.h2T,
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
    const bool tt = tt_on();
    if (!dlog_ready && (rc = pdual(la, dlogits, V, ID, R, V, &q.dlog, 0, tt ? nullptr : &q.dlogT, 0, w.colsum_c))) return rc;
    if ((rc = handoff(st, sx, ev++))) return rc;
    if ((rc = pgemm(la, R, H, V, q.dlog, 0, 0, q.woT, 0, 0, w.dh2dec, H, perm(L - 1, B), nullptr, false))) return rc;
    // bf16 mode with the fused criterion backward: the dlogits planes carry the power of two of gout / rows only (split.hip); the
    // mantissa multiplies the two fp32 products of those planes - here and dW_o below (the bias gradient already has it)
    const bool ce_pow2 = dlog_ready && bf;
    if (ce_pow2 && (rc = scale_by_device_scalar(st, w.dh2dec, (int64_t)R * H, w.ce_alpha))) return rc;
    if (out_mask && (rc = mul_vectors(st, w.dh2dec, out_mask, w.dh2dec, (int64_t)R * H))) return rc;      // autograd of out_drop
    if (!bf && (rc = transpose_f32(st, p->word_w_hh, 4 * H, H, w.wt2))) return rc;       // (the bf16 BPTT reads the W_hh^T planes instead)
    // lane B meanwhile: out_linear weight/bias gradients (k = batch-major row index) and W_hh1^T
    {
        const float* h2dec = w.h2 + L * BH;
        if (out_mask) {      // dW_o sees the masked hidden states (dx1 is free until the vid_rnn input gradient)
            if ((rc = mul_vectors(sx, w.h2 + L * BH, out_mask, w.dx1, (int64_t)R * H))) return rc;
            h2dec = w.dx1;
        }
        if (tt) {            // rows in dlogits' (batch-major) order, read transposed by the GEMM
            if ((rc = psplit(lb, q.h2decB, 0, h2dec, H, perm(L - 1, B), R, H))) return rc;
            if ((rc = pgemm_tt(lb, V, H, R, q.dlog, 0, q.h2decB, 0, g->out_w, H, ID, nullptr, false))) return rc;
        } else {
            if ((rc = psplitT(lb, q.h2decT, 0, h2dec, H, perm(L - 1, B), R, H))) return rc;
            if ((rc = pgemm(lb, V, H, R, q.dlogT, 0, 0, q.h2decT, 0, 0, g->out_w, H, ID, nullptr, false))) return rc;
        }
        if (ce_pow2 && (rc = scale_by_device_scalar(sx, g->out_w, (int64_t)V * H, w.ce_alpha))) return rc;
    }
    if ((rc = colsum_finish(sx, w.colsum_c, cdiv(R, 64), V, g->out_b, false))) return rc;
    if ((rc = grads_ready(0, sx))) return rc;
    if (!bf && (rc = transpose_f32(sx, p->vid_w_hh, 4 * H, H, w.wt1))) return rc;
    const bool pbf_bwd = bf && blk > 0 && persist_on() && lstm_seq_bwd_bf16_persist_supported(B, H, q.dg2.kpad) && q.dg2.kpad == q.whh2T.kpad;
    const bool px3_any = !bf && XP == 3 && blk > 0 && w.xnslots > 0 && persist_x3_bwd_on() && lstm_seq_bwd_x3_persist_supported(B, H) &&
                         w.xnslots > (blk < T ? blk : T);
    const bool px3_bwd = px3_any && persist_x3_bwd_mode() == 1, px3_lanes = px3_any && persist_x3_bwd_mode() == 2;
    const std::vector<int> bd = pipe_bounds(T, L, (pbf_bwd || px3_bwd) ? balanced_block(L, blk) : blk);
    if (px3_any) {   // W_hh^T of both layers as planes (each on the lane that transposed it)
        if ((rc = split3_wt(st, w.wt2, H, (int)w.xkp, (int)w.xhp, w.xwt2, w.xkp * 4 * w.xhp))) return rc;
        if ((rc = split3_wt(sx, w.wt1, H, (int)w.xkp, (int)w.xhp, w.xwt1, w.xkp * 4 * w.xhp))) return rc;
    }
    if (px3_bwd || (!bf && blk > 0 && persist_f32_dir_on(1) && lstm_seq_bwd_f32_persist_supported(B, H))) {
        // fp32-equivalent persistent schedule (split precision: lstm_persist_x3.hip; exact-fp32 MFMA: lstm_persist_f32.hip), ONE
        // stream: stage k = word_rnn BPTT of block k next to vid_rnn BPTT of block k+1
        if ((rc = handoff(sx, st, ev++))) return rc;               // W_hh1^T and the out_linear gradients of lane B
        const int nb = (int)bd.size() - 1;
        for (int k = nb - 1; k >= -1; --k) {
            const bool hw = k >= 0, hv = k + 1 <= nb - 1;
            {
                ProfScope ps(st, K_STEP_BWD, (hw ? bd[k + 1] - bd[k] : 0) + (hv ? bd[k + 2] - bd[k + 1] : 0));
                if (px3_bwd) {
                    SeqBwdX3Args aw, av;
                    if (hw) aw = persist_bwd_x3_args(T, bd[k], bd[k + 1], B, H, w.xkp, w.xhp, w.xwt2, w.dh2dec, L, w.c2, w.s2, w.dc2,
                                                     w.xpart2, w.xpslot, w.xnslots, w.psync_a, w.err + 1);
                    if (hv) av = persist_bwd_x3_args(T, bd[k + 1], bd[k + 2], B, H, w.xkp, w.xhp, w.xwt1, w.dh1, 0, w.c1, w.s1, w.dc1,
                                                     w.xpart1, w.xpslot, w.xnslots, w.psync_b, w.err + 1);
                    if (hw && hv) rc = lstm_seq_bwd_x3_persist2(st, aw, &av);
                    else rc = lstm_seq_bwd_x3_persist2(st, hw ? aw : av, nullptr);
                } else {
                    SeqBwdF32Args aw, av;
                    if (hw) aw = persist_bwd_f32_args(T, bd[k], bd[k + 1], B, H, w.wt2, w.dh2dec, L, w.c2, w.s2, w.dc2, w.psync_a, w.err + 1);
                    if (hv) av = persist_bwd_f32_args(T, bd[k + 1], bd[k + 2], B, H, w.wt1, w.dh1, 0, w.c1, w.s1, w.dc1, w.psync_b, w.err + 1);
                    if (hw && hv) rc = lstm_seq_bwd_f32_persist2(st, aw, &av);
                    else rc = lstm_seq_bwd_f32_persist2(st, hw ? aw : av, nullptr);
                }
                if (rc) return rc;
            }
            if (hw) {
                const int t0 = bd[k], t1 = bd[k + 1];
                if ((rc = pdual(la, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, &q.dg2, t0 * B, tt ? nullptr : &q.dg2T, t0 * B,
                                w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
                if ((rc = pgemm(la, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false)))
                    return rc;
            }
            if (hv) {
                const int t0 = bd[k + 1], t1 = bd[k + 2];
                if ((rc = pdual(la, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, (t0 < L || tt) ? &q.dg1 : nullptr, t0 * B,
                                tt ? nullptr : &q.dg1T, t0 * B, w.colsum_b + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
            }
        }
        if ((rc = grads_ready(0, st))) return rc;                  // (see the bf16 branch below)
        if ((rc = handoff(st, sx, ev++))) return rc;
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
                if (rc) return rc;
            }
            if (hw) {
                const int t0 = bd[k], t1 = bd[k + 1];
                if ((rc = pdual(la, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, nullptr, t0 * B, tt ? nullptr : &q.dg2T, t0 * B,
                                w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
                    return rc;
                if ((rc = pgemm(la, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false)))
                    return rc;
            }
            if (hv) {
                const int t0 = bd[k + 1], t1 = bd[k + 2];
                if ((rc = pdual(la, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, nullptr, t0 * B, tt ? nullptr : &q.dg1T, t0 * B,
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
    // one-layer persistent block of the split-precision BPTT on stream s (px3_lanes)
    auto x3_block = [&](hipStream_t s, int t0, int t1, bool word) -> int {
        ProfScope ps(s, K_STEP_BWD, t1 - t0);
        const SeqBwdX3Args a = word ? persist_bwd_x3_args(T, t0, t1, B, H, w.xkp, w.xhp, w.xwt2, w.dh2dec, L, w.c2, w.s2, w.dc2, w.xpart2,
                                                          w.xpslot, w.xnslots, w.psync_a, w.err + 1)
                                    : persist_bwd_x3_args(T, t0, t1, B, H, w.xkp, w.xhp, w.xwt1, w.dh1, 0, w.c1, w.s1, w.dc1, w.xpart1,
                                                          w.xpslot, w.xnslots, w.psync_b, w.err + 1);
        return lstm_seq_bwd_x3_persist2(s, a, nullptr);
    };
    for (size_t k = bd.size() - 1; k >= 1; --k) {
        const int t0 = bd[k - 1], t1 = bd[k];
        if (bf) {
            if ((rc = seq_bwd_bf16(st, T, t0, t1, B, H, q.whh2T, w.dh2dec, L, w.c2, w.s2, q.dg2, w.dc2))) return rc;
        } else if (px3_lanes) {
            if ((rc = x3_block(st, t0, t1, true))) return rc;
        } else {
            if ((rc = seq_bwd(st, T, t0, t1, B, H, w.wt2, w.dh2dec, L, w.c2, w.s2, w.dc2))) return rc;
        }
        // dG2 of this block: row planes (dh1, d-embedding GEMMs), transposed planes (weight gradients) and the
        // bias-gradient partial sums, all from one read
        if ((rc = pdual(la, w.s2 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, bf ? nullptr : &q.dg2, t0 * B, tt ? nullptr : &q.dg2T, t0 * B,
                        w.colsum_a + (int64_t)(t0 * B / 64) * 4 * H)))
            return rc;
        if ((rc = pgemm(la, (t1 - t0) * B, H, 4 * H, q.dg2, t0 * B, 0, q.wvT, 0, 0, w.dh1 + t0 * BH, H, ID, nullptr, false)))
            return rc;
        if ((rc = handoff(st, sx, ev++))) return rc;
        if (bf) {
            if ((rc = seq_bwd_bf16(sx, T, t0, t1, B, H, q.whh1T, w.dh1, 0, w.c1, w.s1, q.dg1, w.dc1))) return rc;
        } else if (px3_lanes) {
            if ((rc = x3_block(sx, t0, t1, false))) return rc;
        } else {
            if ((rc = seq_bwd(sx, T, t0, t1, B, H, w.wt1, w.dh1, 0, w.c1, w.s1, w.dc1))) return rc;
        }
        if ((rc = pdual(lb, w.s1 + t0 * B4H, 4 * H, ID, (t1 - t0) * B, 4 * H, ((t0 < L || tt) && !bf) ? &q.dg1 : nullptr, t0 * B,
                        tt ? nullptr : &q.dg1T, t0 * B, w.colsum_b + (int64_t)(t0 * B / 64) * 4 * H)))
            return rc;
    }
    if (px3_lanes && (rc = grads_ready(0, st))) return rc;         // (as behind the one-stream persistent schedules above)
    }
    // lane A: word_rnn parameter gradients + embedding gradient
    if (tt) {            // dW = dG^T . (h | emb): row planes of both, read transposed
        if ((rc = pgemm_tt(la, 4 * H, H, (T - 1) * B, q.dg2, B, q.h2r, 0, g->word_w_hh, H, ID, nullptr, false))) return rc;
        if ((rc = pgemm_tt(la, 4 * H, H, T * B, q.dg2, 0, q.h1, 0, g->word_w_ih + E, E + H, ID, nullptr, false))) return rc;
        if ((rc = pgemm_tt(la, 4 * H, E, R, q.dg2, L * B, q.emb, 0, g->word_w_ih, E + H, ID, nullptr, false))) return rc;
    } else {
    if ((rc = pgemm(la, 4 * H, H, (T - 1) * B, q.dg2T, 0, B, q.h2T, 0, 0, g->word_w_hh, H, ID, nullptr, false))) return rc;
    if ((rc = pgemm(la, 4 * H, H, T * B, q.dg2T, 0, 0, q.h1T, 0, 0, g->word_w_ih + E, E + H, ID, nullptr, false))) return rc;
    if ((rc = pgemm(la, 4 * H, E, R, q.dg2T, 0, L * B, q.embT, 0, 0, g->word_w_ih, E + H, ID, nullptr, false))) return rc;
    }
    if ((rc = colsum_finish(st, w.colsum_a, T * B / 64, 4 * H, g->word_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->word_b_hh, g->word_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, st));
    if ((rc = pgemm(la, R, E, 4 * H, q.dg2, L * B, 0, q.weT, 0, 0, w.de, E, ID, nullptr, false))) return rc;
    if ((rc = embedding_grad(st, w.de, R, E, w.tok, V, g->emb_w, w.embws))) return rc;
    if ((rc = grads_ready(1, st))) return rc;
    // lane B: vid_rnn and feat_linear parameter gradients (S2VT_SERIAL_TAIL=1: behind lane A's on the caller's stream -
    // an experiment switch: two MFMA-bound GEMMs side by side share the chip, neither gets faster)
    static const bool serial_tail = getenv("S2VT_SERIAL_TAIL") && atoi(getenv("S2VT_SERIAL_TAIL")) != 0;
    const Lane lt = serial_tail ? Lane{st, w.gws_b, w.gws_floats, w.colsum_b} : lb;
    if (tt) {
        if ((rc = pgemm_tt(lt, 4 * H, H, (T - 1) * B, q.dg1, B, q.h1, 0, g->vid_w_hh, H, ID, nullptr, false))) return rc;
        if ((rc = pgemm_tt(lt, 4 * H, H, L * B, q.dg1, 0, q.x1, 0, g->vid_w_ih, H, ID, nullptr, false))) return rc;
    } else {
    if ((rc = pgemm(lt, 4 * H, H, (T - 1) * B, q.dg1T, 0, B, q.h1T, 0, 0, g->vid_w_hh, H, ID, nullptr, false))) return rc;
    if ((rc = pgemm(lt, 4 * H, H, L * B, q.dg1T, 0, 0, q.x1T, 0, 0, g->vid_w_ih, H, ID, nullptr, false))) return rc;
    }
    if ((rc = colsum_finish(lt.s, w.colsum_b, T * B / 64, 4 * H, g->vid_b_ih, false))) return rc;
    S2VT_HIP(hipMemcpyAsync(g->vid_b_hh, g->vid_b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, lt.s));
    if (tt) {
        // dx1 comes out in BATCH-major row order (the order of feats' rows, whose row planes the forward wrote): dW_f = dx1^T feats
        // reads both transposed - no time-major transposed copy of the features, no transposed dx1 (q.x1 is free: dW_ih1 is done)
        if ((rc = pgemm(lt, L * B, H, 4 * H, q.dg1, 0, 0, q.wih1T, 0, 0, w.dx1, H, perm(B, L), nullptr, false))) return rc;
        if ((rc = pdual(lt, w.dx1, H, ID, L * B, H, &q.x1, 0, nullptr, 0, w.colsum_b))) return rc;
        if ((rc = pgemm_tt(lt, H, F, L * B, q.x1, 0, q.feats, 0, g->feat_w, F, ID, nullptr, false))) return rc;
    } else {
    if ((rc = pgemm(lt, L * B, H, 4 * H, q.dg1, 0, 0, q.wih1T, 0, 0, w.dx1, H, ID, nullptr, false))) return rc;
    if ((rc = pdual(lt, w.dx1, H, ID, L * B, H, nullptr, 0, &q.dx1T, 0, w.colsum_b))) return rc;
    if ((rc = psplitT(lt, q.featsT, 0, feats, F, perm(B, L), L * B, F))) return rc;
    if ((rc = pgemm(lt, H, F, L * B, q.dx1T, 0, 0, q.featsT, 0, 0, g->feat_w, F, ID, nullptr, false))) return rc;
    }
    if ((rc = colsum_finish(lt.s, w.colsum_b, L * B / 64, H, g->feat_b, false))) return rc;
    if (dfeats) {   // rarely requested (nothing reads it in the reference): fp32-MFMA GEMM
        if ((rc = lgemm(lt, true, false, L * B, F, H, w.dx1, H, ID, p->feat_w, F, ID, dfeats, F, tt ? ID : perm(B, L), nullptr, false)))
            return rc;
    }
    return handoff(sx, st, ev++);
}

}  // namespace s2vt

using namespace s2vt;

extern "C" {

int s2vt_abi_version(void) { return S2VT_ABI_VERSION; }
const char* s2vt_last_error(void) { return g_err; }

size_t s2vt_train_workspace_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    size_t n = carve_train(*d, nullptr).bytes;
    if (planes_ok(*d)) {
        const int keep = XP;                       // a size query must not change the state of a running path
        XP = (gemm_mode() == 1) ? 1 : 3;
        n += carve_planes(*d, nullptr).bytes;
        XP = keep;
    }
    return n;
}

static int train_forward_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, const int64_t* targets,
                              int64_t targets_ld, float* logits, void* workspace, size_t workspace_bytes, void* stream,
                              const float* out_mask) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && targets && logits && workspace, "s2vt_train_forward: null/invalid argument");
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
            for (int v : {d->B, d->L, d->F, d->H, d->E, d->V, gemm_mode(), persist_mode(), pipe_block()}) key.push_back((uint64_t)v);
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

int s2vt_check_async_error(int32_t wait) { return poll_async_error(wait != 0); }

static int train_backward_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, const float* dlogits,
                               const s2vt_grads* g, float* dfeats, void* workspace, size_t workspace_bytes, void* stream,
                               const float* out_mask) {
    S2VT_REQUIRE(dims_ok(d) && p && feats && g && workspace, "s2vt_train_backward: null/invalid argument");
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
            for (int v : {d->B, d->L, d->F, d->H, d->E, d->V, gemm_mode(), persist_mode(), pipe_block()}) key.push_back((uint64_t)v);
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
struct DecodeConst { PB wf, wih1, wv, wo; float* gtab; unsigned short *xw1, *xw2; PB whh; size_t bytes; };   // xw: W_hh planes [3][4H][Kp]; whh: word_rnn's W_hh, blocked
static DecodeConst carve_decode_const(const s2vt_dims& d, void* base) {
    const size_t F = d.F, H = d.H;
    Carver c{reinterpret_cast<char*>(base), 0, 0};
    DecodeConst k;
    k.gtab = nullptr;
    k.xw1 = k.xw2 = nullptr;
    k.wf = k.wih1 = k.wv = k.wo = k.whh = PB{nullptr, 0, 0};
    if (planes_ok(d)) {
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
    if (cache && planes_ok(*d) && H <= 1024) {
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
        return post_async_error(st, w.err);
    }
    S2VT_REQUIRE(!gx_vid, "s2vt_beam_step_gx: the precomputed vid_rnn half needs the plane path (B % 64 == 0, H <= 1024)");
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
    return post_async_error(st, w.err);
}

int s2vt_backward_wait_grads(int32_t group, void* stream) {
    S2VT_REQUIRE(group == 0 || group == 1, "s2vt_backward_wait_grads: group must be 0 (out_linear) or 1 (word_rnn + embedding)");
    S2VT_REQUIRE(g_grad_ev_set[group], "s2vt_backward_wait_grads: no s2vt_train_backward has run yet");
    S2VT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_grad_ev[group], 0));
    return 0;
}

size_t s2vt_decode_workspace_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    return carve_decode(*d, nullptr).bytes + carve_decode_const(*d, nullptr).bytes;
}
size_t s2vt_decode_cache_bytes(const s2vt_dims* d) {
    if (!dims_ok(d)) return 0;
    return carve_decode_const(*d, nullptr).bytes;
}

struct EncodeOut { float *vid_h, *vid_c, *word_h, *word_c; float* gx_dec; int depth; };      // states [B, H] after the L encode steps;
                                     // optional: word_rnn's vid_out gate input (+ biases) of the first `depth` decode steps [depth][B][4H]
static int greedy_decode_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
                              void* workspace, size_t workspace_bytes, void* cache, size_t cache_bytes, bool cache_valid,
                              void* stream, const EncodeOut* enc = nullptr);
// schedule of the 79 token-dependent decode steps on the plane path: 1 = fused (the next step's recurrent GEMM inside the
// argmax launch + a cell-update launch), 0 = a step kernel and an argmax kernel per step, batch halves as two chains
static int g_decode_schedule = -1;
static int decode_schedule() {
    if (g_decode_schedule < 0) { const char* e = getenv("S2VT_DECODE_FUSED"); g_decode_schedule = (e && atoi(e) == 0) ? 0 : 1; }
    return g_decode_schedule;
}
int s2vt_set_decode_schedule(int32_t schedule) {
    const int prev = decode_schedule();
    if (schedule == 0 || schedule == 1) g_decode_schedule = schedule;
    return prev;
}
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
static int greedy_decode_impl(const s2vt_dims* d, const s2vt_params* p, const float* feats, int32_t sos_ix, int64_t* ids,
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
    static const bool argmax_f32 = getenv("S2VT_ARGMAX_F32") && atoi(getenv("S2VT_ARGMAX_F32")) != 0;   // A/B switch: the fp32-MFMA kernel
    const bool ax3 = x3 && !argmax_f32;
    // per-token gate-input table instead of the embedding K segment of the 79 decode steps: one V x 4H x E GEMM (0.5 ms at
    // V = 12000) against B x 4H x E of MFMA work and E/(E+H) of the operand traffic in EVERY decode step - pays from B ~ 64
    static const bool no_tab = getenv("S2VT_DECODE_TABLE") && atoi(getenv("S2VT_DECODE_TABLE")) == 0;
    const bool use_tab = x3 && !no_tab;
    // encode phase (both layers, L steps) and vid_rnn's input-free decode steps as persistent split-precision launches; only the
    // 79 token-dependent word_rnn steps stay one launch (+ argmax) per step
    const bool use_px = x3 && blk > 0 && w.xkp > 0 && persist_x3_fwd_on() && lstm_seq_fwd_x3_persist_supported(B, H);
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
        auto gx2_block = [&](int t0, int t1) -> int {           // vid_out half of word_rnn's gate input for steps [t0, t1) (+ biases)
            int r;
            if ((r = psplit(la, w.ph1, t0 * B, w.h1 + t0 * BH, H, ID, (t1 - t0) * B, H))) return r;
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
            {
                ProfScope ps(st, K_STEP_FWD, T - tv);
                if ((rc = lstm_seq_fwd_x3_persist2(st, av, nullptr))) return rc;
            }
            if ((rc = gx2_block(tv, T))) return rc;
        }
        // The 79 token-dependent steps.  A decode step is two dependent launches (word_rnn step, out_linear + argmax) that
        // each leave part of the chip idle (188 of 256 compute units in the argmax; launch gaps and tails between the two) and
        // batch rows never interact: at B % 128 == 0 the two halves of the batch run as two INDEPENDENT chains on the two
        // streams, so one half's step kernel fills the other half's gaps (S2VT_DECODE_HALVES=0: one chain)
        // Fused schedule (s2vt_set_decode_schedule(1), the default; S2VT_DECODE_FUSED=0 selects the two-chain schedule below): h_t·W_hh^T of step t+1 does not depend on step t's token - only the
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
        static const bool halves_off = getenv("S2VT_DECODE_HALVES") && atoi(getenv("S2VT_DECODE_HALVES")) == 0;
        const int nh = (!halves_off && ax3 && B % 128 == 0 && sx != st) ? 2 : 1;
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
    S2VT_REQUIRE(!enc, "s2vt_decode_encode_cached: this shape / mode does not take the persistent split-precision encode phase");
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

// ------------------------------------------------------------------ loss
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
        const bool tt = tt_on();
        if ((rc = split_planes_dual(st, XP, logits, V, ID, R, V, q.dlog.p, q.dlog.ld, q.dlog.kpad, tt ? nullptr : q.dlogT.p + koff(0),
                                    tt ? 0 : q.dlogT.ld, tt ? 0 : pad64(R), w.colsum_c, &ce)))
            return rc;
    }
    std::lock_guard<std::mutex> lock(g_fwd_mutex);
    g_fwd_records[workspace].dlog_ready = true;
    return 0;
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
    else if (persist_f32_dir_on(0) && lstm_seq_fwd_f32_persist_supported(B, H)) *fwd = 2;
    if (H <= 1024 && persist_x3_bwd_on() && lstm_seq_bwd_x3_persist_supported(B, H)) *bwd = 3;        // (one-stream or lanes mode)
    else if (persist_f32_dir_on(1) && lstm_seq_bwd_f32_persist_supported(B, H)) *bwd = 2;
    return 0;
}

int s2vt_set_recurrence_mode(int32_t mode) {
    const int prev = persist_mode();
    if (mode >= 0) g_persist = mode > 2 ? 2 : mode;
    return prev;
}

// fp32 persistent recurrence as its own entry points (kernel-level parity tests and benchmarks).
// workspace: [err int x64][sync A][sync B]
size_t s2vt_lstm_persist_workspace_bytes(void) { return 256 + 2 * lstm_persist_sync_bytes(); }
int s2vt_lstm_seq_fwd_persist(int32_t T, int32_t B, int32_t H, float* gx_stash0, float* gx_stash1, int32_t n_gx,
                              const float* bias0, const float* bias1, const float* w_hh0, const float* w_hh1, float* h_all0,
                              float* h_all1, float* c_all0, float* c_all1, int32_t block, void* workspace,
                              size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && gx_stash0 && w_hh0 && h_all0 && c_all0 && workspace && n_gx >= 0 && n_gx <= T &&
                     (n_gx == T || bias0), "s2vt_lstm_seq_fwd_persist: bad arguments");
    S2VT_REQUIRE(workspace_bytes >= s2vt_lstm_persist_workspace_bytes(), "s2vt_lstm_seq_fwd_persist: workspace too small");
    S2VT_REQUIRE(lstm_seq_fwd_f32_persist_supported(B, H), "s2vt_lstm_seq_fwd_persist: shape not supported");
    const bool two = gx_stash1 != nullptr;
    S2VT_REQUIRE(!two || (w_hh1 && h_all1 && c_all1 && (n_gx == T || bias1)), "s2vt_lstm_seq_fwd_persist: second layer incomplete");
    int* err = reinterpret_cast<int*>(workspace);
    unsigned int* sa = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(workspace) + 256);
    unsigned int* sb = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(workspace) + 256 + lstm_persist_sync_bytes());
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = fill_zero(st, err, 256))) return rc;
    const int blk = block > 0 ? block : T;
    for (int t0 = 0; t0 < T; t0 += blk) {
        const int t1 = (t0 + blk < T) ? t0 + blk : T;
        ProfScope ps(st, K_STEP_FWD, (two ? 2 : 1) * (t1 - t0));
        const SeqFwdF32Args a0 = persist_fwd_f32_args(t0, t1, B, H, gx_stash0, n_gx, bias0, w_hh0, h_all0, c_all0, sa, err);
        SeqFwdF32Args a1;
        if (two) a1 = persist_fwd_f32_args(t0, t1, B, H, gx_stash1, n_gx, bias1, w_hh1, h_all1, c_all1, sb, err);
        if ((rc = lstm_seq_fwd_f32_persist2(st, a0, two ? &a1 : nullptr))) return rc;
    }
    return 0;
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
// w_hh_t0/1 [H,4H] and dc0/1 [B,H]: scratch provided by the caller
int s2vt_lstm_seq_bwd_persist(int32_t T, int32_t B, int32_t H, const float* w_hh0, const float* w_hh1, const float* dh_out0,
                              const float* dh_out1, int32_t dh_first, const float* c_all0, const float* c_all1, float* stash_dg0,
                              float* stash_dg1, float* w_hh_t0, float* w_hh_t1, float* dc0, float* dc1, int32_t block,
                              void* workspace, size_t workspace_bytes, void* stream) {
    S2VT_REQUIRE(T > 0 && B > 0 && H > 0 && w_hh0 && c_all0 && stash_dg0 && w_hh_t0 && dc0 && workspace && dh_first >= 0,
                 "s2vt_lstm_seq_bwd_persist: bad arguments");
    S2VT_REQUIRE(workspace_bytes >= s2vt_lstm_persist_workspace_bytes(), "s2vt_lstm_seq_bwd_persist: workspace too small");
    S2VT_REQUIRE(lstm_seq_bwd_f32_persist_supported(B, H), "s2vt_lstm_seq_bwd_persist: shape not supported");
    const bool two = stash_dg1 != nullptr;
    S2VT_REQUIRE(!two || (w_hh1 && c_all1 && w_hh_t1 && dc1), "s2vt_lstm_seq_bwd_persist: second layer incomplete");
    int* err = reinterpret_cast<int*>(workspace);
    unsigned int* sa = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(workspace) + 256);
    unsigned int* sb = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(workspace) + 256 + lstm_persist_sync_bytes());
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = fill_zero(st, err, 256))) return rc;
    if ((rc = transpose_f32(st, w_hh0, 4 * H, H, w_hh_t0))) return rc;
    if (two && (rc = transpose_f32(st, w_hh1, 4 * H, H, w_hh_t1))) return rc;
    const int blk = block > 0 ? block : T;
    for (int t1 = T; t1 > 0; t1 -= blk) {
        const int t0 = (t1 - blk > 0) ? t1 - blk : 0;
        ProfScope ps(st, K_STEP_BWD, (two ? 2 : 1) * (t1 - t0));
        const SeqBwdF32Args a0 = persist_bwd_f32_args(T, t0, t1, B, H, w_hh_t0, dh_out0, dh_first, c_all0, stash_dg0, dc0, sa, err);
        SeqBwdF32Args a1;
        if (two) a1 = persist_bwd_f32_args(T, t0, t1, B, H, w_hh_t1, dh_out1, dh_first, c_all1, stash_dg1, dc1, sb, err);
        if ((rc = lstm_seq_bwd_f32_persist2(st, a0, two ? &a1 : nullptr))) return rc;
    }
    return 0;
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

int s2vt_set_gemm_mode(int32_t mode) {
    const int prev = gemm_mode();
    if (mode >= 0) g_gemm_mode = norm_mode(mode);         // negative: query only
    return prev;
}

int s2vt_gemm_tune(int32_t nplanes, int32_t tile_rows, int32_t nsplit) {
    S2VT_REQUIRE(nplanes == 1 || nplanes == 3, "s2vt_gemm_tune: nplanes must be 1 (bf16 kernel) or 3 (split-precision kernel)");
    if (nplanes == 1) gemm_b1_tune(tile_rows, nsplit);
    else gemm_x3_tune(tile_rows, nsplit);
    return 0;
}

int s2vt_pipeline_overlaps(void) { return g_side_overlaps; }

int s2vt_set_graph_mode(int32_t on) {
    const int prev = graph_on() || g_graph_mode == 1 ? 1 : 0;
    if (on >= 0) g_graph_mode = on ? 1 : 0;
    return prev;
}
int s2vt_graph_stats(int64_t* captures, int64_t* replays) {
    if (captures) *captures = (int64_t)g_graph_captures;
    if (replays) *replays = (int64_t)g_graph_replays;
    return 0;
}

int s2vt_test_occupy_cus(int32_t workgroups, int32_t lds_bytes, int64_t microseconds, void* stream) {
    return occupy_cus((hipStream_t)stream, workgroups, lds_bytes, microseconds);
}

int s2vt_set_pipeline_block(int32_t steps) {
    const int prev = pipe_block();
    g_pipe_block = steps < 0 ? 0 : steps;
    return prev;
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

// Wall-clock time during which AT LEAST ONE bracket of `kind` was open: the union of the recorded intervals on a common
// time axis (the first bracket's start).  Brackets of one kind on the two lanes overlap (e.g. the weight-gradient GEMMs of
// the two layers at the end of the backward share the chip: each launch then lasts about twice as long as alone), so the
// SUM s2vt_prof_read returns counts that time twice; throughput figures must be priced with this one.
int s2vt_prof_read_busy(int32_t kind, double* busy_ms) {
    S2VT_REQUIRE(kind >= 0 && kind < K_NKINDS && busy_ms, "s2vt_prof_read_busy: bad arguments");
    *busy_ms = 0.0;
    if (g_recs.empty()) return 0;
    const hipEvent_t ref = g_recs.front().a;
    S2VT_HIP(hipEventSynchronize(ref));
    std::vector<std::pair<double, double>> iv;
    for (auto& r : g_recs) {
        if (r.kind != kind) continue;
        S2VT_HIP(hipEventSynchronize(r.b));
        float s = 0.f, e = 0.f;
        if (r.a != ref) S2VT_HIP(hipEventElapsedTime(&s, ref, r.a));
        S2VT_HIP(hipEventElapsedTime(&e, ref, r.b));
        if (e > s) iv.emplace_back((double)s, (double)e);
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0.0, hi = -1e300;
    for (auto& x : iv) {
        if (x.first > hi) { busy += x.second - x.first; hi = x.second; }
        else if (x.second > hi) { busy += x.second - hi; hi = x.second; }
    }
    *busy_ms = busy;
    return 0;
}

}  // extern "C"
