// Run-time services of libs2vt_hip.so behind include/s2vt_hip.h: error text, asynchronous device-side errors, live kernel timing,
// launch-sequence capture (hipGraph), the side stream of the two-lane schedules, and the typed views of the option table.
#include "api_internal.h"

namespace s2vt {

static thread_local char g_err[512] = "";
const char* last_error_text() { return g_err; }
#ifdef S2VT_EXPERIMENT_STAMPS
unsigned long long* g_xstamps = nullptr;     // timing experiments only (experiment.h)
int g_xstamp_block = 0;
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
// records 0..2: forward / decode, backward, loss (one call per step each: a record still pending when its kind posts again is a whole
// step old).  Records 3..10: a ring for callers that post many times per search (the beam search's depth steps: kind 3) - a post
// takes the next slot and only ever waits for the record posted eight depth steps earlier, i.e. never in practice.
constexpr int kAsyncKinds = 3, kAsyncRing = 8, kAsyncRecords = kAsyncKinds + kAsyncRing;
static ErrRecord g_async[kAsyncRecords] = {};
static unsigned g_async_ring_next = 0;
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
int poll_async_error(bool wait) {
    std::lock_guard<std::mutex> lock(g_async_mutex);
    int first = 0;
    for (int k = 0; k < kAsyncRecords && !first; ++k) first = read_record(g_async[k], wait);
    if (first) {               // one bad batch flags the forward's AND the loss's record: it is reported once - the records still
        char keep[512];        // pending are awaited and dropped with it (the caller is about to raise; the wait costs nothing then)
        snprintf(keep, sizeof(keep), "%s", s2vt_last_error());
        for (int k = 0; k < kAsyncRecords; ++k) (void)read_record(g_async[k], true);
        set_error("%s", keep);
    }
    return first;
}
int post_async_error(hipStream_t st, const int* dev_flags, int kind) {
    std::lock_guard<std::mutex> lock(g_async_mutex);
    ErrRecord& r = g_async[kind < kAsyncKinds ? kind : kAsyncKinds + (int)(g_async_ring_next++ % kAsyncRing)];
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
int device_flags(int** out) {
    static int* flags_of[64] = {};
    int dev = 0;
    S2VT_HIP(hipGetDevice(&dev));
    S2VT_REQUIRE(dev >= 0 && dev < 64, "device index %d", dev);
    if (!flags_of[dev]) S2VT_HIP(hipMalloc(reinterpret_cast<void**>(&flags_of[dev]), 4 * sizeof(int)));
    *out = flags_of[dev];
    return 0;
}

// ------------------------------------------------------------------ live kernel timing
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;
bool prof_on() { return g_prof; }
ProfScope::ProfScope(hipStream_t stream, int kind, int64_t launches) : s(stream), on(g_prof) {
    if (!on) return;
    if (!g_free.empty()) {
        r.a = g_free.back().first; r.b = g_free.back().second; g_free.pop_back();
    } else {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
    }
    r.kind = kind; r.launches = launches;
    (void)hipEventRecord(r.a, s);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    g_recs.push_back(r);
}

// ------------------------------------------------------------------ launch-sequence capture (hipGraph)
// s2vt_set_graph_mode(1) / S2VT_GRAPH=1: the launch sequence of a whole-path call of the plane drivers (s2vt_train_forward /
// s2vt_train_backward at B % 64 == 0: ~350 launches on two streams each) is captured ONCE per distinct argument set - every
// pointer, the dims, the modes and the stream are the key - and replayed with one hipGraphLaunch afterwards.  The first
// sighting of a key runs eagerly (lazy initialisations: side-stream calibration, occupancy queries, event pool), the second
// is captured, later ones replay.  A training loop presents the same pointers every step once torch's caching allocator
// has settled (parameters, the flat gradient buffer and the batch ring are fixed; workspace and logits come back at the same
// addresses); a key that never repeats simply stays eager.  Not used while live timing is on (the event brackets are not
// capturable).  At most 8 executables are kept (least recently used goes).
static thread_local bool g_capturing = false;      // (the enqueue callback runs on the capturing thread)
struct GraphEntry { hipGraphExec_t exec; unsigned long long last_use; int seen; };
static std::map<std::vector<uint64_t>, GraphEntry> g_graphs;
static std::mutex g_graph_mutex;
static unsigned long long g_graph_tick = 0, g_graph_replays = 0, g_graph_captures = 0;
bool graph_on() { return option(O_GRAPH) == 1 && !g_prof; }
bool graph_capturing() { return g_capturing; }
int run_graphed(hipStream_t st, const std::vector<uint64_t>& key, const std::function<int(hipStream_t)>& enqueue, bool* graphed) {
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

// split-K scratch of the driver that is running (set by the whole-path entry points from their workspace)
thread_local float* g_gws = nullptr;
thread_local size_t g_gws_floats = 0;

int gemm(hipStream_t st, bool ak, bool bk, int M, int N, int K, const float* A, int64_t lda, RowMap am,
                const float* B, int64_t ldb, RowMap bm, float* C, int64_t ldc, RowMap cm, const float* bias,
                bool acc) {
    ProfScope ps(st, K_GEMM, 1);
    return gemm_f32(st, ak, bk, M, N, K, A, lda, am, B, ldb, bm, C, ldc, cm, bias, acc, g_gws, g_gws_floats);
}

// scratch floats for split-K slabs: up to 4 slices of the largest small-grid GEMM output of the path
size_t gemm_ws_floats(const s2vt_dims& d) {
    const size_t B = d.B, L = d.L, F = d.F, H = d.H, E = d.E, T = 2 * L - 1;
    size_t m = T * B * H;                       // dh1 / x1-like activations
    if (4 * H * (E + H) > m) m = 4 * H * (E + H);
    if (H * F > m) m = H * F;
    if (L * B * F / 4 > m) m = L * B * F / 4;   // dfeats (rarely split)
    return 4 * m;
}

// ------------------------------------------------------------------ two-lane execution
// The two LSTM layers are independent except through h1: word_rnn step t needs vid_rnn step t only.  A single
// timestep kernel cannot fill the chip's latency (launch + prologue + epilogue ~4 us of a ~12 us step), so the
// layers run as a software pipeline on TWO streams: while lane A (the caller's stream) runs vid_rnn block k+1,
// lane B runs the batched input GEMM and the word_rnn steps of block k (backward: mirrored).  The step kernels
// are sized (69.6 KB LDS) so that one workgroup of each lane fits a CU.  Events order the hand-offs; nothing is
// allocated per call (stream + events are created once per process).
static hipStream_t g_side = nullptr;
static std::vector<hipEvent_t> g_events;
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
static int g_side_overlaps = -1;
int side_stream_overlaps() { return g_side_overlaps; }    // 1: verified concurrent with the first caller stream, 0: no candidate overlapped
int side_stream(hipStream_t caller, hipStream_t* out) {
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
int get_event(size_t i, hipEvent_t* out) {
    while (g_events.size() <= i) {
        hipEvent_t e;
        S2VT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g_events.push_back(e);
    }
    *out = g_events[i];
    return 0;
}
// `to` waits for everything enqueued so far on `from`
int handoff(hipStream_t from, hipStream_t to, size_t ev_index) {
    if (from == to) return 0;
    hipEvent_t e;
    int rc = get_event(ev_index, &e);
    if (rc) return rc;
    S2VT_HIP(hipEventRecord(e, from));
    S2VT_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
}

int lgemm(const Lane& ln, bool ak, bool bk, int M, int N, int K, const float* A, int64_t lda, RowMap am,
                 const float* B, int64_t ldb, RowMap bm, float* C, int64_t ldc, RowMap cm, const float* bias, bool acc) {
    ProfScope ps(ln.s, K_GEMM, 1);
    return gemm_f32(ln.s, ak, bk, M, N, K, A, lda, am, B, ldb, bm, C, ldc, cm, bias, acc, ln.gws, ln.gws_floats);
}

}  // namespace s2vt

using namespace s2vt;

extern "C" {

int s2vt_abi_version(void) { return S2VT_ABI_VERSION; }
const char* s2vt_last_error(void) { return last_error_text(); }
int s2vt_check_async_error(int32_t wait) { return poll_async_error(wait != 0); }

int s2vt_set_gemm_mode(int32_t mode) {
    return option_set(O_GEMM_MODE, mode);                 // negative: query only
}

int s2vt_gemm_tune(int32_t nplanes, int32_t tile_rows, int32_t nsplit) {
    S2VT_REQUIRE(nplanes == 1 || nplanes == 3, "s2vt_gemm_tune: nplanes must be 1 (bf16 kernel) or 3 (split-precision kernel)");
    if (nplanes == 1) gemm_b1_tune(tile_rows, nsplit);
    else gemm_x3_tune(tile_rows, nsplit);
    return 0;
}

int s2vt_pipeline_overlaps(void) { return g_side_overlaps; }

int s2vt_set_graph_mode(int32_t on) {
    return option_set(O_GRAPH, on < 0 ? -1 : (on ? 1 : 0));
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
    return option_set(O_PIPE_BLOCK, steps < 0 ? 0 : steps);
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
