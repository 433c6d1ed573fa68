// The reference's beam-search queue bookkeeping (S2VTModel.py:186-238) ON THE DEVICE, for all samples of a batch at once.
//
// What the reference does per sample and depth: pop up to beam_width entries from a heapq of (-score, node)
// tuples, CLEAR the queue (:194), re-insert a popped entry that ends in <eos> unchanged (:200-202), expand every other one
// into its 20 most probable next tokens, pushed in ascending token order with score = last log-prob / len**0.7 (:216-223),
// stop the sample when the queue holds <= beam_width entries (:227-228); at the end the best entry is popped and back-traced
// (:231-236).  Because the queue is emptied after every pop phase, the heap of a depth is exactly "this depth's pushes, in push
// order" - and WHICH entries come out first when scores tie depends on the binary heap's internal layout.  So the heap is not
// replaced by a sort: every sample's pushes are replayed into a real binary heap with heapq's own _siftdown / _siftup (strict
// `<` on the key only - BeamSearchNode.__lt__ says "not less" for equal scores, S2VTModel.py:268-274), one lane per sample, the
// heap in LDS.  One call = push(depth d-1) + record the best entry + termination test + pop(depth d): nothing happens between a
// push and the next pop, so the heap never leaves LDS.  Nodes are materialised when popped (<= beam_width per depth) into a
// per-sample table (token, parent) that the back-trace walks.
//
// Rows of the batched device step (s2vt_beam_step) are FIXED: row r = b * beam_width + slot, so no compaction and no host
// round trip per depth: slots that hold nothing to expand carry token 0 / state row 0 and their results are ignored.
#include <math.h>

#include <vector>

#include "common.h"
#include "kernels.h"

namespace s2vt {

constexpr int BQ_FAN = 20;          // topk(20) of the reference (S2VTModel.py:216)
constexpr int BQ_FIN = 31;          // fan-out code of a re-inserted finished entry
constexpr int BQ_MAXBW = 8;

struct BeamQ {
    int B, bw, NC, NN, sos, eos, depth;        // depth: 1 = init + first pop; > 1 = push + pop; 0 = final push only
    int* n_nodes; int* node_tok; int* node_prev;            // [B], [B][NN], [B][NN]
    int* done; int* done_count; int* best_tok; int* best_node;
    float* beam_key; int* beam_nid; int* beam_len; int* beam_flag;      // [B][bw]: 0 nothing, 1 expand, 2 finished (<eos>)
    const int* top_ix; const float* top_lp;                  // [B*bw][20] of the depth just stepped
    const float* pow07;                                      // fp32(len ** 0.7), as the reference evaluates it
    int* row_b; int* row_state; int* row_tok;               // [B*bw] for the next s2vt_beam_step
};

// heapq._siftdown(heap, 0, pos) with `<` on the key
__device__ __forceinline__ void bq_siftdown(float* hk, int* hc, int S, int lane, int start, int pos) {
    const float nk = hk[pos * S + lane];
    const int nc = hc[pos * S + lane];
    while (pos > start) {
        const int parent = (pos - 1) >> 1;
        const float pk = hk[parent * S + lane];
        if (nk < pk) {
            hk[pos * S + lane] = pk;
            hc[pos * S + lane] = hc[parent * S + lane];
            pos = parent;
            continue;
        }
        break;
    }
    hk[pos * S + lane] = nk;
    hc[pos * S + lane] = nc;
}
__device__ __forceinline__ void bq_push(float* hk, int* hc, int S, int lane, int& n, float key, int code) {
    hk[n * S + lane] = key;
    hc[n * S + lane] = code;
    ++n;
    bq_siftdown(hk, hc, S, lane, 0, n - 1);
}
// heapq.heappop
__device__ __forceinline__ void bq_pop(float* hk, int* hc, int S, int lane, int& n, float& key, int& code) {
    --n;
    const float lk = hk[n * S + lane];
    const int lc = hc[n * S + lane];
    if (n == 0) { key = lk; code = lc; return; }
    key = hk[lane]; code = hc[lane];
    // heap[0] = lastelt; _siftup(heap, 0)
    int pos = 0, child = 1;
    while (child < n) {
        const int right = child + 1;
        if (right < n && !(hk[child * S + lane] < hk[right * S + lane])) child = right;
        hk[pos * S + lane] = hk[child * S + lane];
        hc[pos * S + lane] = hc[child * S + lane];
        pos = child;
        child = 2 * pos + 1;
    }
    hk[pos * S + lane] = lk;
    hc[pos * S + lane] = lc;
    bq_siftdown(hk, hc, S, lane, 0, pos);
}

// One WAVE per sample.  The pushes of a depth are laid out in push order (ck / cc); what the reference's heap then yields is
//   (a) its root after the push phase (the entry :231 would pop) and (b) the first beam_width pops.
// A binary heap pops in ascending key order whatever its layout as long as the keys involved are DISTINCT, so the wave selects the
// smallest remaining key beam_width times (butterfly min + a ballot count of the candidates that hold it); the moment two
// candidates share the smallest remaining key - the only case in which the heap's internal layout decides - lane 0 replays
// heapq's own _siftdown / _siftup over the same push sequence instead (the single-lane replay took 86 us per depth for every
// sample; the selection takes ~5).
constexpr int BQ_MAXC = BQ_MAXBW * BQ_FAN + BQ_MAXBW;     // candidates of one depth
__global__ __launch_bounds__(64) void beam_queue_kernel(BeamQ q) {
    __shared__ float ck[BQ_MAXC], hk[BQ_MAXC], selk[BQ_MAXBW];
    __shared__ int cc[BQ_MAXC], hc[BQ_MAXC], selc[BQ_MAXBW];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int bw = q.bw;
    float* bkey = q.beam_key + b * bw;
    int* bnid = q.beam_nid + b * bw;
    int* blen = q.beam_len + b * bw;
    int* bflag = q.beam_flag + b * bw;
    int* ntok = q.node_tok + (int64_t)b * q.NN;
    int* nprev = q.node_prev + (int64_t)b * q.NN;
    int n = 0;                                 // pushes of this depth (wave-uniform)
    if (q.depth == 1) {        // the root: BeamSearchNode(hidden, None, <sos>, 0, 1), key -0.0 (:186-188)
        if (lane == 0) {
            q.n_nodes[b] = 1;
            ntok[0] = q.sos; nprev[0] = -1;
            q.done[b] = 0;
            q.best_tok[b] = q.sos; q.best_node[b] = -1;
            for (int j = 0; j < bw; ++j) {
                q.row_b[b * bw + j] = b;
                bflag[j] = 0;
            }
            ck[0] = -0.0f; cc[0] = -1;
        }
        n = 1;
    } else {
        if (q.done[b]) return;             // frozen: its beam slots were cleared when it stopped
        // ---- push phase of the depth that was just stepped, in beam (= pop) order (:198-223)
        for (int j = 0; j < bw; ++j) {
            const int flag = bflag[j];
            if (flag == 2) {
                if (lane == 0) { ck[n] = bkey[j]; cc[n] = j * 32 + BQ_FIN; }
                n += 1;
            } else if (flag == 1) {
                if (lane < BQ_FAN) {
                    const float div = q.pow07[blen[j] + 1];
                    ck[n + lane] = -(q.top_lp[(int64_t)(b * bw + j) * BQ_FAN + lane] / div);
                    cc[n + lane] = j * 32 + lane;
                }
                n += BQ_FAN;
            }
        }
    }
    __syncthreads();
    // ---- selection: the root, and (unless the sample stops here or this is the final push) the first beam_width pops
    const bool stops = q.depth != 1 && n <= bw;                        // (:227-228)
    const int m = (q.depth == 0 || stops) ? 1 : (n < bw ? n : bw);
    constexpr int CPL = (BQ_MAXC + 63) / 64;                            // candidates per lane
    float myk[CPL];
    bool live[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int pidx = lane + 64 * c;
        live[c] = pidx < n;
        myk[c] = live[c] ? ck[pidx] : INFINITY;
    }
    bool tie = false;
    for (int r = 0; r < m && !tie; ++r) {
        float mn = INFINITY;
#pragma unroll
        for (int c = 0; c < CPL; ++c) mn = (live[c] && myk[c] < mn) ? myk[c] : mn;
        for (int o = 32; o; o >>= 1) { const float ov = __shfl_xor(mn, o); mn = ov < mn ? ov : mn; }
        int holders = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) holders += __popcll(__ballot(live[c] && myk[c] == mn));
        if (holders != 1) { tie = true; break; }                        // (also: no live candidate compares equal - a NaN key)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if (live[c] && myk[c] == mn) { selk[r] = mn; selc[r] = cc[lane + 64 * c]; live[c] = false; }
    }
    __syncthreads();
    if (lane != 0) return;
    int rootc;
    if (tie) {                              // exact replay of the reference's heap
        int hn = 0;
        for (int i = 0; i < n; ++i) bq_push(hk, hc, 1, 0, hn, ck[i], cc[i]);
        rootc = hc[0];
        if (!(q.depth == 0 || stops))
            for (int j = 0; j < m; ++j) bq_pop(hk, hc, 1, 0, hn, selk[j], selc[j]);
    } else {
        rootc = selc[0];
    }
    if (q.depth != 1) {
        // ---- the entry the reference would pop at the end (:231): the heap's root after this push phase
        const int j = rootc >> 5, f = rootc & 31;
        if (f == BQ_FIN) { q.best_tok[b] = q.eos; q.best_node[b] = nprev[bnid[j]]; }
        else { q.best_tok[b] = q.top_ix[(int64_t)(b * bw + j) * BQ_FAN + f]; q.best_node[b] = bnid[j]; }
        if (stops) {
            q.done[b] = 1;
            atomicAdd(q.done_count, 1);
            for (int jj = 0; jj < bw; ++jj) { bflag[jj] = 0; q.row_state[b * bw + jj] = 0; q.row_tok[b * bw + jj] = 0; }
            return;
        }
    }
    if (q.depth == 0) return;              // final push: nothing is popped any more
    // ---- pop phase: up to beam_width entries, queue cleared (:190-194); a popped candidate becomes a node now
    float pk[BQ_MAXBW];
    int ptok[BQ_MAXBW], pnid[BQ_MAXBW], plen[BQ_MAXBW], prow[BQ_MAXBW];
    int nn = q.n_nodes[b];
    for (int j = 0; j < m; ++j) {
        const float key = selk[j];
        const int code = selc[j];
        pk[j] = key;
        if (code < 0) {                    // the root
            ptok[j] = q.sos; pnid[j] = 0; plen[j] = 1; prow[j] = b;
        } else {
            const int jj = code >> 5, f = code & 31;
            if (f == BQ_FIN) {
                ptok[j] = q.eos; pnid[j] = bnid[jj]; plen[j] = blen[jj]; prow[j] = 0;
            } else {
                ptok[j] = q.top_ix[(int64_t)(b * bw + jj) * BQ_FAN + f];
                plen[j] = blen[jj] + 1;
                prow[j] = b * bw + jj;
                pnid[j] = nn;
                if (nn < q.NN) { ntok[nn] = ptok[j]; nprev[nn] = bnid[jj]; }
                ++nn;
            }
        }
    }
    q.n_nodes[b] = nn < q.NN ? nn : q.NN;
    for (int j = 0; j < bw; ++j) {
        int flag = 0, rs = 0, rt = 0;
        if (j < m) {
            const bool fin = ptok[j] == q.eos && nprev[pnid[j]] >= 0;            // (:198-202)
            flag = fin ? 2 : 1;
            bkey[j] = pk[j]; bnid[j] = pnid[j]; blen[j] = plen[j];
            if (!fin) { rs = prow[j]; rt = ptok[j]; }
        }
        bflag[j] = flag;
        q.row_state[b * bw + j] = rs;
        q.row_tok[b * bw + j] = rt;
    }
}

// back-trace (:231-236): out[b][0..len) = <sos> .. best token
__global__ void beam_queue_result_kernel(BeamQ q, int32_t* out, int32_t* out_len, int cap) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= q.B) return;
    const int* ntok = q.node_tok + (int64_t)b * q.NN;
    const int* nprev = q.node_prev + (int64_t)b * q.NN;
    int len = 1;
    for (int node = q.best_node[b]; node >= 0 && len < cap; node = nprev[node]) ++len;
    int32_t* o = out + (int64_t)b * cap;
    int i = len - 1;
    o[i--] = q.best_tok[b];
    for (int node = q.best_node[b]; node >= 0 && i >= 0; node = nprev[node]) o[i--] = ntok[node];
    out_len[b] = len;
}

static BeamQ carve_beamq(int B, int bw, int max_depth, void* base, size_t* bytes) {
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t n) { off = align_up(off, 256); char* r = base ? p + off : nullptr; off += n; return r; };
    BeamQ q;
    q.B = B; q.bw = bw; q.NC = bw * BQ_FAN; q.NN = 2 + max_depth * bw;
    q.done_count = reinterpret_cast<int*>(take(sizeof(int)));         // (first: s2vt_beam_queue_done_offset() == 0)
    q.n_nodes = reinterpret_cast<int*>(take(sizeof(int) * B));
    q.node_tok = reinterpret_cast<int*>(take(sizeof(int) * (size_t)B * q.NN));
    q.node_prev = reinterpret_cast<int*>(take(sizeof(int) * (size_t)B * q.NN));
    q.done = reinterpret_cast<int*>(take(sizeof(int) * B));
    q.best_tok = reinterpret_cast<int*>(take(sizeof(int) * B));
    q.best_node = reinterpret_cast<int*>(take(sizeof(int) * B));
    q.beam_key = reinterpret_cast<float*>(take(sizeof(float) * (size_t)B * bw));
    q.beam_nid = reinterpret_cast<int*>(take(sizeof(int) * (size_t)B * bw));
    q.beam_len = reinterpret_cast<int*>(take(sizeof(int) * (size_t)B * bw));
    q.beam_flag = reinterpret_cast<int*>(take(sizeof(int) * (size_t)B * bw));
    q.pow07 = reinterpret_cast<float*>(take(sizeof(float) * (size_t)(max_depth + 4)));
    if (bytes) *bytes = align_up(off, 256);
    return q;
}

}  // namespace s2vt

using namespace s2vt;

extern "C" {

size_t s2vt_beam_queue_bytes(int32_t B, int32_t beam_width, int32_t max_depth) {
    if (B <= 0 || beam_width <= 0 || beam_width > BQ_MAXBW || max_depth <= 0) return 0;
    size_t bytes = 0;
    carve_beamq(B, beam_width, max_depth, nullptr, &bytes);
    return bytes;
}

int s2vt_beam_queue_step(int32_t B, int32_t beam_width, int32_t max_depth, int32_t sos_ix, int32_t eos_ix, int32_t depth, void* state,
                         size_t state_bytes, const int32_t* top_ix, const float* top_lp, int32_t* row_b, int32_t* row_state,
                         int32_t* row_tok, void* stream) {
    S2VT_REQUIRE(B > 0 && beam_width > 0 && beam_width <= BQ_MAXBW && max_depth > 0 && depth >= 0 && depth <= max_depth && state &&
                     row_b && row_state && row_tok,
                 "s2vt_beam_queue_step: null/invalid argument (beam_width <= %d)", BQ_MAXBW);
    S2VT_REQUIRE(depth == 1 || (top_ix && top_lp), "s2vt_beam_queue_step: the step's top-20 arrays are needed from depth 2 on");
    size_t need = 0;
    BeamQ q = carve_beamq(B, beam_width, max_depth, state, &need);
    S2VT_REQUIRE(state_bytes >= need, "s2vt_beam_queue_step: state %zu < %zu bytes", state_bytes, need);
    q.sos = sos_ix; q.eos = eos_ix; q.depth = depth;
    q.top_ix = top_ix; q.top_lp = top_lp;
    q.row_b = row_b; q.row_state = row_state; q.row_tok = row_tok;
    hipStream_t st = (hipStream_t)stream;
    if (depth == 1) {
        // score divisor len ** 0.7 exactly as the reference evaluates it: Python float pow (libm double), then fp32 (:262-266)
        // The table lives in pinned host memory for the life of the process (it depends on nothing but the length), so the copy
        // is asynchronous: no host synchronisation at the start of a search, and the call can be stream-captured.
        static float* tab = nullptr;
        static int tab_len = 0;
        if (tab_len < max_depth + 4) {
            float* grown = nullptr;
            const int len = (max_depth + 4 + 1023) / 1024 * 1024;
            S2VT_HIP(hipHostMalloc(reinterpret_cast<void**>(&grown), (size_t)len * sizeof(float), hipHostMallocDefault));
            for (int l = 0; l < len; ++l) grown[l] = l > 0 ? (float)pow((double)l, 0.7) : 1.0f;
            tab = grown;            // (an older, shorter table stays allocated: copies from it may still be in flight)
            tab_len = len;
        }
        S2VT_HIP(hipMemcpyAsync(const_cast<float*>(q.pow07), tab, (size_t)(max_depth + 4) * sizeof(float), hipMemcpyHostToDevice, st));
        S2VT_HIP(hipMemsetAsync(q.done_count, 0, sizeof(int), st));
    }
    hipLaunchKernelGGL(beam_queue_kernel, dim3(B), dim3(64), 0, st, q);
    S2VT_LAUNCH_CHECK("beam_queue_kernel");
    return 0;
}

int s2vt_beam_queue_result(int32_t B, int32_t beam_width, int32_t max_depth, void* state, size_t state_bytes, int32_t* out_tokens,
                           int32_t out_cap, int32_t* out_len, void* stream) {
    S2VT_REQUIRE(B > 0 && beam_width > 0 && beam_width <= BQ_MAXBW && max_depth > 0 && state && out_tokens && out_len && out_cap >= max_depth + 2,
                 "s2vt_beam_queue_result: null/invalid argument (out_cap >= max_depth + 2)");
    size_t need = 0;
    BeamQ q = carve_beamq(B, beam_width, max_depth, state, &need);
    S2VT_REQUIRE(state_bytes >= need, "s2vt_beam_queue_result: state %zu < %zu bytes", state_bytes, need);
    hipLaunchKernelGGL(beam_queue_result_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, q, out_tokens, out_len, out_cap);
    S2VT_LAUNCH_CHECK("beam_queue_result_kernel");
    return 0;
}

}  // extern "C"
