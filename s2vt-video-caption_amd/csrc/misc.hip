// Small memory-bound helpers around the S2VT contractions: bias sums, the W_hh transpose used by
// BPTT, deterministic column sums (bias gradients), caption index conversion, the embedding
// scatter-add (autograd of S2VTModel.py:71) and unpacking of the decode argmax words.
#include <math.h>
#include "common.h"
#include "kernels.h"

namespace s2vt {

__global__ void add_vectors_kernel(const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
__global__ void mul_vectors_kernel(const float* a, const float* b, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
// out = a (.) b, element-wise (out may alias a): the out_drop mask of S2VTModel.py:79 and its gradient
int mul_vectors(hipStream_t s, const float* a, const float* b, float* out, int64_t n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mul_vectors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, out, n);
    S2VT_LAUNCH_CHECK("mul_vectors_kernel");
    return 0;
}

int add_vectors(hipStream_t s, const float* a, const float* b, float* out, int n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_vectors_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, a, b, out, n);
    S2VT_LAUNCH_CHECK("add_vectors_kernel");
    return 0;
}

// out[c][r] = in[r][c]; 64x64 tiles through LDS (padded), coalesced on both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, int rows, int cols, float* out) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(int64_t)c * rows + r] = tile[tx][i];
    }
}
int transpose_f32(hipStream_t s, const float* in, int rows, int cols, float* out) {
    dim3 grid(cdiv(cols, 64), cdiv(rows, 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, rows, cols, out);
    S2VT_LAUNCH_CHECK("transpose_kernel");
    return 0;
}

// x[i] *= *alpha (alpha on the device: the mantissa of the mean-CE scale, CeGradArgs::alpha_out)
__global__ __launch_bounds__(256) void scale_by_kernel(float* x, int64_t n4, int64_t n, const float* alpha) {
    const float a = *alpha;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        v[0] *= a; v[1] *= a; v[2] *= a; v[3] *= a;
        reinterpret_cast<f32x4*>(x)[i] = v;
    } else if (i == n4) {
        for (int64_t j = 4 * n4; j < n; ++j) x[j] *= a;
    }
}
int scale_by_device_scalar(hipStream_t s, float* x, int64_t n, const float* alpha) {
    if (n <= 0) return 0;
    S2VT_REQUIRE(x && alpha && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "scale_by_device_scalar: bad arguments");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(scale_by_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, s, x, n4, n, alpha);
    S2VT_LAUNCH_CHECK("scale_by_kernel");
    return 0;
}

// Adam step (train.py:89-93 -> torch.optim.Adam defaults; train.py:126 optimizer.step()) over ONE flat fp32 buffer holding all
// parameters, with flat gradient / first / second moment buffers of the same length: one memory-bound launch (28 bytes per
// parameter) instead of the framework's multi-tensor launches.  torch.optim.Adam's arithmetic, operation for operation:
//   m = m + (g - m) (1 - b1)              (Tensor.lerp_)
//   v = b2 v + (1 - b2) g g
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (weight_decay 0, amsgrad off, maximize off: the reference's configuration).  The bias corrections arrive as host doubles
// rounded to fp32 once, and so do 1 - b1 and 1 - b2 (torch forms them from the Python doubles), which is why the hyper-parameters
// cross the C ABI as doubles.
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n4, int64_t n, float w1, float beta2, float w2,
                                                        float step_size, float bc2_sqrt, float eps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        mm = mm + (gg - mm) * w1;
        vv = beta2 * vv + w2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp - step_size * (mm / denom);
    };
    if (i < n4) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pp = pv[j], mm = mv[j], v2 = vv[j];
            upd(pp, gv[j], mm, v2);
            pv[j] = pp, mv[j] = mm, vv[j] = v2;
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    } else if (i == n4) {
        for (int64_t j = 4 * n4; j < n; ++j) upd(p[j], g[j], m[j], v[j]);
    }
}
int adam_flat(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
              int64_t step) {
    if (n <= 0) return 0;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    S2VT_REQUIRE(p && g && m && v && al(p) && al(g) && al(m) && al(v) && step >= 1, "adam_flat: null / unaligned buffers or step < 1");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, s, p, g, m, v, n4, n, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps);
    S2VT_LAUNCH_CHECK("adam_flat_kernel");
    return 0;
}

// Deterministic column sum in two passes: partial[chunk][col] over CS_ROWS-row chunks, then a fixed-
// order sum over chunks (bias gradients must not depend on atomics' arrival order).
constexpr int CS_ROWS = 64;   // = the row-tile of split_dual_kernel, which can produce the same partials
size_t colsum_partial_floats(int64_t rows, int cols) { return (size_t)((rows + CS_ROWS - 1) / CS_ROWS) * cols; }

__global__ void colsum_partial_kernel(const float* x, int64_t rows, int cols, int64_t ld, float* partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    const int64_t r1 = (r0 + CS_ROWS < rows) ? r0 + CS_ROWS : rows;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += x[r * ld + c];
    partial[(int64_t)blockIdx.y * cols + c] = s;
}
// 16 columns per workgroup; the chunk loop is dealt over 16 thread rows with 4 independent accumulators each (the sum is a
// dependent chain of loads otherwise: 159 chunks took 80 us in one chain; 4 rows x 64 columns left 63 workgroups with 40
// dependent rounds at B = 256 - 27 us per bias gradient), combined in a fixed order -> deterministic.
constexpr int CF_COLS = 16, CF_ROWS = 16;
__global__ __launch_bounds__(CF_COLS * CF_ROWS) void colsum_final_kernel(const float* partial, int nchunks, int cols, float* out,
                                                                         int accumulate) {
    __shared__ float red[CF_ROWS][CF_COLS + 1];
    const int tx = threadIdx.x % CF_COLS, ty = threadIdx.x / CF_COLS;
    const int c = blockIdx.x * CF_COLS + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        const float* q = partial + c;
        int k = ty;
        for (; k + 3 * CF_ROWS < nchunks; k += 4 * CF_ROWS) {
            s0 += q[(int64_t)k * cols];
            s1 += q[(int64_t)(k + CF_ROWS) * cols];
            s2 += q[(int64_t)(k + 2 * CF_ROWS) * cols];
            s3 += q[(int64_t)(k + 3 * CF_ROWS) * cols];
        }
        for (; k < nchunks; k += CF_ROWS) s0 += q[(int64_t)k * cols];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && c < cols) {
        float s = accumulate ? out[c] : 0.f;
#pragma unroll
        for (int r = 0; r < CF_ROWS; ++r) s += red[r][tx];
        out[c] = s;
    }
}
// fixed-order sum of `nchunks` partial rows (written by colsum_partial_kernel or split_dual_kernel)
int colsum_finish(hipStream_t s, const float* partial, int nchunks, int cols, float* out, bool accumulate) {
    if (cols <= 0) return 0;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, CF_COLS)), dim3(CF_COLS * CF_ROWS), 0, s, partial, nchunks, cols, out,
                       accumulate ? 1 : 0);
    S2VT_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

int colsum_f32(hipStream_t s, const float* x, int64_t rows, int cols, int64_t ld, float* partial, float* out,
               bool accumulate) {
    if (cols <= 0) return 0;
    const int nchunks = (int)((rows + CS_ROWS - 1) / CS_ROWS);
    if (nchunks > 0) {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 256), nchunks), dim3(256), 0, s, x, rows, cols, ld,
                           partial);
        S2VT_LAUNCH_CHECK("colsum_partial_kernel");
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, CF_COLS)), dim3(CF_COLS * CF_ROWS), 0, s, partial, nchunks, cols, out,
                       accumulate ? 1 : 0);
    S2VT_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

// targets[b*ld + t] (int64, batch-major) -> out[t*B + b] (int32, time-major); out-of-range ids are
// clamped and flagged (torch's embedding would raise).
__global__ void targets_tm_kernel(const int64_t* targets, int B, int Lm1, int64_t ld, int V, int32_t* out, int* err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Lm1) return;
    const int t = i / B, b = i % B;
    int64_t tok = targets[(int64_t)b * ld + t];
    if (tok < 0 || tok >= V) {
        if (err) atomicExch(err, 1);
        tok = tok < 0 ? 0 : V - 1;
    }
    out[i] = (int32_t)tok;
}
int targets_to_time_major(hipStream_t s, const int64_t* targets, int B, int Lm1, int64_t ld, int V, int32_t* out,
                          int* err_flag) {
    const int n = B * Lm1;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(targets_tm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, targets, B, Lm1, ld, V, out, err_flag);
    S2VT_LAUNCH_CHECK("targets_tm_kernel");
    return 0;
}

// Embedding gradient (autograd of S2VTModel.py:71): d_emb[v, :] = sum over the rows r with tok[r] == v of d_rows[r, :],
// in a FIXED order - no atomics, so the whole training step is bitwise reproducible.
//   emb_count_kernel: how often every token occurs (integer atomics: the counts do not depend on the order).
//   emb_grad_kernel: one workgroup per vocabulary row v.  A token that does not occur (most of a 12000-word vocabulary in
//     any one batch) gets its zeros at once; otherwise the workgroup scans the token list until it has found all count[v]
//     matches (order-preserving compaction by wave ballots); a row with <= EG_CAP matches is summed in ascending r (unused tokens are written as zeros: no separate
//     fill); a row with more (<pad>, <eos>, frequent words: thousands of matches) is only put on the `heavy` list.
//   emb_grad_heavy_kernel: one workgroup per (heavy token, 16 columns): 64 row-lanes sum contiguous slices of the
//     ordered match list, the 64 partials are added in lane order.
constexpr int EG_CAP = 64;
template <int NW>
__device__ __forceinline__ int eg_compact(bool hit, int r, int* list, int n_list, int* wave_cnt, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = n_list, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wave) off += wave_cnt[w];
        total += wave_cnt[w];
    }
    if (hit && list) list[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
    __syncthreads();
    return n_list + total;
}
// (<pad> / <eos> fill most caption rows: counted one atomic per row their thousands of same-address atomics serialise - 0.2 ms
// at B = 256 - and an LDS histogram per workgroup pays for zeroing and flushing V bins: 91 us at B = 64.  Every wave aggregates its 64
// rows first: the lowest lane still uncounted names its token, the lanes holding the same token are counted by one ballot and
// retire - one global atomic per DISTINCT token of a wave, a handful of rounds for caption data)
__global__ __launch_bounds__(256) void emb_count_kernel(const int32_t* tok, int rows, int* count) {
    const int r = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const int mine = (r < rows) ? tok[r] : -1;
    unsigned long long todo = __ballot(mine >= 0);
    while (todo) {                                            // (wave-uniform)
        const int leader = __ffsll((long long)todo) - 1;
        const int t = __shfl(mine, leader);
        const unsigned long long same = __ballot(mine == t);
        if (lane == leader) atomicAdd(&count[t], __popcll(same));
        todo &= ~same;
    }
}
__global__ __launch_bounds__(256) void emb_grad_kernel(const float* d_rows, int rows, int E, const int32_t* tok, float* d_emb,
                                                       int* heavy, int* n_heavy, const int* count) {
    __shared__ int list[EG_CAP + 256];
    __shared__ int wave_cnt[4];
    const int v = blockIdx.x, tid = threadIdx.x;
    const int want = count[v];                                // uniform
    if (want > EG_CAP) {
        if (tid == 0) heavy[atomicAdd(n_heavy, 1)] = v;       // list order does not matter
        return;
    }
    int n = 0;                                                // uniform: matches found so far
    for (int base = 0; base < rows && n < want; base += 256) {
        const int r = base + tid;
        n = eg_compact<4>((r < rows) && (tok[r] == v), r, list, n, wave_cnt, tid);
    }
    float* out = d_emb + (int64_t)v * E;
    for (int c = tid; c < E; c += 256) {
        float sum = 0.f;
        for (int i = 0; i < n; i += 8) {                      // eight loads in flight, added in list order (see the heavy kernel)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = d_rows[(int64_t)list[(i + j < n) ? i + j : n - 1] * E + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (i + j < n) ? v[j] : 0.f;
        }
        out[c] = sum;
    }
}
constexpr int EG_HLIST = 8192;      // matches of a heavy token handled per pass
constexpr int EG_HT = 1024;         // threads: 64 row-lanes x 16 columns (<pad> has ~15000 matches at B = 256: 16 row-lanes
                                    // summed ~940 rows each, one load latency after the other; 64 lanes cut that four-fold)
__global__ __launch_bounds__(EG_HT) void emb_grad_heavy_kernel(const float* d_rows, int rows, int E, const int32_t* tok,
                                                               float* d_emb, const int* heavy, const int* n_heavy) {
    __shared__ int list[EG_HLIST + EG_HT];
    __shared__ int wave_cnt[EG_HT / 64];
    __shared__ float part[EG_HT / 16][17];
    // gridDim.y is a small constant, not the worst-case number of heavy tokens (rows / 64 + 2 = 318 at B = 256: 20 000 workgroups of
    // 1024 threads that found nothing to do were most of this launch's 130 us); a workgroup takes every gridDim.y-th heavy token
    const int nh = *n_heavy, tid = threadIdx.x;
    const int rl = tid >> 4, c = blockIdx.x * 16 + (tid & 15);
    constexpr int NRL = EG_HT / 16;
    for (int hv = blockIdx.y; hv < nh; hv += gridDim.y) {
    const int v = heavy[hv];
    float total = 0.f;
    int base = 0;
    while (base < rows) {
        int n = 0;
        for (; base < rows && n <= EG_HLIST; base += EG_HT) {     // list holds EG_HLIST + EG_HT
            const int r = base + tid;
            n = eg_compact<EG_HT / 64>((r < rows) && (tok[r] == v), r, list, n, wave_cnt, tid);
        }
        // row-lane rl sums the slice [rl*per, (rl+1)*per) of this pass's ordered matches
        const int per = (n + NRL - 1) / NRL, i0 = rl * per, i1 = (i0 + per < n) ? i0 + per : n;
        float sum = 0.f;
        if (c < E) {
            // eight loads in flight (unconditional: an index past the slice re-reads its last row and is not added), added in
            // list order; a loop with the bound test in front of every load issues them one latency after the other
            for (int i = i0; i < i1; i += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ii = (i + j < i1) ? i + j : i1 - 1;
                    v[j] = d_rows[(int64_t)list[ii] * E + c];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += (i + j < i1) ? v[j] : 0.f;
            }
        }
        part[rl][tid & 15] = sum;
        __syncthreads();
        if (rl == 0) {
            float s = part[0][tid & 15];
#pragma unroll
            for (int k = 1; k < NRL; ++k) s += part[k][tid & 15];
            total += s;
        }
        __syncthreads();
    }
    if (rl == 0 && c < E) d_emb[(int64_t)v * E + c] = total;
    }
}
size_t embedding_grad_ws_ints(int64_t rows, int V) { return (size_t)(rows / EG_CAP + 2) + 1 + (size_t)V; }
// ws: embedding_grad_ws_ints(rows, V) ints of scratch (heavy-token list, its counter, per-token counts)
int embedding_grad(hipStream_t s, const float* d_rows, int64_t rows, int E, const int32_t* tok, int V, float* d_emb, int* ws) {
    if (V <= 0 || E <= 0) return 0;
    const int max_heavy = (int)(rows / EG_CAP + 2);           // each heavy token owns > EG_CAP of the `rows` rows
    int* n_heavy = ws + max_heavy;
    int* count = n_heavy + 1;
    S2VT_HIP(hipMemsetAsync(n_heavy, 0, sizeof(int) * (size_t)(1 + V), s));
    if (rows > 0) {
        hipLaunchKernelGGL(emb_count_kernel, dim3(cdiv((int)rows, 256)), dim3(256), 0, s, tok, (int)rows, count);
    }
    S2VT_LAUNCH_CHECK("emb_count_kernel");
    hipLaunchKernelGGL(emb_grad_kernel, dim3((unsigned)V), dim3(256), 0, s, d_rows, (int)rows, E, tok, d_emb, ws, n_heavy, count);
    S2VT_LAUNCH_CHECK("emb_grad_kernel");
    hipLaunchKernelGGL(emb_grad_heavy_kernel, dim3(cdiv(E, 16), max_heavy < 4 ? max_heavy : 4), dim3(EG_HT), 0, s, d_rows, (int)rows, E,
                       tok, d_emb, ws, n_heavy);
    S2VT_LAUNCH_CHECK("emb_grad_heavy_kernel");
    return 0;
}

// out[r, :] = src[idx[r], :]  (embedding rows in time-major caption order, S2VTModel.py:71)
__global__ void gather_rows_kernel(const float* src, int64_t ld, const int32_t* idx, int64_t rows, int cols, float* out) {
    const int64_t r = blockIdx.y;
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (r >= rows || c >= cols) return;
    const float* s = src + (int64_t)idx[r] * ld;
    float* o = out + r * cols;
    if ((cols & 3) == 0 && (ld & 3) == 0) {
        *reinterpret_cast<f32x4*>(o + c) = *reinterpret_cast<const f32x4*>(s + c);
    } else {
        for (int j = 0; j < 4 && c + j < cols; ++j) o[c + j] = s[c + j];
    }
}
int gather_rows_f32(hipStream_t s, const float* src, int64_t ld, const int32_t* idx, int64_t rows, int cols, float* out) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(cdiv(cols, 4), 256), (unsigned)rows), dim3(256), 0, s, src, ld, idx,
                       rows, cols, out);
    S2VT_LAUNCH_CHECK("gather_rows_kernel");
    return 0;
}

// packed[step][b] -> ids[b][step] (int64, the reference's output layout S2VTModel.py:108-110)
__global__ void unpack_tokens_kernel(const unsigned long long* packed, int steps, int B, int64_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= steps * B) return;
    const int st = i / B, b = i % B;
    out[(int64_t)b * steps + st] = (int64_t)(0xFFFFFFFFu - (uint32_t)(packed[i] & 0xFFFFFFFFull));
}
int unpack_tokens(hipStream_t s, const unsigned long long* packed, int steps, int B, int64_t* out_ids) {
    const int n = steps * B;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(unpack_tokens_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, packed, steps, B, out_ids);
    S2VT_LAUNCH_CHECK("unpack_tokens_kernel");
    return 0;
}

// p[r][col0:col1) = 0 for every row (zero padding of bf16 row images whose valid columns are written elsewhere)
__global__ void zero_pad_cols_kernel(unsigned short* p, int64_t rows, int64_t ld, int col0, int ncols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ncols) return;
    p[(i / ncols) * ld + col0 + (i % ncols)] = 0;
}
int zero_pad_cols_u16(hipStream_t s, unsigned short* p, int64_t rows, int64_t ld, int col0, int col1) {
    const int ncols = col1 - col0;
    if (rows <= 0 || ncols <= 0) return 0;
    const int64_t n = rows * ncols;
    hipLaunchKernelGGL(zero_pad_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, rows, ld, col0, ncols);
    S2VT_LAUNCH_CHECK("zero_pad_cols_kernel");
    return 0;
}

// Test support: `workgroups` workgroups that each hold `lds_bytes` of LDS and spin for `microseconds` of wall clock.  The
// co-residency tests of the persistent recurrence use it as the foreign kernel (an RCCL kernel on a communication stream,
// another tenant of the device) that takes LDS away from under a launch whose workgroups wait for each other.
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks_100mhz, int lds_bytes) {
    extern __shared__ unsigned char occ_lds[];
    if (lds_bytes > 0) occ_lds[(threadIdx.x * 64) % lds_bytes] = (unsigned char)threadIdx.x;      // the allocation is used
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(8);
}
int occupy_cus(hipStream_t s, int workgroups, int lds_bytes, long long microseconds) {
    S2VT_REQUIRE(workgroups > 0 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && microseconds >= 0 && microseconds <= 5000000,
                 "s2vt_test_occupy_cus: bad arguments");
    S2VT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(64), lds_bytes, s, (unsigned long long)microseconds * 100ull, lds_bytes);
    S2VT_LAUNCH_CHECK("occupy_kernel");
    return 0;
}

int fill_zero(hipStream_t s, void* p, size_t bytes) {
    if (bytes == 0) return 0;
    S2VT_HIP(hipMemsetAsync(p, 0, bytes, s));
    return 0;
}

}  // namespace s2vt
