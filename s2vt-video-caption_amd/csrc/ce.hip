// Mean cross-entropy over vocabulary logits (utils.py:11,22: nn.CrossEntropyLoss() with the default
// 'mean' reduction over all B*(L-1) rows against target[:, 1:]) and its gradient.
// One 256-thread workgroup per logits row: 16-B vector loads, wave shuffles + LDS for the row max and
// the exp-sum; per-row losses are then summed in a fixed order by a single workgroup (deterministic).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace s2vt {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// target row r = b*Lm1 + j  ->  target[b*ldt + j + 1]
__device__ __forceinline__ int64_t row_target(const int64_t* target, int64_t r, int Lm1, int64_t ldt) {
    const int64_t b = r / Lm1, j = r % Lm1;
    return target[b * ldt + j + 1];
}

__global__ __launch_bounds__(256) void ce_row_kernel(const float* logits, int V, const int64_t* target, int Lm1,
                                                     int64_t ldt, float* lse, float* rowloss, int* err) {
    __shared__ float sred[4];
    const int64_t r = blockIdx.x;
    const float* row = logits + r * V;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    // Rows of up to 16384 logits are held in registers between the max pass and the exp-sum pass (16 x f32x4 per
    // thread): the row is read from memory once.  Same operations in the same order as the two-pass form.
    constexpr int NBUF = 16;
    const bool reg = vec && V <= NBUF * 1024;
    f32x4 buf[NBUF];
    float m = -INFINITY;
    if (reg) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) {
            const int c = tid * 4 + i * 1024;
            if (c < V) {
                buf[i] = *reinterpret_cast<const f32x4*>(row + c);
                m = fmaxf(fmaxf(m, fmaxf(buf[i][0], buf[i][1])), fmaxf(buf[i][2], buf[i][3]));
            }
        }
    } else if (vec) {
        for (int c = tid * 4; c < V; c += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        }
    } else {
        for (int c = tid; c < V; c += 256) m = fmaxf(m, row[c]);
    }
    m = wave_max(m);
    if (lane == 0) sred[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float s = 0.f;
    if (reg) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) {
            const int c = tid * 4 + i * 1024;
            if (c < V) s += expf(buf[i][0] - m) + expf(buf[i][1] - m) + expf(buf[i][2] - m) + expf(buf[i][3] - m);
        }
    } else if (vec) {
        for (int c = tid * 4; c < V; c += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            s += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
        }
    } else {
        for (int c = tid; c < V; c += 256) s += expf(row[c] - m);
    }
    s = wave_sum(s);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    if (tid == 0) {
        const float tot = (sred[0] + sred[1]) + (sred[2] + sred[3]);
        const float l = logf(tot) + m;
        int64_t t = row_target(target, r, Lm1, ldt);
        if (t < 0 || t >= V) {
            if (err) atomicExch(err, 2);
            t = t < 0 ? 0 : V - 1;
        }
        lse[r] = l;
        rowloss[r] = l - row[t];
    }
}

// loss = (sum_r rowloss[r]) / rows, fixed summation order (one workgroup, fp32 pairwise per thread strip).
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* rowloss, int64_t rows, float* loss_out) {
    __shared__ float sred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int64_t r = tid; r < rows; r += 256) s += rowloss[r];
    s = wave_sum(s);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    if (tid == 0) loss_out[0] = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)rows;
}

int mean_ce_fwd(hipStream_t s, const float* logits, int64_t rows, int V, const int64_t* target, int Lm1, int64_t ldt,
                float* lse, float* rowloss, float* loss_out, int* err_flag) {
    S2VT_REQUIRE(rows > 0 && V > 0 && logits && target && lse && rowloss && loss_out, "mean_ce_fwd: bad arguments");
    hipLaunchKernelGGL(ce_row_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, V, target, Lm1, ldt, lse, rowloss,
                       err_flag);
    S2VT_LAUNCH_CHECK("ce_row_kernel");
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, s, rowloss, rows, loss_out);
    S2VT_LAUNCH_CHECK("ce_mean_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------- MaskCriterion's outer arithmetic
// utils.py:23-25 of the reference around the mean CE: w = mask[:, 1:] flattened, loss = sum(mean_ce * w) / sum(w) (NaN for an
// all-zero mask, as upstream), evaluated product by product as the reference does - ONE workgroup behind the per-row kernel
// instead of a dozen elementwise / reduction launches of the host framework.  out[0] = loss, out[1] = mean_ce, out[2] = sum(w).
// Fixed summation order (thread strips, wave butterfly, four waves in order): deterministic.
__device__ __forceinline__ float block_sum_256(float s, float* sred) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s = wave_sum(s);
    __syncthreads();                       // (sred may still be read from the previous reduction)
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    return (sred[0] + sred[1]) + (sred[2] + sred[3]);
}
__global__ __launch_bounds__(256) void mask_criterion_fwd_kernel(const float* rowloss, int64_t rows, const float* mask, int64_t ldm,
                                                                 int Lm1, float* out) {
    __shared__ float sred[4];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int64_t r = tid; r < rows; r += 256) s += rowloss[r];
    const float mean_ce = block_sum_256(s, sred) / (float)rows;          // (ce_mean_kernel's arithmetic)
    float num = 0.f, den = 0.f;
    for (int64_t r = tid; r < rows; r += 256) {
        const float w = mask[(r / Lm1) * ldm + (r % Lm1) + 1];
        num += mean_ce * w;
        den += w;
    }
    num = block_sum_256(num, sred);
    den = block_sum_256(den, sred);
    if (tid == 0) { out[0] = num / den; out[1] = mean_ce; out[2] = den; }
}
// autograd of the same three lines: d loss / d mean_ce = sum_i (gout / sum(w)) * w_i  ->  g_ce[0]
__global__ __launch_bounds__(256) void mask_criterion_bwd_kernel(const float* mask, int64_t ldm, int64_t rows, int Lm1, const float* fwd_out,
                                                                 const float* gout, float* g_ce) {
    __shared__ float sred[4];
    const int tid = threadIdx.x;
    const float q = gout[0] / fwd_out[2];
    float s = 0.f;
    for (int64_t r = tid; r < rows; r += 256) s += q * mask[(r / Lm1) * ldm + (r % Lm1) + 1];
    s = block_sum_256(s, sred);
    if (tid == 0) g_ce[0] = s;
}

int mask_criterion_fwd(hipStream_t s, const float* logits, int64_t rows, int V, const int64_t* target, int Lm1, int64_t ldt,
                       const float* mask, int64_t ldm, float* lse, float* rowloss, float* out3, int* err_flag) {
    S2VT_REQUIRE(rows > 0 && V > 0 && logits && target && mask && lse && rowloss && out3, "mask_criterion_fwd: bad arguments");
    hipLaunchKernelGGL(ce_row_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, V, target, Lm1, ldt, lse, rowloss, err_flag);
    S2VT_LAUNCH_CHECK("ce_row_kernel");
    hipLaunchKernelGGL(mask_criterion_fwd_kernel, dim3(1), dim3(256), 0, s, rowloss, rows, mask, ldm, Lm1, out3);
    S2VT_LAUNCH_CHECK("mask_criterion_fwd_kernel");
    return 0;
}
int mask_criterion_bwd(hipStream_t s, const float* mask, int64_t ldm, int64_t rows, int Lm1, const float* fwd_out, const float* gout,
                       float* g_ce) {
    S2VT_REQUIRE(rows > 0 && mask && fwd_out && gout && g_ce, "mask_criterion_bwd: bad arguments");
    hipLaunchKernelGGL(mask_criterion_bwd_kernel, dim3(1), dim3(256), 0, s, mask, ldm, rows, Lm1, fwd_out, gout, g_ce);
    S2VT_LAUNCH_CHECK("mask_criterion_bwd_kernel");
    return 0;
}

// dlogits[r][v] = (exp(logit - lse_r) - [v == target_r]) * gout / rows
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, int64_t rows, int V, const int64_t* target,
                                                     int Lm1, int64_t ldt, const float* lse, const float* gout,
                                                     float* dlogits) {
    const int64_t r = blockIdx.x;
    const float* row = logits + r * V;
    float* drow = dlogits + r * V;
    const float l = lse[r];
    const float scale = gout[0] / (float)rows;
    int64_t t = row_target(target, r, Lm1, ldt);
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const int tid = threadIdx.x;
    const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(drow) & 15) == 0);
    if (vec) {
        for (int c = tid * 4; c < V; c += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            f32x4 d;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = (expf(v[j] - l) - ((c + j) == t ? 1.f : 0.f)) * scale;
            *reinterpret_cast<f32x4*>(drow + c) = d;
        }
    } else {
        for (int c = tid; c < V; c += 256) drow[c] = (expf(row[c] - l) - (c == t ? 1.f : 0.f)) * scale;
    }
}

int mean_ce_bwd(hipStream_t s, const float* logits, int64_t rows, int V, const int64_t* target, int Lm1, int64_t ldt,
                const float* lse, const float* gout, float* dlogits) {
    S2VT_REQUIRE(rows > 0 && V > 0 && logits && target && lse && gout && dlogits, "mean_ce_bwd: bad arguments");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, rows, V, target, Lm1, ldt, lse,
                       gout, dlogits);
    S2VT_LAUNCH_CHECK("ce_bwd_kernel");
    return 0;
}


// ---------------------------------------------------------------------------------- beam-search fan-out
// Per logits row: log_softmax (S2VTModel.py:214) and the 20 most probable tokens with their log-probs, returned in
// ASCENDING token order - the order in which the reference pushes them (it scans the vocabulary and keeps the ids that
// are in topk(20), S2VTModel.py:216-219).  One workgroup per row: the row is staged in LDS, reduced to max and
// sum-exp, then the largest remaining element is extracted 20 times (ties -> the lower token id).
constexpr int TOPK_N = 20;
__global__ __launch_bounds__(256) void top20_logprob_kernel(const float* logits, int64_t ld, int V, int32_t* top_ix,
                                                            float* top_lp) {
    extern __shared__ float row[];                  // V floats, then reduction scratch
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ float sel_v[TOPK_N];
    __shared__ int sel_i[TOPK_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = logits + (int64_t)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) {
        const float v = src[i];
        row[i] = v;
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red_v[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf(row[i] - mx);
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red_v[wave] = sum;
    __syncthreads();
    const float lse = mx + logf((red_v[0] + red_v[1]) + (red_v[2] + red_v[3]));
    __syncthreads();
    // every thread keeps the best of its own elements; only the owner of an extracted element rescans
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    auto rescan = [&]() {
        bv = -INFINITY; bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float v = row[i];
            if (v > bv) { bv = v; bi = i; }          // ascending i: the first (lowest id) of equal values is kept
        }
    };
    rescan();
    for (int k = 0; k < TOPK_N; ++k) {
        float v = bv;
        int ix = bi;
        for (int o = 32; o; o >>= 1) {
            const float ov = __shfl_xor(v, o);
            const int oi = __shfl_xor(ix, o);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if (lane == 0) { red_v[wave] = v; red_i[wave] = ix; }
        __syncthreads();
        v = red_v[0]; ix = red_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red_v[w] > v || (red_v[w] == v && red_i[w] < ix)) { v = red_v[w]; ix = red_i[w]; }
        if (tid == 0) { sel_v[k] = v; sel_i[k] = ix; }
        if ((ix & 255) == tid && ix < V) {           // owner: remove it and find the next best of its elements
            row[ix] = -INFINITY;
            rescan();
        }
        __syncthreads();
    }
    if (tid < TOPK_N) {                              // rank by token id (ids are distinct)
        const int my = sel_i[tid];
        int rank = 0;
#pragma unroll
        for (int j = 0; j < TOPK_N; ++j) rank += (sel_i[j] < my);
        top_ix[(int64_t)blockIdx.x * TOPK_N + rank] = my;
        top_lp[(int64_t)blockIdx.x * TOPK_N + rank] = sel_v[tid] - lse;
    }
}

// The same selection with the row in REGISTERS (VPT elements per thread, element i = tid + 256 j): the owner's rescan after an
// extraction is VPT register compares instead of VPT dependent LDS reads (the LDS form spent 1.4 of its 4.7 us per extraction
// there: 95 us per launch at V = 12000, 640 rows; this one ~25).  Same arithmetic in the same order: max, then the sum of
// expf(x - max) per thread in ascending i, wave butterflies, (w0 + w1) + (w2 + w3); ties -> the lower token id.
template <int VPT>
__global__ __launch_bounds__(256) void top20_logprob_reg_kernel(const float* logits, int64_t ld, int V, int32_t* top_ix,
                                                                float* top_lp) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ float sel_v[TOPK_N];
    __shared__ int sel_i[TOPK_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = logits + (int64_t)blockIdx.x * ld;
    float vals[VPT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + 256 * j;
        vals[j] = (i < V) ? src[i] : -INFINITY;
        mx = fmaxf(mx, vals[j]);
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red_v[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
        if (tid + 256 * j < V) sum += expf(vals[j] - mx);
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red_v[wave] = sum;
    __syncthreads();
    const float lse = mx + logf((red_v[0] + red_v[1]) + (red_v[2] + red_v[3]));
    __syncthreads();
    float bv;
    int bj;
    auto rescan = [&]() {
        bv = -INFINITY; bj = -1;
#pragma unroll
        for (int j = 0; j < VPT; ++j)
            if (vals[j] > bv) { bv = vals[j]; bj = j; }      // ascending i: the first (lowest id) of equal values is kept
    };
    rescan();
    for (int k = 0; k < TOPK_N; ++k) {
        float v = bv;
        int ix = bj >= 0 ? tid + 256 * bj : 0x7fffffff;
        for (int o = 32; o; o >>= 1) {
            const float ov = __shfl_xor(v, o);
            const int oi = __shfl_xor(ix, o);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if (lane == 0) { red_v[wave] = v; red_i[wave] = ix; }
        __syncthreads();
        v = red_v[0]; ix = red_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red_v[w] > v || (red_v[w] == v && red_i[w] < ix)) { v = red_v[w]; ix = red_i[w]; }
        if (tid == 0) { sel_v[k] = v; sel_i[k] = ix; }
        if ((ix & 255) == tid && ix < V) {           // owner: remove it and find the next best of its elements
            const int jj = ix >> 8;
#pragma unroll
            for (int j = 0; j < VPT; ++j)
                if (j == jj) vals[j] = -INFINITY;
            rescan();
        }
        __syncthreads();
    }
    if (tid < TOPK_N) {                              // rank by token id (ids are distinct)
        const int my = sel_i[tid];
        int rank = 0;
#pragma unroll
        for (int j = 0; j < TOPK_N; ++j) rank += (sel_i[j] < my);
        top_ix[(int64_t)blockIdx.x * TOPK_N + rank] = my;
        top_lp[(int64_t)blockIdx.x * TOPK_N + rank] = sel_v[tid] - lse;
    }
}

int top20_logprob(hipStream_t s, const float* logits, int64_t ld, int64_t rows, int V, int32_t* top_ix, float* top_lp) {
    if (rows <= 0) return 0;
    S2VT_REQUIRE(V >= TOPK_N && (size_t)V * sizeof(float) <= 150 * 1024, "top20_logprob: 20 <= vocab_size <= 38400");
    if (V <= 256 * 64) {            // the row in registers; larger vocabularies: staged in LDS (top20_logprob_kernel)
        const dim3 grid((unsigned)rows), block(256);
        if (V <= 256 * 16) hipLaunchKernelGGL(top20_logprob_reg_kernel<16>, grid, block, 0, s, logits, ld, V, top_ix, top_lp);
        else if (V <= 256 * 32) hipLaunchKernelGGL(top20_logprob_reg_kernel<32>, grid, block, 0, s, logits, ld, V, top_ix, top_lp);
        else if (V <= 256 * 48) hipLaunchKernelGGL(top20_logprob_reg_kernel<48>, grid, block, 0, s, logits, ld, V, top_ix, top_lp);
        else hipLaunchKernelGGL(top20_logprob_reg_kernel<64>, grid, block, 0, s, logits, ld, V, top_ix, top_lp);
        S2VT_LAUNCH_CHECK("top20_logprob_reg_kernel");
        return 0;
    }
    if ((size_t)V * sizeof(float) > 48 * 1024)
        S2VT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(top20_logprob_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)V * sizeof(float))));
    hipLaunchKernelGGL(top20_logprob_kernel, dim3((unsigned)rows), dim3(256), (size_t)V * sizeof(float), s, logits, ld, V,
                       top_ix, top_lp);
    S2VT_LAUNCH_CHECK("top20_logprob_kernel");
    return 0;
}

}  // namespace s2vt
