// fp32 -> bf16 plane splitting (see gemm_bf16.hip): memory-bound kernels that write, for each consumer GEMM, the
// operand as NP bf16 planes in k-major layout — optionally transposed (for the X^T·Y weight-gradient GEMMs) and
// with gathered / permuted source rows (embedding rows, batch-major <-> time-major) — so the MFMA kernel only
// ever sees the k-contiguous "NT" form.  Rows and columns are padded with zeros to multiples of 8 elements.
#include "common.h"
#include "kernels.h"

namespace s2vt {

__device__ __forceinline__ unsigned short bf16_rn(float x) {
    unsigned int u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

template <int NP>
__device__ __forceinline__ void split3(float x, unsigned short (&o)[3]) {
    o[0] = bf16_rn(x);
    if (NP == 3) {
        const float r1 = x - bf16_f32(o[0]);          // exact
        o[1] = bf16_rn(r1);
        const float r2 = r1 - bf16_f32(o[1]);         // exact
        o[2] = bf16_rn(r2);
    }
}

// LDS staging tile of the transposing kernels: 64 x 64 fp32, row stride 68 floats (rows 16-byte aligned: a 16-byte global
// load lands as one ds_write_b128, an 8-element run leaves as two ds_read_b128) with element (r, c) at column c ^ TSW(r):
// the column reads of the plain-layout transposed path (lanes = 8 consecutive c x 8 row octets) then hit 64 distinct banks
// (8 * 68 = 32 mod 64 alone would fold the octets onto two bank groups); runs of 8 columns stay contiguous.
#define TSW(r) ((((r) >> 3) & 7) << 3)

// Plane layouts.  1 plane (bf16 mode, gemm_bf16.hip / lstm_bf16.hip): plain k-major rows, element (r, k) at r*ldo + k.
// 3 planes (split precision, gemm_x3.hip): BLOCKED - per 64-row block and 16-wide k chunk one 6-KB record of six
// 1-KB pieces (plane, k half), each piece = 64 rows x 8 consecutive k:
//   (r/64)*(64*ldo) + (k/16)*3072 + (pl*2 + (k%16)/8)*512 + (r%64)*8 + k%8,   ldo = 3*kpad.
__device__ __forceinline__ int64_t packed_off(int64_t r, int k, int pl, int64_t ldo, int np) {
    if (np == 3)
        return (r >> 6) * (64 * ldo) + (int64_t)(k >> 4) * 3072 + (pl * 2 + ((k >> 3) & 1)) * 512 + (r & 63) * 8 + (k & 7);
    return r * ldo + k;
}

// out(r, c, pl) = plane pl of in[map(r)][c]; r < rows_pad, c < kpad (zeros beyond rows/cols). 8 columns per thread.
template <int NP>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* in, int64_t ld, RowMap imap, int rows, int cols,
                                                         unsigned short* out, int64_t ldo, int kpad, int rows_pad) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nq = kpad / 8;
    if (q >= (int64_t)rows_pad * nq) return;
    const int r = (int)(q / nq), c0 = (int)(q % nq) * 8;
    unsigned short o[3][8];
    const bool rok = r < rows;
    const float* src = in + (int64_t)(rok ? map_row(imap, r) : 0) * ld;
    const bool vec = rok && (c0 + 7 < cols) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    float v[8];
    if (vec) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + c0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + c0 + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (rok && c0 + j < cols) ? src[c0 + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned short t[3];
        split3<NP>(v[j], t);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) o[pl][j] = t[pl];
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        u_int32_t w0 = o[pl][0] | ((u_int32_t)o[pl][1] << 16), w1 = o[pl][2] | ((u_int32_t)o[pl][3] << 16);
        u_int32_t w2 = o[pl][4] | ((u_int32_t)o[pl][5] << 16), w3 = o[pl][6] | ((u_int32_t)o[pl][7] << 16);
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(out + packed_off(r, c0, pl, ldo, NP)) = u32x4{w0, w1, w2, w3};
    }
}

// Transposed planes: out(c, r, pl) = plane pl of in[map(r)][c]: output row c < cols_pad, output k index r < kpad.
// 64x64 tiles through LDS so both the fp32 reads and the bf16 writes are coalesced.
template <int NP>
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* in, int64_t ld, RowMap imap, int rows, int cols,
                                                              unsigned short* out, int64_t ldo, int kpad,
                                                              int cols_pad) {
    __shared__ __attribute__((aligned(16))) float tile[64][68];       // (row stride 68: see split_dual_kernel)
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {          // 16 bytes per lane and load
        const int i = it * 256 + (int)threadIdx.x;
        const int rl = i >> 4, cl = (i & 15) * 4;
        const int r = r0 + rl, c = c0 + cl;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (r < rows && c < cols) {
            const float* src = in + (int64_t)map_row(imap, r) * ld + c;
            if (vec_ok && c + 3 < cols) {
                x = *reinterpret_cast<const f32x4*>(src);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = (c + j < cols) ? src[j] : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(&tile[rl][cl ^ TSW(rl)]) = x;
    }
    __syncthreads();
    // each thread: one output row (c) segment of 8 consecutive r
    for (int s = threadIdx.x; s < 64 * 8; s += 256) {
        const int cl = s >> 3, rq = (s & 7) * 8;
        const int c = c0 + cl, r = r0 + rq;
        if (c >= cols_pad || r >= kpad) continue;
        unsigned short o[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned short t[3];
            split3<NP>(tile[rq + j][cl ^ TSW(rq + j)], t);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) o[pl][j] = t[pl];
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            u_int32_t w0 = o[pl][0] | ((u_int32_t)o[pl][1] << 16), w1 = o[pl][2] | ((u_int32_t)o[pl][3] << 16);
            u_int32_t w2 = o[pl][4] | ((u_int32_t)o[pl][5] << 16), w3 = o[pl][6] | ((u_int32_t)o[pl][7] << 16);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4*>(out + packed_off(c, r, pl, ldo, NP)) = u32x4{w0, w1, w2, w3};
        }
    }
}

// One pass over a 64x64 tile of in[rows][cols] producing any of: (a) row planes (operand rows = input rows),
// (b) transposed planes (operand rows = input columns, k = input row index), (c) per-tile column sums
// partial[tile_row][col] (the bias gradients: summed afterwards in a fixed order by colsum_final_kernel).
// Tensors that feed both a data-gradient GEMM (row planes) and a weight-gradient GEMM (transposed planes) — dG,
// dlogits, h, x1, the weights themselves — are read from HBM once.
// CE = true: the input is the LOGITS and the value that is split is the mean-CE gradient (utils.py:22 under loss.backward())
//   d[r][c] = (exp(logit[r][c] - lse[r]) - [c == target(r)]) * gout / rows
// - the expression of ce_bwd_kernel (ce.hip), evaluated here so that the fp32 dlogits tensor is never written and re-read
// (971 MB each way at B = 256): the planes and the bias-gradient partial sums come out bit for bit as from the two-kernel route.
template <int NP, bool CE>
__global__ __launch_bounds__(256) void split_dual_kernel(const float* in, int64_t ld, RowMap imap, int rows, int cols,
                                                         unsigned short* out_r, int64_t ldo_r, int kpad_r,
                                                         unsigned short* out_t, int64_t ldo_t, int kpad_t,
                                                         float* colpart, CeGradArgs ce) {
    // 64 x 64 fp32 tile, row stride 68 floats: rows stay 16-byte aligned (16-byte global loads land as one ds_write_b128, the
    // row-plane path reads two ds_read_b128 per 8-element run: conflict-free per 16-lane group) and the transposed path's
    // scalar column reads see consecutive banks
    __shared__ __attribute__((aligned(16))) float tile[64][68];
    __shared__ float row_lse[64];          // CE only: per tile row, fetched once (the index arithmetic of a target is two
    __shared__ int row_tgt[64];            // integer divisions: per ELEMENT they cost more than the split itself)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    float ce_scale = 0.f, ce_alpha = 1.f;
    if (CE) {
        if (threadIdx.x < 64) {
            const int r = r0 + (int)threadIdx.x;
            int64_t t = 0;
            float l = 0.f;
            if (r < rows) {
                t = ce.target[(int64_t)(r / ce.Lm1) * ce.ldt + (r % ce.Lm1) + 1];
                t = t < 0 ? 0 : (t >= cols ? cols - 1 : t);
                l = ce.lse[r];
            }
            row_tgt[threadIdx.x] = (int)t;
            row_lse[threadIdx.x] = l;
        }
        ce_scale = ce.gout[0] / (float)rows;
        if (ce.alpha_out) {         // sign and exponent of the scale only (see CeGradArgs::alpha_out)
            const float s2 = __uint_as_float(__float_as_uint(ce_scale) & 0xFF800000u);
            ce_alpha = (s2 != 0.f) ? ce_scale / s2 : 0.f;
            ce_scale = s2;
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *ce.alpha_out = ce_alpha;
        }
        __syncthreads();
    }
    // 16 bytes per lane and load (a wave covers 4 rows x 256 B): 4-byte loads cannot keep HBM busy
    const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 256 + (int)threadIdx.x;
        const int rl = i >> 4, cl = (i & 15) * 4;
        const int r = r0 + rl, c = c0 + cl;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (r < rows && c < cols) {
            const float* src = in + (int64_t)map_row(imap, r) * ld + c;
            if (vec_ok && c + 3 < cols) {
                x = *reinterpret_cast<const f32x4*>(src);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = (c + j < cols) ? src[j] : 0.f;
            }
            if (CE) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    x[j] = (c + j < cols) ? (expf(x[j] - row_lse[rl]) - (c + j == row_tgt[rl] ? 1.f : 0.f)) * ce_scale : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(&tile[rl][cl ^ TSW(rl)]) = x;
    }
    __syncthreads();
    if (colpart && threadIdx.x < 64 && c0 + (int)threadIdx.x < cols) {
        float sum = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) sum += tile[i][(int)threadIdx.x ^ TSW(i)];
        colpart[(int64_t)blockIdx.y * cols + c0 + threadIdx.x] = CE ? sum * ce_alpha : sum;
    }
    for (int s = threadIdx.x; s < 64 * 8; s += 256) {
        // a: line index, bq: start of an 8-element run.  Blocked layout: a wave writes one 1-KB piece (64 lines x 16 B)
        const int a = (NP == 3) ? (s & 63) : (s >> 3), bq = (NP == 3) ? (s >> 6) * 8 : (s & 7) * 8;
        if (out_r) {                                   // operand row = input row r0+a, k = input cols c0+bq..+7
            const int r = r0 + a, c = c0 + bq;
            if (r < rows && c < kpad_r) {
                unsigned short o[3][8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[a][bq ^ TSW(a)]), hi = *reinterpret_cast<const f32x4*>(&tile[a][(bq ^ TSW(a)) + 4]);
                const float run[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned short t[3];
                    split3<NP>(run[j], t);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) o[pl][j] = t[pl];
                }
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    *reinterpret_cast<u32x4*>(out_r + packed_off(r, c, pl, ldo_r, NP)) =
                        u32x4{o[pl][0] | ((u_int32_t)o[pl][1] << 16), o[pl][2] | ((u_int32_t)o[pl][3] << 16),
                              o[pl][4] | ((u_int32_t)o[pl][5] << 16), o[pl][6] | ((u_int32_t)o[pl][7] << 16)};
            }
        }
        if (out_t) {                                   // operand row = input col c0+a, k = input rows r0+bq..+7
            const int c = c0 + a, r = r0 + bq;
            if (c < cols && r < kpad_t) {
                unsigned short o[3][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned short t[3];
                    split3<NP>(tile[bq + j][a ^ TSW(bq + j)], t);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) o[pl][j] = t[pl];
                }
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    *reinterpret_cast<u32x4*>(out_t + packed_off(c, r, pl, ldo_t, NP)) =
                        u32x4{o[pl][0] | ((u_int32_t)o[pl][1] << 16), o[pl][2] | ((u_int32_t)o[pl][3] << 16),
                              o[pl][4] | ((u_int32_t)o[pl][5] << 16), o[pl][6] | ((u_int32_t)o[pl][7] << 16)};
            }
        }
    }
}

// Dual split.  out_r (nullable): row planes, rows [0, rows) of the operand at out_r, k padded to kpad_r >= cols.
// out_t (nullable): transposed planes, operand rows = input columns, k range [0, kpad_t >= rows) at out_t.
// colpart (nullable): cdiv(rows, 64) x cols partial column sums.  Padding (k beyond the data) is zero-filled as long
// as kpad_r <= 64*cdiv(cols,64) and kpad_t <= 64*cdiv(rows,64), which holds for kpad = pad64(.).
int split_planes_dual(hipStream_t s, int nplanes, const float* in, int64_t ld, RowMap imap, int rows, int cols,
                      unsigned short* out_r, int64_t ldo_r, int kpad_r, unsigned short* out_t, int64_t ldo_t, int kpad_t,
                      float* colpart, const CeGradArgs* ce) {
    S2VT_REQUIRE(nplanes == 1 || nplanes == 3, "split_planes_dual: planes must be 1 or 3");
    if (rows <= 0 || cols <= 0) return 0;
    S2VT_REQUIRE(!out_r || (kpad_r % 64 == 0 && kpad_r >= cols && kpad_r <= 64 * cdiv(cols, 64) && ldo_r >= (int64_t)nplanes * kpad_r),
                 "split_planes_dual: bad row-plane geometry");
    S2VT_REQUIRE(!out_t || (kpad_t % 64 == 0 && kpad_t >= rows && kpad_t <= 64 * cdiv(rows, 64) && ldo_t >= (int64_t)nplanes * kpad_t),
                 "split_planes_dual: bad transposed-plane geometry");
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 64));
    CeGradArgs none = {nullptr, nullptr, nullptr, 1, 0, nullptr};
    if (ce) {
        S2VT_REQUIRE(ce->lse && ce->target && ce->gout && ce->Lm1 > 0 && imap.idx == nullptr && imap.inner == 0,
                     "split_planes_dual: bad CE-gradient arguments");
        if (nplanes == 3)
            hipLaunchKernelGGL((split_dual_kernel<3, true>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out_r, ldo_r, kpad_r,
                               out_t, ldo_t, kpad_t, colpart, *ce);
        else
            hipLaunchKernelGGL((split_dual_kernel<1, true>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out_r, ldo_r, kpad_r,
                               out_t, ldo_t, kpad_t, colpart, *ce);
    } else if (nplanes == 3)
        hipLaunchKernelGGL((split_dual_kernel<3, false>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out_r, ldo_r, kpad_r, out_t,
                           ldo_t, kpad_t, colpart, none);
    else
        hipLaunchKernelGGL((split_dual_kernel<1, false>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out_r, ldo_r, kpad_r, out_t,
                           ldo_t, kpad_t, colpart, none);
    S2VT_LAUNCH_CHECK("split_dual_kernel");
    return 0;
}

// in: fp32 [rows][cols] (row stride ld, rows mapped through imap) -> packed planes of a k-major operand:
//   transpose == false: operand rows = input rows (out_rows_pad >= rows), k = input columns (kpad >= cols);
//   transpose == true : operand rows = input columns (out_rows_pad >= cols), k = input rows (kpad >= rows).
// kpad % 32 == 0, ldo >= nplanes * kpad.
int split_planes(hipStream_t s, int nplanes, bool transpose, const float* in, int64_t ld, RowMap imap, int rows, int cols,
                 unsigned short* out, int64_t ldo, int kpad, int out_rows_pad) {
    S2VT_REQUIRE(nplanes == 1 || nplanes == 3, "split_planes: planes must be 1 or 3");
    S2VT_REQUIRE(kpad % 64 == 0 && ldo % 8 == 0 && ldo >= (int64_t)nplanes * kpad &&
                     (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "split_planes: kpad must be a multiple of 64, ldo >= nplanes*kpad, output 16-B aligned");
    if (rows <= 0 || cols <= 0) return 0;
    if (nplanes == 3) {     // blocked layout: the tiled kernel writes whole 1-KB pieces in either orientation
        S2VT_REQUIRE(kpad >= (transpose ? rows : cols) && out_rows_pad >= (transpose ? cols : rows), "split_planes: output too small");
        return transpose ? split_planes_dual(s, 3, in, ld, imap, rows, cols, nullptr, 0, 0, out, ldo, kpad, nullptr, nullptr)
                         : split_planes_dual(s, 3, in, ld, imap, rows, cols, out, ldo, kpad, nullptr, 0, 0, nullptr, nullptr);
    }
    if (!transpose) {
        S2VT_REQUIRE(kpad >= cols && out_rows_pad >= rows, "split_planes: output too small");
        const int64_t nq = (int64_t)out_rows_pad * (kpad / 8);
        const dim3 grid((unsigned)((nq + 255) / 256));
        if (nplanes == 3) hipLaunchKernelGGL((split_rows_kernel<3>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out, ldo, kpad, out_rows_pad);
        else hipLaunchKernelGGL((split_rows_kernel<1>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out, ldo, kpad, out_rows_pad);
    } else {
        S2VT_REQUIRE(kpad >= rows && out_rows_pad >= cols, "split_planes: output too small");
        const dim3 grid(cdiv(out_rows_pad, 64), cdiv(kpad, 64));
        if (nplanes == 3) hipLaunchKernelGGL((split_transpose_kernel<3>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out, ldo, kpad, out_rows_pad);
        else hipLaunchKernelGGL((split_transpose_kernel<1>), grid, dim3(256), 0, s, in, ld, imap, rows, cols, out, ldo, kpad, out_rows_pad);
    }
    S2VT_LAUNCH_CHECK("split_planes");
    return 0;
}

}  // namespace s2vt
