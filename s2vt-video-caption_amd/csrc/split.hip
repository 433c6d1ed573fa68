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

// Packed plane layout (see gemm_bf16.hip): element (r, k, pl) at r*ldo + (k/32)*(32*NP) + pl*32 + k%32, ldo = NP*kpad.
__device__ __forceinline__ int64_t packed_off(int64_t r, int k, int pl, int64_t ldo, int np) {
    return r * ldo + (int64_t)(k >> 5) * (32 * np) + pl * 32 + (k & 31);
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
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)map_row(imap, r) * ld + c] : 0.f;
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
            split3<NP>(tile[rq + j][cl], t);
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
