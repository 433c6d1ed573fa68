// GEMMs on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate): the arithmetic and the
// dispatch; the kernels live in gemm_x3.hip (3 planes, blocked layout) and gemm_b1.hip (1 plane, bf16 rows).
//
// An fp32 value x is carried as NP bf16 "planes": x = p0 + p1 + p2 exactly to 24 mantissa bits
// (p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)).  A product a*b is then the sum of plane products;
// bf16*bf16 is exact in fp32 and the MFMA accumulates in fp32, so keeping the six products whose weight is
// >= 2^-16 of the leading one (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) reproduces an fp32 product to
// ~2^-23 relative - fp32-equivalent arithmetic at 16/6 = 2.7x the fp32-MFMA rate.  NP = 1 is the plain bf16 GEMM
// (storage bf16, accumulate fp32) of BASELINE config 3.
//
// C[M,N] (+)= A[M,K] · B[N,K]^T (+ bias): ONLY the "NT" form - both operands k-contiguous.  Layout changes
// (transposes, gathers, batch-major <-> time-major) are done by the memory-bound plane-splitting kernels (split.hip).
#include "common.h"
#include "kernels.h"

namespace s2vt {

int gemm_bf16_nt(hipStream_t stream, int nplanes, int M, int N, int K, const unsigned short* A, int64_t lda,
                 const unsigned short* B, int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias,
                 bool accumulate, float* splitk_ws, size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    S2VT_REQUIRE(nplanes == 1 || nplanes == 3, "gemm_bf16_nt: planes must be 1 or 3");
    if (nplanes == 3)
        return gemm_x3(stream, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
    return gemm_b1(stream, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
}

}  // namespace s2vt
