// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma
// chain).  Serves every batched contraction of the S2VT path: frame-feature projection
// (S2VTModel.py:54), the hoisted x·W_ih^T input GEMMs of both LSTMs (:67,:77), the vocab logits
// (:80) and all weight/activation gradient GEMMs of the train step (train.py:124).
//
// C[M,N] (+)= A[M,K] · B[K,N] (+ bias[N]), fp32 in / fp32 accumulate.
//   A_KMAJOR : A stored as rows of m, k contiguous  (else rows of k, m contiguous: "A^T stored")
//   B_KMAJOR : B stored as rows of n, k contiguous  (x·W^T form; else rows of k, n contiguous)
// Stored rows of A, B and C can be gathered / permuted through RowMap (embedding rows,
// batch-major <-> time-major), so no transposed or gathered copy is ever materialised.
//
// Tiling: 128x128x32 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles,
// 64 accumulator VGPRs).  Global -> register prefetch of tile t+1 overlaps the MFMAs of tile t;
// LDS images are padded so that both the 16-B staging writes and the MFMA operand reads are
// bank-conflict free (k-major image [128][36]: ds_read_b128; m-major image [32][132]: ds_read_b32).
// Inside each 8-wide k block lane-half h owns k = 4h..4h+3 for BOTH operands, so MFMA j sums
// k = 8c+j and 8c+4+j: a fixed, deterministic summation order.
#include "common.h"
#include "kernels.h"

namespace s2vt {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;    // k-major image row stride (floats)
constexpr int LDM = BM + 4;    // m-major image row stride (floats)
constexpr int OPER_FLOATS = BM * LDK;  // 4608 >= BK*LDM = 4224

struct GemmArgs {
    int M, N, K;
    const float* A; int64_t lda; RowMap amap;
    const float* B; int64_t ldb; RowMap bmap;
    float* C; int64_t ldc; RowMap cmap;
    const float* bias;
    int accumulate;
    int vecA, vecB;   // 16-byte vector loads legal (ld % 4 == 0 and base 16-B aligned)
};

__device__ __forceinline__ f32x4 load4_guard(const float* row, int c, int limit, bool vec) {
    // row may be nullptr (out-of-range stored row) -> zeros
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row == nullptr || c >= limit) return v;
    if (vec && c + 3 < limit) {
        v = *reinterpret_cast<const f32x4*>(row + c);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < limit) v[j] = row[c + j];
    }
    return v;
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * OPER_FLOATS];
    float* sA = smem;
    float* sB = smem + OPER_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = (p.N + BN - 1) / BN;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging coordinates
    // k-major operand: 4 x (row = tid/8 + 32 i, kq = (tid%8)*4)
    // m-major operand: 4 x (krow = tid/32 + 8 i, mq = (tid%32)*4)
    const float* a_rows[4];
    const float* b_rows[4];
    if (A_KMAJOR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = m0 + (tid >> 3) + 32 * i;
            a_rows[i] = (m < p.M) ? p.A + (int64_t)map_row(p.amap, m) * p.lda : nullptr;
        }
    }
    if (B_KMAJOR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int n = n0 + (tid >> 3) + 32 * i;
            b_rows[i] = (n < p.N) ? p.B + (int64_t)map_row(p.bmap, n) * p.ldb : nullptr;
        }
    }

    f32x4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
        if (A_KMAJOR) {
            const int kq = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load4_guard(a_rows[i], kq, p.K, p.vecA);
        } else {
            const int mq = m0 + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int k = k0 + (tid >> 5) + 8 * i;
                const float* row = (k < p.K) ? p.A + (int64_t)map_row(p.amap, k) * p.lda : nullptr;
                ra[i] = load4_guard(row, mq, p.M, p.vecA);
            }
        }
        if (B_KMAJOR) {
            const int kq = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = load4_guard(b_rows[i], kq, p.K, p.vecB);
        } else {
            const int nq = n0 + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int k = k0 + (tid >> 5) + 8 * i;
                const float* row = (k < p.K) ? p.B + (int64_t)map_row(p.bmap, k) * p.ldb : nullptr;
                rb[i] = load4_guard(row, nq, p.N, p.vecB);
            }
        }
    };
    auto store_tile = [&]() {
        if (A_KMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(&sA[((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4]) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(&sA[((tid >> 5) + 8 * i) * LDM + (tid & 31) * 4]) = ra[i];
        }
        if (B_KMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(&sB[((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(&sB[((tid >> 5) + 8 * i) * LDM + (tid & 31) * 4]) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile((kt + 1) * BK);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wm * 64 + mi * 32 + li;
                if (A_KMAJOR) {
                    a[mi] = *reinterpret_cast<const f32x4*>(&sA[row * LDK + 8 * c + 4 * lh]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[mi][j] = sA[(8 * c + 4 * lh + j) * LDM + row];
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + li;
                if (B_KMAJOR) {
                    b[ni] = *reinterpret_cast<const f32x4*>(&sB[row * LDK + 8 * c + 4 * lh]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[ni][j] = sB[(8 * c + 4 * lh + j) * LDM + row];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= p.M) continue;
            float* crow = p.C + (int64_t)map_row(p.cmap, m) * p.ldc;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = n0 + wn * 64 + ni * 32 + li;
                if (n >= p.N) continue;
                float v = acc[mi][ni][r];
                if (p.bias) v += p.bias[n];
                if (p.accumulate) v += crow[n];
                crow[n] = v;
            }
        }
    }
}

static inline bool vec_ok(const void* ptr, int64_t ld) {
    return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0);
}

int gemm_f32(hipStream_t stream, bool a_kmajor, bool b_kmajor, int M, int N, int K,
             const float* A, int64_t lda, RowMap amap, const float* B, int64_t ldb, RowMap bmap,
             float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate) {
    if (M <= 0 || N <= 0) return 0;
    S2VT_REQUIRE(K >= 0 && A && B && C, "gemm_f32: bad arguments (M=%d N=%d K=%d)", M, N, K);
    S2VT_REQUIRE(a_kmajor || !b_kmajor, "gemm_f32: A^T * B^T form is not used by the S2VT path");
    GemmArgs p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda; p.amap = amap;
    p.B = B; p.ldb = ldb; p.bmap = bmap;
    p.C = C; p.ldc = ldc; p.cmap = cmap;
    p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    p.vecA = vec_ok(A, lda); p.vecB = vec_ok(B, ldb);
    const int grid = cdiv(M, BM) * cdiv(N, BN);
    if (a_kmajor && b_kmajor)
        hipLaunchKernelGGL((gemm_f32_kernel<true, true>), dim3(grid), dim3(256), 0, stream, p);
    else if (a_kmajor && !b_kmajor)
        hipLaunchKernelGGL((gemm_f32_kernel<true, false>), dim3(grid), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3(grid), dim3(256), 0, stream, p);
    S2VT_LAUNCH_CHECK("gemm_f32_kernel");
    return 0;
}

}  // namespace s2vt
