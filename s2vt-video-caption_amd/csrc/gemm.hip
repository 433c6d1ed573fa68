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
#ifndef S2VT_GEMM_NBUF
#define S2VT_GEMM_NBUF 1
#endif
#ifndef S2VT_GEMM_PF
#define S2VT_GEMM_PF 1
#endif
constexpr int GPF = S2VT_GEMM_PF;      // staging register sets (tiles of global-load lookahead)
constexpr int NBUF = S2VT_GEMM_NBUF;   // LDS images per operand (2 = double buffered, one barrier per k-tile)

struct GemmArgs {
    int M, N, K;
    const float* A; int64_t lda; RowMap amap;
    const float* B; int64_t ldb; RowMap bmap;
    float* C; int64_t ldc; RowMap cmap;
    const float* bias;
    int accumulate;
    int vecA, vecB;   // 16-byte vector loads legal (ld % 4 == 0 and base 16-B aligned)
    int ksplit;       // K range per blockIdx.y slice (multiple of BK); == K when not split
    float* slabs;     // split-K: slice s writes its partial tile to slabs + s*M*N (row-major [M][N])
};

// Branch-free guarded load of 4 consecutive floats of a stored row.  hipcc turns a load under a runtime
// condition into a branch around it plus a vmcnt(0) wait per element (cdna_hip_programming.md §5 trap (c)),
// which serialises the whole staging burst; so out-of-range accesses are redirected to a safe address
// (`base`) and zeroed with a select instead.  VEC: limit % 4 == 0, c % 4 == 0 and 16-B aligned rows, so a
// float4 starting below `limit` is entirely in range.
template <bool VEC>
__device__ __forceinline__ f32x4 load4_guard(const float* base, const float* row, int c, int limit) {
    // Out-of-range accesses read a 16-byte block of zeros instead of being masked afterwards: the loaded value
    // then has NO consumer before the LDS staging store, so the loads stay in flight across the MFMA phase
    // (a select on the result would pull the vmcnt wait in front of the MFMAs).
    f32x4 v;
    if (VEC) {
        const bool ok = (row != nullptr) && (c < limit);
        const float* q = ok ? row + c : g_zero4;
        v = *reinterpret_cast<const f32x4*>(q);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (row != nullptr) && (c + j < limit);
            const float* q = ok ? row + c + j : g_zero4;
            v[j] = *q;
        }
    }
    return v;
}

template <bool A_KMAJOR, bool B_KMAJOR, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[NBUF * 2 * OPER_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (speed only): blocks are dealt round-robin over the 8 XCDs, so XCD x is handed the
    // contiguous tile range [x*cpx, (x+1)*cpx); inside it tiles walk 8-row groups column by column, so the
    // ~100 tiles resident on one XCD form a compact rectangle and share their A/B panels through that XCD's L2.
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int nwg = ntm * ntn, cpx = (nwg + 7) >> 3;
    const int t = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= cpx || t >= nwg) return;
    constexpr int GM = 8;
    const int gsz = GM * ntn, grp = t / gsz, first_m = grp * GM;
    const int gm = (ntm - first_m < GM) ? (ntm - first_m) : GM;
    const int tm = first_m + (t % gsz) % gm, tn = (t % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;

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

    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];   // two staging register sets: tiles kt+1 and kt+2 in flight
    auto load_tile = [&](int k0, f32x4 (&ra)[4], f32x4 (&rb)[4]) {
        if (A_KMAJOR) {
            const int kq = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load4_guard<VEC>(p.A, a_rows[i], kq, kend);
        } else {
            const int mq = m0 + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int k = k0 + (tid >> 5) + 8 * i;
                const float* row = (k < kend) ? p.A + (int64_t)map_row_perm(p.amap, k) * p.lda : nullptr;
                ra[i] = load4_guard<VEC>(p.A, row, mq, p.M);
            }
        }
        if (B_KMAJOR) {
            const int kq = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = load4_guard<VEC>(p.B, b_rows[i], kq, kend);
        } else {
            const int nq = n0 + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int k = k0 + (tid >> 5) + 8 * i;
                const float* row = (k < kend) ? p.B + (int64_t)map_row_perm(p.bmap, k) * p.ldb : nullptr;
                rb[i] = load4_guard<VEC>(p.B, row, nq, p.N);
            }
        }
    };
    auto store_tile = [&](int buf, const f32x4 (&ra)[4], const f32x4 (&rb)[4]) {
        float* sA = smem + buf * 2 * OPER_FLOATS;
        float* sB = sA + OPER_FLOATS;
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

    auto compute_tile = [&](int buf) {
        const float* sA = smem + buf * 2 * OPER_FLOATS;
        const float* sB = sA + OPER_FLOATS;
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
    };

    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (NBUF == 1 && GPF == 1) {
        // One LDS image, one staging register set: stage -> barrier -> request tile kt+1 -> MFMAs -> barrier.
        // With 3-4 workgroups per CU sharing the matrix pipes, one tile of lookahead covers the load latency
        // (80 VGPRs keep 4 workgroups resident; the two-set form below needs ~200 and halves residency).
        load_tile(kbeg, ra0, rb0);
        for (int kt = 0; kt < nkt; ++kt) {
            store_tile(0, ra0, rb0);
            __syncthreads();
            load_tile(kbeg + (kt + 1) * BK, ra0, rb0);       // past-the-end tiles read the zero block
            compute_tile(0);
            __syncthreads();
        }
    } else if (NBUF == 1) {
        // One LDS image, two register sets: tile kt is staged from the set requested TWO iterations earlier;
        // unrolled by two so both sets are statically indexed.  Best for a workgroup alone on its CU.
        load_tile(kbeg, ra0, rb0);
        load_tile(kbeg + BK, ra1, rb1);
        for (int kt = 0; kt < nkt; kt += 2) {
            store_tile(0, ra0, rb0);
            __syncthreads();
            load_tile(kbeg + (kt + 2) * BK, ra0, rb0);
            compute_tile(0);
            __syncthreads();
            if (kt + 1 < nkt) {
                store_tile(0, ra1, rb1);
                __syncthreads();
                load_tile(kbeg + (kt + 3) * BK, ra1, rb1);
                compute_tile(0);
                __syncthreads();
            }
        }
    } else {
        // two LDS images, ONE barrier per k-tile (kept as an experiment; measured slower than the form above)
        load_tile(kbeg, ra0, rb0);
        store_tile(0, ra0, rb0);
        load_tile(kbeg + BK, ra0, rb0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            compute_tile(kt & 1);
            if (kt + 1 < nkt) {
                store_tile((kt + 1) & 1, ra0, rb0);
                load_tile(kbeg + (kt + 2) * BK, ra0, rb0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= p.M) continue;
            if (p.slabs) {   // split-K partial: plain [M][N] slab, combined by splitk_reduce_kernel
                float* srow = p.slabs + ((int64_t)blockIdx.y * p.M + m) * p.N;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int n = n0 + wn * 64 + ni * 32 + li;
                    if (n < p.N) srow[n] = acc[mi][ni][r];
                }
                continue;
            }
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

// C[map(m)][n] (+)= sum_s slabs[s][m][n] (+ bias[n]), fixed summation order (deterministic: s = 0, 1, ... whatever the grid).
// One float4 of a row per thread; the slab loop runs in groups of four whose loads are all issued before the first add (a
// dependent load-add chain per slab left the kernel at 1.1-1.7 TB/s: 12 launches, 0.58 ms of a C2 step), 16-byte stores.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit, int M, int N, float* __restrict__ C,
                                                            int64_t ldc, RowMap cmap, const float* __restrict__ bias, int accumulate, int vec_in,
                                                            int vec_c) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one float4 of a row
    const int nq = (N + 3) / 4;
    if (q >= (int64_t)M * nq) return;
    const int m = (int)(q / nq), n = (int)(q % nq) * 4;
    const int64_t MN = (int64_t)M * N;
    const float* src = slabs + (int64_t)m * N + n;
    float* crow = C + (int64_t)map_row(cmap, m) * ldc;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec_in) {      // N % 4 == 0, slabs and bias 16-byte aligned
        f32x4 cv = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
        if (accumulate && vec_c) cv = *reinterpret_cast<const f32x4*>(crow + n);      // (requested with the first slabs)
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
        for (int s = 0; s < nsplit; s += 4) {
            f32x4 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = (s + j < nsplit) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)(s + j) * MN)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s + j < nsplit) v += x[j];
        }
        if (bias) v += bv;
        if (vec_c) {
            if (accumulate) v += cv;
            *reinterpret_cast<f32x4*>(crow + n) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) crow[n + j] = accumulate ? v[j] + crow[n + j] : v[j];
        }
        return;
    }
    for (int s = 0; s < nsplit; ++s)
        for (int j = 0; j < 4; ++j)
            if (n + j < N) v[j] += src[s * MN + j];
    for (int j = 0; j < 4; ++j) {
        if (n + j >= N) break;
        float o = v[j];
        if (bias) o += bias[n + j];
        if (accumulate) o += crow[n + j];
        crow[n + j] = o;
    }
}

int splitk_reduce(hipStream_t stream, const float* slabs, int nsplit, int M, int N, float* C, int64_t ldc, RowMap cmap,
                  const float* bias, bool accumulate) {
    const int64_t nq = (int64_t)M * ((N + 3) / 4);
    // 16-byte accesses where the operands allow them (every row start of C = C + map(m) * ldc)
    const int vec_in = (N % 4 == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) ? 1 : 0;
    const int vec_c = (vec_in && vec_ok(C, ldc)) ? 1 : 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, slabs, nsplit, M,
                       N, C, ldc, cmap, bias, accumulate ? 1 : 0, vec_in, vec_c);
    S2VT_LAUNCH_CHECK("splitk_reduce_kernel");
    return 0;
}

int gemm_f32(hipStream_t stream, bool a_kmajor, bool b_kmajor, int M, int N, int K,
             const float* A, int64_t lda, RowMap amap, const float* B, int64_t ldb, RowMap bmap,
             float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate,
             float* splitk_ws, size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    S2VT_REQUIRE(K >= 0 && A && B && C, "gemm_f32: bad arguments (M=%d N=%d K=%d)", M, N, K);
    S2VT_REQUIRE(a_kmajor || !b_kmajor, "gemm_f32: A^T * B^T form is not used by the S2VT path");
    S2VT_REQUIRE((a_kmajor || !amap.idx) && (b_kmajor || !bmap.idx),
                 "gemm_f32: a gather index is only supported on operands whose stored rows are m / n (gather k-rows "
                 "with gather_rows_f32 first)");
    GemmArgs p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda; p.amap = amap;
    p.B = B; p.ldb = ldb; p.bmap = bmap;
    p.C = C; p.ldc = ldc; p.cmap = cmap;
    p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    // vector path: every 16-B load of both operands is aligned and never straddles the end of a stored row
    const bool vec = vec_ok(A, lda) && vec_ok(B, ldb) && ((a_kmajor ? K : M) % 4 == 0) && ((b_kmajor ? K : N) % 4 == 0);
    p.vecA = p.vecB = vec ? 1 : 0;
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    // Small grids (< 2 workgroups per CU) leave the matrix pipes idle during every staging phase: split K so
    // that ~3 workgroups per CU are in flight, partial tiles to slabs, fixed-order reduce (deterministic).
    // The grid is sized in units of 256 CUs: a workgroup count that is not close to a multiple of 256 (with at
    // least ~3 per CU) leaves CUs idle during the last round.  Pick the smallest K split that fills >= 92 % of
    // its last round; slices write slabs that are combined in a fixed order (deterministic).
    int nsplit = 1;
    if (splitk_ws && tiles < 1024 && K >= 8 * BK) {
        double best_eff = 0.0;
        for (int n = 1; n <= 8; ++n) {
            if (n > 1 && (K / n < 4 * BK || (size_t)n * M * N > splitk_ws_floats)) break;
            const int total = tiles * n;
            const double rounds = total / 256.0;
            double eff = rounds / (double)((total + 255) / 256);
            if (total < 768) eff *= total / 768.0;          // fewer than 3 workgroups per CU: latency exposed
            if (eff > best_eff + 0.02) { best_eff = eff; nsplit = n; }
            if (eff >= 0.92) break;
        }
    }
    p.ksplit = (nsplit > 1) ? cdiv(cdiv(K, nsplit), BK) * BK : (K > 0 ? K : 1);
    if (nsplit > 1) nsplit = cdiv(K, p.ksplit);
    p.slabs = (nsplit > 1) ? splitk_ws : nullptr;
    const dim3 grid(cdiv(tiles, 8) * 8, nsplit);
#define S2VT_GEMM_LAUNCH(AK, BK_, V) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, V>), grid, dim3(256), 0, stream, p)
    if (a_kmajor && b_kmajor) { if (vec) S2VT_GEMM_LAUNCH(true, true, true); else S2VT_GEMM_LAUNCH(true, true, false); }
    else if (a_kmajor) { if (vec) S2VT_GEMM_LAUNCH(true, false, true); else S2VT_GEMM_LAUNCH(true, false, false); }
    else { if (vec) S2VT_GEMM_LAUNCH(false, false, true); else S2VT_GEMM_LAUNCH(false, false, false); }
#undef S2VT_GEMM_LAUNCH
    S2VT_LAUNCH_CHECK("gemm_f32_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
