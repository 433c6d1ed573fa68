// GEMMs on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate): the arithmetic and the
// dispatch; the production kernels live in gemm_x3.hip (3 planes) and gemm_b1.hip (1 plane).
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
//
// gemm_bf16_nt_kernel below is the first-generation register-staged kernel (128x128 tile, 4 waves x (2x2) MFMA tiles,
// padded LDS rows, register prefetch, XCD-aware tile order, deterministic split-K).  It is kept for the 1-plane row
// layout as the A/B reference of gemm_b1.hip (S2VT_B1_OLD=1); its 3-plane form was superseded together with the row
// layout of the planes (724 us vs 620 us on the logits shape).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HBM_ = 128, HBN = 128;
// Packed plane layout (written by split.hip): element (row r, k, plane pl) of an operand lives at
//   r * ld + (k / 32) * (32 * NP) + pl * 32 + (k % 32),   ld = NP * Kpad, Kpad = K rounded up to 32 (zero filled)
// so the NP planes of one 32-wide k chunk of a row are adjacent: a k-tile reads NP*64 contiguous bytes per row
// (192 B for NP = 3) instead of NP scattered 64-B segments.
template <int NP, int WM> struct BfCfg {
    static constexpr int TM = 64 * WM;                     // tile rows: WM x 2 waves, each 64x64
    static constexpr int NTH = 128 * WM;                   // threads
    static constexpr int BK = (NP == 3) ? 32 : 64;         // k per tile (NP = 3: LDS budget; NP = 1: full 128-B lines)
    static constexpr int ROWB = NP * BK * 2;               // data bytes per tile row
    static constexpr int HROW = ROWB + 16;                 // + 16 B pad: odd number of 16-B slots, conflict-free b128
    static constexpr int PPR = ROWB / 16;                  // 16-B pieces per row
    static constexpr int NPA = TM * PPR / NTH;             // pieces per thread, A operand
    static constexpr int NPB = HBN * PPR / NTH;            // pieces per thread, B operand
    static constexpr int IMGA = TM * HROW, IMGB = HBN * HROW;
};

struct GemmBfArgs {
    int M, N, K;
    const unsigned short* A; int64_t lda;     // packed planes, lda = NP * Kpad elements
    const unsigned short* B; int64_t ldb;
    float* C; int64_t ldc; RowMap cmap;
    const float* bias;
    int accumulate;
    int ksplit;
    float* slabs;
};

template <int NP, int WM>
__global__ __launch_bounds__(128 * WM) void gemm_bf16_nt_kernel(GemmBfArgs p) {
    typedef BfCfg<NP, WM> Cf;
    constexpr int HBK = Cf::BK, HROW = Cf::HROW, PPR = Cf::PPR, NPA = Cf::NPA, NPB = Cf::NPB, NTH = Cf::NTH, TM = Cf::TM;
#ifndef S2VT_X3_LDS_PAD
#define S2VT_X3_LDS_PAD 0
#endif
    __shared__ __attribute__((aligned(16))) unsigned char smem[Cf::IMGA + Cf::IMGB + S2VT_X3_LDS_PAD];
    unsigned char* sA = smem;
    unsigned char* sB = smem + Cf::IMGA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = (p.N + HBN - 1) / HBN, ntm = (p.M + TM - 1) / TM;
    const int nwg = ntm * ntn, cpx = (nwg + 7) >> 3;
    const int t = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= cpx || t >= nwg) return;
    constexpr int GM = (WM == 2) ? 8 : 4;
    const int gsz = GM * ntn, grp = t / gsz, first_m = grp * GM;
    const int gm = (ntm - first_m < GM) ? (ntm - first_m) : GM;
    const int tm = first_m + (t % gsz) % gm, tn = (t % gsz) / gm;
    const int m0 = tm * TM, n0 = tn * HBN;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;      // multiples of HBK (K is padded)

    // staging: rows x PPR 16-B pieces per operand; thread -> pieces tid + NTH i; piece q: row q / PPR, byte column
    // (q % PPR) * 16 of the row's contiguous NP*BK*2-byte run -> consecutive lanes sweep whole rows
    const unsigned short* a_src[NPA];
    const unsigned short* b_src[NPB];
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int q = tid + NTH * i, row = q / PPR;
        a_src[i] = (m0 + row < p.M) ? p.A + (int64_t)(m0 + row) * p.lda + (q % PPR) * 8 : nullptr;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int q = tid + NTH * i, row = q / PPR;
        b_src[i] = (n0 + row < p.N) ? p.B + (int64_t)(n0 + row) * p.ldb + (q % PPR) * 8 : nullptr;
    }

    u32x4 ra0[NPA], rb0[NPB], ra1[NPA], rb1[NPB];   // two staging sets: tiles kt+1 and kt+2 in flight
    auto load_tile = [&](int k0, u32x4 (&ra)[NPA], u32x4 (&rb)[NPB]) {
        const bool kin = k0 < kend;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const unsigned short* qa = (kin && a_src[i]) ? a_src[i] + (int64_t)k0 * NP : zero;
            ra[i] = *reinterpret_cast<const u32x4*>(qa);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const unsigned short* qb = (kin && b_src[i]) ? b_src[i] + (int64_t)k0 * NP : zero;
            rb[i] = *reinterpret_cast<const u32x4*>(qb);
        }
    };
    auto store_tile = [&](const u32x4 (&ra)[NPA], const u32x4 (&rb)[NPB]) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int q = tid + NTH * i;
            *reinterpret_cast<u32x4*>(sA + (q / PPR) * HROW + (q % PPR) * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int q = tid + NTH * i;
            *reinterpret_cast<u32x4*>(sB + (q / PPR) * HROW + (q % PPR) * 16) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto compute_tile = [&]() {
#pragma unroll
        for (int c = 0; c < HBK / 16; ++c) {
            // A operand lane map of 32x32x16 bf16: lane (r = l&31, h = l>>5) holds A[row r][k = 8h .. 8h+7].
            // byte offset of (plane pl, k16 block c, half h) inside a tile row: 32-chunk (c/2), then plane, then k
            bf16x8 a[NP][2], b[NP][2];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const int koff = (c >> 1) * (64 * NP) + pl * 64 + (c & 1) * 32 + lh * 16;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    a[pl][mi] = *reinterpret_cast<const bf16x8*>(sA + (wm * 64 + mi * 32 + li) * HROW + koff);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    b[pl][ni] = *reinterpret_cast<const bf16x8*>(sB + (wn * 64 + ni * 32 + li) * HROW + koff);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (NP == 3) {   // smallest terms first
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[1][ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[NP - 1][ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[1][ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
                    }
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
                }
        }
    };
    // The L2->LDS path of this kernel runs near the L2 bandwidth limit, so a load's latency exceeds one tile's MFMA
    // phase: tiles are requested TWO iterations ahead (two register sets, loop unrolled by two so both are
    // statically indexed).  LDS (61 KB at NP = 3) already limits residency to 2 workgroups/CU, so the extra
    // registers cost no occupancy.
    const int nkt = (kend - kbeg + HBK - 1) / HBK;
    load_tile(kbeg, ra0, rb0);
    load_tile(kbeg + HBK, ra1, rb1);               // past-the-end tiles read the zero block
    for (int kt = 0; kt < nkt; kt += 2) {
        store_tile(ra0, rb0);
        __syncthreads();
        load_tile(kbeg + (kt + 2) * HBK, ra0, rb0);
        compute_tile();
        __syncthreads();
        if (kt + 1 < nkt) {
            store_tile(ra1, rb1);
            __syncthreads();
            load_tile(kbeg + (kt + 3) * HBK, ra1, rb1);
            compute_tile();
            __syncthreads();
        }
    }

#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= p.M) continue;
            if (p.slabs) {
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

int splitk_reduce(hipStream_t stream, const float* slabs, int nsplit, int M, int N, float* C, int64_t ldc, RowMap cmap,
                  const float* bias, bool accumulate);

int gemm_bf16_nt(hipStream_t stream, int nplanes, int M, int N, int K, const unsigned short* A, int64_t lda,
                 const unsigned short* B, int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias,
                 bool accumulate, float* splitk_ws, size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    S2VT_REQUIRE(nplanes == 1 || nplanes == 3, "gemm_bf16_nt: planes must be 1 or 3");
    if (nplanes == 3)           // split precision: the LDS-DMA kernel on the blocked plane layout (gemm_x3.hip)
        return gemm_x3(stream, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
    static int b1_old = -1;        // S2VT_B1_OLD=1: the superseded register-staged 128x128 kernel below
    if (b1_old < 0) { const char* e = getenv("S2VT_B1_OLD"); b1_old = e ? atoi(e) : 0; }
    if (!b1_old)
        return gemm_b1(stream, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
    const int BKc = 64;
    S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                 "gemm_bf16_nt: K must be the zero-padded multiple of 64 of the bf16 rows, rows 16-B aligned");
    GemmBfArgs p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda;
    p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.cmap = cmap; p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    // 256x128 tiles (8 waves) are kept as an option (S2VT_X3_WM=4); measured 5-10 % slower than 128x128 with two
    // workgroups per CU on every shape of the path.
    static int force_wm = -1;
    if (force_wm < 0) { const char* e = getenv("S2VT_X3_WM"); force_wm = e ? atoi(e) : 2; }
    const bool wm4 = (force_wm == 4);
    const int tiles = cdiv(M, wm4 ? 256 : HBM_) * cdiv(N, HBN);
    int nsplit = 1;
    if (splitk_ws && tiles < 1024 && K >= 16 * BKc) {
        double best_eff = 0.0;
        for (int n = 1; n <= 8; ++n) {
            if (n > 1 && (K / n < 8 * BKc || (size_t)n * M * N > splitk_ws_floats)) break;
            const int total = tiles * n;
            double eff = (total / 256.0) / (double)((total + 255) / 256);
            if (total < 512) eff *= total / 512.0;
            if (eff > best_eff + 0.02) { best_eff = eff; nsplit = n; }
            if (eff >= 0.92) break;
        }
    }
    p.ksplit = (nsplit > 1) ? cdiv(cdiv(K, nsplit), 64) * 64 : K;
    if (nsplit > 1) nsplit = cdiv(K, p.ksplit);
    p.slabs = (nsplit > 1) ? splitk_ws : nullptr;
    const dim3 grid(cdiv(tiles, 8) * 8, nsplit);
    if (wm4) hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, 4>), grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, 2>), grid, dim3(256), 0, stream, p);
    S2VT_LAUNCH_CHECK("gemm_bf16_nt_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
