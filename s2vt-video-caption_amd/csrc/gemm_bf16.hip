// Split-precision GEMM on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate).
//
// An fp32 value x is carried as NP bf16 "planes": x = p0 + p1 + p2 exactly to 24 mantissa bits
// (p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)).  A product a*b is then the sum of plane products;
// bf16*bf16 is exact in fp32 and the MFMA accumulates in fp32, so keeping the six products whose weight is
// >= 2^-16 of the leading one (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) reproduces an fp32 product to
// ~2^-23 relative — fp32-equivalent arithmetic at 16/6 = 2.7x the fp32-MFMA rate.  NP = 1 is the plain bf16 GEMM
// (storage bf16, accumulate fp32) of BASELINE config 3.
//
// C[M,N] (+)= A[M,K] · B[N,K]^T (+ bias): ONLY the "NT" form — both operands k-contiguous.  Layout changes
// (transposes, gathers, batch-major <-> time-major) are done by the memory-bound plane-splitting kernels
// (split.hip), which write each consumer's k-major planes once, so this kernel stays a pure streaming MFMA loop:
// 128x128x32 tile, 4 waves x (2x2) 32x32 MFMA tiles, padded LDS rows (80 B: conflict-free ds_read_b128),
// register prefetch of the next k-tile, XCD-aware grouped tile order, optional deterministic split-K.
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
template <int NP> struct BfCfg {
    static constexpr int BK = (NP == 3) ? 32 : 64;         // k per tile (NP = 3: 2 workgroups/CU by LDS; NP = 1: full lines)
    static constexpr int ROWB = NP * BK * 2;               // data bytes per tile row
    static constexpr int HROW = ROWB + 16;                 // + 16 B pad: odd number of 16-B slots, conflict-free b128
    static constexpr int PPR = ROWB / 16;                  // 16-B pieces per row
    static constexpr int NPC = HBM_ * PPR / 256;           // pieces per thread per operand
    static constexpr int IMG = HBM_ * HROW;                // bytes per operand image
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

template <int NP>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(GemmBfArgs p) {
    typedef BfCfg<NP> Cf;
    constexpr int HBK = Cf::BK, HROW = Cf::HROW, PPR = Cf::PPR, NPC = Cf::NPC;
#ifndef S2VT_X3_LDS_PAD
#define S2VT_X3_LDS_PAD 0
#endif
    // S2VT_X3_LDS_PAD > 0 (experiment): inflate the LDS footprint so fewer GEMM workgroups share a CU, leaving room
    // for timestep workgroups of the other pipeline lane to co-reside
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * Cf::IMG + S2VT_X3_LDS_PAD];
    unsigned char* sA = smem;
    unsigned char* sB = smem + Cf::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = (p.N + HBN - 1) / HBN, ntm = (p.M + HBM_ - 1) / HBM_;
    const int nwg = ntm * ntn, cpx = (nwg + 7) >> 3;
    const int t = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= cpx || t >= nwg) return;
    constexpr int GM = 8;
    const int gsz = GM * ntn, grp = t / gsz, first_m = grp * GM;
    const int gm = (ntm - first_m < GM) ? (ntm - first_m) : GM;
    const int tm = first_m + (t % gsz) % gm, tn = (t % gsz) / gm;
    const int m0 = tm * HBM_, n0 = tn * HBN;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;      // multiples of HBK (K is padded)

    // staging: 128 rows x PPR 16-B pieces per operand; thread -> pieces tid + 256 i; piece q: row q / PPR,
    // byte column (q % PPR) * 16 of the row's contiguous NP*BK*2-byte run -> consecutive lanes sweep whole rows
    const unsigned short* a_src[NPC];
    const unsigned short* b_src[NPC];
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int q = tid + 256 * i, row = q / PPR;
        a_src[i] = (m0 + row < p.M) ? p.A + (int64_t)(m0 + row) * p.lda + (q % PPR) * 8 : nullptr;
        b_src[i] = (n0 + row < p.N) ? p.B + (int64_t)(n0 + row) * p.ldb + (q % PPR) * 8 : nullptr;
    }

    u32x4 ra[NPC], rb[NPC];
    auto load_tile = [&](int k0) {
        const bool kin = k0 < kend;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const unsigned short* qa = (kin && a_src[i]) ? a_src[i] + (int64_t)k0 * NP : zero;
            const unsigned short* qb = (kin && b_src[i]) ? b_src[i] + (int64_t)k0 * NP : zero;
            ra[i] = *reinterpret_cast<const u32x4*>(qa);
            rb[i] = *reinterpret_cast<const u32x4*>(qb);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int q = tid + 256 * i;
            const int off = (q / PPR) * HROW + (q % PPR) * 16;
            *reinterpret_cast<u32x4*>(sA + off) = ra[i];
            *reinterpret_cast<u32x4*>(sB + off) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nkt = (kend - kbeg + HBK - 1) / HBK;
    load_tile(kbeg);
    for (int kt = 0; kt < nkt; ++kt) {
        store_tile();
        __syncthreads();
        load_tile(kbeg + (kt + 1) * HBK);      // past-the-end tiles read the zero block
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
        __syncthreads();
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
    const int BKc = (nplanes == 3) ? 32 : 64;
    S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= (int64_t)nplanes * K &&
                     ldb >= (int64_t)nplanes * K && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                 "gemm_bf16_nt: K must be the zero-padded multiple of 64 of the packed plane layout, rows 16-B aligned");
    GemmBfArgs p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda;
    p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.cmap = cmap; p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    const int tiles = cdiv(M, HBM_) * cdiv(N, HBN);
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
    if (nplanes == 3) hipLaunchKernelGGL((gemm_bf16_nt_kernel<3>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<1>), grid, dim3(256), 0, stream, p);
    S2VT_LAUNCH_CHECK("gemm_bf16_nt_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
