// Plain bf16 GEMM (BASELINE config 3: bf16 operands, fp32 accumulate) with LDS-DMA staging:
// C[M,N] (+)= A[M,K] · B[N,K]^T (+ bias), both operands k-major ROWS of bf16 (element (r, k) at r*ld + k, K % 64 == 0) -
// the layout the bf16 timestep kernels write, so no repacking pass sits between the recurrence and its batched GEMMs.
//
// 256x256 tile, k chunk 64 per stage: the stage image is [512 rows][128 B], filled by global_load_lds_dwordx4 with
// one wave-instruction per 8 rows (8 lanes x 16 B = one full 128-B line per row).  An LDS-DMA writes lane-linear, so the
// XOR swizzle that keeps the fragment reads conflict-free is applied to the per-lane SOURCE address instead: position p
// of row r holds the row's 16-B piece p ^ ((r >> 1) & 7); a reader wanting piece q of row r reads position
// q ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) then take the 16 different
// (row parity, (r >> 1) & 7) slots of the 256-B bank row.  (The first version XORed with r & 7: rows 12 and 20 of a group
// met on one slot - PMC showed half of the kernel's LDS cycles as bank conflicts.)
// Two stages (128 KB LDS, one workgroup per CU), one barrier per stage: after the barrier that opens stage s every wave
// is done with stage s-1, whose slot takes the requests of stage s+1 while stage s is multiplied.  8 waves as 2x4,
// wave tile 128x64: per k16 block 6 ds_read_b128 feed 8 v_mfma_f32_32x32x16_bf16; reads are inline asm with
// hand-counted lgkmcnt, double-buffered by k16 block.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BT = 256;                    // tile rows = tile cols
constexpr int B_IMG = BT * 128;            // one operand image of a stage: 256 rows x 128 B
constexpr int B_STAGE = 2 * B_IMG;         // A image + B image = 64 KB

struct GemmB1Args {
    int M, N, K;                              // K: multiple of 64 (zero-padded rows)
    const unsigned short* A; int64_t lda;     // bf16 rows
    const unsigned short* B; int64_t ldb;
    float* C; int64_t ldc; RowMap cmap;
    const float* bias;
    int accumulate;
    int ksplit;
    float* slabs;
    unsigned long long* stamps; int stamp_block;    // timing experiments only
};

__device__ __forceinline__ void glds16b(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(512) void gemm_b1_kernel(GemmB1Args p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * B_STAGE];
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = (p.N + BT - 1) / BT, ntm = (p.M + BT - 1) / BT;
    const int nwg = ntm * ntn, cpx = (nwg + 7) >> 3;
    const int t = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= cpx || t >= nwg) return;
    constexpr int GM = 4;
    const int gsz = GM * ntn, grp = t / gsz, first_m = grp * GM;
    const int gm = (ntm - first_m < GM) ? (ntm - first_m) : GM;
    const int tm = first_m + (t % gsz) % gm, tn = (t % gsz) / gm;
    const int m0 = tm * BT, n0 = tn * BT;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;
    const int nk = (kend - kbeg) >> 6;                 // k64 stages

    // loader role: waves 0-3 fill the A image (64 rows each), waves 4-7 the B image; request j of a wave covers rows
    // 8j..8j+7 of its 64: lane -> (row lane/8, position lane%8) <- piece (lane%8) ^ ((row >> 1) & 7); rows past the operand's
    // end are clamped onto its last row (their outputs are never stored)
    const unsigned char* src[8];
    {
        const bool isA = wave < 4;
        const int nrows = isA ? p.M : p.N, r0 = (isA ? m0 : n0) + (wave & 3) * 64 + (lane >> 3);
        const unsigned short* base = isA ? p.A : p.B;
        const int64_t ld = isA ? p.lda : p.ldb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int r = r0 + 8 * j;
            r = r < nrows ? r : nrows - 1;
            // tile row 8j + lane/8 of this wave's 64: (row >> 1) & 7 = (4j + lane/16) & 7
            const int piece = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
            src[j] = reinterpret_cast<const unsigned char*>(base + (int64_t)r * ld + kbeg) + piece * 16;
        }
    }
    unsigned char* const ldst = smem + (wave >> 2) * B_IMG + (wave & 3) * 64 * 128;     // + stage*B_STAGE + j*1024

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment addresses: row r = tile row of lane ((r >> 1) & 7 == (li >> 1) & 7), k16 block c, half lh -> piece 2c+lh at
    // position (2c+lh) ^ s = (2c) ^ y with s = (li >> 1) & 7, y = lh ^ s
    const int y = lh ^ ((li >> 1) & 7);
    unsigned fa[4], fb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        fa[c] = lbase + (unsigned)((wm * 128 + li) * 128 + (((2 * c) ^ y) * 16));
        fb[c] = lbase + (unsigned)(B_IMG + (wn * 64 + li) * 128 + (((2 * c) ^ y) * 16));
    }

#define B1_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF));
#define B1_READ(FA, FB, C, SO)                                                                                        \
    B1_RD(FA[0], fa[C] + (SO), 0) B1_RD(FA[1], fa[C] + (SO), 4096) B1_RD(FA[2], fa[C] + (SO), 8192)                    \
    B1_RD(FA[3], fa[C] + (SO), 12288) B1_RD(FB[0], fb[C] + (SO), 0) B1_RD(FB[1], fb[C] + (SO), 4096)
#define B1_WAIT(N, FA, FB) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), \
                                        "+v"(FB[0]), "+v"(FB[1]));
#define B1_PROD(FA, FB)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)            \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[mi], FB[ni], acc[mi][ni], 0, 0, 0);
#define B1_FENCE __builtin_amdgcn_sched_barrier(0);

    // MORE: stage s+1 exists and is requested during stage s
    const int xon = (p.stamps && (int)blockIdx.x == p.stamp_block && blockIdx.y == 0) ? 1 : 0;
    auto stage = [&](int s, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int xrec = xon ? s : -1;
        XSTAMP(p.stamps, xrec, 0);
        // this wave's requests of stage s are the only ones outstanding; after the barrier all pieces of stage s are in
        // and every wave has finished reading stage s-1 (its slot takes stage s+1)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        XSTAMP(p.stamps, xrec, 1);
        asm volatile("s_barrier" ::: "memory");
        XSTAMP(p.stamps, xrec, 2);
        const unsigned so = (unsigned)((s & 1) * B_STAGE);
        unsigned char* l2 = ldst + ((s + 1) & 1) * B_STAGE;
        const int64_t g2 = (int64_t)(s + 1) * 128;
#define B1_REQ(J) if (MORE) glds16b(src[J] + g2, l2 + (J) * 1024);
        bf16x8 ax[4], bx[2], ay[4], by[2];
        B1_FENCE
        B1_READ(ax, bx, 0, so) B1_READ(ay, by, 1, so)
        B1_WAIT(6, ax, bx) XSTAMP(p.stamps, xrec, 3); B1_PROD(ax, bx) B1_REQ(0) B1_REQ(1) B1_FENCE
        XSTAMP(p.stamps, xrec, 4);
        B1_READ(ax, bx, 2, so)
        B1_WAIT(6, ay, by) B1_PROD(ay, by) B1_REQ(2) B1_REQ(3) B1_FENCE
        XSTAMP(p.stamps, xrec, 5);
        B1_READ(ay, by, 3, so)
        B1_WAIT(6, ax, bx) B1_PROD(ax, bx) B1_REQ(4) B1_REQ(5) B1_FENCE
        XSTAMP(p.stamps, xrec, 6);
        B1_WAIT(0, ay, by) B1_PROD(ay, by) B1_REQ(6) B1_REQ(7) B1_FENCE
        XSTAMP(p.stamps, xrec, 7);
#undef B1_REQ
    };
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16b(src[j], ldst + j * 1024);
        int s = 0;
        for (; s + 1 < nk; ++s) stage(s, std::true_type{});
        stage(s, std::false_type{});
    }
#undef B1_RD
#undef B1_READ
#undef B1_WAIT
#undef B1_PROD
#undef B1_FENCE

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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

#ifdef S2VT_EXPERIMENT_STAMPS
static unsigned long long* g_b1_stamps = nullptr;
static int g_b1_block = 0;
extern "C" int s2vt_experiment_set_b1_stamps(unsigned long long* buf, int block) { g_b1_stamps = buf; g_b1_block = block; return 0; }
#endif
int gemm_b1(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
            int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
            size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K &&
                     (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                 "gemm_b1: K must be the zero-padded multiple of 64 of the bf16 rows, rows 16-B aligned");
    GemmB1Args p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda;
    p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.cmap = cmap; p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    p.stamps = nullptr; p.stamp_block = -1;
#ifdef S2VT_EXPERIMENT_STAMPS
    p.stamps = g_b1_stamps; p.stamp_block = g_b1_block;
#endif
    const int tiles = cdiv(M, BT) * cdiv(N, BT);
    // split K by the same kind of time model as gemm_x3 (one sixth of its MFMA work per k unit)
    int nsplit = 1;
    if (splitk_ws && K >= 512) {
        double best = 1e30;
        for (int n = 1; n <= 16; ++n) {
            if (n > 1 && (K / n < 256 || (size_t)n * M * N > splitk_ws_floats)) break;
            const int ks = cdiv(cdiv(K, n), 64) * 64, nn = cdiv(K, ks);
            if (nn != n) continue;
            const double rounds = (double)cdiv(tiles * nn, 256);
            const double t = rounds * (ks * 0.035 + 6.0) + (nn > 1 ? (nn + 1.0) * M * (double)N * 4.0 / 3.5e6 + 8.0 : 0.0);
            if (t < best * 0.97) { best = t; nsplit = nn; }
        }
    }
    p.ksplit = (nsplit > 1) ? cdiv(cdiv(K, nsplit), 64) * 64 : K;
    if (nsplit > 1) nsplit = cdiv(K, p.ksplit);
    p.slabs = (nsplit > 1) ? splitk_ws : nullptr;
    const dim3 grid(cdiv(tiles, 8) * 8, nsplit);
    hipLaunchKernelGGL(gemm_b1_kernel, grid, dim3(512), 0, stream, p);
    S2VT_LAUNCH_CHECK("gemm_b1_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
