// Split-precision (3 bf16 planes, 6 plane products = fp32-equivalent, see gemm_bf16.hip) GEMM for gfx950, built around
// LDS-DMA: C[M,N] (+)= A[M,K] · B[N,K]^T (+ bias), operands in the BLOCKED plane layout written by split.hip.
//
// Blocked layout of an operand [rows][k] (ld = 3*kpad, kpad % 64 == 0, buffer holds cdiv(rows,64)*64 rows):
//   element (r, k, plane pl) at  (r/64)*(64*ld) + (k/16)*3072 + (pl*2 + (k%16)/8)*512 + (r%64)*8 + k%8   [bf16 units]
// i.e. for every 64-row block and every 16-wide k chunk a contiguous 6 KB record of six 1-KB "pieces"; piece
// (pl, half) holds, for the 64 rows in order, the 16 bytes the 32x32x16 MFMA wants from lane (row, half).  One
// global_load_lds_dwordx4 wave-instruction moves one piece (64 lanes x 16 B, contiguous in HBM and in LDS), so the
// tile is staged without touching a VGPR, without ds_write instructions and without any address arithmetic beyond a
// running pointer; fragment reads are 512-B contiguous per half-wave (conflict-free, no padding, no swizzle).
//
// Tile 256x256, k chunk 16 per stage (48 KB), 3-stage ring (144 KB of the 160-KB LDS, one workgroup per CU), 8 waves
// as 2(M) x 4(N), wave tile 128x64 = 4x2 MFMA tiles -> 18 ds_read_b128 feed 48 MFMAs per stage (0.375 reads per MFMA;
// the 128x128 / 64x64-per-wave form needs 0.5 and its LDS pipe was the limiter).  Stage s+2 is requested right
// after the barrier that opens stage s and stays in flight across the next barrier (counted vmcnt, raw s_barrier).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XT = 256;                    // tile rows = tile cols
constexpr int X_REC = 6144;                // bytes of one (64-row block, k16 chunk) record: 6 pieces x 1 KB
constexpr int X_NS = 3;                    // ring depth

struct GemmX3Args {
    int M, N, K;                              // K = padded k extent of this call (multiple of 64)
    const unsigned short* A; int64_t lda;     // blocked planes; row-block stride = 64 * lda elements
    const unsigned short* B; int64_t ldb;
    float* C; int64_t ldc; RowMap cmap;
    const float* bias;
    int accumulate;
    int ksplit;
    float* slabs;
    int m_base, m_tiles;                      // this launch covers m_tiles row tiles from row m_base (m_tiles == 0: all of M)
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MI = 32-row MFMA tiles per wave in M: 4 -> 256x256 workgroup tile (wave tile 128x64), 2 -> 128x256 (wave tile 64x64:
// twice the tiles for grids that would otherwise leave CUs idle or need split-K; 12 reads per 24 MFMAs instead of 18
// per 48).  The A operand fills MI 64-row records of a stage, the B operand always four.
template <int MI>
__global__ __launch_bounds__(512) void gemm_x3_kernel(GemmX3Args p) {
    constexpr int NRA = MI, TMR = 64 * MI, X_STAGE = (NRA + 4) * X_REC;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[X_NS * X_STAGE];
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = (p.N + XT - 1) / XT, ntm = p.m_tiles ? p.m_tiles : (p.M + TMR - 1) / TMR;
    const int nwg = ntm * ntn, cpx = (nwg + 7) >> 3;
    const int t = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= cpx || t >= nwg) return;
    constexpr int GM = 4;
    const int gsz = GM * ntn, grp = t / gsz, first_m = grp * GM;
    const int gm = (ntm - first_m < GM) ? (ntm - first_m) : GM;
    const int tm = first_m + (t % gsz) % gm, tn = (t % gsz) / gm;
    const int m0 = p.m_base + tm * TMR, n0 = tn * XT;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;
    const int nk = (kend - kbeg) >> 4;                 // k16 stages

    // loader role: waves 0..NRA-1 stream the A row-blocks of the tile, the next four waves the B row-blocks (wave index
    // = record index inside a stage; with MI = 2 waves 6 and 7 load nothing); a row-block past the operand's last one
    // is clamped onto it (its outputs are never stored)
    const bool loader = (NRA + 4 == 8) || wave < NRA + 4;      // compile-time true at MI = 4
    const unsigned char* src;
    {
        const bool isA = wave < NRA;
        const int nrb = ((isA ? p.M : p.N) + 63) >> 6;
        int rb = isA ? (m0 >> 6) + wave : (n0 >> 6) + (wave - NRA);
        rb = rb < 0 ? 0 : rb;
        rb = rb < nrb ? rb : nrb - 1;
        const unsigned short* base = isA ? p.A : p.B;
        const int64_t ld = isA ? p.lda : p.ldb;
        src = reinterpret_cast<const unsigned char*>(base + (int64_t)rb * 64 * ld) + (int64_t)(kbeg >> 4) * X_REC + lane * 16;
    }
    unsigned char* const ldst = smem + wave * X_REC;   // + stage * X_STAGE + piece * 1024 (+ lane * 16 by the DMA)
    auto request = [&](int s) {                        // stage s of this tile's k range -> ring slot s % 3
        const unsigned char* g = src + (int64_t)s * X_REC;
        unsigned char* l = ldst + (s % X_NS) * X_STAGE;
        if (!loader) return;
#pragma unroll
        for (int j = 0; j < 6; ++j) glds16(g + j * 1024, l + j * 1024);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment addresses inside a stage: wave row wm covers tile rows [wm*32*MI, +32*MI): A record wm*MI/2 + mi/2, row
    // (mi&1)*32 + li; B record NRA + wn, row ni*32 + li
    const int a_off = (wm * (MI / 2)) * X_REC + lh * 1024 + li * 16;
    const int b_off = (NRA + wn) * X_REC + lh * 1024 + li * 16;

    // One k16 stage.  MORE (compile time): stage s+2 exists and is requested during this stage.
    auto stage = [&](int s, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        // stage s has landed for this wave once at most the 6 requests of stage s+1 are outstanding; the barrier makes
        // that true for every wave's pieces and also says everyone is done reading slot (s+2)%3 (= stage s-1)
        if (MORE || s + 1 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned char* g2 = src + (int64_t)(s + 2) * X_REC;
        unsigned char* l2 = ldst + ((s + 2) % X_NS) * X_STAGE;
        const unsigned char* st = smem + (s % X_NS) * X_STAGE;
        bf16x8 a[3][MI], b[3][2];
        // Fragment reads are issued in the order the products consume them - (a1 b1) (a0 b2) (a2 b0) - as inline asm
        // with hand-counted lgkmcnt waits: all 8 waves read right after the barrier, so the 18th read of a wave returns
        // ~1150 cycles later (144 KB through a 128 B/clk LDS) but its 6th after ~400; hipcc would wait for all 18
        // before the first MFMA (lgkmcnt(0)), the counted waits start each product group when ITS operands are in.
        // (LDS returns in order; nothing else in the loop uses lgkmcnt.)
        const unsigned la = lbase + (unsigned)((s % X_NS) * X_STAGE + a_off), lb = lbase + (unsigned)((s % X_NS) * X_STAGE + b_off);
#define X3_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF));
#define X3_LDA(PL) X3_RD(a[PL][0], la, (PL) * 2048) X3_RD(a[PL][1], la, (PL) * 2048 + 512)                        \
    if constexpr (MI == 4) { X3_RD(a[PL][MI - 2], la, X_REC + (PL) * 2048) X3_RD(a[PL][MI - 1], la, X_REC + (PL) * 2048 + 512) }
#define X3_LDB(PL) X3_RD(b[PL][0], lb, (PL) * 2048) X3_RD(b[PL][1], lb, (PL) * 2048 + 512)
        // wait until at most N4 (MI = 4) / N2 (MI = 2) reads are outstanding; the operands tie the products to the wait
#define X3_WAIT(N4, N2, PA, PB)                                                                                          \
    if constexpr (MI == 4) asm volatile("s_waitcnt lgkmcnt(" #N4 ")" : "+v"(a[PA][0]), "+v"(a[PA][1]), "+v"(a[PA][MI - 2]),   \
                                        "+v"(a[PA][MI - 1]), "+v"(b[PB][0]), "+v"(b[PB][1]));                                  \
    else asm volatile("s_waitcnt lgkmcnt(" #N2 ")" : "+v"(a[PA][0]), "+v"(a[PA][1]), "+v"(b[PB][0]), "+v"(b[PB][1]));
        X3_LDA(1) X3_LDB(1) X3_LDA(0) X3_LDB(2) X3_LDA(2) X3_LDB(0)
        // six plane products, smallest terms first; product-major order keeps 8 independent MFMAs between two
        // updates of the same accumulator.  The DMA requests of stage s+2 are spread over the MFMA groups: issuing
        // one costs the wave ~100 cycles, which a running MFMA group hides
#define X3_PROD(PA, PB)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)           \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][mi], b[PB][ni], acc[mi][ni], 0, 0, 0);
#define X3_REQ(J) if (MORE && loader) glds16(g2 + (J) * 1024, l2 + (J) * 1024);
        X3_WAIT(12, 8, 1, 1) X3_PROD(1, 1) X3_REQ(0) X3_WAIT(6, 4, 0, 2) X3_PROD(0, 2) X3_REQ(1) X3_WAIT(0, 0, 2, 0) X3_PROD(2, 0) X3_REQ(2)
        X3_PROD(0, 1) X3_REQ(3) X3_PROD(1, 0) X3_REQ(4) X3_PROD(0, 0) X3_REQ(5)
#undef X3_WAIT
#undef X3_RD
#undef X3_LDA
#undef X3_LDB
        // pin the issue order: 18 fragment reads, then 6 x (8 MFMAs, 1 DMA request)
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * (MI + 2), 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MI, 0);
            if (MORE && MI == 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#undef X3_REQ
#undef X3_PROD
    };
    request(0);
    if (nk > 1) request(1);
    int s = 0;
    for (; s + 2 < nk; ++s) stage(s, std::true_type{});
    for (; s < nk; ++s) stage(s, std::false_type{});

#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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

// A, B: blocked 3-plane operands (see the header of this file); K = their common padded k extent for this call.
int gemm_x3(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
            int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
            size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= 3 * (int64_t)K && ldb >= 3 * (int64_t)K &&
                     (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                 "gemm_x3: K must be the zero-padded multiple of 64 of the blocked plane layout, operands 16-B aligned");
    GemmX3Args p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda;
    p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.cmap = cmap; p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    p.m_base = 0; p.m_tiles = 0;
    // One workgroup per CU.  Tile height (256 or 128 rows) and split-K factor are chosen by a time model: rounds of 256
    // workgroups x per-tile time + the fixed-order slab combine.  Per k unit of a tile ~0.14 us at 256x256 (2.2 us per
    // k16 stage) and ~0.092 us at 128x256 (measured: tools/bench_x3_split.py with S2VT_X3_TM); ~6 us per tile of prologue/epilogue; combine = (n + 1) passes over M x N
    // floats at ~3.5 TB/s + a launch.
    static int force_tm = -1;    // S2VT_X3_TM=128|256: experiment override
    if (force_tm < 0) { const char* e = getenv("S2VT_X3_TM"); force_tm = e ? atoi(e) : 0; }
    int nsplit = 1, tmr = 256;
    {
        double best = 1e30;
        for (int shape = 0; shape < 2; ++shape) {
            const int rows = shape ? 128 : 256;
            if (force_tm && force_tm != rows) continue;
            const double ck = shape ? 0.092 : 0.14;
            const int tiles = cdiv(M, rows) * cdiv(N, XT);
            for (int n = 1; n <= 16; ++n) {
                if (n > 1 && (!splitk_ws || K < 512 || K / n < 256 || (size_t)n * M * N > splitk_ws_floats)) break;
                const int ks = cdiv(cdiv(K, n), 64) * 64, nn = cdiv(K, ks);
                if (nn != n) continue;
                const double rounds = (double)cdiv(tiles * nn, 256);
                const double t = rounds * (ks * ck + 6.0) + (nn > 1 ? (nn + 1.0) * M * (double)N * 4.0 / 3.5e6 + 8.0 : 0.0);
                if (t < best * 0.97) { best = t; nsplit = nn; tmr = rows; }
            }
        }
    }
    // Tile quantisation without split-K: 320 tiles of 256 rows are two rounds of the chip with the second a quarter full.  Two
    // launches instead - `a` row tiles of 256 rows that fill whole rounds, the remaining rows as 128-row tiles - when the same
    // time model says so (gx1 / gxe of config 2: 20 x 16 tiles -> 16 x 16 of 256 rows + 8 x 16 of 128: 299 -> 249 us).
    static int mixed_on = -1;    // S2VT_X3_MIXED=0: off
    if (mixed_on < 0) { const char* e = getenv("S2VT_X3_MIXED"); mixed_on = e ? (atoi(e) != 0) : 1; }
    if (mixed_on && !force_tm && nsplit == 1 && tmr == 256) {
        const int ntn = cdiv(N, XT), ntm = cdiv(M, 256);
        const double c256 = K * 0.14 + 6.0, c128 = K * 0.092 + 6.0;
        const double t_single = (double)cdiv(ntm * ntn, 256) * c256;
        int best_a = 0;
        double best_t = t_single * 0.93;                 // (a second launch has to pay for itself)
        for (int a = 1; a < ntm; ++a) {
            const int rest = cdiv(M - a * 256, 128);
            const double t = (double)cdiv(a * ntn, 256) * c256 + (double)cdiv(rest * ntn, 256) * c128 + 2.0;
            if (t < best_t) { best_t = t; best_a = a; }
        }
        if (best_a > 0) {
            p.ksplit = K; p.slabs = nullptr;
            p.m_base = 0; p.m_tiles = best_a;
            hipLaunchKernelGGL(gemm_x3_kernel<4>, dim3(cdiv(best_a * ntn, 8) * 8, 1), dim3(512), 0, stream, p);
            p.m_base = best_a * 256; p.m_tiles = cdiv(M - best_a * 256, 128);
            hipLaunchKernelGGL(gemm_x3_kernel<2>, dim3(cdiv(p.m_tiles * ntn, 8) * 8, 1), dim3(512), 0, stream, p);
            S2VT_LAUNCH_CHECK("gemm_x3_kernel");
            return 0;
        }
    }
    const int tiles = cdiv(M, tmr) * cdiv(N, XT);
    static int force_n = -1;     // S2VT_X3_NSPLIT=n: experiment override (tools/bench_x3_split.py)
    if (force_n < 0) { const char* e = getenv("S2VT_X3_NSPLIT"); force_n = e ? atoi(e) : 0; }
    if (force_n > 0 && splitk_ws && (size_t)force_n * M * N <= splitk_ws_floats && K / force_n >= 64) nsplit = force_n;
    p.ksplit = (nsplit > 1) ? cdiv(cdiv(K, nsplit), 64) * 64 : K;
    if (nsplit > 1) nsplit = cdiv(K, p.ksplit);
    p.slabs = (nsplit > 1) ? splitk_ws : nullptr;
    const dim3 grid(cdiv(tiles, 8) * 8, nsplit);
    if (tmr == 256) hipLaunchKernelGGL(gemm_x3_kernel<4>, grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(gemm_x3_kernel<2>, grid, dim3(512), 0, stream, p);
    S2VT_LAUNCH_CHECK("gemm_x3_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
