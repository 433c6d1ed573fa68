// Plain bf16 GEMM (BASELINE config 3: bf16 operands, fp32 accumulate) with LDS-DMA staging:
// C[M,N] (+)= A[M,K] · B[N,K]^T (+ bias), both operands k-major ROWS of bf16 (element (r, k) at r*ld + k, K % 64 == 0) -
// the layout the bf16 timestep kernels write, so no repacking pass sits between the recurrence and its batched GEMMs.
//
// Workgroup tile (64 MI) x 256, MI = 2..5 (128 / 192 / 256 / 320 rows: the host picks the height whose tile count fills whole
// rounds of the 256 compute units - 20480 x 1000 is 320 tiles of 256 rows = 1.25 rounds, but 256 tiles of 320 rows = one),
// k chunk 64 per stage: the stage image is [64 MI + 256 rows][128 B], filled by buffer_load_dwordx4 ... lds with one
// wave-instruction per 8 rows (8 lanes x 16 B = one full 128-B line per row).  An LDS-DMA writes lane-linear, so the XOR
// swizzle that keeps the fragment reads conflict-free is applied to the per-lane SOURCE offset instead: position p of row r
// holds the row's 16-B piece p ^ ((r >> 1) & 7); a reader wanting piece q of row r reads position q ^ ((r >> 1) & 7): the 16
// rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) then take the 16 different (row parity, (r >> 1) & 7) slots of
// the 256-B bank row.  Operand rows are addressed through buffer descriptors whose range ends with the tile's last valid
// row: lanes of rows past the operand's end are dropped by the range check (no clamping arithmetic, one 32-bit offset
// register per request, the k position in a scalar offset); what such rows leave in LDS only reaches outputs that are never
// stored.
// Two stages (128-144 KB LDS, one workgroup per CU), one barrier per stage: after the barrier that opens stage s every wave
// is done with stage s-1, whose slot takes the requests of stage s+1 while stage s is multiplied.  8 waves as 2x4, wave tile
// (32 MI) x 64: per k16 block MI + 2 ds_read_b128 feed 2 MI v_mfma_f32_32x32x16_bf16; reads are inline asm with hand-counted
// lgkmcnt, double-buffered by k16 block.
// PERSISTENT: a launch is at most one workgroup per compute unit; each walks its share of the tiles (XCD-aware order: the
// workgroups of one XCD work on neighbouring tiles that share operand panels in its L2) and the stage pipeline runs ON across
// tiles - the last stage of a tile requests the first stage of the next one, so that transfer lands under the epilogue's
// stores and no tile but the first pays a prologue.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

struct GemmB1Args {
    int M, N, K;                              // K: multiple of 64 (zero-padded rows)
    const unsigned short* A; int64_t lda;     // bf16 rows
    const unsigned short* B; int64_t ldb;
    float* C; int64_t ldc; RowMap cmap;
    const float* bias;
    int accumulate;
    int ksplit;                               // k extent of a split-K slice (blockIdx.y), multiple of 64
    float* slabs;
    int ntm, ntn;                             // tile grid of this launch's tile height
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t b1_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// TT = true (MI = 4): C[M,N] = X_A^T X_B with BOTH operands read transposed from bf16 ROW images X_A [K rows][lda >= M],
// X_B [K rows][ldb >= N] (k = the images' rows): the weight-gradient GEMMs dW = dG^T h, whose operands the persistent recurrence
// kernels already wrote as bf16 rows - no transposed copy of dG / dlogits / h is made.  Stage image per operand = [64 k rows]
// [256 columns] (512-B rows: one LDS-DMA instruction = two whole rows); the 32-byte column groups of row k are stored at position
// G ^ 2 (k & 3) (source-side swizzle again), and a fragment is two ds_read_b64_tr_b16 per lane (4 k rows x 16 columns each): the
// 32 lanes of a half-wave then read 4 rows x 64 B that fall on 256 different bytes of the bank row.
template <int MI, bool TT>
__global__ __launch_bounds__(512) void gemm_b1_kernel(GemmB1Args p) {
    static_assert(!TT || MI == 4, "the transposed-read variant is built for 256-column tiles of both images");
    constexpr int TM = 64 * MI, ROWS = TM + 256, STAGE = ROWS * 128, IMG_A = TM * 128, NREQ = MI + 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    // ---- this workgroup's share of the tiles: the XCD of blockIdx % 8 owns a contiguous chunk of the (grouped) tile order,
    // its gridDim / 8 workgroups walk that chunk side by side
    const int items = p.ntm * p.ntn;
    const int cpx = (items + 7) >> 3, gx = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int q_end = ((xcd + 1) * cpx < items) ? (xcd + 1) * cpx : items;
    int q = xcd * cpx + (int)(blockIdx.x >> 3);
    if (q >= q_end) return;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;
    const int nk = (kend - kbeg) >> 6;                 // k64 stages per tile
    auto tile_of = [&](int t, int& m0, int& n0) {
        constexpr int GM = 4;
        const int gsz = GM * p.ntn, grp = t / gsz, first_m = grp * GM;
        const int gm = (p.ntm - first_m < GM) ? (p.ntm - first_m) : GM;
        m0 = (first_m + (t % gsz) % gm) * TM;
        n0 = ((t % gsz) / gm) * 256;
    };

    // ---- loader role: request j of wave w covers stage rows 8w + 64j .. + 7 (j < MI: A rows of the tile, else B rows
    // 8w + 64(j - MI) ..): lane -> (row lane / 8, position lane % 8) <- piece (lane % 8) ^ ((row >> 1) & 7), and
    // (row >> 1) & 7 = (4w + lane / 16) & 7 for every j
    unsigned voa[MI], vob[4];
    if constexpr (TT) {
        // request j of wave w covers image rows 2w + 16j, + 1 of the stage (lane -> (row lane / 32, position lane % 32) <- 16-byte
        // piece (lane % 32) ^ 4 (row & 3)); (row & 3) = (2w + lane / 32) & 3 for every j
        const unsigned r = (unsigned)(2 * wave + (lane >> 5));
        const unsigned piece = (unsigned)((lane & 31) ^ (4 * (r & 3)));
#pragma unroll
        for (int j = 0; j < MI; ++j) voa[j] = (r + 16u * j) * (unsigned)(p.lda * 2) + piece * 16u;
#pragma unroll
        for (int j = 0; j < 4; ++j) vob[j] = (r + 16u * j) * (unsigned)(p.ldb * 2) + piece * 16u;
    } else {
        const unsigned piece = (unsigned)((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7));
        const unsigned r = (unsigned)(8 * wave + (lane >> 3));
#pragma unroll
        for (int j = 0; j < MI; ++j) voa[j] = (r + 64u * j) * (unsigned)(p.lda * 2) + piece * 16u;
#pragma unroll
        for (int j = 0; j < 4; ++j) vob[j] = (r + 64u * j) * (unsigned)(p.ldb * 2) + piece * 16u;
    }
    auto rsrc_a = [&](int m0) {
        if constexpr (TT) {     // the image's columns m0.. over the k range's rows (a lane past the image's end reads nothing)
            int64_t bytes = ((int64_t)(kend - kbeg) * p.lda - m0) * 2;
            if (bytes > 0xFFFFF000ll) bytes = 0xFFFFF000ll;
            return b1_rsrc(p.A + (int64_t)kbeg * p.lda + m0, (unsigned)bytes);
        }
        const int rows = (p.M - m0 < TM) ? p.M - m0 : TM;
        return b1_rsrc(p.A + (int64_t)m0 * p.lda + kbeg, (unsigned)((rows - 1) * (int)(p.lda * 2) + (kend - kbeg) * 2));
    };
    auto rsrc_b = [&](int n0) {
        if constexpr (TT) {
            int64_t bytes = ((int64_t)(kend - kbeg) * p.ldb - n0) * 2;
            if (bytes > 0xFFFFF000ll) bytes = 0xFFFFF000ll;
            return b1_rsrc(p.B + (int64_t)kbeg * p.ldb + n0, (unsigned)bytes);
        }
        const int rows = (p.N - n0 < 256) ? p.N - n0 : 256;
        return b1_rsrc(p.B + (int64_t)n0 * p.ldb + kbeg, (unsigned)((rows - 1) * (int)(p.ldb * 2) + (kend - kbeg) * 2));
    };
    // request j of a stage: source descriptors ra / rb, k64 stage `koff` of the k range (ksa / ksb bytes per stage: 128 along the rows,
    // or 64 image rows in the transposed-read variant), into ring slot `slot`
    const int ksa = TT ? (int)(p.lda * 128) : 128, ksb = TT ? (int)(p.ldb * 128) : 128;
#define B1_REQ(J, RA, RB, KOFF, SLOT)                                                                                       \
    if constexpr ((J) < NREQ) {                                                                                             \
        auto* dst = (__attribute__((address_space(3))) void*)(smem + (SLOT) * STAGE + wave * 1024 + (J) * 8192);            \
        if constexpr ((J) < MI) __builtin_amdgcn_raw_ptr_buffer_load_lds(RA, dst, 16, (int)voa[(J) < MI ? (J) : 0], (KOFF) * ksa, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(RB, dst, 16, (int)vob[(J) >= MI ? (J) - MI : 0], (KOFF) * ksb, 0, 0);         \
    }

    // fragment addresses: row r = tile row of lane ((r >> 1) & 7 == (li >> 1) & 7), k16 block c, half lh -> piece 2c+lh at
    // position (2c+lh) ^ s = (2c) ^ y with s = (li >> 1) & 7, y = lh ^ s
    const int y = lh ^ ((li >> 1) & 7);
    unsigned fa[4], fb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        fa[c] = lbase + (unsigned)((wm * 32 * MI + li) * 128 + (((2 * c) ^ y) * 16));
        fb[c] = lbase + (unsigned)(IMG_A + (wn * 64 + li) * 128 + (((2 * c) ^ y) * 16));
    }

    // TT: lane (li, lh): 16-lane group g = li / 16, row q = (li % 16) / 4 of a 4-row block, 8-byte part p = li % 4; tile X's 32-byte
    // column groups 2X + g sit at position (2X + g) ^ 2q of their row: 64 ((X ^ q)) + 32 g bytes
    unsigned fat[MI], fbt[2];
    {
        const int q_ = (li & 15) >> 2;
        const unsigned lane_part = (unsigned)((8 * lh + q_) * 512 + (li >> 4) * 32 + (li & 3) * 8);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fat[mi] = lbase + lane_part + (unsigned)(64 * ((wm * MI + mi) ^ q_));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fbt[ni] = lbase + lane_part + (unsigned)(IMG_A + 64 * ((wn * 2 + ni) ^ q_));
    }

    f32x16 acc[MI][2];

#define B1_RDT(DST, ADDR, OFF)                                                                                            \
    {                                                                                                                     \
        bf16x4 lo_, hi_;                                                                                                  \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo_) : "v"(ADDR), "n"(OFF));                            \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi_) : "v"(ADDR), "n"((OFF) + 2048));                   \
        DST = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                                                  \
    }
#define B1_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF));
#define B1_READ(FA, FB, C, SO)                                                                                        \
    if constexpr (TT) {                                                                                               \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) { B1_RDT(FA[mi], fat[mi] + (SO), (C) * 8192) }             \
        B1_RDT(FB[0], fbt[0] + (SO), (C) * 8192) B1_RDT(FB[1], fbt[1] + (SO), (C) * 8192)                            \
    } else {                                                                                                          \
        const unsigned a_ = fa[C] + (SO), b_ = fb[C] + (SO);                                                          \
        B1_RD(FA[0], a_, 0) B1_RD(FA[1], a_, 4096)                                                                    \
        if constexpr (MI > 2) { B1_RD(FA[MI > 2 ? 2 : 0], a_, 8192) }                                                 \
        if constexpr (MI > 3) { B1_RD(FA[MI > 3 ? 3 : 0], a_, 12288) }                                                \
        if constexpr (MI > 4) { B1_RD(FA[MI > 4 ? 4 : 0], a_, 16384) }                                                \
        B1_RD(FB[0], b_, 0) B1_RD(FB[1], b_, 4096)                                                                    \
    }
    // wait until at most N fragment reads are outstanding; the empty statements tie the fragments to the wait
#define B1_WAIT(N, FA, FB)                                                                                            \
    {                                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) asm volatile("" : "+v"(FA[i_]));                            \
        asm volatile("" : "+v"(FB[0]));                                                                               \
        asm volatile("" : "+v"(FB[1]));                                                                               \
    }
#define B1_PROD(FA, FB)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)           \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[mi], FB[ni], acc[mi][ni], 0, 0, 0);
#define B1_FENCE __builtin_amdgcn_sched_barrier(0);

    // One k64 stage of the running pipeline (g counts stages across tiles: ring slot g & 1).  more (wave-uniform): a next
    // stage exists - of this tile (ra / rb = its descriptors, koff = its byte offset) or the first of the workgroup's next
    // tile - and is requested into the other slot while this one is multiplied.  wait_all = false: the stage follows an
    // epilogue of >= 8 MI store instructions of this wave, so "at most 8 MI operations outstanding" already says that the
    // requests issued BEFORE those stores have landed (the counter retires in issue order): the wave does not sit out its stores.
    auto stage = [&](int g, auto more_tag, bool more_rt, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int koff, bool wait_all) {
        const bool more = decltype(more_tag)::value || more_rt;       // (compile-time true inside a tile: no branch around the requests)
        // this wave's requests of stage g are the only loads outstanding; after the barrier all pieces of stage g are in and
        // every wave has finished reading stage g-1 (its slot takes stage g+1)
        if (wait_all) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(8 * MI) : "memory");
        asm volatile("s_barrier" ::: "memory");
        const unsigned so = (unsigned)((g & 1) * STAGE);
        const int ns = (g + 1) & 1;
        bf16x8 ax[MI], bx[2], ay[MI], by[2];
        // request set k of the next stage: waves 0-3 issue it BEHIND product group k, waves 4-7 (their SIMD partners) IN FRONT of
        // it - while one wave of a SIMD spends ~100-150 cycles per request at the issue port, its partner feeds the matrix pipe
        // (all eight waves behind the group measured 4-6 % slower on every shape of the config-3 step when this schedule was built;
        // the shipped kernel's per-shape table: profiles/round4_gemm_b1_shapes.txt)
        const bool early = more && wave >= 4, late = more && wave < 4;
#define B1_REQS(K) { B1_REQ(K, ra, rb, koff, ns) B1_REQ((K) + 4, ra, rb, koff, ns) if constexpr ((K) == 0) { B1_REQ(8, ra, rb, koff, ns) } }
        B1_FENCE
        constexpr int NRD = TT ? 2 * (MI + 2) : MI + 2;          // LDS read instructions per k16 block
        B1_READ(ax, bx, 0, so) B1_READ(ay, by, 1, so)
        if (early) B1_REQS(0)
        B1_WAIT(NRD, ax, bx) B1_PROD(ax, bx)
        if (late) B1_REQS(0)
        B1_FENCE
        B1_READ(ax, bx, 2, so)
        if (early) B1_REQS(1)
        B1_WAIT(NRD, ay, by) B1_PROD(ay, by)
        if (late) B1_REQS(1)
        B1_FENCE
        B1_READ(ay, by, 3, so)
        if (early) B1_REQS(2)
        B1_WAIT(NRD, ax, bx) B1_PROD(ax, bx)
        if (late) B1_REQS(2)
        B1_FENCE
        if (early) B1_REQS(3)
        B1_WAIT(0, ay, by) B1_PROD(ay, by)
        if (late) B1_REQS(3)
        B1_FENCE
#undef B1_REQS
    };

    int m0, n0;
    tile_of(q, m0, n0);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(m0), rb = rsrc_b(n0);
    {   // the launch's only prologue: stage 0 of the first tile
        B1_REQ(0, ra, rb, 0, 0) B1_REQ(1, ra, rb, 0, 0) B1_REQ(2, ra, rb, 0, 0) B1_REQ(3, ra, rb, 0, 0) B1_REQ(4, ra, rb, 0, 0)
        B1_REQ(5, ra, rb, 0, 0) B1_REQ(6, ra, rb, 0, 0) B1_REQ(7, ra, rb, 0, 0) B1_REQ(8, ra, rb, 0, 0)
    }
    bool wait_all = true;
    int s = 0;
    int qn = q + gx;
    int m1 = 0, n1 = 0;
    if (qn < q_end) tile_of(qn, m1, n1);
    __amdgpu_buffer_rsrc_t ra1 = rsrc_a(m1), rb1 = rsrc_b(n1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int g = 0;; ++g) {             // the stages of all of this workgroup's tiles, one after the other
        const bool has_next = qn < q_end;
        for (; s + 1 < nk; ++s, ++g) {
            stage(g, std::true_type{}, true, ra, rb, s + 1, wait_all);
            wait_all = true;
        }
        stage(g, std::false_type{}, has_next, ra1, rb1, 0, wait_all);      // the tile's last stage requests the next tile's first
        wait_all = true;

        // ---- epilogue (the next tile's first stage is in flight under these stores).  The 32x32 accumulator layout gives a
        // lane ONE column and 16 rows; stored as it stands that is 32 MI dword store instructions per wave, and a tile's
        // epilogue is bound by their issue (16 us of a 44-us tile at K = 1024: with a quarter of them, timing only, the K = 1000
        // shapes ran 20-25 % faster).  So every 4x4 block (registers 4j..4j+3 x the lanes of a quad) is transposed inside the
        // quad (DPP quad_perm, two butterfly rounds) and a lane stores FOUR consecutive columns of one row as 16 bytes: a wave
        // instruction then writes 8 rows x 128 B, a quarter of the instructions for the same bytes.
        const bool full_m = m0 + TM <= p.M;
        {
            // (the lane's coordinates are made opaque here: address arithmetic of the epilogue that does not depend on the tile
            // would otherwise be hoisted out of the tile loop and held in registers across the stage pipeline)
            int e_li = li, e_lh = lh;
            asm volatile("" : "+v"(e_li), "+v"(e_lh));
            const int t = e_li & 3;
            const bool odd = t & 1, hi = t & 2;
            const int ncol = n0 + wn * 64 + (e_li & ~3);                   // first of this lane's four columns (ni = 0)
            const bool vec = p.slabs ? ((p.N & 3) == 0 && (reinterpret_cast<uintptr_t>(p.slabs) & 15) == 0)
                                     : ((p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0);      // 16-byte rows
            f32x4 bv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if (p.bias && !p.slabs) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int n = ncol + ni * 32 + k;
                        bv[ni][k] = n < p.N ? p.bias[n] : 0.f;
                    }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * 32 * MI + mi * 32 + 8 * j + 4 * e_lh + t;      // this lane's row after the transpose
                    f32x4 v[2];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        float a0 = acc[mi][ni][4 * j], a1 = acc[mi][ni][4 * j + 1], a2 = acc[mi][ni][4 * j + 2], a3 = acc[mi][ni][4 * j + 3];
                        // round 1: lanes t <-> t ^ 1 exchange (a0, a1) and (a2, a3) crosswise; round 2: t <-> t ^ 2, (a0, a2) and (a1, a3)
                        float s, r;
                        s = odd ? a0 : a1; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));
                        a0 = odd ? r : a0; a1 = odd ? a1 : r;
                        s = odd ? a2 : a3; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));
                        a2 = odd ? r : a2; a3 = odd ? a3 : r;
                        s = hi ? a0 : a2; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));
                        a0 = hi ? r : a0; a2 = hi ? a2 : r;
                        s = hi ? a1 : a3; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));
                        a1 = hi ? r : a1; a3 = hi ? a3 : r;
                        v[ni] = f32x4{a0, a1, a2, a3};
                    }
                    if (m >= p.M) continue;
                    float* row = p.slabs ? p.slabs + ((int64_t)blockIdx.y * p.M + m) * p.N : p.C + (int64_t)map_row(p.cmap, m) * p.ldc;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int n = ncol + ni * 32;
                        f32x4 o = v[ni];
                        if (!p.slabs) { o[0] += bv[ni][0]; o[1] += bv[ni][1]; o[2] += bv[ni][2]; o[3] += bv[ni][3]; }
                        if (vec && n + 4 <= p.N) {
                            f32x4* q4 = reinterpret_cast<f32x4*>(row + n);
                            if (p.accumulate && !p.slabs) { const f32x4 c = *q4; o[0] += c[0]; o[1] += c[1]; o[2] += c[2]; o[3] += c[3]; }
                            *q4 = o;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (n + k < p.N) {
                                    float x = o[k];
                                    if (p.accumulate && !p.slabs) x += row[n + k];
                                    row[n + k] = x;
                                }
                        }
                    }
                }
            }
        }
        if (!has_next) break;
        // at least 8 MI vector-memory instructions followed the requests of the next tile's first stage when all rows are valid and
        // both column groups of the wave lie inside N: "at most 8 MI operations outstanding" then says those requests have landed
        // (the counter retires in issue order) without sitting out the stores
        wait_all = !(full_m && n0 + wn * 64 + 64 <= p.N);
        q = qn; m0 = m1; n0 = n1; ra = ra1; rb = rb1;
        qn = q + gx;
        if (qn < q_end) tile_of(qn, m1, n1);
        ra1 = rsrc_a(m1); rb1 = rsrc_b(n1);
        s = 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
#undef B1_REQ
#undef B1_RD
#undef B1_RDT
#undef B1_READ
#undef B1_WAIT
#undef B1_PROD
#undef B1_FENCE
}

int splitk_reduce(hipStream_t stream, const float* slabs, int nsplit, int M, int N, float* C, int64_t ldc, RowMap cmap,
                  const float* bias, bool accumulate);

// Per-tile cost model (us) of the launcher: a k64 stage of a (64 MI) x 256 tile and the tile's epilogue, measured with
// tools/bench_gemm_shapes.py under S2VT_B1_MI (the stage is co-limited by the matrix pipes and by the ~65 GB/s a compute unit
// takes in from L2: a 128-row tile moves 3/4 of the bytes of a 256-row tile for half of its products)
static const double kB1Stage[6] = {0, 0, 1.15, 1.50, 1.70, 2.25};
static const double kB1Epi[6] = {0, 0, 4.5, 6.0, 8.0, 10.0};

static int g_b1_force_mi = 0, g_b1_force_n = 0;
void gemm_b1_tune(int tile_rows, int nsplit) {
    g_b1_force_mi = (tile_rows >= 128 && tile_rows <= 320 && tile_rows % 64 == 0) ? tile_rows / 64 : 0;
    g_b1_force_n = nsplit > 0 ? nsplit : 0;
}

static int gemm_b1_impl(hipStream_t stream, bool tt, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
                       int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
                       size_t splitk_ws_floats);
int gemm_b1(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
            int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
            size_t splitk_ws_floats) {
    return gemm_b1_impl(stream, false, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
}
// C[M,N] (+)= X_A^T X_B from the bf16 ROW images X_A [K][lda >= pad(M)], X_B [K][ldb >= pad(N)]; K % 64 == 0
int gemm_b1_tt(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
               int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
               size_t splitk_ws_floats) {
    return gemm_b1_impl(stream, true, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
}
static int gemm_b1_impl(hipStream_t stream, bool tt, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
                       int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
                       size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    if (tt)
        S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N &&
                         (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                     "gemm_b1_tt: K (image rows) must be a multiple of 64, rows 16-B aligned and at least M / N columns long");
    else
    S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K &&
                     (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                 "gemm_b1: K must be the zero-padded multiple of 64 of the bf16 rows, rows 16-B aligned");
    S2VT_REQUIRE(lda < (1 << 21) && ldb < (1 << 21), "gemm_b1: row stride beyond the 32-bit offsets of a 320-row tile");
    GemmB1Args p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda;
    p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.cmap = cmap; p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    // option "cu_reserve" = n: plan the persistent grids for n compute units fewer.  A launch is sized to ONE workgroup per compute
    // unit with a static share of the tiles each; a long-lived foreign kernel on some of the units (a communication kernel of a
    // data-parallel run) makes the workgroups that find no unit wait for a whole share (multi-GPU).
    const int ncu = planned_compute_units();
    // s2vt_gemm_tune(1, tile_rows, nsplit): overrides of the time model (kernel tests run every tile height,
    // tools/bench_gemm_shapes.py calibrates the model with them)
    const int force_mi = g_b1_force_mi, force_n = g_b1_force_n;
    // tile height, split-K factor and grid by the time model: every workgroup walks ceil(its XCD's chunk / workgroups of the
    // XCD) tiles of nk stages + an epilogue; split-K adds the fixed-order slab combine ((n + 1) passes over M x N floats at
    // ~3.5 TB/s + a launch)
    const int ntn = cdiv(N, 256);
    int best_mi = 4, best_ns = 1, best_g = 8;
    double best = 1e30;
    // (transposed reads: a k slice of a row image behind one descriptor and int offsets stays below 2 GB - larger images are cut
    // into k slices, as in gemm_x3.hip)
    const int64_t ldmax = lda > ldb ? lda : ldb, kTTSpan = 0x7FFFF000ll;
    static const int order[4] = {4, 5, 3, 2};
    for (int oi = 0; oi < 4; ++oi) {
        const int mi = order[oi];
        if (tt ? mi != 4 : (force_mi && force_mi != mi)) continue;
        const int tiles = cdiv(M, 64 * mi) * ntn;
        for (int n = 1; n <= 16; ++n) {
            if (n > 1 && (!splitk_ws || K < 512 || K / n < 256 || (size_t)n * M * N > splitk_ws_floats)) break;
            if (force_n && n != force_n) continue;
            const int ks = cdiv(cdiv(K, n), 64) * 64, nn = cdiv(K, ks);
            if (nn != n) continue;
            if (tt && (int64_t)ks * ldmax * 2 >= kTTSpan) continue;     // (a k slice of a row image must fit the signed 32-bit offsets)
            int g = ncu / nn / 8 * 8;
            if (g < 8) g = 8;
            if (g > cdiv(tiles, 8) * 8) g = cdiv(tiles, 8) * 8;
            const int per_wg = cdiv(cdiv(tiles, 8), g / 8);
            const double rounds = (double)cdiv(g * nn, ncu);           // (more workgroups than compute units: they queue)
            const double t = rounds * per_wg * ((ks / 64) * kB1Stage[mi] + kB1Epi[mi]) + 3.0 +
                             (nn > 1 ? (nn + 1.0) * M * (double)N * 4.0 / 3.5e6 + 8.0 : 0.0);
            if (t < best * 0.98) { best = t; best_mi = mi; best_ns = nn; best_g = g; }
        }
    }
    if (best > 1e29) {      // (an override that no candidate met: one slice of 256-row tiles)
        S2VT_REQUIRE(!tt || (int64_t)K * ldmax * 2 < kTTSpan,
                     "gemm_b1_tt: a row image of %lld bytes needs k slices below 2 GB and split-K scratch for them (%zu floats given)",
                     (long long)((int64_t)K * ldmax * 2), splitk_ws_floats);
        best_mi = 4; best_ns = 1;
        best_g = cdiv(cdiv(M, 256) * ntn, 8) * 8;
        if (best_g > ncu) best_g = ncu;
    }
    p.ntm = cdiv(M, 64 * best_mi); p.ntn = ntn;
    p.ksplit = (best_ns > 1) ? cdiv(cdiv(K, best_ns), 64) * 64 : K;
    const int nsplit = (best_ns > 1) ? cdiv(K, p.ksplit) : 1;
    p.slabs = (nsplit > 1) ? splitk_ws : nullptr;
    const dim3 grid(best_g, nsplit);
    if (tt) {
        hipLaunchKernelGGL((gemm_b1_kernel<4, true>), grid, dim3(512), 0, stream, p);
    } else {
        switch (best_mi) {
            case 2: hipLaunchKernelGGL((gemm_b1_kernel<2, false>), grid, dim3(512), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((gemm_b1_kernel<3, false>), grid, dim3(512), 0, stream, p); break;
            case 5: hipLaunchKernelGGL((gemm_b1_kernel<5, false>), grid, dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_b1_kernel<4, false>), grid, dim3(512), 0, stream, p); break;
        }
    }
    S2VT_LAUNCH_CHECK("gemm_b1_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
