// Split-precision (3 bf16 planes, 6 plane products = fp32-equivalent, see gemm_bf16.hip) GEMM for gfx950, built around
// LDS-DMA: C[M,N] (+)= A[M,K] · B[N,K]^T (+ bias), operands in the BLOCKED plane layout written by split.hip.
//
// Blocked layout of an operand [rows][k] (ld = 3*kpad, kpad % 64 == 0, buffer holds cdiv(rows,64)*64 rows):
//   element (r, k, plane pl) at  (r/64)*(64*ld) + (k/16)*3072 + (pl*2 + (k%16)/8)*512 + (r%64)*8 + k%8   [bf16 units]
// i.e. for every 64-row block and every 16-wide k chunk a contiguous 6 KB record of six 1-KB "pieces"; piece
// (pl, half) holds, for the 64 rows in order, the 16 bytes the 32x32x16 MFMA wants from lane (row, half).  One
// LDS-DMA wave-instruction moves one piece (64 lanes x 16 B, contiguous in HBM and in LDS), so the tile is staged without
// touching a VGPR, without ds_write instructions and without address arithmetic: a record is addressed through a buffer
// descriptor of its row block, the piece and the k chunk through the instruction's SCALAR offset; fragment reads are 512-B
// contiguous per half-wave (conflict-free, no padding, no swizzle).
//
// Tile (64 MI) x 256, MI = 2..4 (the host picks the height whose tile count fills the compute units best), k chunk 16 per
// stage ((MI + 4) records of 6 KB), 3-stage ring (up to 144 KB of the 160-KB LDS, one workgroup per CU), 8 waves as 2(M) x
// 4(N), wave tile (32 MI) x 64 -> 3 (MI + 2) ds_read_b128 feed 12 MI MFMAs per stage.  Stage s+2 is requested while stage s
// is multiplied and stays in flight across the next barrier (counted vmcnt, raw s_barrier).
// PERSISTENT: a launch is at most one workgroup per compute unit; each walks its share of the tiles (XCD-aware order) and
// the ring runs ON across tiles - the last two stages of a tile request the first two of the next one, so no tile but the
// first pays a prologue and the transfers land under the epilogue's stores.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

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
    int ntm, ntn;                             // tile grid of this launch's tile height
};

// 4x4 transpose inside every quad of lanes: lane t of a quad holds (a0..a3) = column t of a 4x4 block whose rows are the four
// registers; afterwards it holds row t (two butterfly rounds of DPP quad_perm exchanges)
__device__ __forceinline__ void quad_transpose4(float& a0, float& a1, float& a2, float& a3, bool odd, bool hi) {
    float s, r;
    s = odd ? a0 : a1; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));
    a0 = odd ? r : a0; a1 = odd ? a1 : r;
    s = odd ? a2 : a3; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));
    a2 = odd ? r : a2; a3 = odd ? a3 : r;
    s = hi ? a0 : a2; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));
    a0 = hi ? r : a0; a2 = hi ? a2 : r;
    s = hi ? a1 : a3; r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));
    a1 = hi ? r : a1; a3 = hi ? a3 : r;
}

// MI = 32-row MFMA tiles per wave in M: workgroup tile (64 MI) x 256, wave tile (32 MI) x 64.  The A operand fills MI 64-row
// records of a stage, the B operand always four.
// TT = true: BOTH operands are read TRANSPOSED from the row images of their source tensors - C[M,N] = A^T B with A stored
// as the blocked planes of X_A [K rows][M columns] and B as those of X_B [K rows][N columns] (k = the images' ROWS: the
// weight-gradient GEMMs dW = dG^T h, whose operands the recurrence / the row split already wrote as row planes; no transposed
// twin of dG / dlogits / h is ever written).  A k16 stage is then a QUARTER of a 64-row block: per (plane, 32 columns) a 1-KB
// block [16 k rows][2 chunks x 2 halves x 16 B], gathered by ONE LDS-DMA instruction from four 256-B runs of the image (lane ->
// (row, chunk parity, half): constant per-lane offset, everything else in the scalar offset), and a fragment is two
// ds_read_b64_tr_b16 per lane (4 k rows x 16 columns each, transposed by the LDS hardware): a half-wave reads 4 rows x 64 B =
// 256 contiguous bytes, conflict-free.  Same bytes staged, twice the (half-size) fragment reads, same MFMAs, same epilogue.
template <int MI, bool TT>
__global__ __launch_bounds__(512) void gemm_x3_kernel(GemmX3Args p) {
    // (TT: the tile's 2 MI + 8 column pairs x 3 planes are 6 requests for each of MI + 4 loader waves, as in the NN form)
    constexpr int NRA = MI, TMR = 64 * MI, X_STAGE = (NRA + 4) * X_REC;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[X_NS * X_STAGE];
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    // ---- this workgroup's share of the tiles (as gemm_b1_kernel): the XCD of blockIdx % 8 owns a contiguous chunk of the
    // grouped tile order, its gridDim / 8 workgroups walk that chunk side by side
    const int items = p.ntm * p.ntn;
    const int cpx = (items + 7) >> 3, gx = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int q_end = ((xcd + 1) * cpx < items) ? (xcd + 1) * cpx : items;
    int q = xcd * cpx + (int)(blockIdx.x >> 3);
    if (q >= q_end) return;
    const int kbeg = blockIdx.y * p.ksplit;
    const int kend = (kbeg + p.ksplit < p.K) ? kbeg + p.ksplit : p.K;
    const int nk = (kend - kbeg) >> 4;                 // k16 stages per tile (a multiple of 4)
    auto tile_of = [&](int t, int& m0, int& n0) {
        constexpr int GM = 4;
        const int gsz = GM * p.ntn, grp = t / gsz, first_m = grp * GM;
        const int gm = (p.ntm - first_m < GM) ? (p.ntm - first_m) : GM;
        m0 = (first_m + (t % gsz) % gm) * TMR;
        n0 = ((t % gsz) / gm) * 256;
    };

    // ---- loader role: waves 0..NRA-1 stream the A row-blocks of the tile, the next four waves the B row-blocks (wave index
    // = record index inside a stage; with MI < 4 the last waves load nothing); a row-block past the operand's last one is
    // clamped onto it (its outputs are never stored).  Descriptor = the record stream of this wave's row block over the k
    // range; a request = (descriptor, lane * 16, scalar offset of stage and piece)
    const bool loader = wave < NRA + 4;
    auto rsrc_of = [&](int m0, int n0) {
        if constexpr (TT) {
            // waves 0-3: the A image's columns m0.., waves 4-7: the B image's columns n0..; descriptor = the image from the k
            // range's first row block and the tile's first column chunk on
            const bool isA = wave < NRA;
            const unsigned short* base = isA ? p.A : p.B;
            const int64_t ld = isA ? p.lda : p.ldb;
            const int c0 = (isA ? m0 : n0) >> 4;
            const int64_t off = (int64_t)(kbeg >> 6) * 64 * ld + (int64_t)c0 * (X_REC / 2);
            int64_t bytes = ((int64_t)((kend - kbeg) >> 6) * 64 * ld - (int64_t)c0 * (X_REC / 2)) * 2;
            if (bytes > 0xFFFFF000ll) bytes = 0xFFFFF000ll;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base + off), 0, (int)bytes, 0x00020000);
        }
        const bool isA = wave < NRA;
        const int nrb = ((isA ? p.M : p.N) + 63) >> 6;
        int rb = isA ? (m0 >> 6) + wave : (n0 >> 6) + (wave - NRA);
        rb = rb < nrb ? rb : nrb - 1;
        const unsigned short* base = isA ? p.A : p.B;
        const int64_t ld = isA ? p.lda : p.ldb;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base + (int64_t)rb * 64 * ld + (int64_t)(kbeg >> 4) * (X_REC / 2)), 0,
                                                 nk * X_REC, 0x00020000);
    };
    const int vlane = TT ? ((lane >> 1) & 1) * X_REC + (lane & 1) * 1024 + (lane >> 2) * 16 : lane * 16;
    const int64_t tt_ld2 = (wave < NRA ? p.lda : p.ldb) * 128;      // bytes between two 64-row blocks of this wave's image (TT)
    const int tt_wq = wave < NRA ? wave : wave - NRA;                // TT: this wave's column pairs 2 wq, 2 wq + 1 of its operand
    // piece J of stage `st` (of the tile behind descriptor R) -> ring slot SLOT.  TT: request J of a wave = (column pair 2 (w & 3) +
    // J / 3, plane J % 3) of its operand: image offset = column pair x 2 records + plane x 2 KB + row quarter x 256 B + row block
#define X3_REQ1(J, R, ST, SLOT)                                                                                           \
    if constexpr (TT) {                                                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(R, (__attribute__((address_space(3))) void*)(smem + (SLOT) * X_STAGE + (wave < NRA ? 0 : NRA * X_REC) + \
                                                     ((2 * tt_wq + (J) / 3) * 3 + (J) % 3) * 1024),                       \
                                                 16, vlane, (int)(((ST) >> 2) * tt_ld2) + (2 * tt_wq + (J) / 3) * 2 * X_REC + ((J) % 3) * 2048 + ((ST) & 3) * 256, 0, 0); \
    } else {                                                                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(R, (__attribute__((address_space(3))) void*)(smem + (SLOT) * X_STAGE + wave * X_REC + (J) * 1024), \
                                                 16, vlane, (ST) * X_REC + (J) * 1024, 0, 0);                             \
    }

    f32x16 acc[MI][2];

    // fragment addresses inside a stage: 32-row tile x = wm MI + mi of the workgroup tile sits in A record x / 2, rows
    // (x & 1) * 32 + li; B record NRA + wn, rows ni * 32 + li
    unsigned la[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int x = wm * MI + mi;
        la[mi] = lbase + (unsigned)((x >> 1) * X_REC + (x & 1) * 512 + lh * 1024 + li * 16);
    }
    const unsigned lb = lbase + (unsigned)((NRA + wn) * X_REC + lh * 1024 + li * 16);
    // TT: lane (li, lh) of a 32-column tile: 16-lane group g = li / 16 (chunk parity), row q = (li % 16) / 4 of a 4-row block, 8-byte
    // part p = li % 4 of the group's 32 bytes; k rows 8 lh + 4 j + q (j = 0, 1: the two transposed reads of a fragment)
    const unsigned tt_lane = (unsigned)((8 * lh + ((li & 15) >> 2)) * 64 + (li >> 4) * 32 + (li & 3) * 8);
    const unsigned lb_t = lbase + tt_lane + (unsigned)(NRA * X_REC + wn * 2 * 3072);
    unsigned lat[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) lat[mi] = lbase + tt_lane + (unsigned)((wm * MI + mi) * 3072);

    // One k16 stage of the running ring (g counts stages across tiles: slot g % 3).  more: stage g + 2 exists - of this tile or
    // of the workgroup's next one (descriptor r2, stage index s2 there) - and is requested while this one is multiplied;
    // nwait = how many of this wave's youngest vector-memory operations may still be outstanding when stage g must have landed
    // (the 6 requests of stage g + 1, plus the store instructions of an epilogue issued between them and now)
    // DEFERRED hi*hi group: the sixth product group of a stage (plane 0 x plane 0, whose operands are in registers) is issued at the
    // START of the next stage, right behind that stage's 3 (MI + 2) fragment reads - the 2 MI MFMAs per wave (both waves of a SIMD:
    // ~512 cycles) run while the reads cross the LDS pipe, where every wave of the workgroup used to sit idle for the ~400 cycles
    // until its first fragments were back.  The order of the MFMAs on each accumulator is unchanged (bitwise the same sums).
    bf16x8 a0p[MI], b0p[2];
    auto stage = [&](int g, auto more_tag, bool more_rt, __amdgpu_buffer_rsrc_t r2, int s2, int nwait, bool deferred) {
        const bool more = decltype(more_tag)::value || more_rt;       // (compile-time true inside a tile: no branch around the requests)
        if (nwait == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (nwait == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(6 + 8 * MI) : "memory");
        const int slot = g % X_NS, slot2 = (g + 2) % X_NS;
        const unsigned so = (unsigned)(slot * X_STAGE);
        bf16x8 a[3][MI], b[3][2];
        // Fragment reads are issued in the order the products consume them - (a1 b1) (a0 b2) (a2 b0) - as inline asm
        // with hand-counted lgkmcnt waits: all 8 waves read right after the barrier, so the last read of a wave returns
        // ~1150 cycles later (144 KB through a 128 B/clk LDS) but its first third after ~400; hipcc would wait for all of them
        // before the first MFMA (lgkmcnt(0)), the counted waits start each product group when ITS operands are in.
        // (LDS returns in order; nothing else in the loop uses lgkmcnt.)
#define X3_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF));
        // transposed fragment: two 8-byte reads (k rows +0..3, +4..7 of the lane's half) -> the 8 k values of the MFMA operand
#define X3_RDT(DST, ADDR, OFF)                                                                                               \
    {                                                                                                                        \
        bf16x4 lo_, hi_;                                                                                                     \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo_) : "v"(ADDR), "n"(OFF));                               \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi_) : "v"(ADDR), "n"((OFF) + 256));                       \
        DST = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                                                     \
    }
#define X3_LDA(PL)                                                                                                           \
    if constexpr (TT) { _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) { X3_RDT(a[PL][mi], lat[mi] + so, (PL) * 1024) } } \
    else { _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) { X3_RD(a[PL][mi], la[mi] + so, (PL) * 2048) } }
#define X3_LDB(PL)                                                                                                           \
    if constexpr (TT) { const unsigned b_ = lb_t + so; X3_RDT(b[PL][0], b_, (PL) * 1024) X3_RDT(b[PL][1], b_, (3 + (PL)) * 1024) } \
    else { const unsigned b_ = lb + so; X3_RD(b[PL][0], b_, (PL) * 2048) X3_RD(b[PL][1], b_, (PL) * 2048 + 512) }
        // wait until at most N reads are outstanding; the empty statements tie the fragments to the wait
#define X3_WAIT(N, PA, PB)                                                                                     \
    {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) asm volatile("" : "+v"(a[PA][i_]));                  \
        asm volatile("" : "+v"(b[PB][0]));                                                                     \
        asm volatile("" : "+v"(b[PB][1]));                                                                     \
    }
        __builtin_amdgcn_sched_barrier(0);
        X3_LDA(1) X3_LDB(1) X3_LDA(0) X3_LDB(2)
        if constexpr (!TT) { X3_LDA(2) X3_LDB(0) }      // (TT: 12 half-size reads per pair - the third pair follows the first product group,
                                                        //  the LGKM counter holds 15)
        // six plane products, smallest terms first; product-major order keeps 2 MI independent MFMAs between two updates of the
        // same accumulator.  The DMA requests of stage g + 2 are spread over the MFMA groups: issuing one costs the wave ~100
        // cycles, which a running MFMA group of its SIMD partner hides (the partner-staggered issue that gemm_b1_kernel uses -
        // waves 4-7 in front of a group, waves 0-3 behind it - measured neutral here, 4.047 vs 4.044 ms over the 14 shapes: one
        // request per 8 MI MFMAs is not what this kernel waits for)
#define X3_PROD(PA, PB)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)           \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][mi], b[PB][ni], acc[mi][ni], 0, 0, 0);
        const bool req = more && loader;
#define X3_GROUP(N, PA, PB, J, DOWAIT)                                     \
    if constexpr (DOWAIT) { X3_WAIT(N, PA, PB) }                           \
    X3_PROD(PA, PB)                                                        \
    if (req) { X3_REQ1(J, r2, s2, slot2) }                                 \
    __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);       // (the deferred group stays BEHIND the fragment reads: it is what covers their latency)
        if (deferred) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0p[mi], b0p[ni], acc[mi][ni], 0, 0, 0);
        }
        if (req) { X3_REQ1(5, r2, s2, slot2) }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TT) {
            X3_GROUP(2 * (MI + 2), 1, 1, 0, true)
            X3_LDA(2) X3_LDB(0)
            X3_GROUP(2 * (MI + 2), 0, 2, 1, true)
            X3_GROUP(0, 2, 0, 2, true)
        } else {
            X3_GROUP(2 * (MI + 2), 1, 1, 0, true)
            X3_GROUP(MI + 2, 0, 2, 1, true)
            X3_GROUP(0, 2, 0, 2, true)
        }
        X3_GROUP(0, 0, 1, 3, false)
        X3_GROUP(0, 1, 0, 4, false)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a0p[mi] = a[0][mi];
        b0p[0] = b[0][0]; b0p[1] = b[0][1];
#undef X3_GROUP
#undef X3_PROD
#undef X3_WAIT
#undef X3_RD
#undef X3_RDT
#undef X3_LDA
#undef X3_LDB
    };

    int m0, n0;
    tile_of(q, m0, n0);
    __amdgpu_buffer_rsrc_t r0 = rsrc_of(m0, n0);
    if (loader) {   // the launch's only prologue: stages 0 and 1 of the first tile
#pragma unroll
        for (int j = 0; j < 6; ++j) { X3_REQ1(j, r0, 0, 0) }
#pragma unroll
        for (int j = 0; j < 6; ++j) { X3_REQ1(j, r0, 1, 1) }
    }
    int s = 0;
    int qn = q + gx;
    int m1 = 0, n1 = 0;
    if (qn < q_end) tile_of(qn, m1, n1);
    __amdgpu_buffer_rsrc_t r1 = rsrc_of(m1, n1);
    int nwait = 6;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int g = 0;; ++g) {             // the stages of all of this workgroup's tiles, one after the other
        const bool has_next = qn < q_end;
        for (; s + 2 < nk; ++s, ++g) {
            stage(g, std::true_type{}, true, r0, s + 2, nwait, s > 0);
            nwait = 6;
        }
        stage(g, std::false_type{}, has_next, r1, 0, nwait, true);      // the tile's last two stages request the next tile's
        ++g;                                                            // first two
        stage(g, std::false_type{}, has_next, r1, 1, has_next ? 6 : 0, true);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)                                 // the last stage's deferred group
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0p[mi], b0p[ni], acc[mi][ni], 0, 0, 0);

        // ---- epilogue (the next tile's first stages are in flight under these stores).  The 32x32 accumulator layout gives a
        // lane ONE column and 16 rows; stored as it stands that is 32 MI dword store instructions per wave, and a tile's
        // epilogue is bound by their issue.  Every 4x4 block (registers 4j..4j+3 x the lanes of a quad) is transposed inside
        // the quad and a lane stores FOUR consecutive columns of one row as 16 bytes: a wave instruction then writes 8 rows x
        // 128 B, a quarter of the instructions for the same bytes.
        bool counted;
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
            // at least 8 MI vector-memory instructions follow the requests of the next tile's first stages: all rows valid and
            // both column groups of the wave inside N (the next stage's counted wait relies on that lower bound)
            counted = m0 + TMR <= p.M && n0 + wn * 64 + 64 <= p.N;
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
                        quad_transpose4(a0, a1, a2, a3, odd, hi);
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
        nwait = counted ? 6 + 8 * MI : 6;       // (6: everything but the youngest six operations - also correct, only later)
        q = qn; m0 = m1; n0 = n1; r0 = r1;
        qn = q + gx;
        if (qn < q_end) tile_of(qn, m1, n1);
        r1 = rsrc_of(m1, n1);
        s = 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
#undef X3_REQ1
}

int splitk_reduce(hipStream_t stream, const float* slabs, int nsplit, int M, int N, float* C, int64_t ldc, RowMap cmap,
                  const float* bias, bool accumulate);

// Per-tile cost model (us) of the launcher: a k16 stage of a (64 MI) x 256 tile and the tile's epilogue (tools/bench_gemm_shapes.py
// under S2VT_X3_MI); the stage is bound by the six plane products (12 MI MFMAs per wave)
static const double kX3Stage[5] = {0, 0, 1.35, 1.75, 2.15};
static const double kX3Epi[5] = {0, 0, 4.5, 6.0, 8.0};
static int g_x3_force_mi = 0, g_x3_force_n = 0;
void gemm_x3_tune(int tile_rows, int nsplit) {
    g_x3_force_mi = (tile_rows >= 128 && tile_rows <= 256 && tile_rows % 64 == 0) ? tile_rows / 64 : 0;
    g_x3_force_n = nsplit > 0 ? nsplit : 0;
}

static int gemm_x3_impl(hipStream_t stream, bool tt, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
                       int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
                       size_t splitk_ws_floats);
// A, B: blocked 3-plane operands (see the header of this file); K = their common padded k extent for this call.
int gemm_x3(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
            int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
            size_t splitk_ws_floats) {
    return gemm_x3_impl(stream, false, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
}
// C[M,N] (+)= X_A^T X_B: A / B = blocked 3-plane ROW images of X_A [K rows][M columns] (lda >= 3 pad64(M)) and X_B [K rows][N
// columns], both starting at a 64-row block; K = rows contracted (a multiple of 64)
int gemm_x3_tt(hipStream_t stream, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
               int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
               size_t splitk_ws_floats) {
    return gemm_x3_impl(stream, true, M, N, K, A, lda, B, ldb, C, ldc, cmap, bias, accumulate, splitk_ws, splitk_ws_floats);
}
static int gemm_x3_impl(hipStream_t stream, bool tt, int M, int N, int K, const unsigned short* A, int64_t lda, const unsigned short* B,
                       int64_t ldb, float* C, int64_t ldc, RowMap cmap, const float* bias, bool accumulate, float* splitk_ws,
                       size_t splitk_ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    if (tt) {
        S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= 3 * (int64_t)((M + 63) / 64 * 64) &&
                         ldb >= 3 * (int64_t)((N + 63) / 64 * 64) && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                     "gemm_x3_tt: K (image rows) must be a multiple of 64, the row images hold pad64(M) / pad64(N) columns");
    } else
    S2VT_REQUIRE(K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= 3 * (int64_t)K && ldb >= 3 * (int64_t)K &&
                     (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
                 "gemm_x3: K must be the zero-padded multiple of 64 of the blocked plane layout, operands 16-B aligned");
    GemmX3Args p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda;
    p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.cmap = cmap; p.bias = bias; p.accumulate = accumulate ? 1 : 0;
    // option "cu_reserve" = n: plan the persistent grids for n compute units fewer.  A launch is sized to ONE workgroup per compute
    // unit with a static share of the tiles each; a long-lived foreign kernel on some of the units (a communication kernel of a
    // data-parallel run) makes the workgroups that find no unit wait for a whole share (multi-GPU).
    const int ncu = planned_compute_units();
    // s2vt_gemm_tune(3, tile_rows, nsplit): overrides of the time model (kernel tests run every tile height, tools/bench_gemm_shapes.py)
    const int force_mi = g_x3_force_mi, force_n = g_x3_force_n;
    // One workgroup per CU.  Tile height, split-K factor and grid by the time model: every workgroup walks ceil(its XCD's chunk /
    // workgroups of the XCD) tiles of nk stages + an epilogue; split-K adds the fixed-order slab combine ((n + 1) passes over
    // M x N floats at ~3.5 TB/s + a launch)
    const int ntn = cdiv(N, 256);
    int best_mi = 4, best_ns = 1, best_g = 8;
    double best = 1e30;
    // Transposed reads address a k slice of a row image through ONE buffer descriptor and 32-bit offsets: a slice (ks image rows of
    // ld elements) must stay below 4 GB.  An image beyond that (dlogits from B = 768 on at V = 12000) is cut into k slices here -
    // the split-K path with its fixed-order combine - instead of being refused.
    const int64_t ldmax = lda > ldb ? lda : ldb, kTTSpan = 0xFFFFF000ll;
    static const int order[3] = {4, 3, 2};
    for (int oi = 0; oi < 3; ++oi) {
        const int mi = order[oi];
        if (force_mi && force_mi != mi) continue;
        const int tiles = cdiv(M, 64 * mi) * ntn;
        for (int n = 1; n <= 16; ++n) {
            if (n > 1 && (!splitk_ws || K < 512 || K / n < 256 || (size_t)n * M * N > splitk_ws_floats)) break;
            if (force_n && n != force_n) continue;
            const int ks = cdiv(cdiv(K, n), 64) * 64, nn = cdiv(K, ks);
            if (nn != n) continue;
            if (tt && (int64_t)ks * ldmax * 2 >= kTTSpan) continue;     // (a k slice of a row image must fit the 32-bit offsets)
            int g = ncu / nn / 8 * 8;
            if (g < 8) g = 8;
            if (g > cdiv(tiles, 8) * 8) g = cdiv(tiles, 8) * 8;
            const int per_wg = cdiv(cdiv(tiles, 8), g / 8);
            const double rounds = (double)cdiv(g * nn, ncu);
            const double t = rounds * per_wg * ((ks / 16) * kX3Stage[mi] + kX3Epi[mi]) + 3.0 +
                             (nn > 1 ? (nn + 1.0) * M * (double)N * 4.0 / 3.5e6 + 8.0 : 0.0);
            if (t < best * 0.98) { best = t; best_mi = mi; best_ns = nn; best_g = g; }
        }
    }
    if (best > 1e29) {      // (an override that no candidate met: one slice of 256-row tiles)
        S2VT_REQUIRE(!tt || (int64_t)K * ldmax * 2 < kTTSpan,
                     "gemm_x3_tt: a row image of %lld bytes needs k slices below 4 GB and split-K scratch for them (%zu floats given)",
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
        switch (best_mi) {
            case 2: hipLaunchKernelGGL((gemm_x3_kernel<2, true>), grid, dim3(512), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((gemm_x3_kernel<3, true>), grid, dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_x3_kernel<4, true>), grid, dim3(512), 0, stream, p); break;
        }
    } else {
        switch (best_mi) {
            case 2: hipLaunchKernelGGL((gemm_x3_kernel<2, false>), grid, dim3(512), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((gemm_x3_kernel<3, false>), grid, dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_x3_kernel<4, false>), grid, dim3(512), 0, stream, p); break;
        }
    }
    S2VT_LAUNCH_CHECK("gemm_x3_kernel");
    if (nsplit > 1) return splitk_reduce(stream, splitk_ws, nsplit, M, N, C, ldc, cmap, bias, accumulate);
    return 0;
}

}  // namespace s2vt
