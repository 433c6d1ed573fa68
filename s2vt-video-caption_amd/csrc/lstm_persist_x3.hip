// Persistent SPLIT-PRECISION LSTM forward recurrence for gfx950 (BASELINE config 2: B = 64, fp32-equivalent arithmetic):
// the third member of the family lstm_persist.hip (bf16) / lstm_persist_f32.hip (exact-fp32 MFMA).  One launch runs a block of
// timesteps of one layer - or of both layers side by side (S2VTModel.py:67 / :77 -> nn.LSTM over the steps) - with every
// workgroup's slice of W_hh resident in registers and the same cross-workgroup hand-off protocol.
//
// Arithmetic.  The contraction h_{t-1} . W_hh^T runs on the bf16 matrix cores in the split-precision form of gemm_x3.hip:
// both operands are three bf16 planes (x = p0 + p1 + p2, 24 mantissa bits) and the six plane products >= 2^-16 are summed in the
// fp32 accumulator (h0w0, h0w1, h1w0, h1w1, h0w2, h2w0): fp32-equivalent (error of the fp32-MFMA kernel) at 6/16 of the exact-fp32
// MFMA's cycles - the exact-fp32 persistent kernel is MFMA-bound (4.6 us of K loop in an 8.6-us timestep).
//
// Decomposition.  Workgroup (rg, cs) = 32-row chains x 16 hidden units (64 gate columns {i,f,g,o} x 16: complete cells), FOUR
// waves, ONE workgroup per compute unit: W_hh[64 gate columns][K] as three planes is 384 KB - three quarters of the CU's register
// file - so a wave owns a k QUARTER (256 k) of both 32-column halves: 96 B operands of v_mfma_f32_32x32x16_bf16 = 384 registers
// (arch + accumulation VGPRs; one wave per SIMD may use all 512).  Each wave streams ITS k quarter of the three h_{t-1} plane
// images (32 rows x 256 k x 3 planes = 48 KB per sub-step) through a private 3-slot LDS ring of 12-KB chunks (k64 x 3 planes;
// LDS-DMA, XOR-swizzled like the bf16 kernel's image): the contraction has no workgroup barrier, three A-fragment reads feed
// twelve MFMAs, the reads of k16 step i+1 and the next chunk's requests are issued between the MFMA groups (one wave per SIMD
// issues in order: nothing may sit in a burst in front of the matrix pipe).  192 MFMAs per wave and sub-step = 6144 cycles,
// against 3 us of ingest (192 KB per sub-step at ~65 GB/s per CU).
// Hand-off: the epilogue (two adjacent cells per thread) splits h_t into planes; the three 32 x 32-B tiles leave as three 16-byte
// write-through (sc1) store instructions of wave 0, which drains and adds 1 to the chain's counter; consumers poll (bounded spin),
// barrier, then sc1 LDS-DMA loads; every step's image has its own address (see lstm_persist.hip for the protocol's argument).
#include <stdlib.h>
#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int X_SR = 32;                          // batch rows per sub-step (= per chain)
constexpr int X_UN = 16;                          // hidden units per workgroup (64 gate columns)
constexpr int X_NT = 256;                         // 4 waves = 4 k quarters
constexpr int X_PLANE = X_SR * 128;               // one plane of a k64 chunk: 32 rows x 128 B
constexpr int X_CHUNK = 3 * X_PLANE;              // 12 KB
constexpr int X_RING = 3 * X_CHUNK;               // per-wave ring: 3 slots
constexpr int X_RLD = 72;                         // row stride of a partial-sum tile (floats): conflict-free f32x2 epilogue reads
constexpr int X_HSM = 4 * X_RING;                 // bf16 h_t planes [3][32][16]
constexpr int X_MAXNS = 4;
constexpr int X_CST = X_HSM + 3 * X_SR * X_UN * 2;        // fp32 c_t of the workgroup's cells, per chain [32][16]
constexpr int X_LDS = X_CST + X_MAXNS * X_SR * X_UN * 4;  // 158720 B
constexpr int X_MAX_WG = 256;                     // design point: one workgroup per CU; capped by coresident_capacity() at launch
constexpr unsigned long long X_SPIN_TICKS = 100000000ull;      // 1 s of the 100-MHz wall clock

__device__ __forceinline__ void glds16x_sc1(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 16 /* sc1 */);
}
__device__ __forceinline__ bool spin_until_x(const unsigned int* cnt, unsigned int target) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned int v = __hip_atomic_load((gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > X_SPIN_TICKS) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}
#define X_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define X_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void seq_fwd_x3_body(const SeqFwdX3Args& p, const int bid, unsigned char* smem, int& s_flag) {
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index = k quarter (scalar: LDS-DMA targets, k offsets)
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, B = p.B, Kp = p.Kp;
    const int nC = (H + X_UN - 1) / X_UN;
    const int cs = bid % nC, rg = bid / nC;
    const int u0 = cs * X_UN, row0 = rg * p.RB;
    const int k0 = kq * 256;                            // this wave's k quarter [k0, k0 + 256)
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);

    // ---- this wave's W_hh slice: wreg[(plane * 2 + cw) * 16 + j] = B operand of (plane, 32-column half cw, k16 step j);
    //      gate column n = g*8 + uu of half cw <-> W_hh row g*H + (u0 + 8 cw + uu)
    bf16x8 wreg[96];
    {
        const int g = li >> 3;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int cw = 0; cw < 2; ++cw) {
                const int unit = u0 + cw * 8 + (li & 7);
                const unsigned short* wrow = p.wp + pl * p.wplane + ((int64_t)g * H + unit) * p.ldw + k0 + lh * 8;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned short* q = (unit < H && k0 + 16 * j < Kp) ? wrow + 16 * j : zero;
                    wreg[(pl * 2 + cw) * 16 + j] = *reinterpret_cast<const bf16x8*>(q);
                }
            }
        // the slice is complete in registers before the step loop (otherwise the waits for these loads land inside it)
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wreg[i]));
#pragma unroll
        for (int i = 32; i < 96; ++i) asm volatile("" : "+a"(wreg[i]));
    }

    // ---- loader role (each wave for itself): piece (plane, i) of a chunk = rows 8i..8i+7 x 128 B; lane -> (row, swizzled piece)
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        voff[i] = (unsigned)(r * (p.ldh * 2) + q * 16);
    }
    unsigned char* ring = smem + kq * X_RING;
    // ---- A-fragment read address of lane (row li, k half lh) for k16 step s of a chunk: piece 2s + lh
    unsigned fa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) fa[s] = lbase + (unsigned)(kq * X_RING + li * 128 + (((2 * s + lh) ^ ((li >> 1) & 7)) * 16));

    // ---- epilogue role: 2 adjacent units of one row per thread, the same (row, units) in every step
    const int erow = tid >> 3, eul = (tid & 7) * 2;
    const int eunit = u0 + eul;
    const bool e_ok0 = eunit < H, e_ok1 = eunit + 1 < H;
    const bool e_vec = e_ok1 && ((H & 1) == 0);
    const int ecol = (eul >> 3) * 32 + (eul & 7);      // + g*8: column of gate g inside the workgroup's 64
    float* cst = reinterpret_cast<float*>(smem + X_CST);
    for (int s = 0; s < p.NS; ++s) {
        const int b = row0 + s * X_SR + erow;
        f32x2 c0 = {0.f, 0.f};
        if (p.t0 > 0 && b < B) {
            const float* q = p.c_all + ((int64_t)(p.t0 - 1) * B + b) * H + eunit;
            if (e_ok0) c0[0] = q[0];
            if (e_ok1) c0[1] = q[1];
        }
        *reinterpret_cast<f32x2*>(cst + (s * X_SR + erow) * X_UN + eul) = c0;
    }
    unsigned short* hsm = reinterpret_cast<unsigned short*>(smem + X_HSM);
    const int64_t H4 = 4 * (int64_t)H;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(g_zero4);
    const int64_t hplane_b = p.hplane * 2;             // bytes between two planes of the h image

    for (int t = p.t0; t < p.t1; ++t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * X_SR;
            unsigned int* cnt = p.sync + (rbase / X_SR) * 32;          // one counter per 32-row chain, whatever NS the launch uses
            const int xrec = (bid == p.stamp_block) ? (t - p.t0) * p.NS + s : -1;
            XSTAMP(p.stamps, xrec, 0);
            if (t > p.t0) {          // h_{t-1} of this chain published by every column slice of the row group?
                if (tid == 0) {
                    const bool ok = spin_until_x(cnt, (unsigned int)(nC * t));
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                X_BARRIER();
                if (s_flag == 0) return;
            }
            XSTAMP(p.stamps, xrec, 1);

            // epilogue operands requested now, consumed after the contraction
            const int eb = rbase + erow;
            const bool rok = eb < B;
            f32x2 gxv[4];
            {
                const float* gsrc = (t < p.n_gx) ? p.gx_stash + ((int64_t)t * B + eb) * H4 : p.bias;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* q = gsrc + (int64_t)g * H + eunit;
                    if (e_vec) {
                        gxv[g] = *reinterpret_cast<const f32x2*>(rok ? q : g_zero4);
                    } else {
                        gxv[g][0] = *((rok && e_ok0) ? q : g_zero4);
                        gxv[g][1] = *((rok && e_ok1) ? q + 1 : g_zero4);
                    }
                }
            }

            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

            if (t > 0) {
                const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.hp + ((int64_t)(t - 1) * B + rbase) * p.ldh);
                // chunk C of this wave = k64 block 4 kq + C of all three planes: 12 requests; a block past Kp reads zeros, so
                // that every wave issues the same number of requests (counted waits)
                // request N (plane N / 4, row octet N % 4) of chunk C
#define X_REQ(C, N)                                                                                             \
                glds16x_sc1((4 * kq + (C)) * 64 < Kp ? abase + (4 * kq + (C)) * 128 + ((N) / 4) * hplane_b + voff[(N) % 4] : zsrc, \
                            ring + ((C) % 3) * X_CHUNK + ((N) / 4) * X_PLANE + ((N) % 4) * 1024);
#define X_ISSUE(C)                                                                                              \
                X_REQ(C, 0) X_REQ(C, 1) X_REQ(C, 2) X_REQ(C, 3) X_REQ(C, 4) X_REQ(C, 5)                         \
                X_REQ(C, 6) X_REQ(C, 7) X_REQ(C, 8) X_REQ(C, 9) X_REQ(C, 10) X_REQ(C, 11)
                // A fragment of plane PL for k16 step I (chunk I / 4 in slot (I / 4) % 3)
#define X_RD1(PL, I) X_DSR(af[PL], fa[(I) & 3], (((I) >> 2) % 3) * X_CHUNK + (PL) * X_PLANE);
#define X_WAIT1(PL, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(af[PL]));
#define X_MF(A, W, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, W, ACC, 0, 0, 0);
                // k16 step I: the six plane products for both column halves, small terms first.  ONE fragment set: a plane's
                // fragment of step I + 1 is requested as soon as its last product of step I is issued (h2 after 2 MFMAs, h1
                // after 6, h0 after 12); the 6 MFMAs in front of the first use of h0 cover its LDS latency.  Reads return in
                // order, so a fragment has landed when at most the two younger reads are outstanding.
#define X_MM2(I)  X_WAIT1(2, 2) X_MF(af[2], wreg[0 * 16 + (I)], acc0) X_MF(af[2], wreg[1 * 16 + (I)], acc1)
#define X_MM1(I)  X_MM1N(I, 2)
#define X_MM1N(I, N) X_WAIT1(1, N) X_MF(af[1], wreg[2 * 16 + (I)], acc0) X_MF(af[1], wreg[3 * 16 + (I)], acc1)                \
                                X_MF(af[1], wreg[0 * 16 + (I)], acc0) X_MF(af[1], wreg[1 * 16 + (I)], acc1)
#define X_MM0(I)  X_WAIT1(0, 2) X_MF(af[0], wreg[4 * 16 + (I)], acc0) X_MF(af[0], wreg[5 * 16 + (I)], acc1)                \
                                X_MF(af[0], wreg[2 * 16 + (I)], acc0) X_MF(af[0], wreg[3 * 16 + (I)], acc1)                \
                                X_MF(af[0], wreg[0 * 16 + (I)], acc0) X_MF(af[0], wreg[1 * 16 + (I)], acc1)
#define X_STEP(I) X_MM2(I) X_RD1(2, (I) + 1) X_MM1(I) X_RD1(1, (I) + 1) X_MM0(I) X_RD1(0, (I) + 1)
                // the same with three LDS-DMA requests of chunk C slipped between the MFMA groups: a request costs ~100 issue
                // cycles (36 in a burst in front of the loop were 1.6 us of a 8.5-us sub-step); behind an MFMA group they are free
#define X_STEP_REQ(I, C)                                                                                        \
                X_MM2(I) X_RD1(2, (I) + 1) X_REQ(C, 3 * ((I) & 3))                                              \
                X_MM1(I) X_RD1(1, (I) + 1) X_REQ(C, 3 * ((I) & 3) + 1)                                          \
                X_MM0(I) X_RD1(0, (I) + 1) X_REQ(C, 3 * ((I) & 3) + 2)
#define X_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");
                bf16x8 af[3];
                X_ISSUE(0) X_ISSUE(1)
                XSTAMP(p.stamps, xrec, 2);
                X_VM(12)
                X_RD1(2, 0) X_RD1(1, 0) X_RD1(0, 0)
                XSTAMP(p.stamps, xrec, 3);
                // chunk 0 (slot 0); chunk 2 -> slot 2 on the way
                X_STEP_REQ(0, 2) X_STEP_REQ(1, 2) X_STEP_REQ(2, 2)
                X_VM(9)                  // chunk 1 landed (9 requests of chunk 2 are younger) before step 4's fragments are read
                X_STEP_REQ(3, 2)
                // chunk 1; slot 0 is read out (step 3's last fragment was awaited in front of its MFMAs): chunk 3 takes it
                X_STEP_REQ(4, 3) X_STEP_REQ(5, 3) X_STEP_REQ(6, 3)
                X_VM(9)                  // chunk 2 landed
                X_STEP_REQ(7, 3)
                X_STEP(8) X_STEP(9) X_STEP(10)
                X_VM(0)                  // chunk 3 landed
                X_STEP(11)
                X_STEP(12) X_STEP(13) X_STEP(14)
                X_MM2(15) X_MM1N(15, 1)      // (no younger reads behind the last step's)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]));
                X_MF(af[0], wreg[4 * 16 + 15], acc0) X_MF(af[0], wreg[5 * 16 + 15], acc1)
                X_MF(af[0], wreg[2 * 16 + 15], acc0) X_MF(af[0], wreg[3 * 16 + 15], acc1)
                X_MF(af[0], wreg[0 * 16 + 15], acc0) X_MF(af[0], wreg[1 * 16 + 15], acc1)
#undef X_VM
#undef X_STEP_REQ
#undef X_STEP
#undef X_MM0
#undef X_MM1
#undef X_MM1N
#undef X_MM2
#undef X_WAIT1
#undef X_RD1
#undef X_STEP
#undef X_MM
#undef X_MF
#undef X_WAIT
#undef X_RD
#undef X_ISSUE
#undef X_REQ
            }
            XSTAMP(p.stamps, xrec, 4);
            {   // partial tile of this wave's k quarter -> its own (idle) ring
                float* rp = reinterpret_cast<float*>(ring);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    rp[row * X_RLD + li] = acc0[r];
                    rp[row * X_RLD + 32 + li] = acc1[r];
                }
            }
            X_BARRIER();
            XSTAMP(p.stamps, xrec, 5);

            f32x2 gate[4], cv, hv;
            {
                f32x2 pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x2 v = gxv[g];
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        v += *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(smem + w * X_RING) + erow * X_RLD + ecol + g * 8);
                    pre[g] = v;
                }
                f32x2* cp = reinterpret_cast<f32x2*>(cst + (s * X_SR + erow) * X_UN + eul);
                const f32x2 cprev = *cp;
                unsigned short pb[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gate[0][j] = sigmoidf_(pre[0][j]);
                    gate[1][j] = sigmoidf_(pre[1][j]);
                    gate[2][j] = tanhf_(pre[2][j]);
                    gate[3][j] = sigmoidf_(pre[3][j]);
                    const bool ok = rok && (j ? e_ok1 : e_ok0);
                    cv[j] = ok ? gate[1][j] * cprev[j] + gate[0][j] * gate[2][j] : 0.f;
                    hv[j] = ok ? gate[3][j] * tanhf_(cv[j]) : 0.f;
                    split3_bits(hv[j], pb[j]);
                }
                *cp = cv;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<unsigned int*>(hsm + pl * (X_SR * X_UN) + erow * X_UN + eul) =
                        (unsigned int)pb[0][pl] | ((unsigned int)pb[1][pl] << 16);
            }
            XSTAMP(p.stamps, xrec, 6);
            X_BARRIER();
            if (kq == 0) {   // the h_t planes: three tiles of 32 rows x 32 B = three 16-byte write-through store instructions,
                             // issued first: they are what the other workgroups wait for
                const int rl = lane >> 1, part = lane & 1;
                unsigned short* dst = p.hp + ((int64_t)t * B + rbase + rl) * p.ldh + u0 + part * 8;
                if (rbase + rl < B) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(hsm + pl * (X_SR * X_UN) + rl * X_UN + part * 8);
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + pl * p.hplane), "v"(v) : "memory");
                    }
                    if (cs == nC - 1) {     // the last column slice keeps the pad columns [16 nC, Kp) of the image at zero
                        const u32x4 z = {0u, 0u, 0u, 0u};
                        unsigned short* zd = dst;
                        asm volatile("" : "+v"(zd));          // (opaque: derived from the payload address, nothing to keep live)
                        for (int c = u0 + X_UN; c < Kp; c += X_UN) {
                            zd += X_UN;
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(zd + pl * p.hplane), "v"(z) : "memory");
                        }
                    }
                }
            }
            if (kq == 1 && p.hblk) {     // the same pieces into the batched GEMMs' row image (plain stores, beside wave 0's hand-off)
                const int rl = lane >> 1, part = lane & 1;
                const int64_t r = (int64_t)t * B + rbase + rl;
                if (rbase + rl < B) {
                    unsigned short* dst = p.hblk + (r >> 6) * (64 * p.ldhblk) + (int64_t)(u0 >> 4) * 3072 + part * 512 + (r & 63) * 8;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        *reinterpret_cast<u32x4*>(dst + pl * 1024) = *reinterpret_cast<const u32x4*>(hsm + pl * (X_SR * X_UN) + rl * X_UN + part * 8);
                }
            }
            XSTAMP(p.stamps, xrec, 7);
            if (kq == 0) {   // the ONE wave that stored the hand-off payload drains (those three stores only: its other stores
                             // of the step come after the signal) and signals for the workgroup
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                XSTAMP(p.stamps, xrec, 8);
                if (lane == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            XSTAMP(p.stamps, xrec, 9);
            if (rok) {
                const int64_t rowi = (int64_t)t * B + eb;
                float* cdst = p.c_all + rowi * H + eunit;
                float* hdst = p.h_all + rowi * H + eunit;
                float* st = p.gx_stash + rowi * H4 + eunit;
                const bool stash = p.no_stash == 0;
                if (e_vec) {
                    *reinterpret_cast<f32x2*>(cdst) = cv;
                    *reinterpret_cast<f32x2*>(hdst) = hv;
                    if (stash) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x2*>(st + (int64_t)g * H) = gate[g];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (j ? e_ok1 : e_ok0) {
                            cdst[j] = cv[j];
                            hdst[j] = hv[j];
                            if (stash) {
#pragma unroll
                                for (int g = 0; g < 4; ++g) st[(int64_t)g * H + j] = gate[g][j];
                            }
                        }
                }
            }
            X_BARRIER();       // hsm / cst / partial sums are free again
        }
    }
}

// xg == 0: grid = [na workgroups of layer pa | workgroups of layer pb] (nb may be 0).
// xg = G > 0 (XCD-aware; both layers of one shape): the hardware deals workgroup b to XCD b % 8 (speed only - nothing depends on
// it for correctness); a GROUP is the nC column slices of one (layer, row group) - the workgroups that exchange one chain's h_t
// tiles - and group g is dealt to the XCDs {g, g + G, ..}: a consumer's sc1 loads then find the tile in the L2 its producers
// wrote through, and a row's lines enter 8 / G L2s instead of all eight.  The grid is 8 * ceil(nC / (8 / G)) blocks; the few
// whose slice index falls past nC exit at once (nobody waits for them: the counters count the nC real slices).
__global__ __launch_bounds__(X_NT, 1) void lstm_seq_fwd_x3_persist_kernel(SeqFwdX3Args pa, SeqFwdX3Args pb, int na, int xg) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[X_LDS];
    __shared__ int s_flag;
    const int bid = (int)blockIdx.x;
    bool lb;
    int vb;
    if (xg > 0) {
        const int nC = (pa.H + X_UN - 1) / X_UN;
        const int x = bid & 7, q = bid >> 3, per = 8 / xg;
        const int g = x % xg, cs = q * per + x / xg;
        if (cs >= nC) return;
        const int rgs = na / nC;                     // row groups of layer A
        lb = g >= rgs;
        vb = (lb ? g - rgs : g) * nC + cs;
    } else {
        lb = bid >= na;
        vb = lb ? bid - na : bid;
    }
    // ONE copy of the body: the layer's arguments are selected field by field (scalar selects on kernel arguments)
    seq_fwd_x3_body(lb ? pb : pa, vb, smem, s_flag);
}

size_t lstm_persist_sync_bytes();

static int fwd_x3_capacity() {
    const int cap = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_fwd_x3_persist_kernel), X_NT);
    return cap < X_MAX_WG ? cap : X_MAX_WG;
}
// number of 32-row chains per workgroup (0: unsupported, or two layers of the shape do not fit the device's resident capacity)
int lstm_seq_fwd_x3_persist_supported(int B, int H) {
    if (!(B > 0 && B % X_SR == 0 && H >= 8 && H <= 1024)) return 0;
    const int cap = fwd_x3_capacity();
    const int nC = cdiv(H, X_UN);
    int R = B / X_SR, ns = 1;
    while (R * nC > cap / 2 && ns < X_MAXNS && R % 2 == 0) { R /= 2; ns *= 2; }
    return (R * nC <= cap / 2 && R <= 64) ? ns : 0;
}

// chains per workgroup of a launch with ONE layer: it may use the whole device (a pair shares it half / half)
static int fwd_x3_single_ns(int B, int H) {
    if (!lstm_seq_fwd_x3_persist_supported(B, H)) return 0;
    const int cap = fwd_x3_capacity(), nC = cdiv(H, X_UN);
    int R = B / X_SR, ns = 1;
    while (R * nC > cap && ns < X_MAXNS && R % 2 == 0) { R /= 2; ns *= 2; }
    return (R * nC <= cap) ? ns : 0;
}
int lstm_seq_fwd_x3_persist_single_workgroups(int B, int H) {
    const int ns = fwd_x3_single_ns(B, H);
    return ns > 0 ? (B / (ns * X_SR)) * cdiv(H, X_UN) : 0;
}
static int prep_x(SeqFwdX3Args& a, bool single = false) {
    const int ns = single ? fwd_x3_single_ns(a.B, a.H) : lstm_seq_fwd_x3_persist_supported(a.B, a.H);
    S2VT_REQUIRE(ns > 0, "lstm_seq_fwd_x3_persist: unsupported shape (B %% 32, H <= 1024) or it does not fit the device's resident capacity");
    S2VT_REQUIRE(a.t1 > a.t0 && a.t0 >= 0 && a.wp && a.hp && a.h_all && a.gx_stash && a.c_all && a.sync && a.err && (a.bias || a.n_gx >= a.t1),
                 "lstm_seq_fwd_x3_persist: bad arguments");
    S2VT_REQUIRE(!a.hblk || (a.B % 64 == 0 && a.ldhblk >= 3 * (int64_t)a.Kp && a.ldhblk % 8 == 0 && (reinterpret_cast<uintptr_t>(a.hblk) & 15) == 0),
                 "lstm_seq_fwd_x3_persist: the h row image needs B %% 64 == 0 and rows of 3 * pad64(H) elements");
    S2VT_REQUIRE(a.Kp == (a.H + 63) / 64 * 64 && a.ldw >= a.Kp && a.ldh >= a.Kp && a.ldw % 8 == 0 && a.ldh % 8 == 0 &&
                     a.wplane % 8 == 0 && a.hplane % 8 == 0 && (reinterpret_cast<uintptr_t>(a.wp) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.hp) & 15) == 0,
                 "lstm_seq_fwd_x3_persist: plane images must be 16-byte aligned with rows of Kp = H rounded up to 64");
    a.NS = ns;
    a.RB = ns * X_SR;
    return 0;
}

int lstm_seq_fwd_x3_persist2(hipStream_t stream, SeqFwdX3Args a, const SeqFwdX3Args* b) {
    int rc;
    if ((rc = prep_x(a, b == nullptr))) return rc;
    SeqFwdX3Args bb = b ? *b : a;
    if (b) {
        if ((rc = prep_x(bb))) return rc;
        S2VT_REQUIRE(bb.sync != a.sync, "lstm_seq_fwd_x3_persist: paired layers need their own counters");
    }
    const int na = (a.B / a.RB) * cdiv(a.H, X_UN), nb = b ? (bb.B / bb.RB) * cdiv(bb.H, X_UN) : 0;
    S2VT_REQUIRE(na + nb <= fwd_x3_capacity(), "lstm_seq_fwd_x3_persist: %d workgroups would not be co-resident (device capacity %d)",
                 na + nb, fwd_x3_capacity());
    // the hand-off counters count finished timesteps of the whole sequence: zeroed with its first block only
    if (a.t0 == 0) S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (b && bb.t0 == 0) S2VT_HIP(hipMemsetAsync(bb.sync, 0, lstm_persist_sync_bytes(), stream));
    // XCD-aware dealing (see the kernel) when the groups divide the 8 XCDs and the padded grid still fits the device
    const int nC = cdiv(a.H, X_UN);
    const int G = (na + nb) / nC;
    int xg = 0, grid = na + nb;
    if ((!b || (bb.B == a.B && bb.H == a.H)) && G > 0 && G <= 8 && 8 % G == 0) {
        const int padded = 8 * cdiv(nC, 8 / G);
        if (padded <= fwd_x3_capacity()) { xg = G; grid = padded; }
    }
    hipLaunchKernelGGL(lstm_seq_fwd_x3_persist_kernel, dim3(grid), dim3(X_NT), 0, stream, a, bb, na, xg);
    S2VT_LAUNCH_CHECK("lstm_seq_fwd_x3_persist_kernel");
    return 0;
}

// fp32 [R][C] (row stride ld) -> three bf16 planes [3][R][Cp] (x = p0 + p1 + p2), columns [C, Cp) zero: W_hh for the kernel above
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ src, int64_t ld, int R, int C, int Cp,
                                                          unsigned short* __restrict__ dst, int64_t plane) {
    const int64_t n = (int64_t)R * (Cp / 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / (Cp / 2)), c = (int)(i % (Cp / 2)) * 2;
        unsigned short a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        if (c < C) split3_bits(src[(int64_t)r * ld + c], a);
        if (c + 1 < C) split3_bits(src[(int64_t)r * ld + c + 1], b);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<unsigned int*>(dst + pl * plane + (int64_t)r * Cp + c) = (unsigned int)a[pl] | ((unsigned int)b[pl] << 16);
    }
}
int split3_rows(hipStream_t stream, const float* src, int64_t ld, int R, int C, int Cp, unsigned short* dst, int64_t plane) {
    S2VT_REQUIRE(src && dst && R > 0 && C > 0 && Cp >= C && Cp % 2 == 0 && plane >= (int64_t)R * Cp, "split3_rows: bad arguments");
    const int64_t n = (int64_t)R * (Cp / 2);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(split3_rows_kernel, dim3(grid), dim3(256), 0, stream, src, ld, R, C, Cp, dst, plane);
    S2VT_LAUNCH_CHECK("split3_rows_kernel");
    return 0;
}


// ------------------------------------------------------------------------------------------------------------- BPTT
// Split-precision persistent BPTT (utils/autograd of S2VTModel.py:67,77: dh_t = dh_out_t + dG_{t+1} . W_hh, gate derivatives).
//
// The forward's decomposition does not carry over: dG_{t+1} is [B, 4H], so a workgroup that owns 16 units of dh would have to
// take in 32 rows x 4H x 3 planes = 768 KB per sub-step (12 us at the ~65 GB/s a compute unit ingests), and W_hh^T for 32 units
// as planes does not fit a register file.  Here the contraction is split over K instead (a reduce-scatter): workgroup
// (chain, cs) owns the 64 gate columns k of ITS 16 units - the dG values it has just computed itself, already in LDS - and the
// matching 64 rows of W_hh for ALL H output columns (64 k x 1024 j x 3 planes = the same 384 registers per lane as the forward).
// Per sub-step it computes the partial products P[b, j] = sum_{k in its 64} dG_t[b, k] W_hh[k, j] for all j (192 MFMAs per
// wave, the six plane products of gemm_x3.hip; transposed tiles: lane = batch row, registers = 4 consecutive j, so a finished
// tile leaves as 16-byte stores) and scatters them as fp32 blocks [32 rows][16 units], one per CONSUMER column slice; after the
// hand-off (all waves drain their stores, barrier, one counter add) every workgroup gathers the nC blocks addressed to it - one
// contiguous nC x 2 KB region - by LDS-DMA and sums them in producer order (fixed order: deterministic).  Per sub-step a
// workgroup moves 128 KB out and 126 KB in instead of 768 KB in; nothing but its own dG tile is ever converted to planes.
// Every partial block has its own address within a launch (a ring of nslots > block length steps), so no compute unit can
// hold a stale copy of a line it is about to read; across launches the kernel boundary orders the accesses.
constexpr int Y_GBUF = 64 * 2048;                 // gather buffer: up to 64 producer blocks of [32][16] fp32
constexpr int Y_TILE = Y_GBUF;                    // own dG_t tile as planes [3][32 rows][64 k] bf16 (swizzled like a forward chunk)
constexpr int Y_DCST = Y_TILE + 3 * X_PLANE;      // dL/dc carry of the workgroup's cells, per chain [32][16]
constexpr int Y_CSUM = Y_DCST + X_MAXNS * X_SR * X_UN * 4;    // per-wave column sums of the dG tile [4 waves][64 gate columns]
constexpr int Y_LDS = Y_CSUM + 4 * 64 * 4;                    // 152576 B

__device__ __forceinline__ void seq_bwd_x3_body(const SeqBwdX3Args& p, const int bid, unsigned char* smem, int& s_flag) {
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int kw = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index = quarter of the output columns j
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, B = p.B;
    const int nC = (H + X_UN - 1) / X_UN;
    const int cs = bid % nC, rg = bid / nC;
    const int u0 = cs * X_UN, row0 = rg * p.RB;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);

    // ---- this wave's W_hh slice as A operands of v_mfma_f32_16x16x32_bf16 (lane (jl, kg): 8 consecutive k of output column
    //      j = 256 kw + 16 jt + jl): wreg[(plane * 2 + ks) * 16 + jt]; k step ks covers the workgroup's k = 32 ks + 8 kg + e, i.e.
    //      gate 2 ks + (kg >> 1), unit u0 + 8 (kg & 1) + e <-> W_hh^T column (2 ks + (kg >> 1)) Hp + u0 + 8 (kg & 1) + e
    const int l16 = lane & 15, kg = lane >> 4;
    bf16x8 wreg[96];
    {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jt = 0; jt < 16; ++jt) {
                    const int j = 256 * kw + 16 * jt + l16;
                    const unsigned short* q = (j < p.Kp) ? p.wtp + pl * p.wplane + (int64_t)j * p.ldw + (2 * ks + (kg >> 1)) * p.Hp + u0 + (kg & 1) * 8 : zero;
                    wreg[(pl * 2 + ks) * 16 + jt] = *reinterpret_cast<const bf16x8*>(q);
                }
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wreg[i]));
#pragma unroll
        for (int i = 32; i < 96; ++i) asm volatile("" : "+a"(wreg[i]));
    }
    // ---- B-fragment read address of lane (batch row 16 bt + l16, k group kg) for k step ks: piece 4 ks + kg of the dG tile
    unsigned fa[2][2];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int b = 16 * bt + l16;
            fa[bt][ks] = lbase + (unsigned)(Y_TILE + b * 128 + (((4 * ks + kg) ^ ((b >> 1) & 7)) * 16));
        }

    // ---- cell role: 2 adjacent units of one row per thread
    const int erow = tid >> 3, eul = (tid & 7) * 2;
    const int eunit = u0 + eul;
    const bool e_ok0 = eunit < H, e_ok1 = eunit + 1 < H;
    const bool e_vec = e_ok1 && ((H & 1) == 0);
    float* dcst = reinterpret_cast<float*>(smem + Y_DCST);
    const bool last_block = (p.t1 == p.T);
    for (int s = 0; s < p.NS; ++s) {
        const int b = row0 + s * X_SR + erow;
        f32x2 d0 = {0.f, 0.f};
        if (!last_block && b < B) {
            const float* q = p.dc + (int64_t)b * H + eunit;
            if (e_ok0) d0[0] = q[0];
            if (e_ok1) d0[1] = q[1];
        }
        *reinterpret_cast<f32x2*>(dcst + (s * X_SR + erow) * X_UN + eul) = d0;
    }
    const int64_t H4 = 4 * (int64_t)H;
    const float* gbuf = reinterpret_cast<const float*>(smem);
    unsigned char* tile = smem + Y_TILE;
    // tile write position of this thread's two cells for gate g: row erow, piece 2g + (eul >> 3), element eul & 7
    unsigned tw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) tw[g] = (unsigned)(erow * 128 + (((2 * g + (eul >> 3)) ^ ((erow >> 1) & 7)) * 16) + (eul & 7) * 2);

    for (int t = p.t1 - 1; t >= p.t0; --t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * X_SR;
            const int chain = rbase / X_SR;
            unsigned int* cnt = p.sync + (rg * X_MAXNS + s) * 32;
            const int xrec = (bid == p.stamp_block) ? (p.t1 - 1 - t) * p.NS + s : -1;
            XSTAMP(p.stamps, xrec, 0);

            // cell operands requested first, consumed after the gather
            const int eb = rbase + erow;
            const bool rok = eb < B;
            const int64_t rowi = (int64_t)t * B + eb;
            f32x2 stv[4], cv, cpv, dhov;
            {
                const float* st = p.stash_dg + rowi * H4 + eunit;
                const float* cq = p.c_all + rowi * H + eunit;
                const bool hasdh = p.dh_out && t >= p.dh_first;
                const float* dq = hasdh ? p.dh_out + ((int64_t)(t - p.dh_first) * B + eb) * H + eunit : g_zero4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* q = st + (int64_t)g * H;
                    if (e_vec) stv[g] = *reinterpret_cast<const f32x2*>(rok ? q : g_zero4);
                    else { stv[g][0] = *((rok && e_ok0) ? q : g_zero4); stv[g][1] = *((rok && e_ok1) ? q + 1 : g_zero4); }
                }
                if (e_vec) {
                    cv = *reinterpret_cast<const f32x2*>(rok ? cq : g_zero4);
                    cpv = *reinterpret_cast<const f32x2*>((rok && t > 0) ? cq - (int64_t)B * H : g_zero4);
                    dhov = *reinterpret_cast<const f32x2*>((rok && hasdh) ? dq : g_zero4);
                } else {
                    cv[0] = *((rok && e_ok0) ? cq : g_zero4); cv[1] = *((rok && e_ok1) ? cq + 1 : g_zero4);
                    cpv[0] = *((rok && e_ok0 && t > 0) ? cq - (int64_t)B * H : g_zero4);
                    cpv[1] = *((rok && e_ok1 && t > 0) ? cq - (int64_t)B * H + 1 : g_zero4);
                    dhov[0] = *((rok && e_ok0 && hasdh) ? dq : g_zero4); dhov[1] = *((rok && e_ok1 && hasdh) ? dq + 1 : g_zero4);
                }
            }

            f32x2 dh = {0.f, 0.f};
            if (t < p.T - 1) {
                // the nC partial blocks of dG_{t+1} . W_hh addressed to this column slice: published by every slice of the chain?
                if (tid == 0) {
                    const bool ok = spin_until_x(cnt, (unsigned int)(nC * (p.T - 1 - t)));
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                X_BARRIER();
                if (s_flag == 0) return;
                XSTAMP(p.stamps, xrec, 1);
                const unsigned char* src = reinterpret_cast<const unsigned char*>(
                    p.part + (int64_t)(t % p.nslots) * p.part_slot + ((int64_t)(chain * nC + cs) * nC) * (X_SR * X_UN)) + lane * 16;
                for (int r = kw; r < 2 * nC; r += 4) glds16x_sc1(src + r * 1024, smem + r * 1024);
                XSTAMP(p.stamps, xrec, 2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                X_BARRIER();
                XSTAMP(p.stamps, xrec, 3);
                const float* gp = gbuf + erow * X_UN + eul;
                int c = 0;                  // summed in producer order; eight LDS reads in flight at a time
                for (; c + 8 <= nC; c += 8) {
                    f32x2 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x2*>(gp + (c + i) * (X_SR * X_UN));
#pragma unroll
                    for (int i = 0; i < 8; ++i) dh += v[i];
                }
                for (; c < nC; ++c) dh += *reinterpret_cast<const f32x2*>(gp + c * (X_SR * X_UN));
            } else {
                XSTAMP(p.stamps, xrec, 1);
                XSTAMP(p.stamps, xrec, 2);
                XSTAMP(p.stamps, xrec, 3);
            }
            XSTAMP(p.stamps, xrec, 4);

            f32x2 dg[4];
            {
                f32x2* dp = reinterpret_cast<f32x2*>(dcst + (s * X_SR + erow) * X_UN + eul);
                const f32x2 dcin = *dp;
                f32x2 dcn;
                unsigned short pb[4][2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = rok && (j ? e_ok1 : e_ok0);
                    const float d = dh[j] + dhov[j];
                    const float ig = stv[0][j], fg = stv[1][j], gg = stv[2][j], og = stv[3][j];
                    const float tc = tanhf_(cv[j]);
                    const float dc = d * og * (1.0f - tc * tc) + dcin[j];
                    const float d_o = d * tc;
                    dg[0][j] = ok ? dc * gg * ig * (1.0f - ig) : 0.f;
                    dg[1][j] = ok ? dc * cpv[j] * fg * (1.0f - fg) : 0.f;
                    dg[2][j] = ok ? dc * ig * (1.0f - gg * gg) : 0.f;
                    dg[3][j] = ok ? d_o * og * (1.0f - og) : 0.f;
                    dcn[j] = ok ? dc * fg : 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) split3_bits(dg[g][j], pb[g][j]);
                }
                *dp = dcn;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        *reinterpret_cast<unsigned int*>(tile + pl * X_PLANE + tw[g]) = (unsigned int)pb[g][0][pl] | ((unsigned int)pb[g][1][pl] << 16);
                if (rok) {
                    float* st = p.stash_dg + rowi * H4 + eunit;
                    const bool keep = p.skip_dg == 0;
                    if (e_vec) {
                        if (keep) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x2*>(st + (int64_t)g * H) = dg[g];
                        }
                        if (t == p.t0) *reinterpret_cast<f32x2*>(p.dc + (int64_t)eb * H + eunit) = dcn;
                    } else {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (j ? e_ok1 : e_ok0) {
                                if (keep) {
#pragma unroll
                                    for (int g = 0; g < 4; ++g) st[(int64_t)g * H + j] = dg[g][j];
                                }
                                if (t == p.t0) p.dc[(int64_t)eb * H + eunit + j] = dcn[j];
                            }
                    }
                }
                if (p.colpart) {    // column sums of the tile over its 32 rows: 8 rows per wave by shuffles, the 4 waves through LDS
                    float* cs_w = reinterpret_cast<float*>(smem + Y_CSUM) + kw * 64;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x2 v = dg[g];
#pragma unroll
                        for (int off = 8; off < 64; off <<= 1) { v[0] += __shfl_xor(v[0], off); v[1] += __shfl_xor(v[1], off); }
                        if (lane < 8) *reinterpret_cast<f32x2*>(cs_w + g * 16 + eul) = v;
                    }
                }
            }
            XSTAMP(p.stamps, xrec, 5);
            X_BARRIER();           // the dG_t tile is complete (and everyone is done with the gather buffer)
            // dG_t as the batched GEMMs' row image + its 32-row column sums: issued BEHIND the hand-off (t > 0: after the signal)
            auto emit_planes = [&]() {
                if (p.colpart && kw == 0) {
                    const int g = lane >> 4, u = lane & 15;
                    const float* cs0 = reinterpret_cast<const float*>(smem + Y_CSUM) + lane;
                    const float sum = ((cs0[0] + cs0[64]) + cs0[128]) + cs0[192];
                    if (u0 + u < H) p.colpart[((int64_t)t * (B / X_SR) + chain) * H4 + (int64_t)g * H + u0 + u] = sum;
                }
                if (p.dgp) {
                    const int row = tid >> 3, p8 = tid & 7;                 // piece p8 = gate 2 bits, unit octet 1 bit
                    const int gq = p8 >> 1, k0 = gq * H + u0 + (p8 & 1) * 8;
                    const int64_t r = (int64_t)t * B + rbase + row;
                    if (rbase + row < B && u0 + (p8 & 1) * 8 < H) {
                        unsigned short* dst = p.dgp + (r >> 6) * (64 * p.lddgp) + (int64_t)(k0 >> 4) * 3072 + ((k0 >> 3) & 1) * 512 + (r & 63) * 8;
                        const unsigned char* src = tile + row * 128 + ((p8 ^ ((row >> 1) & 7)) * 16);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            *reinterpret_cast<u32x4*>(dst + pl * 1024) = *reinterpret_cast<const u32x4*>(src + pl * X_PLANE);
                    }
                }
            };

            if (t > 0) {           // partial products of dG_t for step t - 1 (nobody consumes those of step 0)
                float* pslot = p.part + (int64_t)((t - 1) % p.nslots) * p.part_slot + ((int64_t)chain * nC * nC + cs) * (X_SR * X_UN)
                               + l16 * X_UN + 4 * kg;
                const int64_t cstride = (int64_t)nC * (X_SR * X_UN);           // floats between the blocks of two consumers
                // the whole dG_t tile as B operands, once: gfr[plane][bt][ks] (12 fragments; every output column tile uses all)
                bf16x8 gfr[3][2][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) X_DSR(gfr[pl][bt][ks], fa[bt][ks], pl * X_PLANE);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(gfr[pl][bt][ks]));
#define Y_MF(W, G, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, G, ACC, 0, 0, 0);
                // two output column tiles (jt, jt + 1) x two batch tiles at a time: 48 MFMAs on four accumulators (the same one
                // every fourth instruction), the six plane products small terms first; a finished 16 x 16 tile holds
                // P^T[j = 4 kg + e][b = l16] in register e: ONE 16-byte store per lane = 16 rows x 64 B = 1 KB contiguous of the
                // consumer's block [32 rows][16 units]
#pragma unroll
                for (int jp = 0; jp < 8; ++jp) {
                    f32x4 acc[2][2];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int bt = 0; bt < 2; ++bt) acc[x][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
#define Y_PROD(WP, GP)                                                                                          \
                    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
                        _Pragma("unroll") for (int x = 0; x < 2; ++x)                                           \
                            _Pragma("unroll") for (int bt = 0; bt < 2; ++bt)                                    \
                                Y_MF(wreg[((WP) * 2 + ks) * 16 + 2 * jp + x], gfr[GP][bt][ks], acc[x][bt])
                    Y_PROD(2, 0) Y_PROD(0, 2) Y_PROD(1, 1) Y_PROD(1, 0) Y_PROD(0, 1) Y_PROD(0, 0)
#undef Y_PROD
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        const int cons = 16 * kw + 2 * jp + x;
                        if (cons < nC) {
#pragma unroll
                            for (int bt = 0; bt < 2; ++bt) {
                                float* dst = pslot + cons * cstride + bt * (16 * X_UN);
                                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "a"(acc[x][bt]) : "memory");
                            }
                        }
                    }
                }
#undef Y_MF
                XSTAMP(p.stamps, xrec, 6);
                // hand-off: EVERY wave stored a part of the payload: each drains its own stores, then the barrier, then one add
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                XSTAMP(p.stamps, xrec, 7);
                X_BARRIER();
                if (tid == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                XSTAMP(p.stamps, xrec, 8);
                emit_planes();
                if (p.dgp || p.colpart) X_BARRIER();       // (the tile and the column sums are rewritten by the next sub-step's cells)
            } else {
                emit_planes();
                X_BARRIER();
            }
        }
    }
}

// block roles as in the forward kernel (xg: XCD-aware dealing)
__global__ __launch_bounds__(X_NT, 1) void lstm_seq_bwd_x3_persist_kernel(SeqBwdX3Args pa, SeqBwdX3Args pb, int na, int xg) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[Y_LDS];
    __shared__ int s_flag;
    const int bid = (int)blockIdx.x;
    bool lb;
    int vb;
    if (xg > 0) {
        const int nC = (pa.H + X_UN - 1) / X_UN;
        const int x = bid & 7, q = bid >> 3, per = 8 / xg;
        const int g = x % xg, cs = q * per + x / xg;
        if (cs >= nC) return;
        const int rgs = na / nC;
        lb = g >= rgs;
        vb = (lb ? g - rgs : g) * nC + cs;
    } else {
        lb = bid >= na;
        vb = lb ? bid - na : bid;
    }
    seq_bwd_x3_body(lb ? pb : pa, vb, smem, s_flag);
}

static int bwd_x3_capacity() {
    const int cap = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_bwd_x3_persist_kernel), X_NT);
    return cap < X_MAX_WG ? cap : X_MAX_WG;
}
int lstm_seq_bwd_x3_persist_supported(int B, int H) {
    if (!(B > 0 && B % X_SR == 0 && H >= 8 && H <= 1024)) return 0;
    const int cap = bwd_x3_capacity();
    const int nC = cdiv(H, X_UN);
    int R = B / X_SR, ns = 1;
    while (R * nC > cap / 2 && ns < X_MAXNS && R % 2 == 0) { R /= 2; ns *= 2; }
    return (R * nC <= cap / 2 && R <= 64) ? ns : 0;
}
// floats of one ring slot of the partial-sum buffer: [chains][nC consumers][nC producers][32][16]
size_t lstm_seq_bwd_x3_part_slot_floats(int B, int H) {
    const size_t nC = (size_t)cdiv(H, X_UN);
    return (size_t)(B / X_SR) * nC * nC * X_SR * X_UN;
}

static int prep_y(SeqBwdX3Args& a) {
    const int ns = lstm_seq_bwd_x3_persist_supported(a.B, a.H);
    S2VT_REQUIRE(ns > 0, "lstm_seq_bwd_x3_persist: unsupported shape (B %% 32, H <= 1024) or it does not fit the device's resident capacity");
    S2VT_REQUIRE(a.t1 > a.t0 && a.t0 >= 0 && a.t1 <= a.T && a.wtp && a.stash_dg && a.c_all && a.dc && a.part && a.sync && a.err &&
                     a.dh_first >= 0, "lstm_seq_bwd_x3_persist: bad arguments");
    S2VT_REQUIRE(a.Kp == (a.H + 63) / 64 * 64 && a.Hp == cdiv(a.H, X_UN) * X_UN && a.ldw >= 4 * (int64_t)a.Hp && a.ldw % 8 == 0 &&
                     a.wplane % 8 == 0 && (reinterpret_cast<uintptr_t>(a.wtp) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.part) & 15) == 0,
                 "lstm_seq_bwd_x3_persist: W_hh^T planes must be [3][Kp][4 Hp], 16-byte aligned");
    S2VT_REQUIRE(a.nslots > a.t1 - a.t0 && (size_t)a.part_slot >= lstm_seq_bwd_x3_part_slot_floats(a.B, a.H) && a.part_slot % 4 == 0,
                 "lstm_seq_bwd_x3_persist: the partial-sum ring needs more slots than the launch has timesteps");
    S2VT_REQUIRE(!a.dgp || (a.H % 8 == 0 && a.B % 64 == 0 && a.lddgp >= 3 * (int64_t)((4 * a.H + 63) / 64 * 64) && a.lddgp % 8 == 0 &&
                            (reinterpret_cast<uintptr_t>(a.dgp) & 15) == 0),
                 "lstm_seq_bwd_x3_persist: the dG row image needs H %% 8 == 0, B %% 64 == 0 and rows of 3 * pad64(4H) elements");
    S2VT_REQUIRE(!a.skip_dg || a.dgp, "lstm_seq_bwd_x3_persist: skip_dg without a plane image");
    a.NS = ns;
    a.RB = ns * X_SR;
    return 0;
}

int lstm_seq_bwd_x3_persist2(hipStream_t stream, SeqBwdX3Args a, const SeqBwdX3Args* b) {
    int rc;
    if ((rc = prep_y(a))) return rc;
    SeqBwdX3Args bb = b ? *b : a;
    if (b) {
        if ((rc = prep_y(bb))) return rc;
        S2VT_REQUIRE(bb.sync != a.sync && bb.part != a.part, "lstm_seq_bwd_x3_persist: paired layers need their own counters and partial sums");
    }
    const int nC = cdiv(a.H, X_UN);
    const int na = (a.B / a.RB) * nC, nb = b ? (bb.B / bb.RB) * cdiv(bb.H, X_UN) : 0;
    S2VT_REQUIRE(na + nb <= bwd_x3_capacity(), "lstm_seq_bwd_x3_persist: %d workgroups would not be co-resident (device capacity %d)",
                 na + nb, bwd_x3_capacity());
    // the hand-off counters count finished timesteps of the whole sequence: zeroed with its first block (t1 == T) only
    if (a.t1 == a.T) S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (b && bb.t1 == bb.T) S2VT_HIP(hipMemsetAsync(bb.sync, 0, lstm_persist_sync_bytes(), stream));
    const int G = (na + nb) / nC;
    int xg = 0, grid = na + nb;
    if ((!b || (bb.B == a.B && bb.H == a.H)) && G > 0 && G <= 8 && 8 % G == 0) {
        const int padded = 8 * cdiv(nC, 8 / G);
        if (padded <= bwd_x3_capacity()) { xg = G; grid = padded; }
    }
    hipLaunchKernelGGL(lstm_seq_bwd_x3_persist_kernel, dim3(grid), dim3(X_NT), 0, stream, a, bb, na, xg);
    S2VT_LAUNCH_CHECK("lstm_seq_bwd_x3_persist_kernel");
    return 0;
}

// W_hh^T fp32 [H][4H] (row j: the 4H gate columns k = g H + u) -> three bf16 planes [3][Kp][4 Hp] with the gate blocks re-based
// to k' = g Hp + u (Hp = H rounded up to 16: every workgroup's 16-unit block is 16-byte aligned), pad rows / columns zero
__global__ __launch_bounds__(256) void split3_wt_kernel(const float* __restrict__ wt, int H, int Kp, int Hp,
                                                        unsigned short* __restrict__ dst, int64_t plane) {
    const int64_t n = (int64_t)Kp * (2 * Hp);             // pairs of k'
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / (2 * Hp)), kp = (int)(i % (2 * Hp)) * 2;
        const int g = kp / Hp, u = kp % Hp;
        unsigned short a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        if (j < H && u < H) split3_bits(wt[(int64_t)j * 4 * H + (int64_t)g * H + u], a);
        if (j < H && u + 1 < H) split3_bits(wt[(int64_t)j * 4 * H + (int64_t)g * H + u + 1], b);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<unsigned int*>(dst + pl * plane + (int64_t)j * 4 * Hp + kp) = (unsigned int)a[pl] | ((unsigned int)b[pl] << 16);
    }
}
int split3_wt(hipStream_t stream, const float* wt, int H, int Kp, int Hp, unsigned short* dst, int64_t plane) {
    S2VT_REQUIRE(wt && dst && H > 0 && Kp >= H && Hp >= H && Hp % 2 == 0 && plane >= (int64_t)Kp * 4 * Hp, "split3_wt: bad arguments");
    const int64_t n = (int64_t)Kp * (2 * Hp);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(split3_wt_kernel, dim3(grid), dim3(256), 0, stream, wt, H, Kp, Hp, dst, plane);
    S2VT_LAUNCH_CHECK("split3_wt_kernel");
    return 0;
}

}  // namespace s2vt
