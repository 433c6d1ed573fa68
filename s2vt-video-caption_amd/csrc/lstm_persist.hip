// Persistent bf16 LSTM recurrence for gfx950 (BASELINE config 3: B = 256, bf16 operands, fp32 accumulate):
// ONE launch runs a whole block of timesteps of one layer - or of both layers side by side - instead of one launch
// per timestep (S2VTModel.py:67 / :77 -> nn.LSTM over the steps).
//
// Why: a one-launch-per-timestep kernel re-fetches W_hh (8 MB bf16) through the fabric on every step, because the
// per-XCD L2s do not survive a kernel boundary, and pays launch + prologue + epilogue serially 159 times per layer.
// Here every workgroup keeps ITS slice of W_hh in registers for all steps; what crosses the chip per step is only
// h_{t-1} (512 KB at B = 256).
//
// Decomposition.  Workgroup (rg, cs) owns batch rows [rg*RB, rg*RB+RB) and hidden units [16 cs, 16 cs + 16), i.e. the
// 64 gate columns {i,f,g,o} x 16 units: complete cells.  Its 4 waves are (kw, cw): k half and 8-unit half; each wave
// holds W[its 32 gate columns][its k half] as MFMA B operands in 128 VGPRs (v_mfma_f32_32x32x16_bf16; B operand of lane
// (n, kh) = 8 consecutive k of column n).  A workgroup needs 74 KB of LDS and 4 waves: TWO of them share a compute unit,
// and each hides the other's hand-off latencies (measured on the first version, one 8-wave workgroup per CU: of a
// 5.9-us sub-step 2.0 us were the h_{t-1} transfer, 1.4 us the signal round trip; per-CU ingest from L2 is ~65 GB/s).
// With two layer argument sets in one launch the grid is [layer A workgroups | layer B workgroups]: vid_rnn block k+1
// and word_rnn block k of the layer pipeline run side by side with no second stream.
//
// One sub-step (32 rows x 64 gate columns x Kp) of a workgroup:
//   1. wait until all column slices of the row group have published h_{t-1} (one counter per 32-row chain, one lane
//      polls, bounded spin);
//   2. LDS-DMA the chain's h_{t-1} rows (32 x Kp bf16, <= 64 KB) into LDS: one global_load_lds_dwordx4 per 8 rows x
//      128 B, XOR-swizzled through the per-lane SOURCE address (position q of row r holds piece q ^ ((r>>1)&7)) so the
//      A-fragment ds_read_b128 are conflict-free; all requests are issued up front and consumed k chunk by k chunk
//      behind counted vmcnt waits;
//   3. 32 MFMAs per wave (A from LDS, B from registers); the two k halves are summed through LDS;
//   4. cell epilogue (2 adjacent units per thread, c_t stays in LDS between steps): activated gates -> stash, c_t, h_t
//      (fp32) as plain 8-byte stores; the bf16 h_t tile (32 rows x 32 B) goes through LDS and leaves as ONE 16-byte
//      write-through (sc1) store instruction of wave 0;
//   5. wave 0 drains its stores (vmcnt(0)) and one lane adds 1 to the chain's counter (agent scope).
// Hand-off correctness (cdna guide, Guideline 16): payload stored sc1 by ONE wave and drained before that wave's
// counter add; the consumer polls with an sc1 load, then a workgroup barrier, then loads.  h_t of every step has its
// own address (time-major h image), so no CU ever holds an older copy of a line it is about to read; the LDS-DMA loads
// carry sc1 as well.  Every spin is bounded (1 s of wall clock): on a time-out the workgroup sets *err and exits.
// All workgroups of a launch must be co-resident (2 per CU): the launchers ask the runtime how many workgroups of the
// kernel the device holds at once (compute units x occupancy, coresident_capacity()) and a shape that does not fit is
// reported as unsupported, so the drivers fall back to one launch per timestep (api.hip).
#include <mutex>

#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int P_SR = 32;                          // batch rows per sub-step (= per chain)
constexpr int P_UN = 16;                          // hidden units per workgroup (64 gate columns)
constexpr int P_NT = 256;                         // 4 waves
constexpr int P_KCH = 16;                         // k chunks of 64 (Kp <= 1024)
constexpr int P_SLAB = P_KCH * P_SR * 128;        // h_{t-1} image: 64 KB
constexpr int P_RLD = 72;                         // row stride of the partial-sum image (floats): 8 mod 64, so the epilogue's
                                                  // f32x2 reads (4 rows x {0-7, 32-39} per half-wave) and the partial writes are conflict-free
constexpr int P_HSM = P_SLAB;                     // bf16 h_t tile [32][16]
constexpr int P_MAXNS = 4;
constexpr int P_CST = P_HSM + P_SR * P_UN * 2;    // fp32 c_t of the workgroup's cells, per chain [32][16]
constexpr int P_LDS = P_CST + P_MAXNS * P_SR * P_UN * 4;
constexpr int P_MAX_WG = 512;                     // design point: 2 workgroups on each of 256 CUs (config 3 needs 2 x 252);
                                                  // the launchers cap this by what the device reports (persist_capacity)
constexpr unsigned long long P_SPIN_TICKS = 100000000ull;      // 1 s of the 100-MHz wall clock

__device__ __forceinline__ unsigned short f2bf_rn(float x) {
    unsigned int u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// sigmoid / tanh on the hardware exp and reciprocal (1 ulp each): absolute error ~2e-7, far below the bf16 rounding of
// the operands this kernel works on
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

#ifdef S2VT_EXPERIMENT_PLAIN_LOADS      // timing experiment only (tools/bench_bptt_stamps.py): what would hand-off loads without sc1 cost?
#define P_LOAD_AUX 0
#else
#define P_LOAD_AUX 16 /* sc1 */
#endif
__device__ __forceinline__ void glds16_sc1(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, P_LOAD_AUX);
}

// one lane waits for *cnt >= target (relaxed agent-scope = sc1 loads); false on time-out
__device__ __forceinline__ bool spin_until(const unsigned int* cnt, unsigned int target) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned int v = __hip_atomic_load((gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > P_SPIN_TICKS) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

#define P_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
// workgroup barrier WITHOUT the vmcnt(0) drain __syncthreads() implies: global loads/stores stay in flight across it
#define P_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// FULL: Kp == 1024 (16 k chunks: the config-3 shape), counted vmcnt pipeline; otherwise every request is awaited first
template <bool FULL>
__device__ __forceinline__ void seq_fwd_body(const SeqFwdBf16Args& p, const int bid, unsigned char* smem, int& s_flag) {
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kw = wave >> 1, cw = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, B = p.B;
    const int nC = (H + P_UN - 1) / P_UN;
    const int cs = bid % nC, rg = bid / nC;
    const int u0 = cs * P_UN, row0 = rg * p.RB;
    const int nch = FULL ? P_KCH : (p.Kp >> 6), half = FULL ? P_KCH / 2 : ((nch + 1) >> 1);
    const int cbeg = kw * half, cend = (cbeg + half < nch) ? cbeg + half : nch;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);

    // ---- this wave's W_hh slice: gate column n = g*8 + uu of the wave <-> W_hh row g*H + (u0 + 8 cw + uu)
    bf16x8 wreg[32];
    {
        const int g = li >> 3, unit = u0 + cw * 8 + (li & 7);
        const unsigned short* wrow = p.wb + ((int64_t)g * H + unit) * p.ldwb + lh * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = cbeg + j;
                const unsigned short* q = (unit < H && c < cend) ? wrow + c * 64 + s * 16 : zero;
                wreg[j * 4 + s] = *reinterpret_cast<const bf16x8*>(q);
            }
        // the slice is complete in registers before the step loop (otherwise hipcc places its vmcnt(0) for these loads
        // at their first use, inside the loop, in front of the LDS-DMA pipeline)
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wreg[i]));
    }

    // ---- loader role: wave w moves rows 8w..8w+7 of the 32 of every k chunk (chunk image: 32 rows x 128 B = 4 KB)
    const int lrow = wave * 8 + (lane >> 3);
    const int lpiece = (lane & 7) ^ ((lrow >> 1) & 7);
    // ---- A-fragment read address of lane (row li, k half lh): piece 2s+lh of a chunk
    unsigned fa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        fa[s] = lbase + (unsigned)(cbeg * 4096 + li * 128 + (((2 * s + lh) ^ ((li >> 1) & 7)) * 16));

    // ---- epilogue role: 2 adjacent units of one row per thread, the same (row, units) in every step
    const int erow = tid >> 3, eul = (tid & 7) * 2;
    const int eunit = u0 + eul;
    const bool e_ok0 = eunit < H, e_ok1 = eunit + 1 < H;
    const bool e_vec = e_ok1 && ((H & 1) == 0);        // 8-byte accesses: both units valid and rows 8-byte aligned
    const int ecol = (eul >> 3) * 32 + (eul & 7);      // + g*8: column of gate g inside the workgroup's 64
    // c_t of this thread's cells lives in LDS between steps (only this thread touches its entries)
    float* cst = reinterpret_cast<float*>(smem + P_CST);
    for (int s = 0; s < p.NS; ++s) {
        const int b = row0 + s * P_SR + erow;
        f32x2 c0 = {0.f, 0.f};
        if (p.t0 > 0 && b < B) {
            const float* q = p.c_all + ((int64_t)(p.t0 - 1) * B + b) * H + eunit;
            if (e_ok0) c0[0] = q[0];
            if (e_ok1) c0[1] = q[1];
        }
        *reinterpret_cast<f32x2*>(cst + (s * P_SR + erow) * P_UN + eul) = c0;
    }

    float* red = reinterpret_cast<float*>(smem);
    unsigned short* hsm = reinterpret_cast<unsigned short*>(smem + P_HSM);
    const int64_t H4 = 4 * (int64_t)H;

    for (int t = p.t0; t < p.t1; ++t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * P_SR;            // first batch row of this sub-step
            unsigned int* cnt = p.sync + (rbase / P_SR) * 32;          // one counter per 32-row chain, whatever NS the launch uses
            const int xrec = (bid == p.stamp_block) ? (t - p.t0) * p.NS + s : -1;
            XSTAMP(p.stamps, xrec, 0);

            if (t > p.t0) {          // h_{t-1} of this chain published by every column slice of the row group?
                if (tid == 0) {
                    const bool ok = spin_until(cnt, (unsigned int)(nC * t));     // every workgroup of the chain has finished steps 0..t-1
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                P_BARRIER();
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

            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;

            if (t > 0) {
                // h_{t-1} rows of this chain -> LDS (chunk c at c*4 KB; this wave's piece at + wave*1 KB)
                const unsigned char* src = reinterpret_cast<const unsigned char*>(
                                               p.hb + ((int64_t)(t - 1) * B + rbase + lrow) * p.ldhb) + lpiece * 16;
                unsigned char* ldst = smem + wave * 1024;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (FULL || i < half) glds16_sc1(src + i * 128, ldst + i * 4096);
                    if (FULL || half + i < nch) glds16_sc1(src + (half + i) * 128, ldst + (half + i) * 4096);
                }
                if (!FULL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                XSTAMP(p.stamps, xrec, 2);
                // pair I = chunks (I, half + I): 2 requests of this wave; 2*(7-I) younger ones may stay in flight
#define P_CHUNK(I, VM)                                                                                            \
                if (FULL || (I) < half) {                                                                          \
                    if (FULL) asm volatile("s_waitcnt vmcnt(" #VM ")" ::: "memory");                               \
                    asm volatile("s_barrier" ::: "memory");                                                        \
                    if (FULL || cbeg + (I) < cend) {                                                               \
                        bf16x8 a0, a1, a2, a3;                                                                     \
                        P_DSR(a0, fa[0], (I) * 4096); P_DSR(a1, fa[1], (I) * 4096);                                \
                        P_DSR(a2, fa[2], (I) * 4096); P_DSR(a3, fa[3], (I) * 4096);                                \
                        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a0));                                           \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wreg[(I) * 4 + 0], acc, 0, 0, 0);        \
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a1));                                           \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wreg[(I) * 4 + 1], acc, 0, 0, 0);        \
                        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a2));                                           \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wreg[(I) * 4 + 2], acc, 0, 0, 0);        \
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a3));                                           \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, wreg[(I) * 4 + 3], acc, 0, 0, 0);        \
                    }                                                                                              \
                }
                P_CHUNK(0, 14)
                XSTAMP(p.stamps, xrec, 3);
                P_CHUNK(1, 12) P_CHUNK(2, 10) P_CHUNK(3, 8)
                P_CHUNK(4, 6) P_CHUNK(5, 4) P_CHUNK(6, 2) P_CHUNK(7, 0)
#undef P_CHUNK
            }

            XSTAMP(p.stamps, xrec, 4);
            P_BARRIER();                       // every wave is done with the h image: the partial sums may take its place
            {
                float* rp = red + (kw * P_SR) * P_RLD;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rp[((r & 3) + 8 * (r >> 2) + 4 * lh) * P_RLD + cw * 32 + li] = acc[r];
            }
            P_BARRIER();
            XSTAMP(p.stamps, xrec, 5);

            f32x2 gate[4], cv, hv;
            {
                f32x2 pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pre[g] = *reinterpret_cast<const f32x2*>(red + erow * P_RLD + ecol + g * 8) +
                             *reinterpret_cast<const f32x2*>(red + (P_SR + erow) * P_RLD + ecol + g * 8) + gxv[g];
                f32x2* cp = reinterpret_cast<f32x2*>(cst + (s * P_SR + erow) * P_UN + eul);
                const f32x2 cprev = *cp;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gate[0][j] = fast_sigmoid(pre[0][j]);
                    gate[1][j] = fast_sigmoid(pre[1][j]);
                    gate[2][j] = fast_tanh(pre[2][j]);
                    gate[3][j] = fast_sigmoid(pre[3][j]);
                    const bool ok = rok && (j ? e_ok1 : e_ok0);
                    cv[j] = ok ? gate[1][j] * cprev[j] + gate[0][j] * gate[2][j] : 0.f;
                    hv[j] = gate[3][j] * fast_tanh(cv[j]);
                }
                *cp = cv;
                *reinterpret_cast<unsigned int*>(hsm + erow * P_UN + eul) =
                    (unsigned int)f2bf_rn(hv[0]) | ((unsigned int)f2bf_rn(hv[1]) << 16);
            }
            XSTAMP(p.stamps, xrec, 6);
            P_BARRIER();
            if (wave == 0) {   // bf16 h_t tile: 32 rows x 32 B = ONE 16-byte write-through store instruction, issued first:
                               // it is what the other workgroups wait for
                const int rl = lane >> 1, part = lane & 1;
                const u32x4 v = *reinterpret_cast<const u32x4*>(hsm + rl * P_UN + part * 8);
                unsigned short* dst = p.hb + ((int64_t)t * B + rbase + rl) * p.ldhb + u0 + part * 8;
                if (rbase + rl < B) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
            if (rok) {
                const int64_t rowi = (int64_t)t * B + eb;
                float* cdst = p.c_all + rowi * H + eunit;
                float* hdst = p.h_all ? p.h_all + rowi * H + eunit : nullptr;
                float* st = p.gx_stash + rowi * H4 + eunit;
                if (e_vec) {
                    *reinterpret_cast<f32x2*>(cdst) = cv;
                    if (hdst) *reinterpret_cast<f32x2*>(hdst) = hv;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x2*>(st + (int64_t)g * H) = gate[g];
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (j ? e_ok1 : e_ok0) {
                            cdst[j] = cv[j];
                            if (hdst) hdst[j] = hv[j];
#pragma unroll
                            for (int g = 0; g < 4; ++g) st[(int64_t)g * H + j] = gate[g][j];
                        }
                }
            }
            XSTAMP(p.stamps, xrec, 7);
            if (wave == 0) {   // the ONE wave that stored the hand-off payload drains and signals for the workgroup
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                XSTAMP(p.stamps, xrec, 8);
                if (lane == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            XSTAMP(p.stamps, xrec, 9);
            P_BARRIER();       // hsm / cst / partial sums are free again
        }
    }
}

// Which role a block plays: (layer, workgroup index inside the layer = rg * nC + cs).  xg == 0: blocks [0, na) are layer A in
// order, the rest layer B.  xg = G > 0 (XCD-aware, the launchers choose it when 8 % G == 0 and G * nC % 8 == 0): the hardware
// deals workgroups to the 8 XCDs round-robin (block b -> XCD b % 8, speed only - nothing depends on it for correctness); a GROUP
// is the nC column slices of one (layer, row group), i.e. the workgroups that read the SAME rows in every sub-step, and group g
// is dealt to the XCDs {g, g + G, ..}: its rows then enter 8/G L2s instead of all eight, and a line is shared by nC * G / 8
// readers of one L2 instead of nC / 8.
__device__ __forceinline__ int persist_role(int bid, int na, int nC, int xg, bool& layer_b) {
    if (xg <= 0) { layer_b = bid >= na; return layer_b ? bid - na : bid; }
    const int x = bid & 7, q = bid >> 3, per = 8 / xg;
    const int g = x % xg, cs = q * per + x / xg;
    const int rgs = na / nC;                         // row groups of layer A (= of layer B: the launcher checked)
    layer_b = g >= rgs;
    return (layer_b ? g - rgs : g) * nC + cs;
}
// the launcher's side of it: G groups, or 0 for the plain order
static int xcd_groups(int na, int nb, int nC) {
    if (nC <= 0 || na % nC || nb % nC) return 0;
    if (nb && nb != na) return 0;
    const int G = (na + nb) / nC;
    return (G > 0 && 8 % G == 0 && ((na + nb) % 8) == 0) ? G : 0;
}

// grid = [na workgroups of layer pa | workgroups of layer pb] (nb may be 0), or dealt by persist_role
template <bool FULL>
__global__ __launch_bounds__(P_NT, 2) void lstm_seq_fwd_bf16_persist_kernel(SeqFwdBf16Args pa, SeqFwdBf16Args pb, int na, int xg) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[P_LDS];
    __shared__ int s_flag;                 // poll result of the polling lane
    bool lb;
    const int vb = persist_role((int)blockIdx.x, na, (pa.H + P_UN - 1) / P_UN, xg, lb);
    if (!lb) seq_fwd_body<FULL>(pa, vb, smem, s_flag);
    else seq_fwd_body<FULL>(pb, vb, smem, s_flag);
}

int lstm_seq_fwd_bf16_persist_supported(int B, int H, int Kp);
size_t lstm_persist_sync_bytes();

int coresident_capacity(const void* kernel, int block) {
    struct Entry { int dev; const void* k; int cap; };
    static Entry cache[32];
    static int n = 0;
    static std::mutex mu;               // the forward (caller's thread) and the backward (autograd's thread) both size launches
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    for (int i = 0; i < n; ++i)
        if (cache[i].dev == dev && cache[i].k == kernel) return cache[i].cap;
    int cus = 0, occ = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block, 0) != hipSuccess) occ = 0;
    (void)hipGetLastError();
    const int cap = (cus > 0 && occ > 0) ? cus * occ : 0;
    if (n < 32) cache[n++] = Entry{dev, kernel, cap};
    return cap;
}
// what one launch of the bf16 kernels may hold: the smaller of the three kernels' capacities, at most the design point
static int persist_capacity() {
    int cap = P_MAX_WG;
    const int c1 = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_fwd_bf16_persist_kernel<true>), P_NT);
    const int c2 = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_fwd_bf16_persist_kernel<false>), P_NT);
    if (c1 < cap) cap = c1;
    if (c2 < cap) cap = c2;
    return cap;
}

// =============================================================================================== backward (BPTT)
// Same decomposition and hand-off protocol, time reversed: workgroup (rg, cs) owns rows x 16 hidden units, keeps
// W_hh^T[its 16 units][all 4H] in registers (4 waves split k: 128 VGPRs each, v_mfma_f32_16x16x32_bf16) and per sub-step
//   dh[32 rows x 16 units] = dG_{t+1}[32 rows x 4H] . W_hh[4H x 16 units]  (+ dh_out_t)  ->  cell derivatives  ->  dG_t, dc.
// The A operand is 4x wider than the forward's (4H = 4000 -> 4032 bf16 per row: 258 KB per sub-step, far more than LDS):
// every wave streams ITS k quarter through a private 4-slot LDS ring (4 KB per 64-k chunk, LDS-DMA, XOR-swizzled
// like the forward image), so the contraction needs no workgroup barrier at all; the four partial tiles meet in LDS.
// The sub-step is bound by the ~65 GB/s a CU can take in from L2 (4 us for 258 KB); two workgroups per CU (the other
// chain, or the other layer of a fused launch) overlap their epilogues and hand-offs with it.
// dG_t leaves as fp32 (in place of the gate stash: the batched weight-gradient GEMMs' source) and as bf16 rows
// (next step's operand AND the k-major plane of the batched bf16 GEMMs); the bf16 tile (32 rows x 4 gates x 32 B) goes
// through LDS and is stored write-through by wave 0, which then drains and signals the chain's counter.
constexpr int Q_NSLOT = 4;                           // ring slots per wave
constexpr int Q_RING = Q_NSLOT * 4096;               // 16 KB per wave
constexpr int Q_KCH = 64;                            // k chunks of 64 (4H padded <= 4096)

// Two shapes of the same kernel (template parameters NW = waves, UN = hidden units per workgroup):
//   <4, 16>: round 2's - 16 units, 4 waves (16 k chunks each), 76 KB of LDS, TWO workgroups per compute unit;
//   <8, 32>: 32 units, 8 waves (8 k chunks each, two 16-unit MFMA column tiles per wave), 152 KB of LDS, ONE per compute unit.
// What a sub-step costs is the dG_{t+1} rows a workgroup takes in (32 rows x 4H bf16 = 258 KB, whatever UN is): with twice the
// units per workgroup the layer needs half as many workgroups, so half as many bytes cross the L2 -> LDS paths per timestep
// (B x 4H x 2 B x H/UN: 130 MB per layer timestep at B = 256 with UN = 16 - at the ~16 TB/s all L2s deliver together that
// alone is 7.8 us - and 65 MB with UN = 32).  The W_hh^T slice of a workgroup (UN x 4H bf16 = 258 KB at UN = 32) still fits the
// registers of its waves (128 VGPRs each).
template <int NW, int UN>
struct BwdCfg {
    static constexpr int NT = NW * 64;                       // threads
    static constexpr int CPW = Q_KCH / NW;                   // k chunks per wave
    static constexpr int NTU = UN / 16;                      // 16-unit MFMA column tiles
    static constexpr int RLD = UN + 2;                       // row stride of a partial tile (floats)
    static constexpr int DGSM = NW * Q_RING;                 // bf16 dG_t tile [32][4 * UN]
    static constexpr int DCST = DGSM + P_SR * 4 * UN * 2;    // fp32 dc of the workgroup's cells, per chain [32][UN]
    static constexpr int LDS = DCST + P_MAXNS * P_SR * UN * 4;
    static_assert(CPW * 2 * NTU == 32, "the W_hh^T slice of a wave is 32 bf16x8 registers");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int NW, int UN>
__device__ __forceinline__ void seq_bwd_body(const SeqBwdBf16Args& p, const int bid, unsigned char* smem, int& s_flag) {
    typedef BwdCfg<NW, UN> C;
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, lq = lane >> 4;
    const int H = p.H, B = p.B;
    const int nC = (H + UN - 1) / UN;
    const int cs = bid % nC, rg = bid / nC;
    const int u0 = cs * UN, row0 = rg * p.RB;
    const int nch = p.Kp >> 6;
    const int c0 = wave * C::CPW;                      // this wave's chunks [c0, c0 + CPW)
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);

    // ---- this wave's k range of W_hh^T for the workgroup's units: B operand of step (chunk j, k half ks, unit tile ut)
    bf16x8 wreg[32];
    {
#pragma unroll
        for (int ut = 0; ut < C::NTU; ++ut) {
            const int unit = u0 + ut * 16 + lm;
            const unsigned short* wrow = p.wtb + (int64_t)unit * p.ldwtb + lq * 8;
#pragma unroll
            for (int j = 0; j < C::CPW; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const unsigned short* q = (unit < H && c0 + j < nch) ? wrow + (c0 + j) * 64 + ks * 32 : zero;
                    wreg[(j * 2 + ks) * C::NTU + ut] = *reinterpret_cast<const bf16x8*>(q);
                }
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wreg[i]));
    }

    // ---- loader role (each wave for itself): piece i of a chunk = rows 8i..8i+7; lane -> (row, 16-B position)
    // byte offset of this lane's 16 B inside the 32-row image of a chunk, relative to the chunk's first row:
    // row (8i + lane/8) * row pitch + swizzled piece; (row >> 1) & 7 = (4i + lane/16) & 7, i.e. pieces of odd i differ by ^4
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        voff[i] = (unsigned)((8 * i + (lane >> 3)) * (int)(p.lddgb * 2) + ((((lane & 7) ^ (lane >> 4)) ^ (4 * (i & 1))) * 16));
    unsigned char* ring = smem + wave * Q_RING;
    // ---- A-fragment read addresses: row tile rt, k half ks of a chunk: lane (m, kq) reads piece ks*4 + kq of row rt*16 + m
    unsigned fa[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = rt * 16 + lm;
            fa[rt][ks] = lbase + (unsigned)(wave * Q_RING + row * 128 + (((ks * 4 + lq) ^ ((row >> 1) & 7)) * 16));
        }

    // ---- epilogue role: 2 adjacent units of one row per thread (32 rows x UN / 2 pairs = NT threads)
    const int erow = tid / (UN / 2), eul = (tid % (UN / 2)) * 2;
    const int eunit = u0 + eul;
    const bool e_ok = eunit + 1 < H;                   // H % 8 == 0: a pair is valid or not as a whole
    float* dcst = reinterpret_cast<float*>(smem + C::DCST);
    const bool last_block = (p.t1 == p.T);
    for (int s = 0; s < p.NS; ++s) {
        const int b = row0 + s * P_SR + erow;
        f32x2 d0 = {0.f, 0.f};
        if (!last_block && e_ok && b < B) d0 = *reinterpret_cast<const f32x2*>(p.dc + (int64_t)b * H + eunit);
        *reinterpret_cast<f32x2*>(dcst + (s * P_SR + erow) * UN + eul) = d0;
    }
    unsigned short* dgsm = reinterpret_cast<unsigned short*>(smem + C::DGSM);
    const int64_t H4 = 4 * (int64_t)H;

    for (int t = p.t1 - 1; t >= p.t0; --t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * P_SR;
            unsigned int* cnt = p.sync + (rbase / P_SR) * 32;          // one counter per 32-row chain, whatever NS the launch uses
            const int done = p.t1 - 1 - t;             // steps of this launch already finished by every workgroup?
            const int done_all = p.T - 1 - t;          // ... and of the whole sequence: the counters run on from launch to launch
            const int xrec = (bid == p.stamp_block) ? done * p.NS + s : -1;
            XSTAMP(p.stamps, xrec, 0);
            if (done > 0) {
                if (tid == 0) {
                    const bool ok = spin_until(cnt, (unsigned int)(nC * done_all));
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                P_BARRIER();
                if (s_flag == 0) return;
            }

            XSTAMP(p.stamps, xrec, 1);
            // epilogue operands requested now (older than every ring request: they never hold a counted wait up)
            const int eb = rbase + erow;
            const bool ok = e_ok && eb < B;
            const int64_t rowi = (int64_t)t * B + eb;
            f32x2 stv[4], cv, cpv, dhov;
            {
                const float* st = p.stash_dg + rowi * H4 + eunit;
#pragma unroll
                for (int g = 0; g < 4; ++g) stv[g] = *reinterpret_cast<const f32x2*>(ok ? st + (int64_t)g * H : g_zero4);
                cv = *reinterpret_cast<const f32x2*>(ok ? p.c_all + rowi * H + eunit : g_zero4);
                cpv = *reinterpret_cast<const f32x2*>((ok && t > 0) ? p.c_all + (rowi - B) * H + eunit : g_zero4);
                dhov = *reinterpret_cast<const f32x2*>((ok && p.dh_out && t >= p.dh_first)
                                                           ? p.dh_out + ((int64_t)(t - p.dh_first) * B + eb) * H + eunit : g_zero4);
            }

            f32x4 acc[2][C::NTU];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ut = 0; ut < C::NTU; ++ut) acc[rt][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < p.T - 1) {
                const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.dgb + ((int64_t)(t + 1) * B + rbase) * p.lddgb);
                // a chunk index past the end (the 64th of 63) re-reads the last chunk: its W registers are zero, and every
                // wave issues the same number of requests, so the counted waits below are exact
#define Q_ISSUE(J)                                                                                           \
                {                                                                                             \
                    const int cj = (c0 + (J) < nch) ? c0 + (J) : nch - 1;                                     \
                    const unsigned char* sb = abase + cj * 128;                                               \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                        glds16_sc1(sb + voff[i], ring + ((J) % Q_NSLOT) * 4096 + i * 1024);                   \
                }
#define Q_MFMA(J, KS, A0, A1)                                                                                \
                _Pragma("unroll") for (int ut = 0; ut < C::NTU; ++ut) {                                       \
                    acc[0][ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, wreg[((J) * 2 + (KS)) * C::NTU + ut], acc[0][ut], 0, 0, 0); \
                    acc[1][ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, wreg[((J) * 2 + (KS)) * C::NTU + ut], acc[1][ut], 0, 0, 0); \
                }
#define Q_STEP(J, VM)                                                                                        \
                if ((J) < C::CPW) {                                                                           \
                    asm volatile("s_waitcnt vmcnt(" #VM ")" ::: "memory");                                    \
                    bf16x8 a00, a01, a10, a11;                                                                \
                    P_DSR(a00, fa[0][0], ((J) % Q_NSLOT) * 4096); P_DSR(a10, fa[1][0], ((J) % Q_NSLOT) * 4096); \
                    P_DSR(a01, fa[0][1], ((J) % Q_NSLOT) * 4096); P_DSR(a11, fa[1][1], ((J) % Q_NSLOT) * 4096); \
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a00), "+v"(a10));                              \
                    Q_MFMA(J, 0, a00, a10)                                                                    \
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01), "+v"(a11));                              \
                    /* the slot's four fragment reads are in: it takes chunk J + 4 at once, so that FOUR chunks (16 KB per */ \
                    /* wave, 128 KB per CU at 8 waves) are in flight while the MFMAs run - the dG rows were written by     */ \
                    /* other XCDs a sub-step ago and come over the fabric: bytes in flight / latency is the ingest rate    */ \
                    if ((J) + 4 < C::CPW) Q_ISSUE((J) + 4)                                                    \
                    Q_MFMA(J, 1, a01, a11)                                                                    \
                }
                Q_ISSUE(0) Q_ISSUE(1) Q_ISSUE(2) Q_ISSUE(3)
                XSTAMP(p.stamps, xrec, 2);
                // chunk J landed when at most the requests of the younger chunks (J+1..J+3; fewer at the end) are out
                if (C::CPW == 16) {
                    Q_STEP(0, 12) Q_STEP(1, 12) Q_STEP(2, 12) Q_STEP(3, 12) Q_STEP(4, 12) Q_STEP(5, 12) Q_STEP(6, 12)
                    Q_STEP(7, 12) Q_STEP(8, 12) Q_STEP(9, 12) Q_STEP(10, 12) Q_STEP(11, 12) Q_STEP(12, 12)
                    Q_STEP(13, 8) Q_STEP(14, 4) Q_STEP(15, 0)
                } else {
                    Q_STEP(0, 12) Q_STEP(1, 12) Q_STEP(2, 12) Q_STEP(3, 12) Q_STEP(4, 12)
                    Q_STEP(5, 8) Q_STEP(6, 4) Q_STEP(7, 0)
                }
#undef Q_STEP
#undef Q_MFMA
#undef Q_ISSUE
            }
            XSTAMP(p.stamps, xrec, 3);
            // partial tile of this wave -> its own (idle) ring
            {
                float* rp = reinterpret_cast<float*>(ring);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ut = 0; ut < C::NTU; ++ut)
#pragma unroll
                        for (int r = 0; r < 4; ++r) rp[(rt * 16 + 4 * lq + r) * C::RLD + ut * 16 + lm] = acc[rt][ut][r];
            }
            P_BARRIER();
            XSTAMP(p.stamps, xrec, 4);

            f32x2 dg[4], dcn;
            {
                f32x2 dh = dhov;
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    dh += *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(smem + w * Q_RING) + erow * C::RLD + eul);
                f32x2* dp = reinterpret_cast<f32x2*>(dcst + (s * P_SR + erow) * UN + eul);
                const f32x2 dcv = *dp;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float ig = stv[0][j], fg = stv[1][j], gg = stv[2][j], og = stv[3][j];
                    const float tc = fast_tanh(cv[j]);
                    const float dc = dh[j] * og * (1.0f - tc * tc) + dcv[j];
                    const float d_o = dh[j] * tc;
                    dg[0][j] = dc * gg * ig * (1.0f - ig);
                    dg[1][j] = dc * cpv[j] * fg * (1.0f - fg);
                    dg[2][j] = dc * ig * (1.0f - gg * gg);
                    dg[3][j] = d_o * og * (1.0f - og);
                    dcn[j] = ok ? dc * fg : 0.f;
                }
                *dp = dcn;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<unsigned int*>(dgsm + erow * (4 * UN) + g * UN + eul) =
                        ok ? ((unsigned int)f2bf_rn(dg[g][0]) | ((unsigned int)f2bf_rn(dg[g][1]) << 16)) : 0u;
            }
            XSTAMP(p.stamps, xrec, 5);
            P_BARRIER();
            XSTAMP(p.stamps, xrec, 6);
            if (wave == 0) {   // bf16 dG_t tile: 4 gates x (32 rows x UN*2 B): 16-byte write-through stores of ONE wave
                constexpr int PPR = UN / 8;                                  // 16-byte parts per (row, gate)
#pragma unroll
                for (int pass = 0; pass < (P_SR * PPR) / 64; ++pass) {
                    const int idx = pass * 64 + lane, rl = idx / PPR, part = idx % PPR;
                    if (rbase + rl < B && u0 + part * 8 < H) {
                        unsigned short* drow = p.dgb + ((int64_t)t * B + rbase + rl) * p.lddgb + u0 + part * 8;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(dgsm + rl * (4 * UN) + g * UN + part * 8);
                            unsigned short* dst = drow + (int64_t)g * H;
#ifdef S2VT_EXPERIMENT_PLAIN_STORES     // timing experiment only (tools/bench_bptt_stamps.py): hand-off stores that stay in the producer's L2
                            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
#else
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
#endif
                        }
                    }
                }
            }
            if (ok) {
                float* st = p.stash_dg + rowi * H4 + eunit;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x2*>(st + (int64_t)g * H) = dg[g];
                if (t == p.t0) *reinterpret_cast<f32x2*>(p.dc + (int64_t)eb * H + eunit) = dcn;      // carried to the next launch
            }
            XSTAMP(p.stamps, xrec, 7);
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                XSTAMP(p.stamps, xrec, 8);
                if (lane == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            XSTAMP(p.stamps, xrec, 9);
            P_BARRIER();       // rings (partial tiles), dG tile and dc state are free again
        }
    }
}

template <int NW, int UN>
__global__ __launch_bounds__(NW * 64, (NW == 4) ? 2 : 1) void lstm_seq_bwd_bf16_persist_kernel(SeqBwdBf16Args pa, SeqBwdBf16Args pb, int na, int xg) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BwdCfg<NW, UN>::LDS];
    __shared__ int s_flag;
    bool lb;
    const int vb = persist_role((int)blockIdx.x, na, (pa.H + UN - 1) / UN, xg, lb);
    if (!lb) seq_bwd_body<NW, UN>(pa, vb, smem, s_flag);
    else seq_bwd_body<NW, UN>(pb, vb, smem, s_flag);
}

// chains per workgroup such that TWO layers of this shape (un hidden units per workgroup) fit `cap` co-resident workgroups
// (0: they do not)
// single: a launch with ONE layer (the first / last stage of the layer pipeline, the test entries) may take the whole device:
// half the chains per workgroup, half the sub-steps per timestep.  The hand-off counters are per 32-row chain, so launches of
// one layer with different NS continue each other.
static int chains_for(int B, int H, int cap, int un, bool single = false) {
    const int nC = cdiv(H, un), lim = single ? cap : cap / 2;
    int R = B / P_SR;                       // row groups (one 32-row chain each) ...
    int ns = 1;
    while (R * nC > lim && ns < P_MAXNS && R % 2 == 0) { R /= 2; ns *= 2; }       // ... merged until a layer fits half
    return (R * nC <= lim && R <= 64) ? ns : 0;                                   // the device (two layers co-run)
}

// Which shape of the BPTT kernel a layer of (B, H) runs with: 32 units per workgroup (one workgroup per compute unit)
// wherever two such layers fit the device, else 16 units (two per compute unit).  Option "bptt_units" = 16 | 32 pins one (the
// kernel tests run both; 0: the rule above).
struct BwdPlan { int un, ns, cap; };
static BwdPlan bwd_plan(int B, int H, int Kp4, bool single = false) {
    BwdPlan none = {0, 0, 0};
    if (!(B > 0 && B % P_SR == 0 && H % 8 == 0 && Kp4 % 64 == 0 && Kp4 >= 4 * H && Kp4 <= 64 * Q_KCH)) return none;
    const int pref = option(O_BPTT_UNITS);
    if (pref != 16) {
        int cap = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_bwd_bf16_persist_kernel<8, 32>), 512);
        if (cap > P_MAX_WG / 2) cap = P_MAX_WG / 2;
        const int ns = chains_for(B, H, cap, 32, single);
        if (ns > 0 && chains_for(B, H, cap, 32) > 0) return BwdPlan{32, ns, cap};        // (the kernel shape is the pair's choice)
    }
    if (pref != 32) {
        int cap = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_bwd_bf16_persist_kernel<4, 16>), 256);
        if (cap > P_MAX_WG) cap = P_MAX_WG;
        const int ns = chains_for(B, H, cap, 16, single);
        if (ns > 0 && chains_for(B, H, cap, 16) > 0) return BwdPlan{16, ns, cap};
    }
    return none;
}

int lstm_seq_bwd_bf16_persist_supported(int B, int H, int Kp4) { return bwd_plan(B, H, Kp4).ns; }

static int prep_bwd(SeqBwdBf16Args& a, BwdPlan* plan, bool single = false) {
    *plan = bwd_plan(a.B, a.H, a.Kp, single);
    S2VT_REQUIRE(plan->ns > 0, "lstm_seq_bwd_bf16_persist: unsupported shape (B %% 32, H %% 8, 4H <= 4096) or two layers of it do not "
                 "fit the workgroups this device keeps resident");
    S2VT_REQUIRE(a.T > 0 && a.t1 > a.t0 && a.t0 >= 0 && a.t1 <= a.T && a.wtb && a.dgb && a.stash_dg && a.c_all && a.dc && a.sync && a.err,
                 "lstm_seq_bwd_bf16_persist: bad arguments");
    S2VT_REQUIRE(a.lddgb % 8 == 0 && a.ldwtb % 8 == 0 && a.lddgb >= a.Kp && a.ldwtb >= a.Kp &&
                     (reinterpret_cast<uintptr_t>(a.dgb) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.wtb) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.stash_dg) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.c_all) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.dc) & 7) == 0 && (!a.dh_out || (reinterpret_cast<uintptr_t>(a.dh_out) & 7) == 0),
                 "lstm_seq_bwd_bf16_persist: operands must be aligned bf16 rows zero-padded to Kp / 8-byte aligned fp32 rows");
    a.NS = plan->ns;
    a.RB = plan->ns * P_SR;
    return 0;
}

int lstm_seq_bwd_bf16_persist2(hipStream_t stream, SeqBwdBf16Args a, const SeqBwdBf16Args* b) {
    int rc;
    BwdPlan pa, pb;
    if ((rc = prep_bwd(a, &pa, b == nullptr))) return rc;
    SeqBwdBf16Args bb = b ? *b : a;
    pb = pa;
    if (b) {
        if ((rc = prep_bwd(bb, &pb))) return rc;
        S2VT_REQUIRE(bb.sync != a.sync, "lstm_seq_bwd_bf16_persist: paired layers need their own counters");
        S2VT_REQUIRE(pb.un == pa.un, "lstm_seq_bwd_bf16_persist: paired layers must run the same kernel shape");
    }
    const int na = (a.B / a.RB) * cdiv(a.H, pa.un), nb = b ? (bb.B / bb.RB) * cdiv(bb.H, pb.un) : 0;
    S2VT_REQUIRE(na + nb <= pa.cap, "lstm_seq_bwd_bf16_persist: %d workgroups would not be co-resident (device capacity %d)",
                 na + nb, pa.cap);
    // the hand-off counters count finished timesteps of the whole sequence: zeroed with its first block only (a memset
    // is a 5-us kernel of its own on this stream: 28 of them per train step when every launch zeroed its counters)
    if (a.t1 == a.T) S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (b && bb.t1 == bb.T) S2VT_HIP(hipMemsetAsync(bb.sync, 0, lstm_persist_sync_bytes(), stream));
    const int xg = (!b || (bb.B == a.B && bb.H == a.H)) ? xcd_groups(na, nb, cdiv(a.H, pa.un)) : 0;
    if (pa.un == 32)
        hipLaunchKernelGGL((lstm_seq_bwd_bf16_persist_kernel<8, 32>), dim3(na + nb), dim3(512), 0, stream, a, bb, na, xg);
    else
        hipLaunchKernelGGL((lstm_seq_bwd_bf16_persist_kernel<4, 16>), dim3(na + nb), dim3(256), 0, stream, a, bb, na, xg);
    S2VT_LAUNCH_CHECK("lstm_seq_bwd_bf16_persist_kernel");
    return 0;
}

int lstm_seq_fwd_bf16_persist_supported(int B, int H, int Kp) {
    if (!(B > 0 && B % P_SR == 0 && Kp % 64 == 0 && Kp >= H && Kp <= 64 * P_KCH)) return 0;
    return chains_for(B, H, persist_capacity(), P_UN);
}

size_t lstm_persist_sync_bytes() { return (size_t)64 * P_MAXNS * 32 * sizeof(unsigned int); }   // <= 64 row groups

static int prep(SeqFwdBf16Args& a, const char* who, bool single = false) {
    // (`single` is not used here: a one-layer forward launch with half the chains per workgroup - 504 workgroups of ONE layer,
    // two per CU - measured slower, 1.97 vs 1.74 ms per config-3 forward: the two workgroups of a CU then belong to the same
    // timestep wave and stop hiding each other's latencies; the one-workgroup-per-CU BPTT kernel gains, 2.64 -> 2.47 ms)
    const int ns = lstm_seq_fwd_bf16_persist_supported(a.B, a.H, a.Kp);
    S2VT_REQUIRE(ns > 0, "%s: unsupported shape (B %% 32, Kp <= 1024) or two layers of it do not fit the %d workgroups this device "
                 "keeps resident", who, persist_capacity());
    S2VT_REQUIRE(a.t1 > a.t0 && a.t0 >= 0 && a.wb && a.hb && a.gx_stash && a.c_all && a.sync && a.err && (a.bias || a.n_gx >= a.t1),
                 "%s: bad arguments", who);
    S2VT_REQUIRE(a.ldhb % 8 == 0 && a.ldwb % 8 == 0 && a.ldhb >= a.Kp && a.ldwb >= a.Kp &&
                     (reinterpret_cast<uintptr_t>(a.hb) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.wb) & 15) == 0,
                 "%s: operands must be 16-B aligned bf16 rows zero-padded to Kp", who);
    a.NS = ns;
    a.RB = ns * P_SR;
    return 0;
}

// one layer (b == nullptr) or two layers side by side in one launch
int lstm_seq_fwd_bf16_persist2(hipStream_t stream, SeqFwdBf16Args a, const SeqFwdBf16Args* b) {
    int rc;
    if ((rc = prep(a, "lstm_seq_fwd_bf16_persist", b == nullptr))) return rc;
    SeqFwdBf16Args bb = b ? *b : a;
    if (b) {
        if ((rc = prep(bb, "lstm_seq_fwd_bf16_persist"))) return rc;
        S2VT_REQUIRE(bb.Kp == a.Kp && bb.sync != a.sync, "lstm_seq_fwd_bf16_persist: paired layers need the same Kp and their own counters");
    }
    const int na = (a.B / a.RB) * cdiv(a.H, P_UN), nb = b ? (bb.B / bb.RB) * cdiv(bb.H, P_UN) : 0;
    S2VT_REQUIRE(na + nb <= persist_capacity(), "lstm_seq_fwd_bf16_persist: %d workgroups would not be co-resident (device capacity %d)",
                 na + nb, persist_capacity());
    // (counters: see the BPTT launcher) zeroed with the sequence's first block
    if (a.t0 == 0) S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (b && bb.t0 == 0) S2VT_HIP(hipMemsetAsync(bb.sync, 0, lstm_persist_sync_bytes(), stream));
    // (no XCD-aware dealing for the forward: with TWO workgroups per compute unit the neighbours on a CU must be out of phase to
    // hide each other's hand-off latencies, and dealing a group to one XCD makes them members of the same chain - measured
    // 1.99 vs 1.84 ms per config-3 forward; the one-per-CU BPTT launch gains from it: 2.90 vs 3.12 ms)
    const int xg = 0;
    if (a.Kp == 64 * P_KCH)
        hipLaunchKernelGGL((lstm_seq_fwd_bf16_persist_kernel<true>), dim3(na + nb), dim3(P_NT), 0, stream, a, bb, na, xg);
    else
        hipLaunchKernelGGL((lstm_seq_fwd_bf16_persist_kernel<false>), dim3(na + nb), dim3(P_NT), 0, stream, a, bb, na, xg);
    S2VT_LAUNCH_CHECK("lstm_seq_fwd_bf16_persist_kernel");
    return 0;
}
int lstm_seq_fwd_bf16_persist(hipStream_t stream, SeqFwdBf16Args a) { return lstm_seq_fwd_bf16_persist2(stream, a, nullptr); }

}  // namespace s2vt
