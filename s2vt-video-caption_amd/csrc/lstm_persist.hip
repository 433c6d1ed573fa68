// Persistent bf16 LSTM recurrence for gfx950 (BASELINE config 3: B = 256, bf16 operands, fp32 accumulate):
// ONE launch runs a whole block of timesteps of one layer (S2VTModel.py:67 / :77 -> nn.LSTM over the steps).
//
// Why: a one-launch-per-timestep kernel re-fetches W_hh (8 MB bf16) through the fabric on every step, because the
// per-XCD L2s do not survive a kernel boundary, and pays launch + prologue + epilogue serially 159 times per layer.
// Here every workgroup keeps ITS slice of W_hh in registers for all steps; what crosses the chip per step is only
// h_{t-1} (512 KB at B = 256).
//
// Decomposition.  Workgroup (rg, cs) owns batch rows [rg*RB, rg*RB+RB) and hidden units [16 cs, 16 cs + 16), i.e. the
// 64 gate columns {i,f,g,o} x 16 units: complete cells.  Its 8 waves are (kw, rw, cw): k half, 32-row half of the
// 64-row sub-step, 8-unit half; each wave holds W[its 32 gate columns][its k half] as MFMA B operands in 128 VGPRs
// (v_mfma_f32_32x32x16_bf16; B operand of lane (n, kh) = 8 consecutive k of column n).  The RB rows are cut into
// sub-chains of 64 rows that are independent recurrences; the workgroup works on them round-robin, so the hand-off
// latency of one chain is hidden behind the arithmetic of the other.
//
// One sub-step (64 rows x 64 gate columns x Kp):
//   1. wait until all column slices of the row group have published h_{t-1} of this chain (one counter per chain,
//      one lane polls, bounded spin);
//   2. LDS-DMA the chain's h_{t-1} rows (64 x Kp bf16, <= 128 KB) into LDS: one global_load_lds_dwordx4 per 8 rows x
//      128 B, XOR-swizzled through the per-lane SOURCE address (position q of row r holds piece q ^ ((r>>1)&7)) so the
//      A-fragment ds_read_b128 are conflict-free; all requests are issued up front and consumed k chunk by k chunk
//      behind counted vmcnt waits;
//   3. 32 MFMAs per wave (A from LDS, B from registers), the two k halves are summed through LDS;
//   4. cell epilogue (2 cells per thread, c_t stays in LDS between steps): activated gates -> stash, c_t, h_t
//      (fp32) as plain stores, bf16 h_t tile through LDS as 8-byte WRITE-THROUGH (sc1) stores;
//   5. every wave drains its stores (vmcnt(0)), workgroup barrier, one lane adds 1 to the chain's counter (agent scope).
// Hand-off correctness (cdna guide, Guideline 16): payload stored sc1 and drained before the counter add; the consumer
// polls with an sc1 load, then a workgroup barrier, then loads.  h_t of every step has its own address (time-major
// h image), so no CU ever holds an older copy of a line it is about to read; the LDS-DMA loads carry sc1 as well.
// Every spin is bounded (1 s of wall clock): on a time-out the workgroup sets *err and exits.
#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int P_SR = 64;                          // batch rows per sub-step
constexpr int P_UN = 16;                          // hidden units per workgroup (64 gate columns)
constexpr int P_NT = 512;                         // 8 waves
constexpr int P_KCH = 16;                         // k chunks of 64 (Kp <= 1024)
constexpr int P_SLAB = P_KCH * P_SR * 128;        // h_{t-1} image: 128 KB
constexpr int P_RLD = 68;                         // row stride of the partial-sum image (floats)
constexpr int P_HSM = P_SLAB;                     // bf16 h_t tile [64][16]
constexpr int P_MAXNS = 4;
constexpr int P_CST = P_SLAB + P_SR * P_UN * 2;                // fp32 c_t of the workgroup's cells, per chain [64][16]
constexpr int P_LDS = P_CST + P_MAXNS * P_SR * P_UN * 4;
constexpr unsigned long long P_SPIN_TICKS = 100000000ull;      // 1 s of the 100-MHz wall clock

__device__ __forceinline__ unsigned short f2bf_rn(float x) {
    unsigned int u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ void glds16_sc1(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 16 /* sc1 */);
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
__global__ __launch_bounds__(P_NT) void lstm_seq_fwd_bf16_persist_kernel(SeqFwdBf16Args p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[P_LDS];
    __shared__ int s_flag;                 // poll result of the polling lane
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kw = wave >> 2, rw = (wave >> 1) & 1, cw = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, B = p.B;
    const int nC = (H + P_UN - 1) / P_UN;
    const int cs = blockIdx.x % nC, rg = blockIdx.x / nC;
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

    // ---- loader role: wave w moves rows 8w..8w+7 of the 64 of every k chunk
    const int lrow = wave * 8 + (lane >> 3);
    const int lpiece = (lane & 7) ^ ((lrow >> 1) & 7);
    // ---- A-fragment read address of lane (row li of this wave's 32, k half lh): piece 2s+lh of a chunk
    const int arow = rw * 32 + li;
    unsigned fa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        fa[s] = lbase + (unsigned)(cbeg * 8192 + arow * 128 + (((2 * s + lh) ^ ((arow >> 1) & 7)) * 16));

    // ---- epilogue role: 2 cells per thread, the same (row, unit) in every step
    const int erow = tid >> 4, eul = tid & 15;         // rows erow, erow + 32
    const int eunit = u0 + eul;
    const bool eu_ok = eunit < H;
    const int ecol = (eul >> 3) * 32 + (eul & 7);      // + g*8: column of gate g inside the workgroup's 64
    // c_t of this thread's cells lives in LDS between steps (only this thread touches its entries)
    float* cst = reinterpret_cast<float*>(smem + P_CST);
    for (int s = 0; s < p.NS; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int b = row0 + s * P_SR + erow + 32 * e;
            float c0 = 0.f;
            if (p.t0 > 0 && eu_ok && b < B) c0 = p.c_all[((int64_t)(p.t0 - 1) * B + b) * H + eunit];
            cst[(s * P_SR + erow + 32 * e) * P_UN + eul] = c0;
        }

    float* red = reinterpret_cast<float*>(smem);
    unsigned short* hsm = reinterpret_cast<unsigned short*>(smem + P_HSM);
    const int64_t H4 = 4 * (int64_t)H;

    for (int t = p.t0; t < p.t1; ++t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * P_SR;            // first batch row of this sub-step
            const int xrec = (blockIdx.x == p.stamp_block) ? (t - p.t0) * p.NS + s : -1;
            XSTAMP(p.stamps, xrec, 0);
            unsigned int* cnt = p.sync + (rg * P_MAXNS + s) * 32;

            if (t > p.t0) {          // h_{t-1} of this chain published by every column slice of the row group?
                if (tid == 0) {
                    const bool ok = spin_until(cnt, (unsigned int)(nC * (t - p.t0)));
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                P_BARRIER();
                if (s_flag == 0) return;
            }
            XSTAMP(p.stamps, xrec, 1);

            // epilogue operands requested now, consumed after the contraction
            float gxv[2][4];
            {
                const float* gsrc = (t < p.n_gx) ? p.gx_stash + (int64_t)t * B * H4 : nullptr;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int b = rbase + erow + 32 * e;
                    const bool ok = eu_ok && b < B;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float* q = !ok ? g_zero4
                                             : (gsrc ? gsrc + (int64_t)b * H4 + (int64_t)g * H + eunit
                                                     : p.bias + (int64_t)g * H + eunit);
                        gxv[e][g] = *q;
                    }
                }
            }

            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;

            if (t > 0) {
                // h_{t-1} rows of this chain -> LDS (chunk c at c*8 KB; this wave's piece at + wave*1 KB)
                const unsigned char* src = reinterpret_cast<const unsigned char*>(
                                               p.hb + ((int64_t)(t - 1) * B + rbase + lrow) * p.ldhb) + lpiece * 16;
                unsigned char* ldst = smem + wave * 1024;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (FULL || i < half) glds16_sc1(src + i * 128, ldst + i * 8192);
                    if (FULL || half + i < nch) glds16_sc1(src + (half + i) * 128, ldst + (half + i) * 8192);
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
                        P_DSR(a0, fa[0], (I) * 8192); P_DSR(a1, fa[1], (I) * 8192);                                \
                        P_DSR(a2, fa[2], (I) * 8192); P_DSR(a3, fa[3], (I) * 8192);                                \
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
                    rp[(rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * P_RLD + cw * 32 + li] = acc[r];
            }
            P_BARRIER();
            XSTAMP(p.stamps, xrec, 5);

            float gate[2][4], cv[2], hv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int rl = erow + 32 * e, b = rbase + rl;
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pre[g] = red[rl * P_RLD + ecol + g * 8] + red[(P_SR + rl) * P_RLD + ecol + g * 8] + gxv[e][g];
                gate[e][0] = 1.0f / (1.0f + expf(-pre[0]));
                gate[e][1] = 1.0f / (1.0f + expf(-pre[1]));
                gate[e][2] = tanhf(pre[2]);
                gate[e][3] = 1.0f / (1.0f + expf(-pre[3]));
                float* cp = cst + (s * P_SR + rl) * P_UN + eul;
                const bool ok = eu_ok && b < B;
                cv[e] = ok ? gate[e][1] * *cp + gate[e][0] * gate[e][2] : 0.f;
                hv[e] = gate[e][3] * tanhf(cv[e]);
                *cp = cv[e];
                hsm[rl * P_UN + eul] = ok ? f2bf_rn(hv[e]) : (unsigned short)0;
            }
            XSTAMP(p.stamps, xrec, 6);
            P_BARRIER();
            if (tid < 256) {   // bf16 h_t tile: 64 rows x 32 B, as 8-byte write-through stores (issued first: they are
                               // what the other workgroups wait for)
                const int rl = tid >> 2, part = tid & 3, b = rbase + rl;
                if (b < B) {
                    const unsigned long long v = *reinterpret_cast<const unsigned long long*>(hsm + rl * P_UN + part * 4);
                    unsigned short* dst = p.hb + ((int64_t)t * B + b) * p.ldhb + u0 + part * 4;
                    __hip_atomic_store((gu64*)dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int b = rbase + erow + 32 * e;
                if (eu_ok && b < B) {
                    const int64_t rowi = (int64_t)t * B + b;
                    p.c_all[rowi * H + eunit] = cv[e];
                    if (p.h_all) p.h_all[rowi * H + eunit] = hv[e];
                    float* st = p.gx_stash + rowi * H4 + eunit;
                    st[0] = gate[e][0];
                    st[(int64_t)H] = gate[e][1];
                    st[(int64_t)2 * H] = gate[e][2];
                    st[(int64_t)3 * H] = gate[e][3];
                }
            }
            XSTAMP(p.stamps, xrec, 7);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores
            XSTAMP(p.stamps, xrec, 8);
            P_BARRIER();
            if (tid == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            XSTAMP(p.stamps, xrec, 9);
        }
    }
}

int lstm_seq_fwd_bf16_persist_supported(int B, int H, int Kp) {
    return (B % P_SR == 0 && Kp % 64 == 0 && Kp >= H && Kp <= 64 * P_KCH) ? 1 : 0;
}

size_t lstm_persist_sync_bytes() { return (size_t)64 * P_MAXNS * 32 * sizeof(unsigned int); }   // <= 64 row groups

int lstm_seq_fwd_bf16_persist(hipStream_t stream, SeqFwdBf16Args a) {
    S2VT_REQUIRE(lstm_seq_fwd_bf16_persist_supported(a.B, a.H, a.Kp), "lstm_seq_fwd_bf16_persist: unsupported shape");
    S2VT_REQUIRE(a.t1 > a.t0 && a.t0 >= 0 && a.wb && a.hb && a.gx_stash && a.c_all && a.sync && a.err,
                 "lstm_seq_fwd_bf16_persist: bad arguments");
    S2VT_REQUIRE(a.ldhb % 8 == 0 && a.ldwb % 8 == 0 && a.ldhb >= a.Kp && a.ldwb >= a.Kp &&
                     (reinterpret_cast<uintptr_t>(a.hb) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.wb) & 15) == 0,
                 "lstm_seq_fwd_bf16_persist: operands must be 16-B aligned bf16 rows zero-padded to Kp");
    a.RB = (a.B % 128 == 0) ? 128 : 64;
    a.NS = a.RB / P_SR;
    const int R = a.B / a.RB, nC = cdiv(a.H, P_UN);
    S2VT_REQUIRE(R <= 64, "lstm_seq_fwd_bf16_persist: batch too large for the counter block");
    S2VT_REQUIRE(R * nC <= 256, "lstm_seq_fwd_bf16_persist: %d workgroups would not be co-resident", R * nC);
    S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (a.Kp == 64 * P_KCH)
        hipLaunchKernelGGL((lstm_seq_fwd_bf16_persist_kernel<true>), dim3(R * nC), dim3(P_NT), 0, stream, a);
    else
        hipLaunchKernelGGL((lstm_seq_fwd_bf16_persist_kernel<false>), dim3(R * nC), dim3(P_NT), 0, stream, a);
    S2VT_LAUNCH_CHECK("lstm_seq_fwd_bf16_persist_kernel");
    return 0;
}

}  // namespace s2vt
