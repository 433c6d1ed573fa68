// Decode-step out_linear + argmax (S2VTModel.py:95-96, 105-106) on the bf16 matrix cores, fp32-equivalent:
//   packed[b] = max over n of (ordered(h[b]·W_o[n] + b_o[n]) << 32 | ~n)          (first-max tie rule: lowest index wins)
// with both operands as three bf16 planes and the six plane products >= 2^-16 (the arithmetic of gemm_x3.hip, whose
// blocked plane layout the operands use: split.hip).  W_o is constant over the 79 decode steps of a call, so its planes
// are written once per call; h_t (128 x 1000) is re-split every step by the split kernel.
//
// Why (round 2 profile of the fp32-MFMA kernel this replaces on the plane path, lstm.hip::logits_argmax_kernel): 46 us per
// launch = a third of a greedy decode; its floor on the exact-fp32 MFMA is 19.5 us and every one of its 1500 32x32 tiles
// re-read 256 KB of fp32 operands.  Here the six bf16 plane products cost 7.4 us of matrix time chip-wide and the kernel is
// bound by what a compute unit takes in: one workgroup per 64 vocabulary rows (188 workgroups at V = 12000) streams its
// W_o record (384 KB) once and the h planes of <= 128 batch rows (768 KB).
//
// Tile: 64 vocabulary rows (the MFMA M side) x 64*NB batch rows (N side), k32 stages of (1 + NB) x 2 records of 6 KB in a
// 4-slot ring (144 KB at NB = 2: three stages in flight, ~108 KB per compute unit), filled by LDS-DMA (one wave
// instruction = one 1-KB piece, spread over the 4 waves) behind counted vmcnt waits and one barrier per stage.  4 waves as
// 2 (vocabulary) x 2 (batch): wave tile 32 x 32*NB.  Epilogue: every lane holds 16 vocabulary rows of ONE batch column per
// MFMA tile: bias, running max in registers, one cross-half exchange, one 64-bit atomicMax per (wave, batch column).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int AX_REC = 6144;                    // bytes of one (64-row block, k16 chunk) record
constexpr int AX_SEG = 2 * AX_REC;              // one row block's bytes of a k32 stage (two consecutive records)
constexpr int AX_NS = 4;                        // ring depth (stages)

__device__ __forceinline__ void ax_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ uint32_t ax_ordered_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// experiment switches (S2VT_AX_DBG; tools/bench_argmax_x3_stamps.py): compiled out of the product
#ifdef S2VT_EXPERIMENT_STAMPS
#define AX_DBG(P) ((P).dbg)
#else
#define AX_DBG(P) 0
#endif

template <int NB>
__global__ __launch_bounds__(256) void logits_argmax_x3_kernel(ArgmaxX3Args p) {
    constexpr int NSEG = 1 + NB;                       // row-block segments of a stage: W_o block, then the h blocks
    constexpr int STAGE = NSEG * AX_SEG;
    constexpr int NPIECE = NSEG * 12;                  // 1-KB pieces per stage
    constexpr int RPW = NPIECE / 4;                    // requests per wave and stage (9 at NB = 2, 6 at NB = 1)
    constexpr int RQA = RPW / 2, RQB = RPW - RQA;      // ... issued in the second half of one stage / the first half of the next
    static_assert(NPIECE % 4 == 0 && AX_NS * STAGE <= 160 * 1024 && RQB <= 6, "ring geometry");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[AX_NS * STAGE];
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nvb = (p.V + 63) >> 6;
    const int vb_all = (int)blockIdx.x + p.v_off;      // row block of the launch: vocabulary blocks first, then the second image's
    const bool zrole = vb_all >= nvb;                  // (workgroup-uniform)
    const int vb = zrole ? vb_all - nvb : vb_all;      // row block (64 rows) of this workgroup's weight image
    const int mlim = zrole ? p.M2 : p.V;
    const int bt = blockIdx.y;                         // batch tile (64 * NB rows)
    const int nst = (AX_DBG(p) & 4) ? 4 : (p.K >> 5);      // k32 stages (K % 64 == 0: an even number >= 2)
    const int xrec = p.stamps ? (int)blockIdx.x : -1;
    XSTAMP(p.stamps, xrec, 0);
#ifdef S2VT_EXPERIMENT_STAMPS
    if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x * XSTAMP_SLOTS + 8] = __builtin_amdgcn_s_memtime();
#endif

    // bias of this lane's 16 vocabulary rows (rows 8 q + 4 lh + (0..3), q = 0..3): requested FIRST, so that the loads are
    // older than every ring request and never show up in a counted wait
    const int m_base = vb * 64 + wm * 32 + 4 * lh;
    f32x4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m_base + 8 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = *((!zrole && p.bias && m + e < p.V) ? p.bias + m + e : g_zero4);
    }

    // ---- loader role: piece q = wave + 4 j of a stage lies in segment q / 12 at byte (q % 12) * 1024
    const unsigned char* gsrc[RPW];
    int loff[RPW];
    {
        const int nhb = (p.B + 63) >> 6;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int q = wave + 4 * j, seg = q / 12, off = (q % 12) * 1024;
            const unsigned short* base;
            if (seg == 0) {
                base = zrole ? p.W2 + (int64_t)vb * 64 * p.ldw2 : p.W + (int64_t)vb * 64 * p.ldw;
            } else {
                int hb = bt * NB + seg - 1;            // a batch block past the last one is clamped onto it (never stored)
                hb = hb < nhb ? hb : nhb - 1;
                base = p.Hp + (int64_t)hb * 64 * p.ldh;
            }
            gsrc[j] = reinterpret_cast<const unsigned char*>(base) + off + lane * 16;
            loff[j] = seg * AX_SEG + off;
        }
    }
    // request j of stage s -> ring slot s % AX_NS (nothing for a stage past the last one)
#define AX_REQ(S, J)                                                                                         \
    if ((steady_ || (S) < nst) && !(AX_DBG(p) & 1)) ax_glds16(gsrc[J] + (int64_t)(S) * AX_SEG, smem + ((S) % AX_NS) * STAGE + loff[J]);

    f32x16 acc[NB];
#pragma unroll
    for (int ni = 0; ni < NB; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

    // fragment byte offsets inside a stage (k16 record kk at + kk * AX_REC; plane pl, k half lh at + (pl*2 + lh) * 1024)
    const int a_off = lh * 1024 + (wm * 32 + li) * 16;
    int b_off[NB];
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
        const int rec = (NB == 2) ? 1 + wn : 1, row = (NB == 2) ? ni * 32 + li : wn * 32 + li;
        b_off[ni] = rec * AX_SEG + lh * 1024 + row * 16;
    }

    // Two fragment register sets: the reads of stage s+1 are issued in the MIDDLE of stage s, behind the barrier that says
    // "stage s+1 has landed and everybody has read stage s", and the second half of stage s's MFMAs covers their round
    // trip - with ONE wave per SIMD nothing else would (first version: reads at the top of the stage, 0.63 us per stage for
    // 0.36 us of MFMA; in-kernel stamps, tools/bench_argmax_x3_stamps.py).  The 9 LDS-DMA requests of a wave and stage are
    // slipped in between the MFMA groups (each costs the wave ~40 cycles of issue).
    bf16x8 fa[2][2][3], fb[2][2][3][NB];               // [set][k16 record][plane]([batch tile])
#define AX_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF));
#define AX_LD(SET, KK, PA, PB)                                                  \
    AX_RD(fa[SET][KK][PA], la, (KK) * AX_REC + (PA) * 2048)                     \
    AX_RD(fb[SET][KK][PB][0], lb[0], (KK) * AX_REC + (PB) * 2048)               \
    if constexpr (NB == 2) { AX_RD(fb[SET][KK][PB][NB - 1], lb[NB - 1], (KK) * AX_REC + (PB) * 2048) }
    // all 6 (NB = 1: 4) x 3 fragment reads of stage S into set SET, in the order the products consume them
#define AX_READS(SET, S)                                                                                     \
    {                                                                                                        \
        const unsigned la = lbase + (unsigned)(((S) % AX_NS) * STAGE + a_off);                               \
        unsigned lb[NB];                                                                                     \
        _Pragma("unroll") for (int ni = 0; ni < NB; ++ni) lb[ni] = lbase + (unsigned)(((S) % AX_NS) * STAGE + b_off[ni]); \
        AX_LD(SET, 0, 1, 1) AX_LD(SET, 0, 0, 2) AX_LD(SET, 0, 2, 0) AX_LD(SET, 1, 1, 1) AX_LD(SET, 1, 0, 2) AX_LD(SET, 1, 2, 0) \
    }
    // wait until at most N2 (NB = 2) / N1 (NB = 1) LDS reads are outstanding; the operands tie the products to the wait
#define AX_WAIT(N2, N1, SET, KK, PA, PB)                                                                                      \
    if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(" #N2 ")" : "+v"(fa[SET][KK][PA]), "+v"(fb[SET][KK][PB][0]), "+v"(fb[SET][KK][PB][NB - 1])); \
    else asm volatile("s_waitcnt lgkmcnt(" #N1 ")" : "+v"(fa[SET][KK][PA]), "+v"(fb[SET][KK][PB][0]));
#define AX_PROD(SET, KK, PA, PB)                                                                             \
    _Pragma("unroll") for (int ni = 0; ni < NB; ++ni)                                                        \
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][KK][PA], fb[SET][KK][PB][ni], acc[ni], 0, 0, 0);
    // One stage on fragment set CUR (six plane products per k16 record, smallest terms first: gemm_x3.hip).
    //   first half : record 0's products; requests RQA.. of stage s+3 in between
    //   middle     : all reads of this stage are in (lgkmcnt(0)); stage s+1 landed for this wave (counted vmcnt: the
    //                requests of the stages s+2, s+3 that exist may stay in flight); barrier; reads of stage s+1 -> set NXT
    //   second half: record 1's products; requests 0..RQA-1 of stage s+4 in between (its slot = this stage's, free now)
#define AX_STAGE(CUR, NXT, S, STEADY)                                                                              \
    {                                                                                                        \
        const int s_ = (S);                                                                                  \
        constexpr bool steady_ = (STEADY);      /* s + 4 < nst: every request and the next stage exist */        \
        AX_WAIT(15, 10, CUR, 0, 1, 1) AX_PROD(CUR, 0, 1, 1) if (RQA + 0 < RPW) { AX_REQ(s_ + 3, (RQA + 0 < RPW ? RQA + 0 : 0)) } \
        AX_WAIT(12, 8, CUR, 0, 0, 2)  AX_PROD(CUR, 0, 0, 2) if (RQA + 1 < RPW) { AX_REQ(s_ + 3, (RQA + 1 < RPW ? RQA + 1 : 0)) } \
        AX_WAIT(9, 6, CUR, 0, 2, 0)   AX_PROD(CUR, 0, 2, 0) if (RQA + 2 < RPW) { AX_REQ(s_ + 3, (RQA + 2 < RPW ? RQA + 2 : 0)) } \
        AX_PROD(CUR, 0, 0, 1) if (RQA + 3 < RPW) { AX_REQ(s_ + 3, (RQA + 3 < RPW ? RQA + 3 : 0)) }            \
        AX_PROD(CUR, 0, 1, 0) if (RQA + 4 < RPW) { AX_REQ(s_ + 3, (RQA + 4 < RPW ? RQA + 4 : 0)) }            \
        AX_PROD(CUR, 0, 0, 0)                                                                                \
        AX_WAIT(6, 4, CUR, 1, 1, 1) AX_WAIT(3, 2, CUR, 1, 0, 2) AX_WAIT(0, 0, CUR, 1, 2, 0)                   \
        const bool nx_ = steady_ || s_ + 1 < nst;                                                             \
        if (nx_) {                                                                                           \
            if (steady_ || s_ + 3 < nst) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * RPW) : "memory");    \
            else if (s_ + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RPW) : "memory");   \
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                \
        }                                                                                                    \
        const unsigned la = lbase + (unsigned)(((s_ + 1) % AX_NS) * STAGE + a_off);                          \
        unsigned lb[NB];                                                                                     \
        _Pragma("unroll") for (int ni = 0; ni < NB; ++ni) lb[ni] = lbase + (unsigned)(((s_ + 1) % AX_NS) * STAGE + b_off[ni]); \
        /* the reads of stage s+1 go out between the MFMA groups (a burst of 18 ds_read_b128 in front of them left the   */ \
        /* matrix pipe idle for ~300 cycles per stage: the wave issues in order)                                           */ \
        if (nx_) { AX_LD(NXT, 0, 1, 1) } AX_PROD(CUR, 1, 1, 1) if (0 < RQA) { AX_REQ(s_ + 4, 0) }              \
        if (nx_) { AX_LD(NXT, 0, 0, 2) } AX_PROD(CUR, 1, 0, 2) if (1 < RQA) { AX_REQ(s_ + 4, (1 < RQA ? 1 : 0)) } \
        if (nx_) { AX_LD(NXT, 0, 2, 0) } AX_PROD(CUR, 1, 2, 0) if (2 < RQA) { AX_REQ(s_ + 4, (2 < RQA ? 2 : 0)) } \
        if (nx_) { AX_LD(NXT, 1, 1, 1) } AX_PROD(CUR, 1, 0, 1) if (3 < RQA) { AX_REQ(s_ + 4, (3 < RQA ? 3 : 0)) } \
        if (nx_) { AX_LD(NXT, 1, 0, 2) } AX_PROD(CUR, 1, 1, 0)                                                \
        if (nx_) { AX_LD(NXT, 1, 2, 0) } AX_PROD(CUR, 1, 0, 0)                                                \
    }

    // prologue: stages 0..2 and the second-half share of stage 3 requested; stage 0 landed; its fragments on their way
    constexpr bool steady_ = false;
#pragma unroll
    for (int j = 0; j < RPW; ++j) { AX_REQ(0, j) }
#pragma unroll
    for (int j = 0; j < RPW; ++j) { AX_REQ(1, j) }
#pragma unroll
    for (int j = 0; j < RPW; ++j) { AX_REQ(2, j) }
#pragma unroll
    for (int j = 0; j < RQA; ++j) { AX_REQ(3, j) }
    XSTAMP(p.stamps, xrec, 1);
    if (nst > 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * RPW + RQA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    AX_READS(0, 0)
    XSTAMP(p.stamps, xrec, 2);
    int s = 0;
    for (; s + 5 < nst; s += 2) {          // steady state: s + 1 + 4 < nst
        AX_STAGE(0, 1, s, true)
        AX_STAGE(1, 0, s + 1, true)
        if (s == 14) XSTAMP(p.stamps, xrec, 3);
    }
    for (; s < nst; s += 2) {              // the last stages: requests / barrier / reads only for stages that exist
        AX_STAGE(0, 1, s, false)
        AX_STAGE(1, 0, s + 1, false)
    }
    XSTAMP(p.stamps, xrec, 4);
    XSTAMP(p.stamps, xrec, 5);
#ifdef S2VT_EXPERIMENT_STAMPS
    if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x * XSTAMP_SLOTS + 9] = __builtin_amdgcn_s_memtime();
#endif
#undef AX_STAGE
#undef AX_PROD
#undef AX_WAIT
#undef AX_READS
#undef AX_LD
#undef AX_RD
#undef AX_REQ

    // ---- epilogue: lane (li, lh) of tile ni holds logits of batch column b = ... + li for vocabulary rows m_base + 8 q + e.
    // Per column: running max in registers, the two k-row halves (lh) by one exchange, the two vocabulary halves of the
    // workgroup (wm) through LDS, then ONE atomicMax per column and workgroup (188 per address and launch at V = 12000;
    // one per wave doubled that and cost 3.5 us of a 35-us launch)
    if (zrole) {
        // second role: the products leave as they are.  Lane (li, lh) holds rows m_base + 8 q + (0..3) of batch column li:
        // one 16-byte store per q, the two lh halves of a column side by side (32 contiguous bytes per batch row)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int b = bt * 64 * NB + ((NB == 2) ? wn * 64 + ni * 32 : wn * 32) + li;
            if (b >= p.B) continue;
            float* zr = p.z + (int64_t)b * p.ldz;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m_base + 8 * q;
                if (m + 3 < mlim) {
                    *reinterpret_cast<f32x4*>(zr + m) = f32x4{acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < mlim) zr[m + e] = acc[ni][4 * q + e];
                }
            }
        }
        return;
    }
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);       // [2][64 * NB]
    __syncthreads();                                                               // the ring is dead: every wave is past its last reads
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
        const int col = ((NB == 2) ? wn * 64 + ni * 32 : wn * 32) + li;
        unsigned long long best = 0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            if (m < p.V) {
                const float v = acc[ni][r] + bv[r >> 2][r & 3];
                const unsigned long long key =
                    ((unsigned long long)ax_ordered_bits(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)m);
                best = key > best ? key : best;
            }
        }
        const unsigned long long o = __shfl_xor(best, 32);
        best = o > best ? o : best;
        if (lh == 0) keys[wm * (64 * NB) + col] = best;
    }
    __syncthreads();
    if (tid < 64 * NB) {
        const unsigned long long k0 = keys[tid], k1 = keys[64 * NB + tid];
        const unsigned long long best = k0 > k1 ? k0 : k1;
        const int b = bt * 64 * NB + tid;
        if (b < p.B && best && !(AX_DBG(p) & 2)) atomicMax(&p.packed[b], best);
    }
    XSTAMP(p.stamps, xrec, 6);
}

int logits_argmax_x3(hipStream_t stream, const ArgmaxX3Args& a) {
    S2VT_REQUIRE(a.B > 0 && a.V > 0 && a.K > 0 && a.K % 64 == 0 && a.W && a.Hp && a.packed, "logits_argmax_x3: bad arguments");
    S2VT_REQUIRE(a.ldw >= 3 * (int64_t)a.K && a.ldh >= 3 * (int64_t)a.K && a.ldw % 8 == 0 && a.ldh % 8 == 0 &&
                     (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.Hp) & 15) == 0,
                 "logits_argmax_x3: operands must be blocked 3-plane images (split.hip) with k padded to K");
    S2VT_REQUIRE(a.M2 >= 0 && (a.M2 == 0 || (a.W2 && a.z && a.ldw2 >= 3 * (int64_t)a.K && a.ldw2 % 8 == 0 && a.ldz >= a.M2 && a.ldz % 4 == 0 &&
                                              (reinterpret_cast<uintptr_t>(a.W2) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.z) & 15) == 0)),
                 "logits_argmax_x3: the second image must be a blocked 3-plane image of the same K, z 16-byte aligned rows");
    S2VT_REQUIRE(a.v_off == 0 || (a.v_off == cdiv(a.V, 64) && a.M2 > 0), "logits_argmax_x3: v_off is 0 or every vocabulary block");
    const int vblocks = cdiv(a.V, 64) - a.v_off + cdiv(a.M2, 64);
    ArgmaxX3Args b = a;
    if (a.B > 64) {
        hipLaunchKernelGGL(logits_argmax_x3_kernel<2>, dim3(vblocks, cdiv(a.B, 128)), dim3(256), 0, stream, b);
    } else {
        hipLaunchKernelGGL(logits_argmax_x3_kernel<1>, dim3(vblocks, 1), dim3(256), 0, stream, b);
    }
    S2VT_LAUNCH_CHECK("logits_argmax_x3_kernel");
    return 0;
}

}  // namespace s2vt
