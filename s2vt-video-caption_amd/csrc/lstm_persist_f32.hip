// Persistent fp32 LSTM recurrence for gfx950 (BASELINE config 2: B = 64, fp32): the fp32 counterpart of lstm_persist.hip.
// ONE launch runs a block of timesteps of one layer - or of both layers side by side - with every workgroup's slice of
// W_hh (forward) / W_hh^T (BPTT) resident in registers, exact-fp32 products on v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32, and the same cross-workgroup hand-off protocol (write-through payload stored by one wave,
// drained, agent-scope counter add; consumer polls, barrier, loads; every spin bounded; see lstm_persist.hip).
//
// Forward.  Workgroup (rg, cs) = 32-row chains x 8 hidden units (32 gate columns {i,f,g,o} x 8: complete cells).  Its 4
// waves split k in quarters of 256; wave kw keeps W[32 gate columns][its 256 k] in 128 VGPRs (B operand of lane
// (n, kh) for the MFMA of (chunk j, octet o, r) = W[n][k0 + 32 j + 8 o + 4 kh + r]) and streams ITS quarter of
// h_{t-1} (32 rows x 256 k fp32) through a private 4-slot LDS ring of 4-KB chunks (LDS-DMA, XOR-swizzled like the bf16
// image), so the contraction needs no workgroup barrier; 128 MFMAs per wave and sub-step (the sub-step is MFMA-bound:
// 8192 cycles, against 2 us of transfer).  70 KB of LDS and 4 waves per workgroup: two per CU - the other chain or,
// in a fused launch, the other layer.  h_t (fp32) is the hand-off payload itself: the 32 x 8 tile goes through LDS and
// leaves as ONE 16-byte write-through store instruction of wave 0.
//
// BPTT.  Workgroup = 32-row chains x 16 hidden units, 8 waves (one workgroup per CU): wave kw keeps
// W_hh^T[16 units][its 512 k of 4H] in 128 VGPRs and streams its eighth of dG_{t+1} (32 rows x 512 k fp32) through its
// own ring; 256 v_mfma_f32_16x16x4_f32 per wave and sub-step on two independent accumulators (two 16-row tiles).  The
// fp32 dG_t tile (32 rows x 4 gates x 64 B) is the hand-off payload (eight write-through store instructions of wave 0).
#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int F_SR = 32;                          // batch rows per sub-step (= per chain)
constexpr int F_UN = 8;                           // forward: hidden units per workgroup (32 gate columns)
constexpr int F_NT = 256;
constexpr int F_RING = 4 * 4096;                  // per-wave ring: 4 slots of one 32-row x 32-k fp32 chunk
constexpr int F_PLD = 36;                         // row stride of a partial tile (floats)
constexpr int F_HSM = 4 * F_RING;                 // fp32 h_t tile [32][8]
constexpr int F_MAXNS = 4;
constexpr int F_CST = F_HSM + F_SR * F_UN * 4;    // c_t of the workgroup's cells, per chain [32][8]
constexpr int F_LDS = F_CST + F_MAXNS * F_SR * F_UN * 4;
constexpr int F_MAX_WG = 512;                     // design point (2 per CU); capped by coresident_capacity() at launch
constexpr unsigned long long F_SPIN_TICKS = 100000000ull;

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ void glds16f_sc1(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 16 /* sc1 */);
}
__device__ __forceinline__ bool spin_until_f(const unsigned int* cnt, unsigned int target) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned int v = __hip_atomic_load((gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > F_SPIN_TICKS) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}
// L1-bypassing scalar load (global_load_dword sc1): for words of a buffer that another workgroup's hand-off will overwrite
__device__ __forceinline__ float ld_sc1(const float* q) {
    return __uint_as_float(__hip_atomic_load((gu32*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
#define F_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define F_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ------------------------------------------------------------------------------------------------------ forward
__device__ __forceinline__ void seq_fwd_f32_body(const SeqFwdF32Args& p, const int bid, unsigned char* smem, int& s_flag) {
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, B = p.B;
    const int nC = H / F_UN;
    const int cs = bid % nC, rg = bid / nC;
    const int u0 = cs * F_UN, row0 = rg * p.RB;
    const int k0 = wave * 256;                          // this wave's k quarter [k0, k0 + 256)

    // ---- W_hh slice of this wave: gate column n = g*8 + uu <-> W_hh row g*H + u0 + uu
    f32x4 wreg[32];
    {
        const int g = li >> 3, unit = u0 + (li & 7);
        const float* wrow = p.w_hh + ((int64_t)g * H + unit) * p.ldw;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int k = k0 + 32 * j + 8 * o + 4 * lh;
                wreg[j * 4 + o] = *reinterpret_cast<const f32x4*>((k + 3 < H) ? wrow + k : g_zero4);
            }
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wreg[i]));
    }

    // ---- loader role (each wave for itself): piece i of a chunk = rows 8i..8i+7; lane -> (row, swizzled 16-B piece)
    unsigned voff[4];
    int lk[4];                                          // k offset (floats) of this lane's piece inside the chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = ((lane & 7) ^ (lane >> 4)) ^ (4 * (i & 1));
        lk[i] = 4 * q;
        voff[i] = (unsigned)((8 * i + (lane >> 3)) * (H * 4) + q * 16);
    }
    unsigned char* ring = smem + wave * F_RING;
    unsigned fa[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) fa[o] = lbase + (unsigned)(wave * F_RING + li * 128 + (((2 * o + lh) ^ ((li >> 1) & 7)) * 16));

    // ---- epilogue role: one cell per thread
    const int erow = tid >> 3, eu = tid & 7, eunit = u0 + eu;
    float* cst = reinterpret_cast<float*>(smem + F_CST);
    for (int s = 0; s < p.NS; ++s) {
        const int b = row0 + s * F_SR + erow;
        cst[(s * F_SR + erow) * F_UN + eu] = (p.t0 > 0 && b < B) ? p.c_all[((int64_t)(p.t0 - 1) * B + b) * H + eunit] : 0.f;
    }
    float* hsm = reinterpret_cast<float*>(smem + F_HSM);
    const int64_t H4 = 4 * (int64_t)H;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(g_zero4);

    for (int t = p.t0; t < p.t1; ++t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * F_SR;
            unsigned int* cnt = p.sync + (rg * F_MAXNS + s) * 32;
            const int xrec = (bid == p.stamp_block) ? (t - p.t0) * p.NS + s : -1;
            XSTAMP(p.stamps, xrec, 0);
            if (t > p.t0) {
                if (tid == 0) {
                    const bool ok = spin_until_f(cnt, (unsigned int)(nC * (t - p.t0)));
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                F_BARRIER();
                if (s_flag == 0) return;
            }
            XSTAMP(p.stamps, xrec, 1);
            const int eb = rbase + erow;
            const bool rok = eb < B;
            const int64_t rowi = (int64_t)t * B + eb;
            float gxv[4];
            {
                const float* gsrc = (t < p.n_gx) ? p.gx_stash + rowi * H4 : p.bias;
#pragma unroll
                for (int g = 0; g < 4; ++g) gxv[g] = *(rok ? gsrc + (int64_t)g * H + eunit : g_zero4);
            }

            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (t > 0) {
                const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.h_all + ((int64_t)(t - 1) * B + rbase) * H);
                // every wave issues the same number of requests (counted waits); pieces past the row end read zeros
#define F_ISSUE(J)                                                                                              \
                {                                                                                               \
                    const int kc = k0 + 32 * (J);                                                               \
                    const unsigned char* sb = abase + kc * 4;                                                   \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
                        glds16f_sc1((kc + lk[i] + 3 < H) ? sb + voff[i] : zsrc, ring + ((J) % 4) * 4096 + i * 1024); \
                }
#define F_STEP(J, VM)                                                                                           \
                if ((J) + 3 < 8) F_ISSUE((J) + 3)                                                               \
                asm volatile("s_waitcnt vmcnt(" #VM ")" ::: "memory");                                          \
                {                                                                                               \
                    f32x4 a0, a1, a2, a3;                                                                       \
                    F_DSR(a0, fa[0], ((J) % 4) * 4096); F_DSR(a1, fa[1], ((J) % 4) * 4096);                     \
                    F_DSR(a2, fa[2], ((J) % 4) * 4096); F_DSR(a3, fa[3], ((J) % 4) * 4096);                     \
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a0));                                            \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], wreg[(J) * 4 + 0][r], acc, 0, 0, 0);  \
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a1));                                            \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], wreg[(J) * 4 + 1][r], acc, 0, 0, 0);  \
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a2));                                            \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], wreg[(J) * 4 + 2][r], acc, 0, 0, 0);  \
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a3));                                            \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[r], wreg[(J) * 4 + 3][r], acc, 0, 0, 0);  \
                }
                F_ISSUE(0) F_ISSUE(1) F_ISSUE(2)
                XSTAMP(p.stamps, xrec, 2);
                F_STEP(0, 12)
                XSTAMP(p.stamps, xrec, 3);
                F_STEP(1, 12) F_STEP(2, 12) F_STEP(3, 12) F_STEP(4, 12) F_STEP(5, 8) F_STEP(6, 4) F_STEP(7, 0)
#undef F_STEP
#undef F_ISSUE
            }
            XSTAMP(p.stamps, xrec, 4);
            {   // partial tile of this wave -> its own (idle) ring
                float* rp = reinterpret_cast<float*>(ring);
#pragma unroll
                for (int r = 0; r < 16; ++r) rp[((r & 3) + 8 * (r >> 2) + 4 * lh) * F_PLD + li] = acc[r];
            }
            F_BARRIER();
            XSTAMP(p.stamps, xrec, 5);

            float gate[4], cv, hv;
            {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v = gxv[g];
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += reinterpret_cast<const float*>(smem + w * F_RING)[erow * F_PLD + g * 8 + eu];
                    pre[g] = v;
                }
                gate[0] = sigmoid_f(pre[0]);
                gate[1] = sigmoid_f(pre[1]);
                gate[2] = tanh_f(pre[2]);
                gate[3] = sigmoid_f(pre[3]);
                float* cp = cst + (s * F_SR + erow) * F_UN + eu;
                cv = rok ? gate[1] * *cp + gate[0] * gate[2] : 0.f;
                hv = gate[3] * tanh_f(cv);
                *cp = cv;
                hsm[erow * F_UN + eu] = rok ? hv : 0.f;
            }
            XSTAMP(p.stamps, xrec, 6);
            F_BARRIER();
            if (wave == 0) {   // h_t tile: 32 rows x 32 B = ONE 16-byte write-through store instruction
                const int rl = lane >> 1, part = lane & 1;
                const u32x4 v = *reinterpret_cast<const u32x4*>(hsm + rl * F_UN + part * 4);
                float* dst = p.h_all + ((int64_t)t * B + rbase + rl) * H + u0 + part * 4;
                if (rbase + rl < B) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
            if (rok) {
                p.c_all[rowi * H + eunit] = cv;
                float* st = p.gx_stash + rowi * H4 + eunit;
                st[0] = gate[0];
                st[(int64_t)H] = gate[1];
                st[(int64_t)2 * H] = gate[2];
                st[(int64_t)3 * H] = gate[3];
            }
            XSTAMP(p.stamps, xrec, 7);
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                XSTAMP(p.stamps, xrec, 8);
                if (lane == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            XSTAMP(p.stamps, xrec, 9);
            F_BARRIER();
        }
    }
}

__global__ __launch_bounds__(F_NT, 2) void lstm_seq_fwd_f32_persist_kernel(SeqFwdF32Args pa, SeqFwdF32Args pb, int na) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[F_LDS];
    __shared__ int s_flag;
    if ((int)blockIdx.x < na) seq_fwd_f32_body(pa, blockIdx.x, smem, s_flag);
    else seq_fwd_f32_body(pb, blockIdx.x - na, smem, s_flag);
}

static int fwd_f32_capacity() {
    const int cap = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_fwd_f32_persist_kernel), F_NT);
    return cap < F_MAX_WG ? cap : F_MAX_WG;
}
// returns the number of 32-row chains per workgroup (0: unsupported, or two layers do not fit the device's resident capacity)
int lstm_seq_fwd_f32_persist_supported(int B, int H) {
    if (!(B > 0 && B % F_SR == 0 && H % 8 == 0 && H >= 8 && H <= 1024)) return 0;
    const int cap = fwd_f32_capacity();
    const int nC = H / F_UN;
    int R = B / F_SR, ns = 1;
    while (R * nC > cap / 2 && ns < F_MAXNS && R % 2 == 0) { R /= 2; ns *= 2; }
    return (R * nC <= cap / 2 && R <= 64) ? ns : 0;
}

static int prep_f(SeqFwdF32Args& a) {
    const int ns = lstm_seq_fwd_f32_persist_supported(a.B, a.H);
    S2VT_REQUIRE(ns > 0, "lstm_seq_fwd_f32_persist: unsupported shape (B %% 32, H %% 8, H <= 1024) or it does not fit the device's resident capacity");
    S2VT_REQUIRE(a.t1 > a.t0 && a.t0 >= 0 && a.w_hh && a.h_all && a.gx_stash && a.c_all && a.sync && a.err && (a.bias || a.n_gx >= a.t1),
                 "lstm_seq_fwd_f32_persist: bad arguments");
    S2VT_REQUIRE(a.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(a.w_hh) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.h_all) & 15) == 0,
                 "lstm_seq_fwd_f32_persist: W_hh and h rows must be 16-byte aligned");
    a.NS = ns;
    a.RB = ns * F_SR;
    return 0;
}

int lstm_seq_fwd_f32_persist2(hipStream_t stream, SeqFwdF32Args a, const SeqFwdF32Args* b) {
    int rc;
    if ((rc = prep_f(a))) return rc;
    SeqFwdF32Args bb = b ? *b : a;
    if (b) {
        if ((rc = prep_f(bb))) return rc;
        S2VT_REQUIRE(bb.sync != a.sync, "lstm_seq_fwd_f32_persist: paired layers need their own counters");
    }
    const int na = (a.B / a.RB) * (a.H / F_UN), nb = b ? (bb.B / bb.RB) * (bb.H / F_UN) : 0;
    S2VT_REQUIRE(na + nb <= fwd_f32_capacity(), "lstm_seq_fwd_f32_persist: %d workgroups would not be co-resident (device capacity %d)",
                 na + nb, fwd_f32_capacity());
    S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (b) S2VT_HIP(hipMemsetAsync(bb.sync, 0, lstm_persist_sync_bytes(), stream));
    hipLaunchKernelGGL(lstm_seq_fwd_f32_persist_kernel, dim3(na + nb), dim3(F_NT), 0, stream, a, bb, na);
    S2VT_LAUNCH_CHECK("lstm_seq_fwd_f32_persist_kernel");
    return 0;
}

// --------------------------------------------------------------------------------------------------------- BPTT
constexpr int G_UN = 16;                          // hidden units per workgroup
constexpr int G_NT = 512;                         // 8 waves
constexpr int G_RING = 4 * 4096;
constexpr int G_PLD = 20;                         // row stride of a partial tile (floats)
constexpr int G_DGSM = 8 * G_RING;                // fp32 dG_t tile [32][64]
constexpr int G_DCST = G_DGSM + F_SR * 64 * 4;
constexpr int G_LDS = G_DCST + F_MAXNS * F_SR * G_UN * 4;
constexpr int G_MAX_WG = 256;                     // design point: one workgroup per CU; capped by coresident_capacity()

__device__ __forceinline__ void seq_bwd_f32_body(const SeqBwdF32Args& p, const int bid, unsigned char* smem, int& s_flag) {
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, lq = lane >> 4;
    const int H = p.H, B = p.B, K = 4 * H;
    const int nC = (H + G_UN - 1) / G_UN;
    const int cs = bid % nC, rg = bid / nC;
    const int u0 = cs * G_UN, row0 = rg * p.RB;
    const int k0 = wave * 512;                          // this wave's k eighth [k0, k0 + 512) of 4H (<= 4096)

    // ---- W_hh^T slice: B operand of the MFMA (chunk j, group o, r) = W_hh[k0 + 32 j + 16 o + 4 kq + r][unit n]
    //      taken from the transposed copy w_hh_t [H][4H] (k contiguous)
    f32x4 wreg[32];
    {
        const int unit = u0 + lm;
        const float* wrow = p.w_hh_t + (int64_t)unit * p.ldwt;
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int k = k0 + 32 * j + 16 * o + 4 * lq;
                wreg[j * 2 + o] = *reinterpret_cast<const f32x4*>((unit < H && k + 3 < K) ? wrow + k : g_zero4);
            }
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wreg[i]));
    }

    unsigned voff[4];
    int lk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = ((lane & 7) ^ (lane >> 4)) ^ (4 * (i & 1));
        lk[i] = 4 * q;
        voff[i] = (unsigned)((8 * i + (lane >> 3)) * (K * 4) + q * 16);
    }
    unsigned char* ring = smem + wave * G_RING;
    // A-fragment reads: row tile rt, group o of a chunk: lane (m, kq) reads piece 4 o + kq of row rt*16 + m
    unsigned fa[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int row = rt * 16 + lm;
            fa[rt][o] = lbase + (unsigned)(wave * G_RING + row * 128 + (((4 * o + lq) ^ ((row >> 1) & 7)) * 16));
        }

    // ---- epilogue role: one cell per thread
    const int erow = tid >> 4, eul = tid & 15, eunit = u0 + eul;
    const bool e_ok = eunit < H;
    float* dcst = reinterpret_cast<float*>(smem + G_DCST);
    const bool last_block = (p.t1 == p.T);
    for (int s = 0; s < p.NS; ++s) {
        const int b = row0 + s * F_SR + erow;
        dcst[(s * F_SR + erow) * G_UN + eul] = (!last_block && e_ok && b < B) ? p.dc[(int64_t)b * H + eunit] : 0.f;
    }
    float* dgsm = reinterpret_cast<float*>(smem + G_DGSM);
    const int64_t H4 = 4 * (int64_t)H;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(g_zero4);

    for (int t = p.t1 - 1; t >= p.t0; --t) {
#pragma unroll 1
        for (int s = 0; s < p.NS; ++s) {
            const int rbase = row0 + s * F_SR;
            unsigned int* cnt = p.sync + (rg * F_MAXNS + s) * 32;
            const int done = p.t1 - 1 - t;
            if (done > 0) {
                if (tid == 0) {
                    const bool ok = spin_until_f(cnt, (unsigned int)(nC * done));
                    s_flag = ok ? 1 : 0;
                    if (!ok) atomicExch(p.err, 1);
                }
                F_BARRIER();
                if (s_flag == 0) return;
            }
            const int eb = rbase + erow;
            const bool ok = e_ok && eb < B;
            const int64_t rowi = (int64_t)t * B + eb;
            float stv[4], cv, cpv, dhov;
            {
                const float* st = p.stash_dg + rowi * H4 + eunit;
#pragma unroll
                // the gate stash is overwritten in place by dG_t, which other workgroups then read: this CU must not keep an
                // L1 copy of the old line (sc1 load; the sc1 stores below also drop the line from this XCD's L2)
                for (int g = 0; g < 4; ++g) stv[g] = ld_sc1(ok ? st + (int64_t)g * H : g_zero4);
                cv = *(ok ? p.c_all + rowi * H + eunit : g_zero4);
                cpv = *((ok && t > 0) ? p.c_all + (rowi - B) * H + eunit : g_zero4);
                dhov = *((ok && p.dh_out && t >= p.dh_first) ? p.dh_out + ((int64_t)(t - p.dh_first) * B + eb) * H + eunit : g_zero4);
            }

            f32x4 acc[2];
            acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < p.T - 1) {
                const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.stash_dg + ((int64_t)(t + 1) * B + rbase) * H4);
#define G_ISSUE(J)                                                                                              \
                {                                                                                               \
                    const int kc = k0 + 32 * (J);                                                               \
                    const unsigned char* sb = abase + kc * 4;                                                   \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
                        glds16f_sc1((kc + lk[i] + 3 < K) ? sb + voff[i] : zsrc, ring + ((J) % 4) * 4096 + i * 1024); \
                }
#define G_STEP(J, VM)                                                                                           \
                if ((J) + 3 < 16) G_ISSUE((J) + 3)                                                              \
                asm volatile("s_waitcnt vmcnt(" #VM ")" ::: "memory");                                          \
                {                                                                                               \
                    f32x4 a00, a10, a01, a11;                                                                   \
                    F_DSR(a00, fa[0][0], ((J) % 4) * 4096); F_DSR(a10, fa[1][0], ((J) % 4) * 4096);             \
                    F_DSR(a01, fa[0][1], ((J) % 4) * 4096); F_DSR(a11, fa[1][1], ((J) % 4) * 4096);             \
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a00), "+v"(a10));                                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                             \
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[r], wreg[(J) * 2][r], acc[0], 0, 0, 0);   \
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a10[r], wreg[(J) * 2][r], acc[1], 0, 0, 0);   \
                    }                                                                                           \
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01), "+v"(a11));                                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                             \
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[r], wreg[(J) * 2 + 1][r], acc[0], 0, 0, 0); \
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a11[r], wreg[(J) * 2 + 1][r], acc[1], 0, 0, 0); \
                    }                                                                                           \
                }
                G_ISSUE(0) G_ISSUE(1) G_ISSUE(2)
                G_STEP(0, 12) G_STEP(1, 12) G_STEP(2, 12) G_STEP(3, 12) G_STEP(4, 12) G_STEP(5, 12) G_STEP(6, 12)
                G_STEP(7, 12) G_STEP(8, 12) G_STEP(9, 12) G_STEP(10, 12) G_STEP(11, 12) G_STEP(12, 12)
                G_STEP(13, 8) G_STEP(14, 4) G_STEP(15, 0)
#undef G_STEP
#undef G_ISSUE
            }
            {
                float* rp = reinterpret_cast<float*>(ring);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rp[(rt * 16 + 4 * lq + r) * G_PLD + lm] = acc[rt][r];
            }
            F_BARRIER();

            float dg[4], dcn;
            {
                float dh = dhov;
#pragma unroll
                for (int w = 0; w < 8; ++w) dh += reinterpret_cast<const float*>(smem + w * G_RING)[erow * G_PLD + eul];
                float* dp = dcst + (s * F_SR + erow) * G_UN + eul;
                const float ig = stv[0], fg = stv[1], gg = stv[2], og = stv[3];
                const float tc = tanh_f(cv);
                const float dc = dh * og * (1.0f - tc * tc) + *dp;
                const float d_o = dh * tc;
                dg[0] = dc * gg * ig * (1.0f - ig);
                dg[1] = dc * cpv * fg * (1.0f - fg);
                dg[2] = dc * ig * (1.0f - gg * gg);
                dg[3] = d_o * og * (1.0f - og);
                dcn = ok ? dc * fg : 0.f;
                *dp = dcn;
#pragma unroll
                for (int g = 0; g < 4; ++g) dgsm[erow * 64 + g * 16 + eul] = ok ? dg[g] : 0.f;
            }
            F_BARRIER();
            if (wave == 0) {   // fp32 dG_t tile: 32 rows x 4 gates x 64 B = eight 16-byte write-through store instructions
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = q * 64 + lane;                 // 512 pieces of 16 B: (row, gate, quarter)
                    const int rl = idx >> 4, g = (idx >> 2) & 3, part = idx & 3;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(dgsm + rl * 64 + g * 16 + part * 4);
                    float* dst = p.stash_dg + ((int64_t)t * B + rbase + rl) * H4 + (int64_t)g * H + u0 + part * 4;
                    if (rbase + rl < B && u0 + part * 4 < H)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                }
            }
            if (ok && t == p.t0) p.dc[(int64_t)eb * H + eunit] = dcn;
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            F_BARRIER();
        }
    }
}

__global__ __launch_bounds__(G_NT) void lstm_seq_bwd_f32_persist_kernel(SeqBwdF32Args pa, SeqBwdF32Args pb, int na) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G_LDS];
    __shared__ int s_flag;
    if ((int)blockIdx.x < na) seq_bwd_f32_body(pa, blockIdx.x, smem, s_flag);
    else seq_bwd_f32_body(pb, blockIdx.x - na, smem, s_flag);
}

static int bwd_f32_capacity() {
    const int cap = coresident_capacity(reinterpret_cast<const void*>(&lstm_seq_bwd_f32_persist_kernel), G_NT);
    return cap < G_MAX_WG ? cap : G_MAX_WG;
}
int lstm_seq_bwd_f32_persist_supported(int B, int H) {
    if (!(B > 0 && B % F_SR == 0 && H % 4 == 0 && H >= 4 && 4 * H <= 4096)) return 0;
    const int cap = bwd_f32_capacity();
    const int nC = (H + G_UN - 1) / G_UN;
    int R = B / F_SR, ns = 1;
    while (R * nC > cap / 2 && ns < F_MAXNS && R % 2 == 0) { R /= 2; ns *= 2; }
    return (R * nC <= cap / 2 && R <= 64) ? ns : 0;
}

static int prep_g(SeqBwdF32Args& a) {
    const int ns = lstm_seq_bwd_f32_persist_supported(a.B, a.H);
    S2VT_REQUIRE(ns > 0, "lstm_seq_bwd_f32_persist: unsupported shape (B %% 32, H %% 4, H <= 1024) or it does not fit the device's resident capacity");
    S2VT_REQUIRE(a.T > 0 && a.t1 > a.t0 && a.t0 >= 0 && a.t1 <= a.T && a.w_hh_t && a.stash_dg && a.c_all && a.dc && a.sync && a.err,
                 "lstm_seq_bwd_f32_persist: bad arguments");
    S2VT_REQUIRE(a.ldwt % 4 == 0 && (reinterpret_cast<uintptr_t>(a.w_hh_t) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.stash_dg) & 15) == 0,
                 "lstm_seq_bwd_f32_persist: W_hh^T and dG rows must be 16-byte aligned");
    a.NS = ns;
    a.RB = ns * F_SR;
    return 0;
}

int lstm_seq_bwd_f32_persist2(hipStream_t stream, SeqBwdF32Args a, const SeqBwdF32Args* b) {
    int rc;
    if ((rc = prep_g(a))) return rc;
    SeqBwdF32Args bb = b ? *b : a;
    if (b) {
        if ((rc = prep_g(bb))) return rc;
        S2VT_REQUIRE(bb.sync != a.sync, "lstm_seq_bwd_f32_persist: paired layers need their own counters");
    }
    const int na = (a.B / a.RB) * cdiv(a.H, G_UN), nb = b ? (bb.B / bb.RB) * cdiv(bb.H, G_UN) : 0;
    S2VT_REQUIRE(na + nb <= bwd_f32_capacity(), "lstm_seq_bwd_f32_persist: %d workgroups would not be co-resident (device capacity %d)",
                 na + nb, bwd_f32_capacity());
    S2VT_HIP(hipMemsetAsync(a.sync, 0, lstm_persist_sync_bytes(), stream));
    if (b) S2VT_HIP(hipMemsetAsync(bb.sync, 0, lstm_persist_sync_bytes(), stream));
    hipLaunchKernelGGL(lstm_seq_bwd_f32_persist_kernel, dim3(na + nb), dim3(G_NT), 0, stream, a, bb, na);
    S2VT_LAUNCH_CHECK("lstm_seq_bwd_f32_persist_kernel");
    return 0;
}

}  // namespace s2vt
