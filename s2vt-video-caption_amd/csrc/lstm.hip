// Fused LSTM timestep kernels (forward cell, BPTT cell, decode logits+argmax) for gfx950.
//
// One launch per timestep computes, for a [16*MT batch rows] x [16*NT columns] tile per workgroup,
// the recurrent contraction on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) and the
// whole pointwise cell in the epilogue, so gate pre-activations never touch HBM:
//   forward  (S2VTModel.py:67,77,86,93,103 -> nn.LSTM step):   G = gx_t + h_{t-1} W_hh^T (+ Emb[tok] W_e^T)
//            columns of a tile = {i,f,g,o} x UN hidden units, so one workgroup owns complete cells;
//   backward (autograd of the same, train.py:124):  dh = dh_out_t + dG_{t+1} W_hh, then the gate
//            derivatives -> dG_t, dc_{t-1};
//   decode   (S2VTModel.py:95-96,105-106): logits tile + first-max argmax folded into one 64-bit
//            atomicMax per (row, tile).
// The 8 waves of a workgroup split K in 32-wide chunks (wave w takes chunks w, w+8, ...), each
// staging its operands through a wave-private LDS image with coalesced 16-B loads and a register
// prefetch of its next two chunks, and the 8 partial tiles are summed through LDS.
// Both operands are k-contiguous (h/dG rows, W_hh rows / W_hh^T rows), LDS row stride 36 floats:
// 16-B aligned staging writes and conflict-free ds_read_b128 operand reads.  Within each 16-wide
// k block lane quarter q owns k = 4q..4q+3 for both operands (fixed summation order).
// (These are the launch-per-timestep kernels: the fp32 training path (two layers on two streams, overlapped with the
// batched GEMMs), decode and beam search; the persistent kernels of lstm_persist*.hip are the bf16 configuration's.)
#include <stdlib.h>
#include "common.h"
#include "experiment.h"
#include "kernels.h"

namespace s2vt {

#ifndef S2VT_KC
#define S2VT_KC 32
#endif
constexpr int KC = S2VT_KC;      // k chunk per wave iteration (64: 256-B row segments; 32: half the LDS, one line)
constexpr int SLD = KC + 4;      // LDS row stride in floats (68 / 36: conflict-free ds_read_b128)
constexpr int LPR = KC / 4;      // lanes per staged row (one float4 each)
constexpr int RPL = 64 / LPR;    // rows covered by one wave-wide load
constexpr int LPT = 16 / RPL;    // loads per lane per 16 rows of tile
#ifndef S2VT_NWAVE_FWD
#define S2VT_NWAVE_FWD 8
#endif
#ifndef S2VT_NWAVE_BWD
#define S2VT_NWAVE_BWD 8
#endif
#ifndef S2VT_PF
#define S2VT_PF 2
#endif
// waves per workgroup = K-split factor inside the workgroup.  Both choices keep the LDS footprint at 69.6 KB so
// that two workgroups (e.g. a vid_rnn step and a word_rnn step launched on two streams) fit one CU.
constexpr int NW_FWD = S2VT_NWAVE_FWD;
constexpr int NW_BWD = S2VT_NWAVE_BWD;
constexpr int PF = S2VT_PF;            // staging chunks in flight per wave (register prefetch depth)


// Branch-free guarded 4-float load (see gemm.hip load4_guard): out-of-range accesses read a safe address and
// are zeroed by a select, so the staging burst stays a run of independent loads.
template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* base, const float* row, int c, int limit) {
    // Out-of-range accesses read a 16-byte block of zeros instead of being masked afterwards: the loaded value
    // then has NO consumer before the LDS staging store, so the loads stay in flight across the MFMA phase
    // (a select on the result would pull the vmcnt wait in front of the MFMAs).
    f32x4 v;
    if (VEC) {
        const bool ok = (row != nullptr) && (c < limit);
        const float* q = ok ? row + c : g_zero4;
        v = *reinterpret_cast<const f32x4*>(q);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (row != nullptr) && (c + j < limit);
            const float* q = ok ? row + c + j : g_zero4;
            v[j] = *q;
        }
    }
    return v;
}

// acc[mi][ni][a] += A[16*MT rows, 0:K] · B[16*NT rows, 0:K]^T over this wave's chunks.
// arow/brow: per-lane row pointers for rows (lane/16 + 4 i); sA/sB: wave-private LDS images.
template <int MT, int NT, int NA, bool VEC, int NWAVE>
__device__ __forceinline__ void wave_gemm_nt(f32x4 (&acc)[MT][NT][NA], const float* abase, const float* bbase,
                                             const float* const (&arow)[MT * LPT], const float* const (&brow)[NT * LPT],
                                             int K, float* sA, float* sB, int wave, int lane) {
    // Wave w owns chunks w, w+NWAVE, ...; PF of them are in flight (registers) at any time.  Loads are issued
    // unconditionally (chunks past K read the zero block), so the body is straight-line code and the compiler's
    // counted vmcnt leaves the younger chunks in flight while the oldest is staged and multiplied.
    const int nch = (K + KC - 1) / KC;
    const int per_wave = (nch + NWAVE - 1) / NWAVE;
    const int n_round = (per_wave + PF - 1) / PF;
    const int lrow = lane / LPR, kq = (lane % LPR) * 4;
    const int fi = lane & 15, fq = lane >> 4;
    f32x4 ra[PF][MT * LPT], rb[PF][NT * LPT];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int k0 = (wave + d * NWAVE) * KC + kq;
#pragma unroll
        for (int i = 0; i < MT * LPT; ++i) ra[d][i] = ld4<VEC>(abase, arow[i], k0, K);
#pragma unroll
        for (int i = 0; i < NT * LPT; ++i) rb[d][i] = ld4<VEC>(bbase, brow[i], k0, K);
    }
    for (int r = 0; r < n_round; ++r) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int i = 0; i < MT * LPT; ++i) *reinterpret_cast<f32x4*>(&sA[(lrow + RPL * i) * SLD + kq]) = ra[d][i];
#pragma unroll
            for (int i = 0; i < NT * LPT; ++i) *reinterpret_cast<f32x4*>(&sB[(lrow + RPL * i) * SLD + kq]) = rb[d][i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {   // refill this stage with the chunk PF rounds ahead
                const int k0 = (wave + ((r + 1) * PF + d) * NWAVE) * KC + kq;
#pragma unroll
                for (int i = 0; i < MT * LPT; ++i) ra[d][i] = ld4<VEC>(abase, arow[i], k0, K);
#pragma unroll
                for (int i = 0; i < NT * LPT; ++i) rb[d][i] = ld4<VEC>(bbase, brow[i], k0, K);
            }
#pragma unroll
            for (int s = 0; s < KC / 16; ++s) {
                f32x4 a[MT], b[NT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    a[mi] = *reinterpret_cast<const f32x4*>(&sA[(mi * 16 + fi) * SLD + 16 * s + 4 * fq]);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    b[ni] = *reinterpret_cast<const f32x4*>(&sB[(ni * 16 + fi) * SLD + 16 * s + 4 * fq]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NT; ++ni)
                            acc[mi][ni][j & (NA - 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[mi][j], b[ni][j], acc[mi][ni][j & (NA - 1)], 0, 0, 0);
            }
        }
    }
}

// Sum the NWAVE waves' partial tiles: every wave writes its accumulators to red[wave][row][col],
// after which red holds 4 partials per output.  16x16 C/D layout: col = lane&15, row = 4*(lane>>4)+reg.
// RLD = row stride of the partial tiles in floats, chosen per kernel so that the EPILOGUE's read pattern is free of bank
// conflicts (ds_read_b32: 32 banks, conflicts counted per 32-lane half).
template <int MT, int NT, int NA, int RLD = 16 * NT + 1>
__device__ __forceinline__ void write_partials(const f32x4 (&acc)[MT][NT][NA], float* red, int wave, int lane) {
    constexpr int TM = 16 * MT;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[mi][ni][0][r];
                if (NA == 2) v += acc[mi][ni][NA - 1][r];
                red[(wave * TM + mi * 16 + 4 * (lane >> 4) + r) * RLD + ni * 16 + (lane & 15)] = v;
            }
}

template <int MT, int NT, int NWAVE, int RLD = 16 * NT + 1>
__device__ __forceinline__ float read_sum(const float* red, int row, int col) {
    constexpr int TM = 16 * MT;
    float s = red[row * RLD + col];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) s += red[(w * TM + row) * RLD + col];
    return s;
}

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group).  All NY batch tiles of
// one column tile read the same weight slice, so they are given ids that differ by a multiple of 8: the slice
// is then fetched into ONE XCD's L2 (16 MB of W_hh / 8 XCDs = 2 MB per 4 MB L2) instead of NY of them.
// Speed only: any placement is correct.  Grid = ceil(NX/8)*8*NY blocks; ids with x >= NX exit.
__device__ __forceinline__ bool xcd_tile(int NX, int NY, int& x, int& y, int id = -1) {
    if (id < 0) id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    x = (j / NY) * 8 + xcd;
    y = j % NY;
    return x < NX;
}
static inline int xcd_grid(int NX, int NY) { return ((NX + 7) / 8) * 8 * NY; }

static inline bool vec_ok(const void* ptr, int64_t ld) {
    return ptr != nullptr && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0);
}

// ------------------------------------------------------------------------------ forward step
// (two workgroups must fit a CU: with 8 waves each that is 4 waves per SIMD -> <= 128 VGPRs, see launch bounds)
template <int MT, int NT, bool VEC>
__device__ __forceinline__ void lstm_step_fwd_body(const StepFwdArgs& p, int bid);

// Up to TWO independent timesteps per launch (blocks [0, na): pa, blocks [na, ...): pb), co-resident by construction
// (one workgroup of each per CU).  The training drivers launch one timestep per dispatch on two streams (measured
// faster); the two-step form is exercised by the kernel tests.
template <int MT, int NT, bool VEC>
__global__ __launch_bounds__(NW_FWD * 64, NW_FWD / 2) void lstm_step_fwd_kernel(StepFwdArgs pa, StepFwdArgs pb, int na) {
    if ((int)blockIdx.x < na) lstm_step_fwd_body<MT, NT, VEC>(pa, blockIdx.x);
    else lstm_step_fwd_body<MT, NT, VEC>(pb, blockIdx.x - na);
}

// token of batch row b for the embedding segment / the per-token table; ids outside [0, tok_limit) -> token 0 + error flag
__device__ __forceinline__ int64_t step_token(const StepFwdArgs& p, int b) {
    int64_t tok = p.tok_const;
    if (p.tok_idx) tok = p.tok_idx[b];
    else if (p.tok_packed) tok = (int64_t)(0xFFFFFFFFu - (uint32_t)(p.tok_packed[b] & 0xFFFFFFFFull));
    if ((uint64_t)tok >= (uint64_t)(int64_t)p.tok_limit) {
        if (p.tok_err) *p.tok_err = 1;
        tok = 0;
    }
    return tok;
}

// gates -> (c_t, h_t) and every optional output of a step, for one cell (shared by the fused epilogue and the stand-alone
// cell kernel: one expression, one rounding sequence)
__device__ __forceinline__ void step_cell_outputs(const StepFwdArgs& p, int b, int unit, const float (&pre)[4], float cpv) {
    const float ig = sigmoidf_(pre[0]);
    const float fg = sigmoidf_(pre[1]);
    const float gg = tanhf_(pre[2]);
    const float og = sigmoidf_(pre[3]);
    const float c = fg * cpv + ig * gg;
    const float h = og * tanhf_(c);
    p.h_out[(int64_t)b * p.ldho + unit] = h;
    if (p.h_out2) p.h_out2[(int64_t)b * p.ldho2 + unit] = h;
    p.c_out[(int64_t)b * p.ldco + unit] = c;
    if (p.stash) {
        float* st = p.stash + (int64_t)b * p.ldst + unit;
        st[0] = ig;
        st[(int64_t)p.H] = fg;
        st[(int64_t)2 * p.H] = gg;
        st[(int64_t)3 * p.H] = og;
    }
    if (p.h_planes) {
        // blocked plane layout (split.hip): element (row b, k = unit, plane pl) at
        //   (b/64)*(64*ld) + (k/16)*3072 + (pl*2 + (k%16)/8)*512 + (b%64)*8 + k%8.
        // The 8 threads of a row hold the 8 consecutive units u0..u0+7 (u0 % 8 == 0): their 2-byte stores fill one 16-byte
        // slot, the rows of the tile consecutive slots of the same piece
        unsigned short pl3[3];
        split3_bits(h, pl3);
        unsigned short* q = p.h_planes + (int64_t)(b >> 6) * (64 * p.ldhp) + (int64_t)(unit >> 4) * 3072 +
                            ((unit >> 3) & 1) * 512 + (b & 63) * 8 + (unit & 7);
        q[0] = pl3[0];
        q[1024] = pl3[1];
        q[2048] = pl3[2];
    }
}

template <int MT, int NT, bool VEC>
__device__ __forceinline__ void lstm_step_fwd_body(const StepFwdArgs& p, int bid) {
    constexpr int TM = 16 * MT, TN = 16 * NT, UN = TN / 4;
    constexpr int NWAVE = NW_FWD, NTHR = NWAVE * 64;
    constexpr int NA = (MT * NT == 1) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float smem[NWAVE * (TM + TN) * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sA = smem + wave * (TM + TN) * SLD;
    float* sB = sA + TM * SLD;
    int tx, ty;
    if (!xcd_tile((p.H + UN - 1) / UN, (p.B + TM - 1) / TM, tx, ty, bid)) return;
    const int b0 = ty * TM, u0 = tx * UN;
    const int lrow = lane / LPR;

    f32x4 acc[MT][NT][NA];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[mi][ni][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue operands (gate inputs, c_{t-1}) are requested NOW, ahead of the K loop, so their HBM/MALL latency
    // is hidden behind the contraction instead of being exposed after it (one output element per thread).
    static_assert(TM * UN <= NTHR, "one epilogue element per thread");
    const int ebl = tid / UN, eu = tid % UN;
    const int eb = b0 + ebl, eunit = u0 + eu;
    const bool evalid = (tid < TM * UN) && (eb < p.B) && (eunit < p.H);
    float gxv[4] = {0.f, 0.f, 0.f, 0.f}, cpv = 0.f;
    if (!p.z_out) {
        const float* gsrc = p.gx ? p.gx + (int64_t)((p.gx_idx && evalid) ? p.gx_idx[eb] : eb) * p.ldgx : p.bias;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* q = (evalid && gsrc) ? gsrc + (int64_t)g * p.H + eunit : g_zero4;
            gxv[g] = *q;
        }
        const float* q = (evalid && p.c_prev) ? p.c_prev + (int64_t)eb * p.ldc + eunit : g_zero4;
        cpv = *q;
    }
    float gtv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.gx_tab && !p.z_out) {       // embedded-word half of the gate input from the per-token table (two dependent loads, behind the K loop)
        const int64_t tok = evalid ? step_token(p, eb) : 0;
        const float* trow = p.gx_tab + tok * p.ldtab;
#pragma unroll
        for (int g = 0; g < 4; ++g) gtv[g] = *(evalid ? trow + (int64_t)g * p.H + eunit : g_zero4);
    }

    if (p.h_prev) {
        const float* arow[MT * LPT];
        const float* brow[NT * LPT];
#pragma unroll
        for (int i = 0; i < MT * LPT; ++i) {
            const int b = b0 + lrow + RPL * i;
            arow[i] = (b < p.B) ? p.h_prev + (int64_t)b * p.ldh : nullptr;
        }
#pragma unroll
        for (int i = 0; i < NT * LPT; ++i) {
            const int r = lrow + RPL * i, g = r / UN, u = u0 + r % UN;
            brow[i] = (u < p.H) ? p.w_hh + ((int64_t)g * p.H + u) * p.ldw : nullptr;
        }
        wave_gemm_nt<MT, NT, NA, VEC, NWAVE>(acc, p.h_prev, p.w_hh, arow, brow, p.H, sA, sB, wave, lane);
    }
    if (p.x2 && !p.z_out) {
        const float* arow[MT * LPT];
        const float* brow[NT * LPT];
#pragma unroll
        for (int i = 0; i < MT * LPT; ++i) {
            const int b = b0 + lrow + RPL * i;
            if (b < p.B) {
                arow[i] = p.x2 + step_token(p, b) * p.ldx2;
            } else {
                arow[i] = nullptr;
            }
        }
#pragma unroll
        for (int i = 0; i < NT * LPT; ++i) {
            const int r = lrow + RPL * i, g = r / UN, u = u0 + r % UN;
            brow[i] = (u < p.H) ? p.w2 + ((int64_t)g * p.H + u) * p.ldw2 : nullptr;
        }
        wave_gemm_nt<MT, NT, NA, VEC, NWAVE>(acc, p.x2, p.w2, arow, brow, p.K2, sA, sB, wave, lane);
    }

    // a half-wave of the epilogue reads 4 rows x UN = 8 consecutive columns per gate: row stride 8 (mod 32) puts the 32
    // lanes on 32 banks (stride 33 put them on 11: 4-way conflicts on each of the 32 reads of a thread)
    constexpr int RLD = TN + 8;
    static_assert(UN == 8 && NWAVE * TM * RLD <= NWAVE * (TM + TN) * SLD, "partial tiles fit the staging area");
    __syncthreads();
    float* red = smem;
    write_partials<MT, NT, NA, RLD>(acc, red, wave, lane);
    __syncthreads();

    if (evalid && p.z_out) {           // contraction only: the cell update is lstm_cell_pointwise's
#pragma unroll
        for (int g = 0; g < 4; ++g)
            p.z_out[(int64_t)eb * p.ldz + (int64_t)g * p.H + eunit] = read_sum<MT, NT, NWAVE, RLD>(red, ebl, g * UN + eu);
        return;
    }
    if (evalid) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = read_sum<MT, NT, NWAVE, RLD>(red, ebl, g * UN + eu) + gxv[g] + gtv[g];
        step_cell_outputs(p, eb, eunit, pre, cpv);
    }
}

// The epilogue of lstm_step_fwd_body on a contraction that ran as its own launch (z_out): one thread per NU consecutive cells
// of a batch row (NU = 4: 16-byte loads and stores, H % 4 == 0 and 16-byte aligned rows; NU = 1 otherwise).
template <int NU>
__global__ __launch_bounds__(256) void lstm_cell_pointwise_kernel(StepFwdArgs p) {
    const int upr = p.H / NU;                          // threads per batch row
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= p.B * upr) return;
    const int b = i / upr, unit = (i % upr) * NU;
    typedef float vec __attribute__((ext_vector_type(NU)));
    const float* gsrc = p.gx ? p.gx + (int64_t)(p.gx_idx ? p.gx_idx[b] : b) * p.ldgx : p.bias;
    const float* zsrc = p.z_out + (int64_t)b * p.ldz + unit;
    const float* trow = p.gx_tab ? p.gx_tab + step_token(p, b) * p.ldtab + unit : nullptr;
    vec zv[4], gxv[4], gtv[4], cpv;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        zv[g] = *reinterpret_cast<const vec*>(zsrc + (int64_t)g * p.H);
        gxv[g] = *reinterpret_cast<const vec*>(gsrc + (int64_t)g * p.H + unit);
        gtv[g] = trow ? *reinterpret_cast<const vec*>(trow + (int64_t)g * p.H) : (vec)(0.f);
    }
    cpv = p.c_prev ? *reinterpret_cast<const vec*>(p.c_prev + (int64_t)b * p.ldc + unit) : (vec)(0.f);
#pragma unroll
    for (int e = 0; e < NU; ++e) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = zv[g][e] + gxv[g][e] + gtv[g][e];
        step_cell_outputs(p, b, unit + e, pre, cpv[e]);
    }
}

static bool step_fwd_vec(const StepFwdArgs& a) {
    // vector path: 16-B aligned rows whose length is a multiple of 4 floats, for every operand in use
    return (!a.h_prev || (vec_ok(a.h_prev, a.ldh) && vec_ok(a.w_hh, a.ldw) && a.H % 4 == 0)) &&
           (!a.x2 || (vec_ok(a.x2, a.ldx2) && vec_ok(a.w2, a.ldw2) && a.K2 % 4 == 0));
}

// b == nullptr: one timestep; otherwise two independent timesteps of the same (B, H) in one launch
int lstm_step_fwd2(hipStream_t stream, const StepFwdArgs& a, const StepFwdArgs* b) {
    S2VT_REQUIRE(a.B > 0 && a.H > 0 && ((a.h_out && a.c_out) || a.z_out), "lstm_step_fwd: bad arguments");
    S2VT_REQUIRE(a.gx || a.bias || a.z_out, "lstm_step_fwd: need gx or bias");
    S2VT_REQUIRE(!a.z_out || (a.h_prev && !b && a.ldz >= 4 * (int64_t)a.H), "lstm_step_fwd: a contraction-only step needs h_prev and runs alone");
    S2VT_REQUIRE(!(a.x2 || a.gx_tab) || a.tok_limit > 0, "lstm_step_fwd: a token segment needs tok_limit (rows of the table)");
    S2VT_REQUIRE(!b || (b->B == a.B && b->H == a.H && b->h_out && b->c_out && (b->gx || b->bias)),
                 "lstm_step_fwd: paired steps must have the same batch and hidden size");
    // B <= 4: gate GEMVs (lstm_gemv.hip) - measured faster than the 16-row tile up to there (a B = 1 greedy decode 3.07 vs 3.67 ms,
    // B = 4 3.47 vs 3.73, B = 6 4.01 vs 3.81: profiles/round5_gemv_small_batch.txt); option gemv = 2 sends every B <= 8 there
    if (!b && lstm_step_fwd_gemv_ok(a) && (option(O_GEMV) == 2 || (option(O_GEMV) == 1 && a.B <= 4))) return lstm_step_fwd_gemv(stream, a);
    const bool vec = step_fwd_vec(a) && (!b || step_fwd_vec(*b));
    const StepFwdArgs& bb = b ? *b : a;
    if (a.B <= 16) {
        const int na = xcd_grid(cdiv(a.H, 8), cdiv(a.B, 16));
        dim3 grid(b ? 2 * na : na);
        if (vec) hipLaunchKernelGGL((lstm_step_fwd_kernel<1, 2, true>), grid, dim3(NW_FWD * 64), 0, stream, a, bb, na);
        else hipLaunchKernelGGL((lstm_step_fwd_kernel<1, 2, false>), grid, dim3(NW_FWD * 64), 0, stream, a, bb, na);
    } else {
        const int na = xcd_grid(cdiv(a.H, 8), cdiv(a.B, 32));
        dim3 grid(b ? 2 * na : na);
        if (vec) hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 2, true>), grid, dim3(NW_FWD * 64), 0, stream, a, bb, na);
        else hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 2, false>), grid, dim3(NW_FWD * 64), 0, stream, a, bb, na);
    }
    S2VT_LAUNCH_CHECK("lstm_step_fwd_kernel");
    return 0;
}
int lstm_step_fwd(hipStream_t stream, const StepFwdArgs& a) { return lstm_step_fwd2(stream, a, nullptr); }

int lstm_cell_pointwise(hipStream_t stream, const StepFwdArgs& a) {
    S2VT_REQUIRE(a.B > 0 && a.H > 0 && a.h_out && a.c_out && a.z_out && a.ldz >= 4 * (int64_t)a.H, "lstm_cell_pointwise: bad arguments");
    S2VT_REQUIRE(a.gx || a.bias, "lstm_cell_pointwise: need gx or bias");
    S2VT_REQUIRE(!a.x2, "lstm_cell_pointwise: the token segment must be the per-token table (gx_tab), not a second K segment");
    S2VT_REQUIRE(!a.gx_tab || a.tok_limit > 0, "lstm_cell_pointwise: a token segment needs tok_limit (rows of the table)");
    const bool v4 = a.H % 4 == 0 && vec_ok(a.z_out, a.ldz) && (!a.gx || vec_ok(a.gx, a.ldgx)) && (!a.bias || vec_ok(a.bias, 4)) &&
                    (!a.gx_tab || vec_ok(a.gx_tab, a.ldtab)) && (!a.c_prev || vec_ok(a.c_prev, a.ldc));
    if (v4) hipLaunchKernelGGL(lstm_cell_pointwise_kernel<4>, dim3((unsigned)cdiv(a.B * (a.H / 4), 256)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(lstm_cell_pointwise_kernel<1>, dim3((unsigned)cdiv(a.B * a.H, 256)), dim3(256), 0, stream, a);
    S2VT_LAUNCH_CHECK("lstm_cell_pointwise_kernel");
    return 0;
}

// ----------------------------------------------------------------------------- backward step
template <int MT, int NT, bool VEC>
__device__ __forceinline__ void lstm_step_bwd_body(const StepBwdArgs& p, int bid);

template <int MT, int NT, bool VEC>
__global__ __launch_bounds__(NW_BWD * 64) void lstm_step_bwd_kernel(StepBwdArgs pa, StepBwdArgs pb, int na) {
    if ((int)blockIdx.x < na) lstm_step_bwd_body<MT, NT, VEC>(pa, blockIdx.x);
    else lstm_step_bwd_body<MT, NT, VEC>(pb, blockIdx.x - na);
}

template <int MT, int NT, bool VEC>
__device__ __forceinline__ void lstm_step_bwd_body(const StepBwdArgs& p, int bid) {
    constexpr int TM = 16 * MT, TN = 16 * NT;
    constexpr int NWAVE = NW_BWD, NTHR = NWAVE * 64;
    constexpr int NA = (MT * NT == 1) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float smem[NWAVE * (TM + TN) * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sA = smem + wave * (TM + TN) * SLD;
    float* sB = sA + TM * SLD;
    int tx, ty;
    if (!xcd_tile((p.H + TN - 1) / TN, (p.B + TM - 1) / TM, tx, ty, bid)) return;
    const int b0 = ty * TM, n0 = tx * TN;
    const int lrow = lane / LPR;

    f32x4 acc[MT][NT][NA];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[mi][ni][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue operands requested ahead of the K loop (see the forward kernel)
    static_assert(TM * TN <= NTHR, "one epilogue element per thread");
    const int ebl = tid / TN, eul = tid % TN;
    const int eb = b0 + ebl, eunit = n0 + eul;
    const bool evalid = (tid < TM * TN) && (eb < p.B) && (eunit < p.H);
    float stv[4], cv, cpv, dcv, dhov;
    {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* q = evalid ? p.stash + (int64_t)eb * p.ldst + (int64_t)g * p.H + eunit : g_zero4;
            stv[g] = *q;
        }
        const float* q1 = evalid ? p.c + (int64_t)eb * p.ldc + eunit : g_zero4;
        const float* q2 = (evalid && p.c_prev) ? p.c_prev + (int64_t)eb * p.ldcp + eunit : g_zero4;
        const float* q3 = (evalid && !p.dc_is_zero) ? p.dc + (int64_t)eb * p.lddc + eunit : g_zero4;
        const float* q4 = (evalid && p.dh_out) ? p.dh_out + (int64_t)eb * p.lddho + eunit : g_zero4;
        cv = *q1; cpv = *q2; dcv = *q3; dhov = *q4;
    }

    if (p.dg_next) {
        const float* arow[MT * LPT];
        const float* brow[NT * LPT];
#pragma unroll
        for (int i = 0; i < MT * LPT; ++i) {
            const int b = b0 + lrow + RPL * i;
            arow[i] = (b < p.B) ? p.dg_next + (int64_t)b * p.lddg : nullptr;
        }
#pragma unroll
        for (int i = 0; i < NT * LPT; ++i) {
            const int n = n0 + lrow + RPL * i;
            brow[i] = (n < p.H) ? p.w_hh_t + (int64_t)n * p.ldwt : nullptr;
        }
        wave_gemm_nt<MT, NT, NA, VEC, NWAVE>(acc, p.dg_next, p.w_hh_t, arow, brow, 4 * p.H, sA, sB, wave, lane);
    }
    __syncthreads();
    float* red = smem;
    write_partials<MT, NT, NA>(acc, red, wave, lane);
    __syncthreads();

    if (evalid) {
        const int b = eb, unit = eunit;
        const float dh = read_sum<MT, NT, NWAVE>(red, ebl, eul) + dhov;
        const float ig = stv[0], fg = stv[1], gg = stv[2], og = stv[3];
        const float tc = tanhf_(cv);
        const float dc = dh * og * (1.0f - tc * tc) + dcv;
        const float d_o = dh * tc;
        float* dg = p.dg + (int64_t)b * p.lddg_out + unit;
        dg[0] = dc * gg * ig * (1.0f - ig);
        dg[(int64_t)p.H] = dc * cpv * fg * (1.0f - fg);
        dg[(int64_t)2 * p.H] = dc * ig * (1.0f - gg * gg);
        dg[(int64_t)3 * p.H] = d_o * og * (1.0f - og);
        p.dc[(int64_t)b * p.lddc + unit] = dc * fg;
    }
}

int lstm_step_bwd2(hipStream_t stream, const StepBwdArgs& a, const StepBwdArgs* b) {
    S2VT_REQUIRE(a.B > 0 && a.H > 0 && a.stash && a.c && a.dc && a.dg, "lstm_step_bwd: bad arguments");
    S2VT_REQUIRE(!b || (b->B == a.B && b->H == a.H && b->stash && b->c && b->dc && b->dg),
                 "lstm_step_bwd: paired steps must have the same batch and hidden size");
    auto okv = [](const StepBwdArgs& x) { return !x.dg_next || (vec_ok(x.dg_next, x.lddg) && vec_ok(x.w_hh_t, x.ldwt)); };
    const bool vec = okv(a) && (!b || okv(*b));
    const StepBwdArgs& bb = b ? *b : a;
    // 32-row tiles where they still fill the chip (B >= 128 at H = 1000): a workgroup then takes
    // in dG of 32 rows + one W_hh^T slice (768 KB) where two 16-row workgroups take in 1 MB - config-3-shard step (B = 128)
    // 20.56 -> 20.11 ms; at B = 64 they would leave half the compute units idle (11.6 -> 12.0 ms)
    const bool wide = vec && cdiv(a.H, 16) * cdiv(a.B, 32) >= 240;
    if (wide) {
        const int na = xcd_grid(cdiv(a.H, 16), cdiv(a.B, 32));
        hipLaunchKernelGGL((lstm_step_bwd_kernel<2, 1, true>), dim3(b ? 2 * na : na), dim3(NW_BWD * 64), 0, stream, a, bb, na);
        S2VT_LAUNCH_CHECK("lstm_step_bwd_kernel");
        return 0;
    }
    const int na = xcd_grid(cdiv(a.H, 16), cdiv(a.B, 16));
    dim3 grid(b ? 2 * na : na);
    if (vec) hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 1, true>), grid, dim3(NW_BWD * 64), 0, stream, a, bb, na);
    else hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 1, false>), grid, dim3(NW_BWD * 64), 0, stream, a, bb, na);
    S2VT_LAUNCH_CHECK("lstm_step_bwd_kernel");
    return 0;
}
int lstm_step_bwd(hipStream_t stream, const StepBwdArgs& a) { return lstm_step_bwd2(stream, a, nullptr); }

// -------------------------------------------------------------------- decode: logits + argmax
__device__ __forceinline__ uint32_t ordered_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int MT, int NT, bool VEC, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 4) void logits_argmax_kernel(LogitsArgmaxArgs p) {
    constexpr int TM = 16 * MT, TN = 16 * NT;
    constexpr int NTHR = NWAVE * 64;
    constexpr int NA = (MT * NT == 1) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float smem[NWAVE * (TM + TN) * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sA = smem + wave * (TM + TN) * SLD;
    float* sB = sA + TM * SLD;
    int tx, ty;
    if (!xcd_tile((p.V + TN - 1) / TN, (p.B + TM - 1) / TM, tx, ty)) return;
    const int xrec = p.stamps ? (int)blockIdx.x : -1;
    XSTAMP(p.stamps, xrec, 0);
    const int b0 = ty * TM, n0 = tx * TN;
    const int lrow = lane / LPR;

    // epilogue role: 8 threads per batch row, TN/8 columns each; the bias of those columns is requested NOW, ahead of the
    // contraction (in-kernel stamps: fetched in the epilogue it cost every tile ~1.5 us of exposed latency)
    constexpr int CPT = TN / 8;
    static_assert(TM * 8 == 256 && TM * 8 <= NTHR, "one pass over the tile");
    const int bl = (tid >> 3) % TM, sub = tid & 7;
    const bool active = tid < TM * 8;
    float bv[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int n = n0 + sub * CPT + j;
        bv[j] = *((active && p.b_out && n < p.V) ? p.b_out + n : g_zero4);
    }

    f32x4 acc[MT][NT][NA];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[mi][ni][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* arow[MT * LPT];
        const float* brow[NT * LPT];
#pragma unroll
        for (int i = 0; i < MT * LPT; ++i) {
            const int b = b0 + lrow + RPL * i;
            arow[i] = (b < p.B) ? p.h + (int64_t)b * p.ldh : nullptr;
        }
#pragma unroll
        for (int i = 0; i < NT * LPT; ++i) {
            const int n = n0 + lrow + RPL * i;
            brow[i] = (n < p.V) ? p.w_out + (int64_t)n * p.ldw : nullptr;
        }
        wave_gemm_nt<MT, NT, NA, VEC, NWAVE>(acc, p.h, p.w_out, arow, brow, p.H, sA, sB, wave, lane);
    }
    XSTAMP(p.stamps, xrec, 1);
    __syncthreads();
    XSTAMP(p.stamps, xrec, 2);
    float* red = smem;
    write_partials<MT, NT, NA>(acc, red, wave, lane);
    __syncthreads();
    XSTAMP(p.stamps, xrec, 3);

    // 8 threads per batch row, TN/8 columns each; first-max (lowest index) wins ties.
    const int b = b0 + bl;
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int nl = sub * CPT + j, n = n0 + nl;
        if (active && b < p.B && n < p.V) {
            const float v = read_sum<MT, NT, NWAVE>(red, bl, nl) + bv[j];
            const unsigned long long key =
                ((unsigned long long)ordered_bits(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)n);
            best = key > best ? key : best;
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (active && sub == 0 && b < p.B && best) atomicMax(&p.packed[b], best);
    XSTAMP(p.stamps, xrec, 4);
}

int logits_argmax(hipStream_t stream, const LogitsArgmaxArgs& a) {
    S2VT_REQUIRE(a.B > 0 && a.H > 0 && a.V > 0 && a.h && a.w_out && a.packed, "logits_argmax: bad arguments");
    const bool vec = vec_ok(a.h, a.ldh) && vec_ok(a.w_out, a.ldw) && a.H % 4 == 0;
    dim3 grid(xcd_grid(cdiv(a.V, 32), cdiv(a.B, 32)));
    // 4 waves per tile (K split four ways, 36.9 KB LDS, four workgroups per CU): with eight (two workgroups per CU) every
    // workgroup of a CU sat in its reduction / argmax phase at about the same time and the CU's ingest idled meanwhile
    // (in-kernel stamps of all 1500 workgroups, tools/bench_argmax_stamps.py; 2 % on a greedy decode)
    if (vec) hipLaunchKernelGGL((logits_argmax_kernel<2, 2, true, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((logits_argmax_kernel<2, 2, false, 4>), grid, dim3(256), 0, stream, a);
    S2VT_LAUNCH_CHECK("logits_argmax_kernel");
    return 0;
}

}  // namespace s2vt
