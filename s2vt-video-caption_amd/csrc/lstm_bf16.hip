// bf16-storage LSTM timestep kernels for gfx950 (BASELINE config 3: B = 256, bf16 operands, fp32 accumulate).
//
// Same structure as lstm.hip — one launch per timestep, a workgroup owns a [batch rows] x [gate columns] tile, its
// waves split K, the partial tiles are summed through LDS and the whole cell runs in the epilogue — but the
// recurrent contraction reads bf16 operands (h_{t-1} / dG_{t+1} rows and W_hh / W_hh^T rows, zero-padded to a
// multiple of 64 in k) and runs on v_mfma_f32_32x32x16_bf16.  Cell state, gate inputs, gate stash and gradients
// stay fp32; every kernel also writes the bf16 row image of its output (h_t or dG_t), which is both the next
// timestep's A operand and, unchanged, the k-major bf16 plane of the batched GEMMs that consume it.
#include "common.h"
#include "kernels.h"

namespace s2vt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BKC = 64;                 // bf16 elements of k per staging chunk (128 B per row: full lines)
constexpr int BROWB = BKC * 2 + 16;     // LDS bytes per staged row (9 x 16-B slots: conflict-free ds_read_b128)
constexpr int BNW = 4;                  // waves per workgroup = K split
constexpr int BNT = BNW * 64;

__device__ __forceinline__ unsigned short f2bf(float x) {
    unsigned int u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// acc[mi][ni] += A[32*MT rows, 0:K) · B[32*NT rows, 0:K)^T over this wave's chunks (K multiple of 64, zero padded).
// arow/brow: per-lane row pointers for rows (lane/8 + 8 i); each lane moves the 16-B piece (lane % 8) of its rows.
// PF chunks of the wave are in flight at any time (PF register sets, loops fully unrolled so they are statically
// indexed).  Measured at B=256: PF = 3 in the BPTT step (K = 4H, 16 rounds per wave) is 24 % SLOWER than PF = 1 - the
// step is bound by its L2->CU traffic (256 workgroups x 516 KB = 132 MB per launch at 32x32 tiles), not by the round
// trips; larger tiles (fewer bytes) are the lever there, deeper prefetch only adds contention.
template <int MT, int NT, int PF>
__device__ __forceinline__ void wave_gemm_bf16(f32x16 (&acc)[MT][NT], const unsigned short* const (&arow)[MT * 4],
                                               const unsigned short* const (&brow)[NT * 4], int K, unsigned char* sA,
                                               unsigned char* sB, int wave, int lane) {
    const int nch = K / BKC;
    const int per_wave = (nch + BNW - 1) / BNW;
    const int n_round = (per_wave + PF - 1) / PF;
    const int lrow = lane >> 3, piece = lane & 7;
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero4);
    u32x4 ra[PF][MT * 4], rb[PF][NT * 4];
    auto load = [&](int c, u32x4 (&qa)[MT * 4], u32x4 (&qb)[NT * 4]) {      // past-the-end chunks read the zero block
        const bool in = c < nch;
        const int k0 = c * BKC + piece * 8;
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) qa[i] = *reinterpret_cast<const u32x4*>((in && arow[i]) ? arow[i] + k0 : zero);
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) qb[i] = *reinterpret_cast<const u32x4*>((in && brow[i]) ? brow[i] + k0 : zero);
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) load(wave + d * BNW, ra[d], rb[d]);
    for (int r = 0; r < n_round; ++r) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int i = 0; i < MT * 4; ++i) *reinterpret_cast<u32x4*>(sA + (lrow + 8 * i) * BROWB + piece * 16) = ra[d][i];
#pragma unroll
            for (int i = 0; i < NT * 4; ++i) *reinterpret_cast<u32x4*>(sB + (lrow + 8 * i) * BROWB + piece * 16) = rb[d][i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            load(wave + ((r + 1) * PF + d) * BNW, ra[d], rb[d]);       // refill this set
#pragma unroll
            for (int s = 0; s < BKC / 16; ++s) {
                bf16x8 a[MT], b[NT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    a[mi] = *reinterpret_cast<const bf16x8*>(sA + (mi * 32 + fr) * BROWB + s * 32 + fh * 16);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    b[ni] = *reinterpret_cast<const bf16x8*>(sB + (ni * 32 + fr) * BROWB + s * 32 + fh * 16);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
}

// 32x32 C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int MT, int NT>
__device__ __forceinline__ void write_partials32(const f32x16 (&acc)[MT][NT], float* red, int wave, int lane) {
    constexpr int TM = 32 * MT, RLD = 32 * NT + 1;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wave * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * RLD + ni * 32 + (lane & 31)] =
                    acc[mi][ni][r];
}
template <int MT, int NT>
__device__ __forceinline__ float read_sum32(const float* red, int row, int col) {
    constexpr int TM = 32 * MT, RLD = 32 * NT + 1;
    float s = red[row * RLD + col];
#pragma unroll
    for (int w = 1; w < BNW; ++w) s += red[(w * TM + row) * RLD + col];
    return s;
}

__device__ __forceinline__ bool xcd_tile_b(int NX, int NY, int& x, int& y) {
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    x = (j / NY) * 8 + xcd;
    y = j % NY;
    return x < NX;
}
static inline int xcd_grid_b(int NX, int NY) { return ((NX + 7) / 8) * 8 * NY; }

template <int MT, int NT>
constexpr int smem_bytes() {
    return (BNW * 32 * (MT + NT) * BROWB > BNW * 32 * MT * (32 * NT + 1) * 4) ? BNW * 32 * (MT + NT) * BROWB
                                                                              : BNW * 32 * MT * (32 * NT + 1) * 4;
}

// ------------------------------------------------------------------------------ forward step
template <int MT, int NT>
__global__ __launch_bounds__(BNT) void lstm_step_fwd_bf16_kernel(StepFwdBf16Args p) {
    constexpr int TM = 32 * MT, TN = 32 * NT, UN = TN / 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[smem_bytes<MT, NT>()];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* sA = smem + wave * 32 * (MT + NT) * BROWB;
    unsigned char* sB = sA + 32 * MT * BROWB;
    int tx, ty;
    if (!xcd_tile_b((p.H + UN - 1) / UN, (p.B + TM - 1) / TM, tx, ty)) return;
    const int b0 = ty * TM, u0 = tx * UN;
    const int lrow = lane >> 3;

    // epilogue operands requested ahead of the K loop (TM*UN cells, NE per thread)
    constexpr int NE = (TM * UN + BNT - 1) / BNT;
    float gxv[NE][4], cpv[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * BNT, bl = idx / UN, u = idx % UN;
        const int b = b0 + bl, unit = u0 + u;
        const bool ok = (idx < TM * UN) && (b < p.B) && (unit < p.H);
        const float* gsrc = p.gx ? p.gx + (int64_t)b * p.ldgx : p.bias;
#pragma unroll
        for (int g = 0; g < 4; ++g) gxv[e][g] = *((ok && gsrc) ? gsrc + (int64_t)g * p.H + unit : g_zero4);
        cpv[e] = *((ok && p.c_prev) ? p.c_prev + (int64_t)b * p.ldc + unit : g_zero4);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    if (p.hb_prev) {
        const unsigned short* arow[MT * 4];
        const unsigned short* brow[NT * 4];
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            const int b = b0 + lrow + 8 * i;
            arow[i] = (b < p.B) ? p.hb_prev + (int64_t)b * p.ldhb : nullptr;
        }
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) {
            const int r = lrow + 8 * i, g = r / UN, u = u0 + r % UN;
            brow[i] = (u < p.H) ? p.wb + ((int64_t)g * p.H + u) * p.ldwb : nullptr;
        }
        wave_gemm_bf16<MT, NT, (MT * NT >= 4) ? 1 : 2>(acc, arow, brow, p.Kp, sA, sB, wave, lane);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    write_partials32<MT, NT>(acc, red, wave, lane);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * BNT, bl = idx / UN, u = idx % UN;
        const int b = b0 + bl, unit = u0 + u;
        if (idx >= TM * UN || b >= p.B || unit >= p.H) continue;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = read_sum32<MT, NT>(red, bl, g * UN + u) + gxv[e][g];
        const float ig = sigmoidf_(pre[0]);
        const float fg = sigmoidf_(pre[1]);
        const float gg = tanhf_(pre[2]);
        const float og = sigmoidf_(pre[3]);
        const float c = fg * cpv[e] + ig * gg;
        const float h = og * tanhf_(c);
        if (p.h_out) p.h_out[(int64_t)b * p.ldho + unit] = h;
        p.hb_out[(int64_t)b * p.ldhbo + unit] = f2bf(h);
        p.c_out[(int64_t)b * p.ldco + unit] = c;
        if (p.stash) {
            float* st = p.stash + (int64_t)b * p.ldst + unit;
            st[0] = ig;
            st[(int64_t)p.H] = fg;
            st[(int64_t)2 * p.H] = gg;
            st[(int64_t)3 * p.H] = og;
        }
    }
}

int lstm_step_fwd_bf16(hipStream_t stream, const StepFwdBf16Args& a) {
    S2VT_REQUIRE(a.B > 0 && a.H > 0 && a.hb_out && a.c_out && (a.gx || a.bias), "lstm_step_fwd_bf16: bad arguments");
    S2VT_REQUIRE(!a.hb_prev || (a.Kp % 64 == 0 && a.Kp >= a.H && a.ldhb % 8 == 0 && a.ldwb % 8 == 0 &&
                                (reinterpret_cast<uintptr_t>(a.hb_prev) & 15) == 0 &&
                                (reinterpret_cast<uintptr_t>(a.wb) & 15) == 0),
                 "lstm_step_fwd_bf16: operands must be 16-B aligned bf16 rows zero-padded to a multiple of 64");
    if (a.B <= 32) {
        dim3 grid(xcd_grid_b(cdiv(a.H, 16), cdiv(a.B, 32)));
        hipLaunchKernelGGL((lstm_step_fwd_bf16_kernel<1, 2>), grid, dim3(BNT), 0, stream, a);
    } else {
        dim3 grid(xcd_grid_b(cdiv(a.H, 16), cdiv(a.B, 64)));
        hipLaunchKernelGGL((lstm_step_fwd_bf16_kernel<2, 2>), grid, dim3(BNT), 0, stream, a);
    }
    S2VT_LAUNCH_CHECK("lstm_step_fwd_bf16_kernel");
    return 0;
}

// ----------------------------------------------------------------------------- backward step
template <int MT, int NT>
__global__ __launch_bounds__(BNT) void lstm_step_bwd_bf16_kernel(StepBwdBf16Args p) {
    constexpr int TM = 32 * MT, TN = 32 * NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[smem_bytes<MT, NT>()];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* sA = smem + wave * 32 * (MT + NT) * BROWB;
    unsigned char* sB = sA + 32 * MT * BROWB;
    int tx, ty;
    if (!xcd_tile_b((p.H + TN - 1) / TN, (p.B + TM - 1) / TM, tx, ty)) return;
    const int b0 = ty * TM, n0 = tx * TN;
    const int lrow = lane >> 3;

    constexpr int NE = (TM * TN + BNT - 1) / BNT;
    float stv[NE][4], cv[NE], cpv[NE], dcv[NE], dhov[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * BNT, bl = idx / TN, ul = idx % TN;
        const int b = b0 + bl, unit = n0 + ul;
        const bool ok = (idx < TM * TN) && (b < p.B) && (unit < p.H);
#pragma unroll
        for (int g = 0; g < 4; ++g) stv[e][g] = *(ok ? p.stash + (int64_t)b * p.ldst + (int64_t)g * p.H + unit : g_zero4);
        cv[e] = *(ok ? p.c + (int64_t)b * p.ldc + unit : g_zero4);
        cpv[e] = *((ok && p.c_prev) ? p.c_prev + (int64_t)b * p.ldcp + unit : g_zero4);
        dcv[e] = *((ok && !p.dc_is_zero) ? p.dc + (int64_t)b * p.lddc + unit : g_zero4);
        dhov[e] = *((ok && p.dh_out) ? p.dh_out + (int64_t)b * p.lddho + unit : g_zero4);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    if (p.dgb_next) {
        const unsigned short* arow[MT * 4];
        const unsigned short* brow[NT * 4];
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            const int b = b0 + lrow + 8 * i;
            arow[i] = (b < p.B) ? p.dgb_next + (int64_t)b * p.lddgb : nullptr;
        }
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) {
            const int n = n0 + lrow + 8 * i;
            brow[i] = (n < p.H) ? p.wtb + (int64_t)n * p.ldwtb : nullptr;
        }
        wave_gemm_bf16<MT, NT, 1>(acc, arow, brow, p.Kp, sA, sB, wave, lane);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    write_partials32<MT, NT>(acc, red, wave, lane);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * BNT, bl = idx / TN, ul = idx % TN;
        const int b = b0 + bl, unit = n0 + ul;
        if (idx >= TM * TN || b >= p.B || unit >= p.H) continue;
        const float dh = read_sum32<MT, NT>(red, bl, ul) + dhov[e];
        const float ig = stv[e][0], fg = stv[e][1], gg = stv[e][2], og = stv[e][3];
        const float tc = tanhf_(cv[e]);
        const float dc = dh * og * (1.0f - tc * tc) + dcv[e];
        const float d_o = dh * tc;
        const float d4[4] = {dc * gg * ig * (1.0f - ig), dc * cpv[e] * fg * (1.0f - fg), dc * ig * (1.0f - gg * gg),
                             d_o * og * (1.0f - og)};
        float* dg = p.dg + (int64_t)b * p.lddg + unit;
        unsigned short* dgb = p.dgb + (int64_t)b * p.lddgbo + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            dg[(int64_t)g * p.H] = d4[g];
            dgb[(int64_t)g * p.H] = f2bf(d4[g]);
        }
        p.dc[(int64_t)b * p.lddc + unit] = dc * fg;
    }
}

int lstm_step_bwd_bf16(hipStream_t stream, const StepBwdBf16Args& a) {
    S2VT_REQUIRE(a.B > 0 && a.H > 0 && a.stash && a.c && a.dc && a.dg && a.dgb, "lstm_step_bwd_bf16: bad arguments");
    S2VT_REQUIRE(!a.dgb_next || (a.Kp % 64 == 0 && a.Kp >= 4 * a.H && a.lddgb % 8 == 0 && a.ldwtb % 8 == 0 &&
                                 (reinterpret_cast<uintptr_t>(a.dgb_next) & 15) == 0 &&
                                 (reinterpret_cast<uintptr_t>(a.wtb) & 15) == 0),
                 "lstm_step_bwd_bf16: operands must be 16-B aligned bf16 rows zero-padded to a multiple of 64");
    dim3 grid(xcd_grid_b(cdiv(a.H, 32), cdiv(a.B, 32)));
    hipLaunchKernelGGL((lstm_step_bwd_bf16_kernel<1, 1>), grid, dim3(BNT), 0, stream, a);
    S2VT_LAUNCH_CHECK("lstm_step_bwd_bf16_kernel");
    return 0;
}

}  // namespace s2vt
