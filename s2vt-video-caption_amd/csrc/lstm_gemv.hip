// Small-batch LSTM timestep (B <= 8) as gate GEMVs - north_star's formulation of the per-timestep cell for the regime where a
// 16-row MFMA tile would be mostly padding (eval.py:27 decodes 10 clips at a time, S2VTModel.py:98-107 one step per launch):
//   G[b][g H + u] = gx[b][g H + u] + sum_k h_{t-1}[b][k] W_hh[g H + u][k] (+ sum_e Emb[tok_b][e] W_e[g H + u][e])
// i.e. per layer the 4 gate GEMVs of the recurrent weights and the 4 of the (embedded) input weights, fused with the cell.
//
// One workgroup = 4 waves = 4 hidden units (complete cells: the 16 gate rows of W_hh / W_e).  The batch's h_{t-1} rows (and
// embedded words) are staged ONCE per workgroup in LDS (coalesced 16-byte loads; <= 32 KB at B = 8, H = 1000); a wave then
// streams the four gate rows of ITS unit straight from memory into registers - lane l takes k = 4 l + 256 i, a wave
// instruction reads 1 KB contiguous, all loads of a row quartet are issued before the first use (no LDS round trip for the
// weights: each byte is used once per workgroup) - and multiplies them with the NB staged rows (conflict-free ds_read_b128:
// consecutive lanes, consecutive 16 bytes).  The 4 x NB partial dot products per lane are summed over the wave by a halving
// butterfly of xor shuffles (4 NB + 1 shuffles instead of 24 NB), lanes 0..NB-1 each finish one batch row's cell and write h_t,
// c_t, the gate stash and - for the decode - the h_t planes.  Same outputs as lstm_step_fwd_kernel (lstm.hip) up to the order of fp32
// additions; the fp32-MFMA tile kernel stays the path of B > 8, of paired steps and of contraction-only steps.
#include "common.h"
#include "kernels.h"

namespace s2vt {

// Sum M = 2^h values (h <= 5) per lane over the 64 lanes of a wave with M - 1 + (6 - h) shuffles instead of 6 M: at every step the
// two halves of the lanes exchange the half of the values they do not keep (a halving butterfly), until one value per lane is left,
// which the remaining steps sum plainly.  Lane l ends with the total of value index l >> (6 - h).
template <int M, int STEP>
__device__ __forceinline__ void wave_reduce_scatter_step(float (&v)[M], int lane) {
    constexpr int o = 32 >> STEP;
    constexpr int m = (M >> STEP) > 1 ? (M >> STEP) : 1;        // values per lane entering this step
    if constexpr (m > 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int j = 0; j < m / 2; ++j) {
            float a = v[j], b = v[j + m / 2];
            asm volatile("" : "+v"(a), "+v"(b));                // (two registers, then a select - not a select of the register INDEX,
            const float keep = up ? b : a;                      //  which hipcc turns into a 32-way compare chain per access)
            const float send = up ? a : b;
            v[j] = keep + __shfl_xor(send, o);
        }
    } else {
        v[0] += __shfl_xor(v[0], o);
    }
}
template <int M>
__device__ __forceinline__ float wave_reduce_scatter(float (&v)[M], int lane) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "power of two, at most 32 values");
    wave_reduce_scatter_step<M, 0>(v, lane);
    wave_reduce_scatter_step<M, 1>(v, lane);
    wave_reduce_scatter_step<M, 2>(v, lane);
    wave_reduce_scatter_step<M, 3>(v, lane);
    wave_reduce_scatter_step<M, 4>(v, lane);
    wave_reduce_scatter_step<M, 5>(v, lane);
    return v[0];
}

__device__ __forceinline__ int64_t gemv_token(const StepFwdArgs& p, int b) {
    int64_t tok = p.tok_const;
    if (p.tok_idx) tok = p.tok_idx[b];
    else if (p.tok_packed) tok = (int64_t)(0xFFFFFFFFu - (uint32_t)(p.tok_packed[b] & 0xFFFFFFFFull));
    if ((uint64_t)tok >= (uint64_t)(int64_t)p.tok_limit) {
        if (p.tok_err) *p.tok_err = 1;
        tok = 0;
    }
    return tok;
}

constexpr int GV_UNITS = 4;            // hidden units (= waves) per workgroup
constexpr int GV_MAXK = 4096;          // staged row length in floats (h and embedded word separately)

// one K segment: acc[g][b] += sum_k x[b][k] * W[(g H + unit)][k] for this wave's unit; x rows staged in LDS (row stride xs)
template <int NB>
__device__ __forceinline__ void gemv_segment(float (&acc)[4][NB], const float* xs, int xld, const float* w, int64_t ldw, int H, int unit, int K,
                                             int lane) {
    const float* wrow[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wrow[g] = w + ((int64_t)g * H + unit) * ldw;
    for (int k0 = 0; k0 < K; k0 += 1024) {            // four 256-wide slabs per trip: 16 row loads in flight per lane
        f32x4 wv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 256 * i + 4 * lane;
#pragma unroll
            for (int g = 0; g < 4; ++g) wv[i][g] = *reinterpret_cast<const f32x4*>(k < K ? wrow[g] + k : g_zero4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 256 * i + 4 * lane;
            if (k0 + 256 * i >= K) break;               // (wave-uniform)
            const int kk = k < K ? k : 0;               // lanes past the row's end multiply zeros (wv) with a valid LDS address
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + b * xld + kk);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g][b] += wv[i][g][0] * xv[0] + wv[i][g][1] * xv[1] + wv[i][g][2] * xv[2] + wv[i][g][3] * xv[3];
            }
        }
    }
}

template <int NB>
__global__ __launch_bounds__(64 * GV_UNITS) void lstm_step_fwd_gemv_kernel(StepFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float gv_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, K2 = p.x2 ? p.K2 : 0;
    const int hld = (H + 3) & ~3, xld = (K2 + 3) & ~3;
    float* hs = gv_smem;                      // [NB][hld]
    float* xs = gv_smem + NB * hld;           // [NB][xld]
    // ---- stage the batch's operand rows (zero rows past B; 16-byte loads: the launcher checked alignment and H % 4 == K2 % 4 == 0)
    if (p.h_prev) {
        for (int i = tid; i < NB * (hld >> 2); i += 64 * GV_UNITS) {
            const int b = i / (hld >> 2), k = (i % (hld >> 2)) * 4;
            *reinterpret_cast<f32x4*>(hs + b * hld + k) = *reinterpret_cast<const f32x4*>(b < p.B ? p.h_prev + (int64_t)b * p.ldh + k : g_zero4);
        }
    }
    if (p.x2) {
        for (int i = tid; i < NB * (xld >> 2); i += 64 * GV_UNITS) {
            const int b = i / (xld >> 2), k = (i % (xld >> 2)) * 4;
            const float* row = b < p.B ? p.x2 + gemv_token(p, b) * p.ldx2 + k : g_zero4;
            *reinterpret_cast<f32x4*>(xs + b * xld + k) = *reinterpret_cast<const f32x4*>(row);
        }
    }
    const int unit = (int)blockIdx.x * GV_UNITS + wave;
    const bool uok = unit < H;
    const int un = uok ? unit : 0;
    // ---- epilogue operands of lane b (batch row b), requested before the contraction
    const int eb = lane;
    const bool evalid = uok && eb < p.B && eb < NB;
    float gxv[4] = {0.f, 0.f, 0.f, 0.f}, gtv[4] = {0.f, 0.f, 0.f, 0.f}, cpv = 0.f;
    {
        const float* gsrc = p.gx ? p.gx + (int64_t)((p.gx_idx && evalid) ? p.gx_idx[eb] : eb) * p.ldgx : p.bias;
#pragma unroll
        for (int g = 0; g < 4; ++g) gxv[g] = *((evalid && gsrc) ? gsrc + (int64_t)g * H + un : g_zero4);
        cpv = *((evalid && p.c_prev) ? p.c_prev + (int64_t)eb * p.ldc + un : g_zero4);
        if (p.gx_tab) {
            const int64_t tok = evalid ? gemv_token(p, eb) : 0;
            const float* trow = p.gx_tab + tok * p.ldtab;
#pragma unroll
            for (int g = 0; g < 4; ++g) gtv[g] = *(evalid ? trow + (int64_t)g * H + un : g_zero4);
        }
    }
    __syncthreads();
    float acc[4][NB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[g][b] = 0.f;
    if (p.h_prev) gemv_segment<NB>(acc, hs, hld, p.w_hh, p.ldw, H, un, H, lane);
    if (p.x2) gemv_segment<NB>(acc, xs, xld, p.w2, p.ldw2, H, un, K2, lane);
    // ---- wavefront reduction of the 4 NB partial sums (value index 4 b + g): lane l ends with the total of index l >> SH; lane b
    // then fetches row b's four gate totals from the lanes (4 b + g) << SH
    float pre[4];
    {
        constexpr int M = 4 * NB;
        constexpr int SH = (M == 4) ? 4 : (M == 8) ? 3 : (M == 16) ? 2 : 1;
        float v[M];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[4 * b + g] = acc[g][b];
        const float tot = wave_reduce_scatter<M>(v, lane);
        const int src = (lane < NB ? lane : 0) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = __shfl(tot, (src + g) << SH);
    }
    if (!evalid) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = pre[g] + gxv[g] + gtv[g];
    // cell (lstm.hip::step_cell_outputs: one expression, one rounding sequence)
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
    const float c = fg * cpv + ig * gg;
    const float h = og * tanhf_(c);
    p.h_out[(int64_t)eb * p.ldho + unit] = h;
    if (p.h_out2) p.h_out2[(int64_t)eb * p.ldho2 + unit] = h;
    p.c_out[(int64_t)eb * p.ldco + unit] = c;
    if (p.stash) {
        float* st = p.stash + (int64_t)eb * p.ldst + unit;
        st[0] = ig; st[(int64_t)H] = fg; st[(int64_t)2 * H] = gg; st[(int64_t)3 * H] = og;
    }
    if (p.h_planes) {       // blocked plane layout (split.hip), as lstm.hip writes it
        unsigned short pl3[3];
        split3_bits(h, pl3);
        unsigned short* q = p.h_planes + (int64_t)(eb >> 6) * (64 * p.ldhp) + (int64_t)(unit >> 4) * 3072 + ((unit >> 3) & 1) * 512 +
                            (eb & 63) * 8 + (unit & 7);
        q[0] = pl3[0]; q[1024] = pl3[1]; q[2048] = pl3[2];
    }
}

// the shapes this kernel takes (everything else: the MFMA tile kernel of lstm.hip)
bool lstm_step_fwd_gemv_ok(const StepFwdArgs& a) {
    auto al = [](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; };
    return a.B >= 1 && a.B <= 8 && !a.z_out && a.H % 4 == 0 && a.H <= GV_MAXK && (!a.h_prev || (al(a.h_prev, a.ldh) && al(a.w_hh, a.ldw))) &&
           (!a.x2 || (a.K2 % 4 == 0 && a.K2 <= GV_MAXK && al(a.x2, a.ldx2) && al(a.w2, a.ldw2)));
}

int lstm_step_fwd_gemv(hipStream_t stream, const StepFwdArgs& a) {
    S2VT_REQUIRE(lstm_step_fwd_gemv_ok(a) && a.h_out && a.c_out && (a.gx || a.bias), "lstm_step_fwd_gemv: unsupported shape / arguments");
    S2VT_REQUIRE(!(a.x2 || a.gx_tab) || a.tok_limit > 0, "lstm_step_fwd_gemv: a token segment needs tok_limit (rows of the table)");
    const int nb = a.B <= 1 ? 1 : a.B <= 2 ? 2 : a.B <= 4 ? 4 : 8;
    const size_t lds = (size_t)nb * (((a.H + 3) & ~3) + (a.x2 ? ((a.K2 + 3) & ~3) : 0)) * sizeof(float);
    const dim3 grid((unsigned)cdiv(a.H, GV_UNITS)), block(64 * GV_UNITS);
    static size_t lds_allowed[9] = {};              // per instantiation (NB): the largest dynamic-LDS size requested so far
    auto launch = [&](auto kern) -> int {
        if (lds > 48 * 1024 && lds > lds_allowed[nb]) {
            S2VT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_allowed[nb] = lds;
        }
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
        return 0;
    };
    int rc;
    switch (nb) {
        case 1: rc = launch(lstm_step_fwd_gemv_kernel<1>); break;
        case 2: rc = launch(lstm_step_fwd_gemv_kernel<2>); break;
        case 4: rc = launch(lstm_step_fwd_gemv_kernel<4>); break;
        default: rc = launch(lstm_step_fwd_gemv_kernel<8>); break;
    }
    if (rc) return rc;
    S2VT_LAUNCH_CHECK("lstm_step_fwd_gemv_kernel");
    return 0;
}

}  // namespace s2vt
