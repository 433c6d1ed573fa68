// Experiment: can a LOW-FOOTPRINT timestep-like kernel (4 waves, <= 80 VGPRs, 12 KB LDS, operands straight from global
// memory into MFMA fragments) run INSIDE the shadow of the split-precision GEMM (one 8-wave workgroup per CU, 144 KB LDS,
// ~215 VGPRs), which excludes the shipped timestep kernels (73 KB LDS) from its CUs?  tools/bench_coresident.py builds this
// file as a small shared library and times the chain alone, the GEMM alone and both on two streams.
// The kernel does the contraction of lstm_step_fwd_kernel (32 x 32 tile of h W_hh^T over K, v_mfma_f32_16x16x4_f32) with its
// 4 waves splitting K, a 3-tile LDS reduction and a 1-KB store per workgroup; no cell math.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 6) void light_step_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                            float* __restrict__ out, int B, int H, int K) {
    __shared__ float red[3][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nx = (4 * H) / 32;
    const int tx = blockIdx.x % nx, ty = blockIdx.x / nx;
    const int fi = lane & 15, fq = lane >> 4;
    const float* a0 = h + (int64_t)(ty * 32 + fi) * K + 4 * fq;          // row tile 0; tile 1 is 16 rows below
    const float* b0 = w + (int64_t)(tx * 32 + fi) * K + 4 * fq;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kq = K / 4;                                                 // this wave's quarter of K, in k16 blocks of 16
    const int kb = wave * kq, ke = kb + kq;
    f32x4 a[2][2], b[2][2];                                               // [buffer][tile]
    auto load = [&](int buf, int k) {
        const bool ok = k + 15 < ke + 0 && k < K;
        const int kk = ok ? k : kb;
        a[buf][0] = *reinterpret_cast<const f32x4*>(a0 + kk);
        a[buf][1] = *reinterpret_cast<const f32x4*>(a0 + 16 * (int64_t)K + kk);
        b[buf][0] = *reinterpret_cast<const f32x4*>(b0 + kk);
        b[buf][1] = *reinterpret_cast<const f32x4*>(b0 + 16 * (int64_t)K + kk);
    };
    load(0, kb);
    load(1, kb + 16);
    for (int k = kb; k + 15 < ke; k += 32) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 ca[2] = {a[u][0], a[u][1]}, cb[2] = {b[u][0], b[u][1]};
            load(u, k + 16 * u + 32);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[mi][j], cb[ni][j], acc[mi][ni], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][mi * 16 + 4 * fq + r][ni * 16 + fi] = acc[mi][ni][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mi * 16 + 4 * fq + r, col = ni * 16 + fi;
                    const float v = acc[mi][ni][r] + red[0][row][col] + red[1][row][col] + red[2][row][col];
                    out[(int64_t)(ty * 32 + row) * (4 * H) + tx * 32 + col] = 1.0f / (1.0f + __expf(-v));
                }
    }
}

extern "C" int light_step_launch(void* stream, const float* h, const float* w, float* out, int B, int H, int K) {
    const int grid = ((4 * H) / 32) * (B / 32);
    hipLaunchKernelGGL(light_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w, out, B, H, K);
    return (int)hipGetLastError();
}
