// fp32-input MFMA issue rate on gfx950: v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, by number of independent
// accumulators and waves per SIMD (random-ish operands in registers, no memory traffic).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_rate.hip -o tools/micro/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.f) out[0] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.f) out[0] = s;
}
template <typename K>
static void run(const char* name, K kern, int nacc, int threads, double flop_per_mfma) {
    float* d; hipMalloc(&d, 64);
    const int iters = 2000, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * 8 * nacc * (threads / 64) * grid;
    printf("%-40s acc=%d waves/CU=%d: %.1f TFLOP/s, %.1f ns per MFMA per wave\n", name, nacc, threads / 64,
           n * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 8 * nacc));
    hipFree(d);
}
int main() {
    run("v_mfma_f32_16x16x4_f32", k16<1>, 1, 256, 2048.); run("v_mfma_f32_16x16x4_f32", k16<2>, 2, 256, 2048.);
    run("v_mfma_f32_16x16x4_f32", k16<3>, 3, 256, 2048.); run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 256, 2048.);
    run("v_mfma_f32_16x16x4_f32", k16<1>, 1, 512, 2048.); run("v_mfma_f32_16x16x4_f32", k16<3>, 3, 512, 2048.);
    run("v_mfma_f32_32x32x2_f32", k32<1>, 1, 256, 4096.); run("v_mfma_f32_32x32x2_f32", k32<2>, 2, 256, 4096.);
    run("v_mfma_f32_32x32x2_f32", k32<1>, 1, 512, 4096.); run("v_mfma_f32_32x32x2_f32", k32<2>, 2, 512, 4096.);
    return 0;
}
