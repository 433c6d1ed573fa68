// What does a dependent same-stream kernel boundary cost on this system, and which launch property moves it?
// (reconcile the 3.9 us exit->entry gap of the timestep kernel with the guide's 1.45-1.9 us "boundary" row.)
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/launch_boundary.hip -o /tmp/launch_boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Big { float v[96]; };       // 384-byte kernarg

__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_bigarg(Big a, Big b, float* p) { if (p && threadIdx.x == 9999) p[0] = a.v[0] + b.v[1]; }
template <int LDSB>
__global__ void k_lds(float* p) {
    __shared__ float s[LDSB / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (p && threadIdx.x == 9999) p[0] = s[5];
}
// a kernel that does ~T us of work per workgroup (spin on the 100-MHz clock)
__global__ void k_work(float* p, unsigned ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (p && threadIdx.x == 9999) p[0] = 1.f;
}

template <typename F>
static double chain(hipStream_t st, int n, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) launch();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / n;
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    float* d = nullptr;
    hipMalloc(&d, 1 << 20);
    const int N = 3000;
    Big A = {}, Bq = {};
    printf("empty  grid 256 x 64 : %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(64), 0, st, d); }));
    printf("empty  grid 256 x 512: %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, d); }));
    printf("empty  grid 1024 x 256: %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, d); }));
    printf("bigarg grid 256 x 512 (784-B kernarg): %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL(k_bigarg, dim3(256), dim3(512), 0, st, A, Bq, d); }));
    printf("lds 8 KB  grid 256 x 512: %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL((k_lds<8192>), dim3(256), dim3(512), 0, st, d); }));
    printf("lds 70 KB grid 256 x 512: %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL((k_lds<71680>), dim3(256), dim3(512), 0, st, d); }));
    printf("lds 70 KB grid 512 x 512 (2 per CU): %.2f us/launch\n", chain(st, N, [&] { hipLaunchKernelGGL((k_lds<71680>), dim3(512), dim3(512), 0, st, d); }));
    for (unsigned ticks : {100u, 400u, 800u}) {
        double t = chain(st, N, [&] { hipLaunchKernelGGL(k_work, dim3(256), dim3(512), 0, st, d, ticks); });
        printf("work %.1f us grid 256 x 512: %.2f us/launch -> boundary %.2f us\n", ticks * 0.01, t, t - ticks * 0.01);
    }
    // the same through a captured graph (one graph = 200 launches)
    {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_work, dim3(256), dim3(512), 0, st, d, 400u);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        double t = chain(st, 15, [&] { hipGraphLaunch(ge, st); }) / 200.0;
        printf("work 4.0 us grid 256 x 512, hipGraph of 200: %.2f us/launch -> boundary %.2f us\n", t, t - 4.0);
    }
    // two streams interleaved (the layer pipeline): per-stream chain of 4-us kernels, both streams busy
    {
        hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, st);
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(128), dim3(512), 0, st, d, 400u);
            hipLaunchKernelGGL(k_work, dim3(128), dim3(512), 0, s2, d + 64, 400u);
        }
        hipEventRecord(b, st);
        hipEventSynchronize(b); hipStreamSynchronize(s2);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("two streams, 4.0-us kernels of 128 WGs each: %.2f us per launch per stream -> boundary %.2f us\n", ms * 1e3 / N, ms * 1e3 / N - 4.0);
    }
    return 0;
}
