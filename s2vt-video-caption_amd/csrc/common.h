// Shared device/host helpers for the gfx950 (CDNA4, wave64) S2VT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace s2vt {

// ---- error plumbing (no C++ exceptions cross the C ABI) -------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define S2VT_HIP(call)                                        \
    do {                                                      \
        int _rc = ::s2vt::check_hip((call), #call);           \
        if (_rc) return _rc;                                  \
    } while (0)

#define S2VT_LAUNCH_CHECK(name)                               \
    do {                                                      \
        int _rc = ::s2vt::check_hip(hipGetLastError(), name); \
        if (_rc) return _rc;                                  \
    } while (0)

#define S2VT_REQUIRE(cond, ...)                               \
    do {                                                      \
        if (!(cond)) {                                        \
            ::s2vt::set_error(__VA_ARGS__);                   \
            return -1;                                        \
        }                                                     \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- run-time options (options.hip): one table; S2VT_<NAME> from the environment at first read, s2vt_set_option afterwards
enum Option { O_GEMM_MODE, O_PERSIST, O_PERSIST_X3_FWD, O_PERSIST_X3_BWD, O_PIPE_BLOCK, O_GRAPH, O_DECODE_FUSED, O_CU_RESERVE, O_BPTT_UNITS, O_GEMV, O_PAD_MIN_BATCH, O_CORUN, O_BPTT_SOLO, O_COUNT };
int option(int id);
int option_set(int id, int value);      // returns the previous value; value < 0 only queries
// compute units the persistent GEMMs size their grids for: the device's, minus option "cu_reserve", rounded down to the 8 XCDs
int planned_compute_units();
// cap of the calling thread's persistent-GEMM grids (0: none): a driver that launches a GEMM BESIDE a persistent recurrence kernel
// which holds only part of the compute units plans it for the units left over.  Returns the previous cap.
int cu_plan_cap(int n);
int cu_plan_cap_current();
// option cu_reserve applies while a collective may be running beside the calling thread's launches: inside s2vt_train_backward from
// the release of gradient group 0 (s2vt_backward_wait_grads) to the end of the backward.  Returns the previous state.
bool cu_reserve_window(bool on);
struct CuPlanCap {
    int prev;
    explicit CuPlanCap(int n) : prev(cu_plan_cap(n)) {}
    ~CuPlanCap() { cu_plan_cap(prev); }
};

// Workgroups of `kernel` (block size `block`, static LDS only) that can be RESIDENT AT ONCE on the current device: compute
// units x occupancy per compute unit, as the runtime reports them (cached per device and kernel; 0 when the query fails).
// The persistent recurrence kernels wait for each other inside a launch, so a launch may never be larger than this.
int coresident_capacity(const void* kernel, int block);
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Row map of a stored matrix: stored_row = idx ? idx[r] : (inner ? (r % inner) * outer + r / inner : r).
// `inner/outer` converts between batch-major (b*L + l) and time-major (l*B + b) row orders.
struct RowMap {
    const int32_t* idx;  // optional gather index
    int inner, outer;    // optional permutation (inner == 0: identity)
};

__device__ __forceinline__ int map_row(const RowMap& m, int r) {
    if (m.idx) return m.idx[r];
    if (m.inner) return (r % m.inner) * m.outer + r / m.inner;
    return r;
}

// Permutation-only row map (no gather): pure integer arithmetic, usable inside a software-pipelined loop without
// introducing a dependent load (a gather index would make every staging load wait for the index load, and with
// the in-order vmcnt counter, for every older prefetch as well).
__device__ __forceinline__ int map_row_perm(const RowMap& m, int r) {
    const int inner = m.inner ? m.inner : 1;
    const int pr = (r % inner) * m.outer + r / inner;
    return m.inner ? pr : r;
}

// 16 bytes of zeros in device memory: the target of redirected out-of-range loads (see load4_guard).
static __device__ __attribute__((aligned(16))) float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};

// Cell activations of the fp32 timestep kernels: hardware exp (v_exp_f32, ~1 ulp) + IEEE division; absolute error ~1.5e-7
// for both (tanh as 1 - 2 / (1 + e^{2x}): exact limits +-1, no cancellation blow-up in absolute terms).  libm's
// expf / tanhf cost ~3x the instructions in the epilogue of a latency-bound kernel for the same fp32-level accuracy.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f / (1.0f + __expf(2.0f * x)); }

// fp32 -> bf16 (round to nearest even; NaN stays NaN) and the three-plane split x = p0 + p1 + p2 (24 mantissa bits) of the
// split-precision operands (split.hip, gemm_x3.hip)
__device__ __forceinline__ unsigned short bf16_rn_bits(float x) {
    unsigned int u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void split3_bits(float x, unsigned short (&o)[3]) {
    o[0] = bf16_rn_bits(x);
    const float r1 = x - __uint_as_float((unsigned int)o[0] << 16);          // exact
    o[1] = bf16_rn_bits(r1);
    const float r2 = r1 - __uint_as_float((unsigned int)o[1] << 16);         // exact
    o[2] = bf16_rn_bits(r2);
}

}  // namespace s2vt
