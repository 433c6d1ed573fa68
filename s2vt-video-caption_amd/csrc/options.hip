// Run-time options of libs2vt_hip.so: ONE table.  Every switch has a name, a default and a range; its first read takes
// S2VT_<NAME> from the environment (if set and in range), s2vt_set_option(name, value) changes it at any time (a negative
// value only queries).  The typed setters of include/s2vt_hip.h (s2vt_set_gemm_mode, s2vt_set_recurrence_mode,
// s2vt_set_pipeline_block, s2vt_set_graph_mode, s2vt_set_decode_schedule) are views of the same table.
#include <ctype.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/s2vt_hip.h"
#include "common.h"

namespace s2vt {

struct OptionDef { const char* name; int def, lo, hi; };
static const OptionDef kOptions[O_COUNT] = {
    /* O_GEMM_MODE      */ {"gemm_mode", 3, 0, 3},          // 3 split precision (fp32-equivalent), 1 bf16 operands, 0 exact-fp32 MFMA
    /* O_PERSIST        */ {"persist", 1, 0, 1},            // persistent recurrence kernels (0: launches per timestep everywhere)
    /* O_PERSIST_X3_FWD */ {"persist_x3_fwd", 1, 0, 1},     // split-precision persistent forward (gemm mode 3)
    /* O_PERSIST_X3_BWD */ {"persist_x3_bwd", 2, 0, 2},     // split-precision persistent BPTT: 0 off, 1 on, 2 where one chain per workgroup fits (B = 64)
    /* O_PIPE_BLOCK     */ {"pipe_block", 32, 0, 4096},     // timesteps per pipeline block (0: one stream, no layer pipeline)
    /* O_GRAPH          */ {"graph", 0, 0, 1},              // hipGraph replay of the train forward / backward launch sequences
    /* O_DECODE_FUSED   */ {"decode_fused", 1, 0, 1},       // greedy decode: recurrent GEMM inside the argmax launch
    /* O_CU_RESERVE     */ {"cu_reserve", 0, 0, 128},       // persistent GEMMs of the backward plan for this many compute units fewer once gradient group 0 is released
    /* O_BPTT_UNITS     */ {"bptt_units", 0, 0, 32},        // persistent bf16 BPTT: 32 | 16 hidden units per workgroup (0: 32 where it fits)
    /* O_GEMV           */ {"gemv", 1, 0, 2},               // launch-per-timestep forward step as gate GEMVs (lstm_gemv.hip): 1 at B <= 4, 2 at B <= 8, 0 never
    /* O_PAD_MIN_BATCH  */ {"pad_min_batch", 33, 1, 64},    // ragged batches of at least this size are padded to a multiple of 64 (plane path)
    /* O_CORUN          */ {"corun", 3, 0, 5},              // tenths of a GEMM that nothing waits for beside EACH one-layer stage of the persistent split-precision schedules (0: off)
    /* O_BPTT_SOLO      */ {"bptt_solo", 1, 0, 1},          // co-run schedule of the BPTT: every persistent launch carries ONE layer, GEMMs on the other half of the device
};
static int g_value[O_COUNT];
static bool g_read[O_COUNT];

static int clamp_to(const OptionDef& d, int v) { return v < d.lo ? d.lo : (v > d.hi ? d.hi : v); }

int option(int id) {
    if (id < 0 || id >= O_COUNT) return 0;
    if (!g_read[id]) {
        const OptionDef& d = kOptions[id];
        char env[64] = "S2VT_";
        size_t n = strlen(env);
        for (const char* c = d.name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = 0;
        const char* e = getenv(env);
        int v = d.def;
        if (e && *e) {
            v = atoi(e);
            if (id == O_GEMM_MODE && v != 1 && v != 3) v = 0;
            v = clamp_to(d, v);
        }
        g_value[id] = v;
        g_read[id] = true;
    }
    return g_value[id];
}

int option_set(int id, int value) {
    const int prev = option(id);
    if (id >= 0 && id < O_COUNT && value >= 0) {
        if (id == O_GEMM_MODE && value != 1 && value != 3) value = 0;
        g_value[id] = clamp_to(kOptions[id], value);
    }
    return prev;
}

static thread_local int t_cu_cap = 0;
static thread_local bool t_reserve_window = false;

bool cu_reserve_window(bool on) {
    const bool prev = t_reserve_window;
    t_reserve_window = on;
    return prev;
}

int planned_compute_units() {
    static int device_cus = 0;
    if (!device_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        (void)hipGetLastError();
        device_cus = n;
    }
    int n = device_cus;
    const int reserve = t_reserve_window ? option(O_CU_RESERVE) : 0;
    if (reserve > 0 && reserve < n - 8) n -= reserve;
    if (t_cu_cap >= 8 && t_cu_cap < n) n = t_cu_cap;
    return n / 8 * 8;
}

int cu_plan_cap_current() { return t_cu_cap; }

int cu_plan_cap(int n) {
    const int prev = t_cu_cap;
    t_cu_cap = n > 0 ? n : 0;
    return prev;
}

int option_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < O_COUNT; ++i)
        if (strcmp(name, kOptions[i].name) == 0) return i;
    return -1;
}

}  // namespace s2vt

extern "C" {

int32_t s2vt_set_option(const char* name, int32_t value) {
    const int id = s2vt::option_id(name);
    if (id < 0) {
        s2vt::set_error("s2vt_set_option: unknown option '%s'", name ? name : "(null)");
        return INT_MIN;
    }
    return s2vt::option_set(id, value);
}

int32_t s2vt_option_count(void) { return s2vt::O_COUNT; }
const char* s2vt_option_name(int32_t index) { return (index >= 0 && index < s2vt::O_COUNT) ? s2vt::kOptions[index].name : nullptr; }

}
