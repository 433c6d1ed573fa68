// Timing experiments only (tools/bench_persist_stamps.py builds a separate library with -DS2VT_EXPERIMENT_STAMPS;
// in the product build every macro below is empty and no stamp code exists).
#pragma once
#ifdef S2VT_EXPERIMENT_STAMPS
#define XSTAMP_SLOTS 16
// stamp i of record `rec` <- 100-MHz wall clock (one lane); the buffer is read by nothing else in the kernel
#define XSTAMP(buf, rec, i)                                                                        \
    do {                                                                                           \
        if ((buf) != nullptr && threadIdx.x == 0 && (rec) >= 0 && (rec) < 4096)                    \
            (buf)[(rec) * XSTAMP_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime();                  \
    } while (0)
#else
#define XSTAMP(buf, rec, i) do { } while (0)
#endif
