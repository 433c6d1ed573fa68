"""MI355X-native (gfx950) implementation of the S2VT hot path.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C-ABI
library ``libs2vt_hip.so``), ``capi`` (ctypes binding of ``include/s2vt_hip.h``),
``functional`` (autograd glue used by the drop-in ``S2VTModel.S2VT``),
``synth`` (synthetic weights / inputs recipe shared by tests and bench) and
``dp`` (one-process-per-GPU data parallel step over RCCL).
"""
__version__ = "0.1.0"
