"""sha1 of every kernel source under s2vt-video-caption_amd/csrc: written into the PMC summaries (tools/pmc_busy.py,
tools/pmc_traffic.py) when they are collected, compared by bench.py when it quotes them - a counter file measured before the last
change of a kernel's source is dropped from the bench line instead of being quoted as if it were current."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "s2vt-video-caption_amd", "csrc")

# kernel family -> the sources whose change makes its counters stale (common.h: cell math, row maps, load helpers)
KERNEL_SOURCES = {
    "gemm_x3_kernel": ["gemm_x3.hip"], "gemm_b1_kernel": ["gemm_b1.hip"], "gemm_f32_kernel": ["gemm.hip"],
    "lstm_seq_fwd_x3_persist_kernel": ["lstm_persist_x3.hip"], "lstm_seq_bwd_x3_persist_kernel": ["lstm_persist_x3.hip"],
    "lstm_seq_fwd_bf16_persist_kernel": ["lstm_persist.hip"], "lstm_seq_bwd_bf16_persist_kernel": ["lstm_persist.hip"],
    "lstm_step_fwd_kernel": ["lstm.hip"], "lstm_step_bwd_kernel": ["lstm.hip"],
    "lstm_step_fwd_bf16_kernel": ["lstm_bf16.hip"], "lstm_step_bwd_bf16_kernel": ["lstm_bf16.hip"],
    "logits_argmax_x3_kernel": ["argmax_x3.hip"], "logits_argmax_kernel": ["lstm.hip"],
    "split_dual_kernel": ["split.hip"], "split3_rows_kernel": ["split.hip"],
}


def digests():
    out = {}
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        out[os.path.basename(f)] = hashlib.sha1(open(f, "rb").read()).hexdigest()[:12]
    return out


def stale_sources(recorded, kernel):
    """names of the sources of `kernel` that changed since `recorded` (the digests a PMC file carries) was written; None if the file
    carries no digests (collected before round 5)"""
    if not recorded:
        return None
    now = digests()
    fam = kernel.split("<")[0]
    return [f for f in KERNEL_SOURCES.get(fam, []) + ["common.h"] if recorded.get(f) != now.get(f)]
