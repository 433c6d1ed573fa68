"""Reference point for the hand-written bf16 GEMM (gemm_b1.hip): the library GEMM torch dispatches to (hipBLASLt / rocBLAS)
on the same 14 shapes of a config-3 train step, bf16 operands.  NT form (both operands k-contiguous, what gemm_b1 reads)
and, for the weight-gradient shapes, the TN form a library takes straight from the row images (no transposed planes).
Not part of the product: a yardstick only.  usage: python tools/bench_blaslt_shapes.py [B]   (GPU box)"""
import sys
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, F, H, V, BLK = 80, 4096, 1000, 12000, 32
R = B * (L - 1)
T = 2 * L - 1
SHAPES = [("x1 = feats Wf^T", B * L, H, F), ("gx1 = x1 Wih1^T", B * L, 4 * H, H), ("gx2 block = h1 Wv^T", BLK * B, 4 * H, H),
          ("gxe = emb We^T", R, 4 * H, H), ("logits = h2 Wo^T", R, V, H), ("dh2 = dlogits Wo", R, H, V),
          ("dWo = dlogits^T h2", V, H, R), ("dh1 block = dG2 Wv", BLK * B, H, 4 * H), ("dWhh = dG^T h", 4 * H, H, T * B),
          ("dWe = dG2^T emb", 4 * H, H, R), ("demb = dG2 We", R, H, 4 * H), ("dx1 = dG1 Wih1", B * L, H, 4 * H),
          ("dWih1 = dG1^T x1", 4 * H, H, B * L), ("dWf = dx1^T feats", H, F, B * L)]
dev = "cuda:0"


def timed(fn):
    best = 1e9
    for it in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        if it:
            best = min(best, e0.elapsed_time(e1))
    return best


tot = 0.0
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    bt = b.t()
    t_nt = timed(lambda: torch.mm(a, bt))                      # C = A B^T, both k-contiguous; bf16 output
    try:
        t_f32 = timed(lambda: torch.mm(a, bt, out_dtype=torch.float32))
    except Exception:
        t_f32 = float("nan")
    at = torch.randn(K, M, device=dev).bfloat16()              # k-outer images (what a weight-gradient GEMM has)
    bk = torch.randn(K, N, device=dev).bfloat16()
    t_tn = timed(lambda: torch.mm(at.t(), bk))
    tot += t_nt
    print("%-22s M=%6d N=%6d K=%6d  NT %7.1f us %7.1f TF | NT fp32-out %7.1f us | TN (k-outer operands) %7.1f us %7.1f TF" %
          (name, M, N, K, t_nt * 1e3, 2.0 * M * N * K / t_nt / 1e9, t_f32 * 1e3, t_tn * 1e3, 2.0 * M * N * K / t_tn / 1e9), flush=True)
print("sum NT: %.3f ms" % tot)
