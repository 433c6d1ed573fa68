"""debug driver for the plain-bf16 LDS-DMA GEMM: growing shapes, result check after each (prints before each launch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s2vt_video_caption_amd import capi, ops
capi.load()
dev = "cuda:0"
shapes = [(256, 256, 128), (256, 256, 64), (200, 300, 104), (512, 768, 1024), (5056, 12000, 1000), (4000, 1000, 10112)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for M, N, K in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    pa, pb = ops.split_planes(a, nplanes=1), ops.split_planes(b, nplanes=1)
    torch.cuda.synchronize()
    print("launch", M, N, K, flush=True)
    ws = torch.empty(64 << 20, device=dev)
    c = ops.gemm_planes(pa, pb, M, N, nplanes=1, splitk_ws=ws)
    torch.cuda.synchronize()
    ref = a.bfloat16().float() @ b.bfloat16().float().t()
    err = (c - ref).abs().max().item() / ref.abs().max().item()
    print("  done, rel err vs bf16-rounded operands in fp32: %.2e" % err, flush=True)
