"""Run the split-precision GEMM a few times on one shape (for rocprofv3 --pmc).  usage: prof_gemm_x3.py [M N K]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s2vt_video_caption_amd import capi
lib = capi.load()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (5056, 12000, 1024)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev = "cuda:0"
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05
pa = torch.empty((M + 63) // 64 * 64, 3 * K, dtype=torch.int16, device=dev); pb = torch.empty((N + 63) // 64 * 64, 3 * K, dtype=torch.int16, device=dev)
c = torch.empty(M, N, device=dev)
st = vp(torch.cuda.current_stream().cuda_stream)
lib.s2vt_split_planes(3, 0, vp(a.data_ptr()), i64(K), M, K, vp(pa.data_ptr()), i64(3 * K), K, M, st)
lib.s2vt_split_planes(3, 0, vp(b.data_ptr()), i64(K), N, K, vp(pb.data_ptr()), i64(3 * K), K, N, st)
for _ in range(3):
    lib.s2vt_gemm_bf16_nt(3, M, N, K, vp(pa.data_ptr()), i64(3 * K), vp(pb.data_ptr()), i64(3 * K), vp(c.data_ptr()), i64(N),
                          vp(0), 0, vp(0), ctypes.c_size_t(0), st)
torch.cuda.synchronize()
