"""What a long-lived foreign kernel on some compute units costs the train step (a stand-in for RCCL's all-reduce kernels of a
data-parallel run, which cannot be run here with more than one rank): per step, a kernel of C one-wave workgroups that each hold
20 KB of LDS (a persistent GEMM workgroup - 144 KB - cannot share their unit, a timestep kernel - 70 KB - can) is started on
a side stream right before loss.backward() and spins for DUR microseconds.  Prints the step time without and with it.
usage: [S2VT_CU_RESERVE=n] python tools/bench_shared_device.py [C] [DUR_us] [B] [gemm_mode] [after]   (GPU box)
after = 1: the foreign kernel starts where a data-parallel step's first collective would - when the library releases gradient group 0
(s2vt_backward_wait_grads(0): behind the last persistent BPTT launch) - instead of at the start of the backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import S2VTModel, utils
from s2vt_video_caption_amd import capi, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
DUR = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 3
AFTER = (int(sys.argv[5]) if len(sys.argv) > 5 else 0) != 0
lib = capi.load()
lib.s2vt_set_gemm_mode(mode)
d = synth.CONFIGS["c2"]
dev = "cuda:0"
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
m.load_state_dict(synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=0)); m.to(dev).train()
feats, caps, mask = (t.to(dev) for t in synth.make_batch(B, d["L"], d["F"], d["V"], seed=1))
crit = utils.MaskCriterion(); opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
side = torch.cuda.Stream()
def run(n, occupy):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
        if occupy and not AFTER:
            side.wait_stream(torch.cuda.current_stream())
            capi.check(lib.s2vt_test_occupy_cus(C, 20 * 1024, DUR, side.cuda_stream), "occupy")
        loss.backward()
        if occupy and AFTER:
            capi.check(lib.s2vt_backward_wait_grads(0, capi.c_void_p(side.cuda_stream)), "s2vt_backward_wait_grads")
            capi.check(lib.s2vt_test_occupy_cus(C, 20 * 1024, DUR, side.cuda_stream), "occupy")
        if occupy:
            torch.cuda.current_stream().wait_stream(side)
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(8, False)
a = run(30, False); b = run(30, True); a2 = run(30, False); b2 = run(30, True)
capi.check_async_error()
print("B=%d mode %d, S2VT_CU_RESERVE=%s: step %.2f / %.2f ms alone; %.2f / %.2f ms with %d compute units held for %d us from %s" %
      (B, mode, os.environ.get("S2VT_CU_RESERVE", "0"), a, a2, b, b2, C, DUR,
       "the release of gradient group 0 (behind the last persistent BPTT launch)" if AFTER else "the start of the backward"))
