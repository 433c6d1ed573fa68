"""The reference's optimizer (train.py:89-93: torch.optim.Adam with default betas / eps) as ONE kernel launch per step.

`FlatAdam(model, lr)` moves the parameters of the HIP S2VT module into one flat fp32 buffer (every `param.data` becomes a
view into it: names, shapes, state_dict and pickles are unchanged), keeps the gradients in a second flat buffer that the backward
WRITES into directly (`functional.set_grad_sink`, the mechanism of `dp.FlatGradAllReducer.attach`; `param.grad` are views of it,
so hooks / inspection still see them) and the two moments in two more; `step()` is `s2vt_adam_step` - 28 bytes per parameter
in one launch instead of torch's multi-tensor launches.  Same arithmetic as `torch.optim.Adam` operation for operation
(`tests/test_gpu_kernels.py::test_flat_adam_step_is_torch_adam`, `::test_flat_adam_trains_the_model_as_torch_adam_does`); `lr` may be changed through `param_groups[0]["lr"]`, so
`ReduceLROnPlateau` works on it unchanged.
"""
import ctypes

import torch

from . import capi, functional


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, reducer=None):
        hip = list(model._hip_params())
        if any(p.dtype != torch.float32 or not p.is_cuda for p in hip):
            raise capi.S2VTHipError("FlatAdam needs the fp32 HIP parameters of an S2VT module")
        if reducer is not None:                    # data-parallel: the all-reduce buffer IS the gradient buffer, in its parameter order
            if reducer.model is not model or not reducer.flat.is_cuda:
                raise capi.S2VTHipError("FlatAdam(reducer=...): attach the reducer to this model first (FlatGradAllReducer.attach)")
            params, slices = list(reducer.params), [tuple(s) for s in reducer.slices]
        else:
            params, slices, off = hip, [], 0
            for p in params:
                slices.append((off, off + p.numel()))
                off += p.numel()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        n = slices[-1][1]
        dev = params[0].device
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, (lo, hi) in zip(params, slices):
                self.flat_p[lo:hi].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[lo:hi].view_as(p)
        if reducer is not None:
            self.flat_g = reducer.flat
        else:
            self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
            for p, (lo, hi) in zip(params, slices):
                p.grad = self.flat_g[lo:hi].view_as(p)
            functional.set_grad_sink(model, [p.grad for p in hip])
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.n = n
        self.steps = 0
        self._lib = capi.load()

    def zero_grad(self, set_to_none=False):
        pass                                       # the backward overwrites every gradient in the flat buffer

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self.steps += 1
        dev = self.flat_p.device
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            capi.check(self._lib.s2vt_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.n, float(g["lr"]),
                                                float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self.steps,
                                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "s2vt_adam_step")
        # the kernel wrote through raw pointers: bump the version counters, as an in-place torch op would have (autograd's saved-tensor
        # check and the decode-image cache of functional.py key on them)
        torch.autograd.graph.increment_version(g["params"])

    # (the reference checkpoints the module only, train.py:160-175; a resumable optimizer costs two methods)
    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.flat_m.detach().clone(), "exp_avg_sq": self.flat_v.detach().clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    @torch.no_grad()
    def load_state_dict(self, state):
        if state["exp_avg"].numel() != self.n or state["exp_avg_sq"].numel() != self.n:
            raise capi.S2VTHipError("FlatAdam.load_state_dict: the state belongs to a model of another size")
        self.steps = int(state["steps"])
        self.flat_m.copy_(state["exp_avg"])
        self.flat_v.copy_(state["exp_avg_sq"])
        for k, v in state["param_groups"][0].items():
            self.param_groups[0][k] = v
