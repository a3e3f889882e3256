"""Fused optimizer step for the hot path's parameters -- host-side mirror of

    optimizer   = torch.optim.AdamW(param_groups, lr, betas)                      Painter/main_train.py:344-348
    loss_scaler = NativeScalerWithGradNormCount()                                  Painter/util/misc.py:252-285
    grad_norm   = loss_scaler(loss, optimizer, clip_grad=3.0, parameters=..., update_grad=...)   Painter/engine_train.py:85-88

`AdamW` is a torch.optim.Optimizer (same constructor, `param_groups` -- so `util/lr_sched.adjust_learning_rate` and the
`lr_scale` groups of `util/lr_decay.param_groups_lrd` work unchanged -- and the same `state_dict()` layout: `step`,
`exp_avg`, `exp_avg_sq`, so `misc.save_model / load_model` checkpoints are interchangeable with torch.optim.AdamW).
The arithmetic runs in libpainter_hip.so (csrc/optim.hip): one streaming pass for the gradient norm / finiteness, one for the
update; unscale, clip coefficient and the inf/nan skip are consumed on the device, nothing synchronises with the host.
It also honours torch.cuda.amp.GradScaler's fused-optimizer protocol (`_step_supports_amp_scaling`: `grad_scale`, `found_inf`),
so the unchanged reference scaler can drive it.  `NativeScalerWithGradNormCount` below is the two-pass replacement of the
reference class with the same call signature.

There is no CPU path: parameters must live on the MI355X (the CPU restatement used by the tests is oracle/optim_oracle.py).
"""
import ctypes

import numpy as np
import torch

from . import ops
from ._lib import check, lib

_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("w16", "<u8"), ("n", "<i8"), ("group", "<i4"), ("first_chunk", "<i4")])
MAX_GROUPS = 64


class _Groups(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_float * 64), ("wd", ctypes.c_float * 64), ("active", ctypes.c_int32 * 64)]


class AdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True        # torch GradScaler then hands grad_scale / found_inf over instead of unscaling itself
    # no class-level `grad_scale` / `found_inf`: torch's GradScaler.step() multiplies `getattr(optimizer, "grad_scale", 1)` into its scale and
    # deletes both attributes afterwards; step() reads them with getattr(..., None)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > MAX_GROUPS:
            raise NotImplementedError("at most %d parameter groups (the reference builds 52: util/lr_decay.py)" % MAX_GROUPS)
        b = {tuple(g["betas"]) for g in self.param_groups}
        e = {g["eps"] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise NotImplementedError("betas and eps must be shared by all groups (they are in the reference)")
        self._norm_info = None
        self._cached = None
        self._tab_dev = None
        self._steps = None                   # device float32 [64]: per-group step counts (a skipped step must not advance them)
        self._model = None

    def bind_model(self, model):
        """Let the update refresh the model's cached bf16 weight copies in the same pass (saves the per-step cast kernels).
        `model` is a painter_amd Painter / SegGPT module; optional."""
        self._model = model
        return self

    # ------------------------------------------------------------------ state / table
    def _steps_for(self, device):
        if self._steps is None or self._steps.device != device:
            self._steps = torch.zeros(MAX_GROUPS, dtype=torch.float32, device=device)
        return self._steps

    def _init_state(self, p, gi):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = self._steps_for(p.device)[gi]                   # 0-d device view (torch's capturable/fused layout)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def state_dict(self):
        """torch.optim layout.  `step` is stored per parameter as its own 0-d tensor: inside this class the steps of a group are
        views of one element of a shared device array, and torch.save would keep that aliasing -- a torch.optim.AdamW resuming
        from such a checkpoint would then advance the shared element once per parameter of the group."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone().reshape(())
        return sd

    def zero_grad(self, set_to_none=True):
        self._cached, self._norm_info = None, None            # the cached table points at the gradients being dropped
        return super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict):
        """Checkpoints written by torch.optim.AdamW or by this class: re-home the per-parameter `step` scalars in the per-group
        device array."""
        super().load_state_dict(state_dict)
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                st = self.state.get(p, {})
                if "step" in st:
                    steps = self._steps_for(p.device)
                    steps[gi] = float(st["step"])
                    st["step"] = steps[gi]
                    for k in ("exp_avg", "exp_avg_sq"):
                        st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()

    def _table(self):
        """Device table of (param, grad, exp_avg, exp_avg_sq) records in flat chunk order; rebuilt every step because the
        gradient tensors are fresh allocations (a 14 KB pinned -> device copy on the step's stream)."""
        chunk = int(lib.pa_opt_chunk_elems())
        recs, dev = [], None
        nchunks = 0
        self._fresh = []
        shadows = None
        if self._model is not None and getattr(self._model, "_hot", None) is not None:
            shadows = self._model._hot.shadow_buffers()
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("painter_amd.optim.AdamW: parameters must be contiguous fp32 tensors on the MI355X")
                dev = p.device
                st = self._init_state(p, gi)
                g = p.grad
                gptr = 0
                if g is not None:
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                        raise RuntimeError("painter_amd.optim.AdamW: gradients must be dense contiguous fp32")
                    gptr = g.data_ptr()
                n = p.numel()
                w16 = 0
                if shadows is not None and gptr:
                    buf = shadows.get(p.data_ptr())
                    if buf is not None and buf.numel() == n:
                        w16 = buf.data_ptr()
                        self._fresh.append(p)
                recs.append((p.data_ptr(), gptr, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), w16, n, gi, nchunks))
                nchunks += (n + chunk - 1) // chunk
        if not recs:
            return None, 0, 0, None
        arr = np.array(recs, dtype=_REC)
        nbytes = arr.nbytes
        if self._tab_dev is None or self._tab_dev.numel() < nbytes or self._tab_dev.device != dev:
            self._tab_dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # pageable source: the runtime stages the 14 KB itself, so the host array may be dropped right away; the copy is
        # ordered on the step's stream behind the previous step's kernels that still read the table
        self._tab_dev[:nbytes].copy_(torch.from_numpy(arr.view(np.uint8).reshape(-1)))
        return self._tab_dev, len(recs), nchunks, dev

    # ------------------------------------------------------------------ passes
    @torch.no_grad()
    def grad_sumsq(self):
        """-> device tensor [sum of squares of the gradients as stored (still loss-scaled), non-finite count]; kept for step()."""
        tab, nt, nchunks, dev = self._table()
        if tab is None:
            return None
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ws = ops.workspace(lib.pa_grad_sumsq_workspace_bytes(nchunks), dev, slot=2)
        check(lib.pa_grad_sumsq(tab.data_ptr(), nt, nchunks, out.data_ptr(), ws.data_ptr(), ops.stream()), "pa_grad_sumsq")
        self._norm_info = out
        self._cached = (tab, nt, nchunks, dev, self._grad_ptrs())
        return out

    def _grad_ptrs(self):
        return tuple(0 if p.grad is None else p.grad.data_ptr() for g in self.param_groups for p in g["params"] if p.requires_grad)

    @torch.no_grad()
    def step(self, closure=None, *, grad_scale=None, found_inf=None, max_norm=None):
        """AdamW update.  grad_scale / found_inf: device scalars (keyword or the attributes torch's GradScaler sets);
        max_norm: clip the global gradient norm (uses the result of grad_sumsq(), which is run first if needed)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale = grad_scale if grad_scale is not None else getattr(self, "grad_scale", None)
        found_inf = found_inf if found_inf is not None else getattr(self, "found_inf", None)
        norm_info = None
        if max_norm is not None and max_norm > 0:
            if self._norm_info is None:
                self.grad_sumsq()
            norm_info = self._norm_info
        if self._cached is not None and self._cached[4] != self._grad_ptrs():
            self._cached, self._norm_info, norm_info = None, None, None      # gradients were replaced since grad_sumsq(): start over
            if max_norm is not None and max_norm > 0:
                self.grad_sumsq()
                norm_info = self._norm_info
        tab, nt, nchunks, dev = self._cached[:4] if self._cached is not None else self._table()
        self._cached, self._norm_info = None, None
        if tab is None:
            return loss
        gs = _Groups()
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for gi, group in enumerate(self.param_groups):
            gs.lr[gi] = float(group["lr"])
            gs.wd[gi] = float(group["weight_decay"])
            gs.active[gi] = int(any(p.requires_grad and p.grad is not None for p in group["params"]))
        f32 = lambda x: 0 if x is None else x.data_ptr()
        as_dev = lambda x: None if x is None else (x if torch.is_tensor(x) else torch.tensor(float(x))).to(device=dev, dtype=torch.float32).reshape(-1)
        grad_scale, found_inf = as_dev(grad_scale), as_dev(found_inf)
        self._keep = (grad_scale, found_inf, norm_info)       # keep the scalars alive until the kernel has been enqueued
        check(lib.pa_adamw_step(tab.data_ptr(), nt, nchunks, ctypes.byref(gs), float(b1), float(b2), float(eps),
                                self._steps_for(dev).data_ptr(), f32(norm_info), f32(grad_scale), f32(found_inf),
                                float(max_norm) if max_norm else 0.0, ops.stream()), "pa_adamw_step")
        # the kernel rewrote the parameters behind autograd's back: bump their version counters so that everything keyed on
        # them (the engine's cached bf16 operand copies, autograd's saved-tensor checks) sees the update
        torch.autograd.graph.increment_version([p for g in self.param_groups for p in g["params"] if p.requires_grad and p.grad is not None])
        if self._model is not None and self._fresh:
            self._model._hot.mark_fresh(self._fresh)     # parameter and bf16 copy were rewritten together (or, on a skip, neither)
        return loss


class NativeScalerWithGradNormCount:
    """Two-pass replacement of util/misc.py:252-285 with the same interface.  `__call__` returns the (unscaled, pre-clip) total
    gradient norm as a 0-d device tensor, like clip_grad_norm_ / get_grad_norm_ in the reference; the loss-scale bookkeeping
    (growth / back-off) is torch.cuda.amp.GradScaler's own scalar update, fed with the device-side finiteness flag."""
    state_dict_key = "amp_scaler"

    def __init__(self, **scaler_kwargs):
        self._scaler = torch.amp.GradScaler("cuda", **scaler_kwargs)

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        if not isinstance(optimizer, AdamW):
            raise TypeError("painter_amd.optim.NativeScalerWithGradNormCount drives painter_amd.optim.AdamW")
        self._scaler.scale(loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        info = optimizer.grad_sumsq()                               # pass 1: ||scale * g||^2 and the inf/nan count
        scale = self._scaler._scale if self._scaler.is_enabled() else None
        if scale is None:
            scale = torch.ones((), dtype=torch.float32, device=info.device)
        # scalar plumbing for GradScaler.update(); an overflowed sum of squares counts as inf (pa_adamw_step skips on it too)
        # (ONE predicate on both sides: csrc/optim.hip skip_step() skips on `!(sumsq <= 3.0e38)`, which also covers NaN / inf)
        found_inf = ((info[1:2] != 0) | ~(info[0:1] <= 3.0e38)).to(torch.float32)
        norm = torch.sqrt(info[0]) / scale.reshape(())
        optimizer.step(grad_scale=scale, found_inf=found_inf, max_norm=clip_grad)    # pass 2
        if self._scaler.is_enabled():
            # GradScaler.update() with our found_inf: same growth / back-off rule (torch/amp/grad_scaler.py, update())
            torch._amp_update_scale_(self._scaler._scale, self._scaler._growth_tracker, found_inf, self._scaler._growth_factor,
                                     self._scaler._backoff_factor, self._scaler._growth_interval)
        return norm

    def state_dict(self):
        return self._scaler.state_dict()

    def load_state_dict(self, state_dict):
        self._scaler.load_state_dict(state_dict)
