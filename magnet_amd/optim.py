"""Fused Adam for the training step (reference ``configure_optimizers``, models/mpnn.py:206-214:
``torch.optim.Adam(lr, weight_decay)`` + ``StepLR(step_size, gamma=factor)``).

All parameters of the model are moved into ONE flat fp32 buffer (each ``p.data`` becomes a view of it), with flat
moment buffers beside it, so that an optimizer step is a single kernel launch (``magnet_adam_f32``) instead of one
launch group per parameter tensor, and the gradient all-reduce (``magnet_amd.sharding.allreduce_gradients``) is one
RCCL call on one buffer.  The class is a ``torch.optim.Optimizer`` - ``param_groups[0]['lr']`` is what ``StepLR``
rewrites, ``state_dict`` / ``zero_grad`` come from the base class - so Lightning drives it like the reference's Adam.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import ops


class FlatGrads:
    """A flat gradient buffer laid out like the flat parameter buffer; ``views[i]`` is the slice of parameter i."""

    def __init__(self, params):
        self.sizes = [p.numel() for p in params]
        self.total = sum(self.sizes)
        dev = params[0].device
        self.flat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.views, o = [], 0
        for p, n in zip(params, self.sizes):
            self.views.append(self.flat[o:o + n].view(p.shape))
            o += n


def flat_gradient(params):
    """the flat buffer behind ``[p.grad for p in params]`` if they are consecutive slices of one, else None"""
    g0 = params[0].grad
    if g0 is None:
        return None
    base, o = g0.data_ptr(), 0
    for p in params:
        g = p.grad
        if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.data_ptr() != base + 4 * o:
            return None
        o += g.numel()
    store = g0.untyped_storage()
    off = (base - store.data_ptr()) // 4
    if store.nbytes() < 4 * (off + o):
        return None
    return torch.empty(0, device=g0.device, dtype=torch.float32).set_(store, off, (o,))


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        every = list(params)
        # the reference builds Adam(self.parameters()) (models/mpnn.py:206-214): its checkpoints index ALL parameters, frozen ones
        # included; `_index[i]` is the position of our i-th (trainable) parameter in that list
        self._index = [i for i, p in enumerate(every) if p.requires_grad]
        self._n_all = len(every)
        params = [every[i] for i in self._index]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._params = params
        if not params[0].is_cuda:
            raise RuntimeError("magnet_amd.optim.FusedAdam: parameters must live on a HIP device (no CPU path)")
        n = sum(p.numel() for p in params)
        self.flat_p = torch.empty(n, device=params[0].device, dtype=torch.float32)
        o = 0
        with torch.no_grad():
            for p in params:                                   # parameters become views of the flat buffer
                k = p.numel()
                self.flat_p[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
                o += k
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self._scratch_g = None
        self._gbuf = None                                      # FlatGrads behind p.grad (zero_grad)
        self.steps = 0
        self.on_step = None

    def zero_grad(self, set_to_none: bool = True):
        """Gradients of this optimizer's parameters live in ONE flat buffer laid out like the flat parameter buffer: ``p.grad`` becomes
        (stays) a zeroed view of it instead of None.  Autograd's AccumulateGrad then adds in place, the accumulating backward kernels
        (autograd.accumulate_param_grads) add their share straight into it, and ``step()`` / the gradient all-reduce find one
        contiguous array instead of gathering ~230 tensors with as many ``copy_`` launches (a ``gnn_64`` step: 415 -> 185).  The sum
        0 + g1 + g2 has the bits of g1 + g2.  MAGNET_FLAT_GRADS=0: torch's behaviour (``set_to_none`` honoured)."""
        import os
        if os.environ.get("MAGNET_FLAT_GRADS", "1") == "0":
            return super().zero_grad(set_to_none=set_to_none)
        # (``set_to_none`` is not honoured on this path: a None gradient would break the flat layout.  Code that tells "no gradient
        #  this step" from ``p.grad is None`` sees zeros instead; a parameter frozen AFTER construction is the exception below.)
        gb = self._gbuf
        if gb is None or any(v.device != p_.device or v.shape != p_.shape or p_.dtype != torch.float32
                             for p_, v in zip(self._params, gb.views)):
            gb = self._gbuf = FlatGrads(self._params)          # first call, or the model moved / changed dtype since: stale views
        else:
            gb.flat.zero_()
        for p_, v in zip(self._params, gb.views):
            if not p_.requires_grad:                           # frozen after construction: no gradient, and (zero slice) no Adam motion
                p_.grad = None                                 # beyond what its decaying moments still carry - as torch's Adam skips it
            elif p_.grad is not v:
                p_.grad = v

    def _spans(self):
        o = 0
        for p in self._params:
            yield o, p.numel()
            o += p.numel()

    def _gradient(self):
        g = flat_gradient(self._params)
        if g is not None:
            return g
        if self._scratch_g is None:
            self._scratch_g = torch.empty_like(self.flat_p)
        o = 0
        for p in self._params:                                 # gradients from elsewhere (not our backward): gather
            k = p.numel()
            if p.grad is None:
                self._scratch_g[o:o + k].zero_()
            else:
                self._scratch_g[o:o + k].copy_(p.grad.reshape(-1))
            o += k
        return self._scratch_g

    # ---- checkpointing: the moments live in the flat buffers, the checkpoint is torch.optim.Adam's format
    def state_dict(self):
        """The layout ``torch.optim.Adam.state_dict()`` writes (what the reference's Lightning checkpoints hold,
        configure_optimizers models/mpnn.py:206-214): per parameter index ``step`` / ``exp_avg`` / ``exp_avg_sq`` in the
        parameter's own shape, so a checkpoint written here resumes under torch.optim.Adam and vice versa.  Indices count
        every parameter the optimizer was constructed with (as ``Adam(model.parameters())`` does); frozen parameters have
        no state entry, exactly like parameters torch's Adam never saw a gradient for."""
        sd = super().state_dict()
        sd["param_groups"][0]["params"] = list(range(self._n_all))
        state, o = {}, 0
        if self.steps > 0:
            for i, p in zip(self._index, self._params):
                k = p.numel()
                state[i] = {"step": torch.tensor(float(self.steps)),
                            "exp_avg": self.flat_m[o:o + k].view(p.shape).clone(),
                            "exp_avg_sq": self.flat_v[o:o + k].view(p.shape).clone()}
                o += k
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        """accepts torch.optim.Adam's layout over ALL parameters (the reference's checkpoints: no entry for a parameter that
        never received a gradient, frozen parameters counted in the indices), the same layout over the trainable parameters
        only, and round-1 ``magnet_flat`` checkpoints.  Missing entries start from zero moments at the common step count."""
        flat = state_dict.get("magnet_flat")              # round-1 checkpoints of this class
        per_param = state_dict.get("state", {})
        groups = state_dict["param_groups"]
        if len(groups) != 1:
            raise ValueError(f"FusedAdam.load_state_dict: {len(groups)} parameter groups in the checkpoint, one here")
        for k, v in groups[0].items():                    # lr, betas, eps, weight_decay, the scheduler's initial_lr ...
            if k != "params":
                self.param_groups[0][k] = v
        if flat is not None:
            self.flat_m.copy_(flat["exp_avg"])
            self.flat_v.copy_(flat["exp_avg_sq"])
            self.steps = int(flat["step"])
            return
        saved = list(groups[0]["params"])
        if len(saved) == self._n_all:
            ids = [saved[i] for i in self._index]         # checkpoint id of our i-th trainable parameter
        elif len(saved) == len(self._params):
            ids = saved
        else:
            raise ValueError(f"FusedAdam.load_state_dict: the checkpoint's group has {len(saved)} parameters; this optimizer "
                             f"was built over {self._n_all} ({len(self._params)} trainable)")
        stray = set(per_param) - set(ids)
        if stray:
            raise ValueError(f"FusedAdam.load_state_dict: optimizer state for frozen / unknown parameters {sorted(stray)[:4]}")
        self.flat_m.zero_()
        self.flat_v.zero_()
        steps, o = set(), 0
        for cid, p in zip(ids, self._params):
            k = p.numel()
            st = per_param.get(cid)
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdam.load_state_dict: moment shape {tuple(st['exp_avg'].shape)} of parameter {cid} "
                                     f"does not match {tuple(p.shape)}")
                self.flat_m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            o += k
        if len(steps) > 1:
            raise ValueError(f"FusedAdam.load_state_dict: parameters carry different step counts {sorted(steps)}; "
                             "one bias correction per step is applied to the whole flat buffer")
        self.steps = steps.pop() if steps else 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        g = self._gradient()
        self.steps += 1
        frozen = [(o, k) for (o, k), p_ in zip(self._spans(), self._params) if not p_.requires_grad]
        keep = [(o, k, self.flat_p[o:o + k].clone(), self.flat_m[o:o + k].clone(), self.flat_v[o:o + k].clone()) for o, k in frozen]
        L.check(L.lib().magnet_adam_f32(ops._p(self.flat_p), ops._p(g), ops._p(self.flat_m), ops._p(self.flat_v),
                                        C.c_long(self.flat_p.numel()), C.c_float(grp["lr"]), C.c_float(grp["betas"][0]),
                                        C.c_float(grp["betas"][1]), C.c_float(grp["eps"]), C.c_float(grp["weight_decay"]),
                                        self.steps, ops._stream()), "adam")
        for o, k, p0, m0, v0 in keep:                          # frozen after construction: torch's Adam skips a parameter without
            self.flat_p[o:o + k].copy_(p0)                     # a gradient - value and moments stay as they were
            self.flat_m[o:o + k].copy_(m0)
            self.flat_v[o:o + k].copy_(v0)
        if self.on_step is not None:                           # the kernel wrote behind torch's version counters:
            self.on_step()                                     # tell the model its packed-weight caches are stale
        return loss
