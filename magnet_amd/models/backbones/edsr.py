"""EDSR feature encoder (reference models/backbones/edsr.py:32-60; effective res_scale is always 1 because the
reference passes it positionally into ``bias``).

Parameters keep the reference's names and torch ``Conv1d`` / ``Conv2d`` layouts (checkpoints load unchanged); the
arithmetic runs in libmagnet_hip.so: every "same" convolution is ONE launch of the fused MLP
kernel as an implicit GEMM - the k^d shifted neighbours of a pixel are gathered as k tiles through tap index tables
(``magnet_conv_tap_index``), zero padding is an all-zero row appended to the channels-last map, ReLU and the skip
connections are the kernel's activation / residual epilogues.  No im2col buffer, no MIOpen.
"""
from __future__ import annotations

import torch
from torch import nn

from ... import ops
from ...graphs import hold
from ...packing import default_gemm, pack, pad_to, param_list


class ResBlock(nn.Module):
    def __init__(self, n_chan, kernel_size, mode="1d"):
        super().__init__()
        conv = nn.Conv2d if mode == "2d" else nn.Conv1d
        self.conv_1 = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.act = nn.ReLU(True)
        self.conv_2 = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)


def _as_linear(conv, c_pad):
    """conv weight [Co, Ci, k(, k)] -> Linear weight [Co, taps * c_pad], tap-major (row-major dy, dx), channels inside"""
    w = conv.weight.detach().float()
    co, ci = w.shape[:2]
    w = w.reshape(co, ci, -1).permute(0, 2, 1)                  # [Co, taps, Ci]
    if c_pad != ci:
        w = torch.cat([w, torch.zeros(co, w.shape[1], c_pad - ci, device=w.device)], 2)
    return w.reshape(co, -1)


class EDSR(nn.Module):
    def __init__(self, in_chan, n_chan=64, res_layers=16, kernel_size=3, res_scale=1, mode="1d"):
        super().__init__()
        conv = nn.Conv2d if mode == "2d" else nn.Conv1d
        self.head_conv = conv(in_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.res_layers = nn.Sequential(*[ResBlock(n_chan, kernel_size, mode) for _ in range(res_layers)])
        self.tail_conv = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.out_dim, self.in_chan, self.n_chan = n_chan, in_chan, n_chan
        self.kernel_size, self.dim = kernel_size, 2 if mode == "2d" else 1
        if n_chan % 32 or n_chan // 32 not in (2, 4):
            raise NotImplementedError("the convolution kernels are instantiated for n_chan = 64 or 128")
        if (kernel_size ** self.dim) * (n_chan // 32) > 36 or not kernel_size & 1:
            raise NotImplementedError("odd kernel sizes with k^d * n_chan / 32 <= 36 k tiles are supported")
        self._pack_key, self._pack, self._pack_gen, self._taps = None, None, 0, {}
        self.resident_conv = __import__("os").environ.get("MAGNET_RESIDENT_CONV", "1") != "0"      # A/B knob
        self._plan_key, self._plan_val = None, None
        self._pack_id = 0
        self._pack_replay = __import__("os").environ.get("MAGNET_PACK_REPLAY", "1") != "0"      # A/B knob (packing.PackRecorder)

    def __getstate__(self):
        """pickle / copy.deepcopy: parameters travel, the per-instance device caches (packed weights, tap tables, launch plans) do not"""
        from ...packing import state_without_device_caches
        return state_without_device_caches(self)

    # ------------------------------------------------------------------ packed weights / tap tables (cached)
    def packed(self, mode=None):
        mode = mode or default_gemm()
        params = param_list(self)
        # _pack_gen: bumped by the owning model when an optimizer wrote the parameters behind torch's version counters (FusedAdam)
        key = (params[0].device, sum(p._version for p in params), mode, self._pack_gen)
        if self._pack_key != key:
            from ...packing import PackRecorder, derived, recording, weights_check_due
            rec, rkey = self.__dict__.get("_pack_rec"), (params[0].device, mode)
            if self._pack is not None and rec is not None and rec.key == rkey and rec.usable() and self._pack_replay:
                # the weights changed, the structure did not: the recorded pack launches again, into the same buffers - the launch
                # plans built on them (_plan) and everything cached on the pack dict stay valid (packing.PackRecorder)
                rec.replay()
            else:
                cin = pad_to(self.in_chan)
                rec = PackRecorder(params, rkey) if params[0].is_cuda else None

                def mk(conv, cp):
                    """(whole pack, per-32-output-column packs): the second form feeds the column-split launch"""
                    w, b = derived(lambda: _as_linear(conv, cp)), conv.bias.detach().float()
                    return pack(w, b, mode), [pack(w[g:g + 32], b[g:g + 32], mode) for g in range(0, w.shape[0], 32)]

                with recording(rec):
                    self._pack = dict(head=mk(self.head_conv, cin), tail=mk(self.tail_conv, self.n_chan),
                                      blocks=[(mk(b.conv_1, self.n_chan), mk(b.conv_2, self.n_chan)) for b in self.res_layers])
                self.__dict__["_pack_rec"] = rec
                self._pack_id += 1                       # (the launch plans are keyed on it: new buffers, new plans)
            self._pack_key = key
            if params[0].is_cuda and weights_check_due(params[0].device):   # (inside a rollout scope the owning model reads it at the end)
                from ...packing import bad_weights, widen
                if bad_weights(params[0].device):        # fp16x3 range check of the packs just made: one host read per re-pack
                    self._pack_key = None
                    if widen("EDSR: a convolution weight does not fit the two fp16 planes (|w| >= 65504)"):
                        return self.packed()             # the same packs in the full-range split
                    raise ValueError("EDSR: non-finite convolution weights")
        return self._pack

    def _tap_table(self, B, spatial, device):
        key = (B, tuple(spatial), str(device))
        if key not in self._taps:
            H, W = (spatial[0], spatial[1]) if self.dim == 2 else (1, spatial[0])
            if len(self._taps) > 4:
                self._taps.clear()
            self._taps[key] = ops.conv_tap_index(B, H, W, self.kernel_size, self.dim, device)
        return self._taps[key]

    def weights_changed(self):
        """parameters were rewritten in place without a version bump: drop the packs and every launch plan built on them"""
        self._pack_gen += 1
        self._pack_key = None                    # (the plans go with the pack BUFFERS, _pack_id, not with their contents)

    # ------------------------------------------------------------------ forward
    def _plan(self, B, cin, spatial, dev):
        """buffers + the 2 * res_layers + 2 prebuilt launches for one input geometry (cached with the packed weights)"""
        pk = self.packed()
        g = ops.RangeGuard.current                                # (the prebuilt launches report into the armed guard's status word)
        key = (B, cin, tuple(spatial), str(dev), self._pack_id, id(g) if g is not None and g.active else None)
        if self._plan_key == key:
            return self._plan_val
        P = B * int(torch.tensor(spatial).prod())
        C, ot = self.n_chan, self.n_chan // 32
        taps = self._tap_table(B, spatial, dev)
        cp = pad_to(cin)
        xin = torch.zeros(P + 1, cp, device=dev, dtype=torch.float32)           # + the zero-padding row
        head = torch.zeros(P + 1, C, device=dev, dtype=torch.float32)           # row P of every map stays zero
        work = [torch.zeros_like(head) for _ in range(3)]
        launches = []

        split = P < 16384                     # fewer than 128 workgroups of 128 rows: split the output columns as well

        def conv(src, width, layer, dst, relu=False, res=None):
            kts = ops.ktiles(*[ops.Term(src, width, idx=taps[t]) for t in range(taps.shape[0])])
            kw = dict(nl=1, act=ops.ACT_RELU if relu else ops.ACT_NONE, last_act=relu,
                      epi=ops.EPI_STORE_RES if res is not None else ops.EPI_STORE, res=res, y=dst, ldy=C)
            fast = ops.conv_resident_plan(src, width, taps, layer[1], relu, res, dst, C) if split and self.resident_conv else None
            if fast is not None:               # small map, 3 x 3, fp16x3: the column group's planes resident in LDS
                launches.append(fast)
            elif split:
                launches.append(ops.mlp_plan(P, kts, layer[1], ht=1, ot=1, ncols=32, colgroups=ot, **kw))
            else:
                launches.append(ops.mlp_plan(P, kts, [layer[0]], ht=ot, ot=ot, **kw))
            return dst

        cur = conv(xin, cp, pk["head"], head)
        for c1, c2 in pk["blocks"]:                                              # conv_2(relu(conv_1(x))) + x
            mid, out = [w for w in work if w is not cur][:2]
            conv(cur, C, c1, mid, relu=True)
            cur = conv(mid, C, c2, out, res=cur)
        out = conv(cur, C, pk["tail"], [w for w in work if w is not cur][0], res=head)   # tail_conv(res) + head
        self._plan_key, self._plan_val = key, (xin, out, launches, P)
        return self._plan_val

    def rows(self, x):
        """x [B, in_chan, W(, W)] -> channels-last feature rows [B * W^d, n_chan] (row-major pixels).  The returned
        tensor is a view of a buffer that the next call overwrites."""
        x = ops._f32(x, "EDSR input")
        B, cin = x.shape[:2]
        xin, out, launches, P = self._plan(B, cin, x.shape[2:], x.device)
        # a hipGraph capture in progress bakes in this plan's buffers (feature maps with their zero-padding rows, tap tables,
        # packed weights): the capture keeps them alive although `_plan` holds one geometry at a time (magnet_amd/graphs.py)
        hold(self._plan_val, self._pack, self._taps.get((B, tuple(x.shape[2:]), str(x.device))))
        xin[:P, :cin] = x.reshape(B, cin, -1).permute(0, 2, 1).reshape(P, cin)
        for go in launches:
            go()
        return out[:P]

    def rows_grad(self, x):
        """``rows`` with gradients (training): every convolution is a differentiable gathered-tap Linear
        (magnet_amd.autograd.mlp, HIP forward and backward); the conv weight reaches it as a torch view
        [Co, taps, Ci] -> [Co, taps * Ci] of the parameter, so its gradient lands in the parameter's own layout."""
        from ... import autograd as AG
        import types
        x = ops._f32(x, "EDSR input")
        B, cin = x.shape[:2]
        spatial = x.shape[2:]
        P = B * int(torch.tensor(spatial).prod())
        dev, C, ot = x.device, self.n_chan, self.n_chan // 32
        pk = self.packed()
        taps = self._tap_table(B, spatial, dev)
        ntaps = taps.shape[0]

        def as_rows(t_rows):                                   # append the all-zero padding row
            return torch.cat([t_rows, torch.zeros(1, t_rows.shape[1], device=dev, dtype=torch.float32)], 0)

        rev = self._taps.setdefault(("rev", B, tuple(spatial), str(dev)), None)
        if rev is None:
            rev = self._taps[("rev", B, tuple(spatial), str(dev))] = taps.flip(0).contiguous()      # tap T - 1 - t undoes tap t

        def conv(src, width, layer, conv_mod, relu=False, res=None):
            ci = conv_mod.weight.shape[1]
            W2 = conv_mod.weight.reshape(conv_mod.weight.shape[0], ci, ntaps).permute(0, 2, 1).reshape(conv_mod.weight.shape[0], -1)
            # one namespace per convolution for as long as its pack lives: the backward caches its transposed pack on it
            cache = pk.setdefault("_grad_ns", {})
            pkc = cache.get((id(conv_mod), width))
            if pkc is None:
                cmap = None if ci == width else \
                    (torch.arange(ntaps, device=dev)[:, None] * width + torch.arange(ci, device=dev)[None, :]).reshape(-1)
                pkc = cache[(id(conv_mod), width)] = types.SimpleNamespace(layers=[layer[0]], ln=None, ht=ot, ot=ot, out_dim=C, col_map=cmap,
                                                                           _rec=self.__dict__.get("_pack_rec"))
            return AG.conv(src, taps, rev, pkc, W2, conv_mod.bias, relu=relu, res=res)

        cp = pad_to(cin)
        xin = torch.zeros(P, cp, device=dev, dtype=torch.float32)
        xin = torch.cat([x.reshape(B, cin, -1).permute(0, 2, 1).reshape(P, cin), xin[:, cin:]], 1) if cp != cin else \
            x.reshape(B, cin, -1).permute(0, 2, 1).reshape(P, cin)
        head = conv(as_rows(xin), cp, pk["head"], self.head_conv)
        cur = head
        for blk, (c1, c2) in zip(self.res_layers, pk["blocks"]):
            mid = conv(as_rows(cur), C, c1, blk.conv_1, relu=True)
            cur = conv(as_rows(mid), C, c2, blk.conv_2, res=cur)
        return conv(as_rows(cur), C, pk["tail"], self.tail_conv, res=head)

    @torch.no_grad()
    def forward(self, x):
        """[B, in_chan, W(, W)] -> [B, n_chan, W(, W)] (the reference's layout; the models consume ``rows`` directly)."""
        B = x.shape[0]
        r = self.rows(x)
        return r.reshape(B, -1, self.n_chan).permute(0, 2, 1).reshape(B, self.n_chan, *x.shape[2:]).contiguous()
