"""Backward pass of the MPNN rollout on MI355X.

The reference trains by letting torch autograd differentiate ``training_step`` (models/mpnn.py:254-297): the loss
of the concatenated rollout, back through every ``forward`` call (:175-203) and - when teacher forcing is off -
through the predicted window that became the next graph's ``x``.  Here the same gradients are produced by explicit
HIP kernels (csrc/train_kernels.hip + the MLP kernels with transposed weights):

  forward of a training step   the inference kernels, unchanged, plus a *tape* of per-layer node tensors
                               (h_in, A, Bm, agg, InstanceNorm variance / output)
  backward of one forward call read-out -> [InstanceNorm -> update MLP -> mean -> message MLP -> node halves] x layers
                               -> embedding; edge-granular tensors (p, Swish(p), q) are recomputed, not stored
  across forward calls         d out_i = d loss / d out_i + d u_{i+1}  (free running), in the [B*N, tw] node layout

Gradients are accumulated in the padded layouts the kernels use and mapped to the parameters' shapes at the end.
``rollout_with_grad`` wraps all of it in one ``torch.autograd.Function`` so that ``loss.backward()`` - what Lightning
calls on the value ``training_step`` returns - fills ``.grad`` exactly as it does for the reference.

Arithmetic: backward-data GEMMs run in the ``bwd_gemm`` mode (default "bf16x6": exact 3-piece bf16 split, fp32
exponent range - gradients of a mean loss are ~1e-7 and would fall into fp16's subnormals); weight gradients on the bf16
matrix cores through the same split with a fixed-order two-stage reduction (deterministic).  Where a gradient goes through
the fp16 split after all it is scaled by an exact power of two first: per row in the data GEMMs of the fused kernels, per
launch (from block maxima handed from kernel to kernel) where it is an operand of a weight gradient.

Round 3: the edge block's backward also forms message_net_2's weight / bias gradient
(``ops.mpnn_edge_bwd_w``: s1 and dq never reach memory); on large graphs update_net_2 and the agg block of update_net_1 are
walked back by ``ops.linear_bwd128`` (data + weight + bias gradient in one kernel), and the forward launch keeps the update
net's hidden activation and raw output.
"""
from __future__ import annotations

import types

import torch

from .. import _lib as L
from .. import ops
from ..packing import pack
from .mpnn import _READOUT


def _packed_bwd(m):
    """transposed / concatenated packs for the backward-data products (cached with the forward packs)."""
    fwd = m._packed()
    if getattr(m, "_pack_bwd_key", None) == m._pack_key:
        return m._pack_bwd
    H, tw, dp, kin = m.hidden_features, m.time_window, m.dpos, m.kin
    mode = m.bwd_gemm
    dev = next(m.parameters()).device
    from ..packing import wide
    f16_planes = not wide()
    Q = types.SimpleNamespace(layers=[], mode=mode)
    padk = lambda w: torch.cat([w, torch.zeros(w.shape[0], kin - w.shape[1], device=dev)], 1)  # noqa: E731
    for lyr in m.gnn_layers:
        W1 = lyr.message_net_1[0].weight.detach().float()
        W2 = lyr.message_net_2[0].weight.detach().float()
        W3 = lyr.update_net_1[0].weight.detach().float()
        W4 = lyr.update_net_2[0].weight.detach().float()
        tail = W1[:, 2 * H:]                                             # columns of (u_i - u_j, x_i - x_j, var)
        wan = padk(tail)                                                 # d A / d node_in
        wbn = padk(-tail[:, :tw + dp])                                   # d Bm / d node_in
        q = types.SimpleNamespace()
        # dh = [dz1 | dA | dBm] . [W3_h ; W1a ; W1b]
        q.dh = pack(torch.cat([W3[:, :H].t(), W1[:, :H].t(), W1[:, H:2 * H].t()], 1), None, mode)
        q.dnin = pack(torch.cat([wan.t(), wbn.t()], 1), None, mode)     # [kin, 2H]
        q.dagg = pack(W3[:, H:2 * H].t(), None, mode)
        q.w2t = pack(W2.t(), None, mode)
        # The fused kernels of the backward hold weights as fp16 planes whatever `mode` is (their activations are recomputed from
        # post-InstanceNorm data): not built once the process has been widened to the full-range split (packing.widen) - the
        # backward then takes the materialising bf16x6 path for the edge block and the per-Linear path for the node block.
        if f16_planes:
            from ..packing import flag_if_abs_reaches
            q.w2_f16 = pack(W2, lyr.message_net_2[0].bias, "fp16x3")     # fused edge backward: both packings as fp16 planes
            q.w2t_f16 = pack(W2.t(), None, "fp16x3")
            if m.edge_bwd == "fused_w":                                  # that kernel scales the fp16 high planes by 2^11
                flag_if_abs_reaches(W2, 31.98)
        q.w4t = pack(W4.t(), None, mode)
        if mode == "bf16x6" and f16_planes and _NODE_ONE_KERNEL:         # large graphs: update_net_2 and the agg block of update_net_1
            q.w4t_f16 = pack(W4.t(), None, "fp16x3")[0]                  # walked back by k_linear_bwd128 (planes pre-scaled by 2^11)
            q.daggt_f16 = pack(W3[:, H:2 * H].t(), None, "fp16x3")[0]
            flag_if_abs_reaches(W4, 31.98)
            flag_if_abs_reaches(W3[:, H:2 * H], 31.98)
        Q.layers.append(q)
    e0, e1 = m.embedding_mlp[0].weight.detach().float(), m.embedding_mlp[2].weight.detach().float()
    Q.e1t = pack(e1.t(), None, mode)
    Q.e0t = pack(padk(e0).t(), None, mode)                               # [kin, H]
    m._pack_bwd, m._pack_bwd_key = Q, m._pack_key
    from ..graphs import capturing
    if dev.type == "cuda" and f16_planes and not capturing():      # (a captured step - graphs.TrainStepGraph - reads the word behind its replay)
        from ..packing import bad_weights
        if bad_weights(dev):                   # the fp16 planes of the fused backward kernels: one host read per re-pack
            # |w| >= 31.98 in message_net_2 / update_net_2 / the agg block of update_net_1 (or >= 65504 anywhere): the planes those
            # kernels pre-scale by 2^11 would overflow.  The forward of this step is valid (its own packs passed their check); the
            # backward - and everything from the next step on - runs in the full-range split.
            m._widen("a weight does not fit the pre-scaled fp16 planes of the fused backward kernels (|w| >= 31.98; "
                     "MAGNET_NODE_ONE_KERNEL=0 / edge_bwd='fused' avoid those kernels)")
            return _packed_bwd(m)
    return Q


def _grad_buffers(m, dev):
    H, kin = m.hidden_features, m.kin
    z = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)  # noqa: E731
    k1, _, k2 = _READOUT[m.time_window]
    G = types.SimpleNamespace(layers=[], e0=z(H, kin), be0=z(H), e1=z(H, H), be1=z(H), conv=z(8 * k1 + 8 + 8 * k2 + 1))
    for _ in m.gnn_layers:
        G.layers.append(types.SimpleNamespace(w1a=z(H, H + kin), w1b=z(H, H + kin), b1=z(H), w2=z(H, H), b2=z(H),
                                              w3=z(H, 2 * H + kin), b3=z(H), w4=z(H, H), b4=z(H)))
    return G


_NODE_ONE_KERNEL = __import__("os").environ.get("MAGNET_NODE_ONE_KERNEL", "1") != "0"      # A/B knob
_NORM_SWISH = __import__("os").environ.get("MAGNET_NORM_SWISH", "1") != "0"                # A/B knob: 0 = InstanceNorm backward and Swish backward as two passes
_WGRAD_PAIR = __import__("os").environ.get("MAGNET_WGRAD_PAIR", "1") != "0"                # A/B knob: 0 = two launches per [h | node_in] weight gradient


def _source_csr(topo):
    """edges grouped by *source* node (for the scatter of d Bm): rowptr + permutation into the by-target order."""
    if getattr(topo, "by_source", None) is None:
        c = topo.csr
        topo.by_source = ops.csr_by_target(c.tgt, c.src, c.N)           # "targets" of the flipped graph = sources
    return topo.by_source


def backward_call(m, tape, dout, G, ws):
    """Backward of one ``forward`` call.  dout [N, tw] -> d u [N, tw]; parameter gradients accumulate into G."""
    P, Q = m._packed(), _packed_bwd(m)
    H, kin, tw = m.hidden_features, m.kin, m.time_window
    topo = tape.topo
    csr, B, Ns = topo.csr, topo.n_graphs, topo.nodes_per_graph
    bs = _source_csr(topo)
    N, E, dev = csr.N, csr.E, dout.device
    T = ops.Term
    nin = tape.node_in
    lin = lambda rows, kts, layer, y, o=4: ops.mlp(rows, kts, [layer], nl=1, ht=o, ot=o, act=ops.ACT_NONE, y=y)  # noqa: E731
    # ... the same with y = res + x W in the GEMM's own epilogue (no separate add pass over [N, width]) where that instantiation exists
    def lin_add(rows, kts, layer, acc, tmp, o=4):
        """acc += x W"""
        try:
            ops.mlp(rows, kts, [layer], nl=1, ht=o, ot=o, act=ops.ACT_NONE, y=acc, epi=ops.EPI_STORE_RES, res=acc)
        except NotImplementedError:
            lin(rows, kts, layer, tmp, o)
            ops.axpy(acc, tmp, 1.0, out=acc)
        return acc
    def wg2(dz, x, dW, wcol, wcol_nin, db):
        """dW[:, wcol ..] += dz^T x and dW[:, wcol_nin ..] += dz^T node_in: one pass over dz where node_in is one k tile wide"""
        if kin == 32 and H == 128 and _WGRAD_PAIR:
            ops.wgrad_pair(dz, x, 0, nin, 0, dW, wcol, wcol_nin, db, ws)
        else:
            ops.wgrad(dz, x, 0, H, dW, wcol, db, ws)
            ops.wgrad(dz, nin, 0, kin, dW, wcol_nin, None, ws)
    new = lambda: torch.empty(N, H, device=dev, dtype=torch.float32)  # noqa: E731
    enew = lambda: torch.empty(max(E, 1), H, device=dev, dtype=torch.float32)  # noqa: E731

    k1, s1, k2 = _READOUT[tw]
    dh, du_last, gp = ops.mpnn_readout_bwd(tape.h, dout, tw, P.conv[0], P.conv[1], k1, s1, P.conv[2], k2, m.readout_swish,
                                           tape.dt)
    ops.reduce_partials(gp, gp.shape[0], gp.shape[1], G.conv)
    dnin = torch.zeros(N, kin, device=dev, dtype=torch.float32)
    dnin_l = torch.empty_like(dnin)
    z1_buf, r1_buf, z2_buf, t1, t2, dagg, dA, dBm = (new() for _ in range(8))
    r1, z2 = r1_buf, z2_buf
    has_f16 = hasattr(Q.layers[0], "w2_f16") if Q.layers else False       # not after packing.widen (see _packed_bwd)
    fused = m.edge_bwd in ("fused", "fused_w") and has_f16
    fused_w = m.edge_bwd == "fused_w" and has_f16     # dW2 / db2 formed inside the edge kernel: no s1 / dq arrays at all
    d2 = enew()
    sp, d1 = (None, None) if fused_w else (enew(), enew())
    p, q = (None, None) if fused else (enew(), enew())
    partials = torch.empty(((E + 31) // 32) * 2 * H + 1, device=dev, dtype=torch.float32) if fused else None
    ews = torch.empty(L.lib().magnet_mpnn_edge_bwd_w_workspace(), device=dev, dtype=torch.float32) if fused_w else None
    o_k = kin // 32

    lbw = None
    for li in reversed(range(len(P.layers))):
        lyr, qb, tp, g = P.layers[li], Q.layers[li], tape.layers[li], G.layers[li]
        ragged = getattr(tape, "gptr", None) is not None                 # foreign Data with graphs of unequal size
        fuse_swish = not ragged and _NORM_SWISH and getattr(tp, "z2", None) is not None
        if ragged:
            dg = ops.instance_norm_ragged_bwd(dh, tp.h_out, tp.var, tape.gptr)
        elif not fuse_swish:
            dg = ops.instance_norm_bwd(dh, tp.h_out, tp.var, B, Ns)
        # ---- update MLP (recomputed pre-activations)                                 mpnn.py:81-90
        z1 = getattr(tp, "z1", None)
        if z1 is None:
            z1 = z1_buf
            lin(N, ops.ktiles(T(tp.h_in, H), T(tp.agg, H), T(nin, kin)), lyr.upd1, z1)
        if getattr(tp, "r1", None) is not None:                           # kept by the forward launch
            r1, z2 = tp.r1, tp.z2
        else:
            r1, z2 = r1_buf, z2_buf
            ops.swish_fwd(z1, r1)
            lin(N, ops.ktiles(T(r1, H)), lyr.upd2, z2)
        one_kernel = N >= 4096 and hasattr(qb, "w4t_f16") and _NODE_ONE_KERNEL     # data + weight + bias gradient of a 128 x 128 Linear in one kernel
        am = torch.zeros(2, 256, device=dev, dtype=torch.float32) if one_kernel else None     # block maxima of |dz2|, |dz1| from the Swish backward
        if fuse_swish:                                                    # the norm's backward and update_net_2's Swish backward in one apply pass
            dg, dz2 = ops.instance_norm_bwd_swish(dh, tp.h_out, tp.var, B, Ns, z2, t1, amax=am[0] if one_kernel else None)
        else:
            dz2 = ops.swish_bwd(dg, z2, t1, amax=am[0] if one_kernel else None)
        if one_kernel:
            if lbw is None:
                lbw = (torch.empty(L.lib().magnet_linear_bwd128_workspace(), device=dev, dtype=torch.float32),
                       torch.empty(H, H, device=dev, dtype=torch.float32), torch.empty(H, device=dev, dtype=torch.float32))
            ops.linear_bwd128(dz2, r1, qb.w4t_f16, t2, g.w4, g.b4, lbw[0], mask=False, want_amax=False, accumulate=True, amax_in=(am[0], 256))
        else:
            ops.wgrad(dz2, r1, 0, H, g.w4, 0, g.b4, ws)
            lin(N, ops.ktiles(T(dz2, H)), qb.w4t, t2)
        dz1 = ops.swish_bwd(t2, z1, t2, amax=am[1] if one_kernel else None)
        wg2(dz1, tp.h_in, g.w3, 0, 2 * H, g.b3)
        if one_kernel:                                                   # dagg = dz1 W3[:, agg block], dW3[:, agg block] += dz1^T agg
            dagg_amax = ops.linear_bwd128(dz1, tp.agg, qb.daggt_f16, dagg, lbw[1], lbw[2], lbw[0], mask=False, amax_in=(am[1], 256))
            ops.copy_cols(lbw[1], 0, g.w3, H, H, accumulate=True)
        else:
            dagg_amax = None
            ops.wgrad(dz1, tp.agg, 0, H, g.w3, H, None, ws)
            lin(N, ops.ktiles(T(dz1, H)), qb.dagg, dagg)
        # ---- mean aggregation + message MLP at edge granularity                      mpnn.py:73-79, :46
        tA, tBm = tp.A, tp.Bm
        l2e = bool(getattr(tape, "l2e", False))                          # the tape holds the forward's log2(e)-scaled halves
        if l2e and not (E > 0 and fused_w and ops.mpnn_edge_bwd_w_takes_l2e(N)):
            # (the forward expected generation 3 of the fused kernel and something changed in between - the process was widened, the
            #  generation knob moved: the other paths read plain halves)
            tA, tBm, l2e = tp.A * 0.6931471805599453, tp.Bm * 0.6931471805599453, False
        if E > 0 and fused_w:
            ops.mpnn_edge_bwd_w(tA, tBm, dagg, csr, qb.w2_f16[0], qb.w2t_f16[0], qb.w2_f16[1], d2, dA, partials, g.w2, g.b2, ews,
                                amax_in=dagg_amax, l2e=l2e)
            ops.segment_sum128(d2, bs.rowptr, N, perm=bs.perm, out=dBm)
        elif E > 0 and fused:
            ops.mpnn_edge_bwd(tA, tBm, dagg, csr, qb.w2_f16[0], qb.w2t_f16[0], qb.w2_f16[1], sp, d1, d2, dA, partials)
            ops.wgrad(d1, sp, 0, H, g.w2, 0, g.b2, ws, M=E)
            ops.segment_sum128(d2, bs.rowptr, N, perm=bs.perm, out=dBm)
        elif E > 0:
            ops.edge_mean_bwd(dagg, csr, d1)
            ops.edge_pre(tA, tBm, csr, p)
            ops.swish_fwd(p, sp)
            lin(E, ops.ktiles(T(sp, H)), lyr.msg2, q)
            dq = ops.swish_bwd(d1, q, d1)
            ops.wgrad(dq, sp, 0, H, g.w2, 0, g.b2, ws, M=E)
            lin(E, ops.ktiles(T(dq, H)), qb.w2t, d2)
            dp = ops.swish_bwd(d2, p, d2)
            ops.segment_sum128(dp, csr.rowptr, N, out=dA)
            ops.segment_sum128(dp, bs.rowptr, N, perm=bs.perm, out=dBm)
        else:
            dA.zero_()
            dBm.zero_()
        # ---- per-node halves of message_net_1
        wg2(dA, tp.h_in, g.w1a, 0, H, g.b1)
        wg2(dBm, tp.h_in, g.w1b, 0, H, None)
        dh = lin_add(N, ops.ktiles(T(dz1, H), T(dA, H), T(dBm, H)), qb.dh, dg, t1)
        lin_add(N, ops.ktiles(T(dA, H), T(dBm, H)), qb.dnin, dnin, dnin_l, o_k)

    # ---- embedding MLP                                                                mpnn.py:131-136
    z1 = z1_buf                                                          # (not the last layer's tape entry)
    lin(N, ops.ktiles(T(nin, kin)), P.emb[0], z1)
    r1, z2 = r1_buf, z2_buf
    ops.swish_fwd(z1, r1)
    lin(N, ops.ktiles(T(r1, H)), P.emb[1], z2)
    dz2 = ops.swish_bwd(dh, z2, t1)
    ops.wgrad(dz2, r1, 0, H, G.e1, 0, G.be1, ws)
    lin(N, ops.ktiles(T(dz2, H)), Q.e1t, t2)
    dz1 = ops.swish_bwd(t2, z1, t2)
    ops.wgrad(dz1, nin, 0, kin, G.e0, 0, G.be0, ws)
    lin_add(N, ops.ktiles(T(dz1, H)), Q.e0t, dnin, dnin_l, o_k)

    du = torch.empty(N, tw, device=dev, dtype=torch.float32)
    ops.copy_cols(dnin, 0, du, 0, tw)
    ops.copy_cols(du_last[:, None], 0, du, tw - 1, 1, accumulate=True)
    return du


def parameter_grads(m, G, inplace=False):
    """kernel-layout gradient buffers -> one tensor per parameter, in ``m.parameters()`` order.  ``inplace`` (only what
    ``training_step`` decided in its forward, see ``_Rollout.forward``): add to the flat ``p.grad`` views instead and return Nones."""
    H, tw, dp = m.hidden_features, m.time_window, m.dpos
    nin = tw + dp + 1
    k1, _, k2 = _READOUT[tw]
    by_name = {}
    for i, g in enumerate(G.layers):
        tail = g.w1a[:, H:H + nin].clone()
        tail[:, :tw + dp] -= g.w1b[:, H:H + tw + dp]
        pre = f"gnn_layers.{i}."
        by_name[pre + "message_net_1.0.weight"] = torch.cat([g.w1a[:, :H], g.w1b[:, :H], tail], 1)
        by_name[pre + "message_net_1.0.bias"] = g.b1
        by_name[pre + "message_net_2.0.weight"], by_name[pre + "message_net_2.0.bias"] = g.w2, g.b2
        by_name[pre + "update_net_1.0.weight"] = torch.cat([g.w3[:, :2 * H], g.w3[:, 2 * H + tw + dp:2 * H + nin]], 1)
        by_name[pre + "update_net_1.0.bias"] = g.b3
        by_name[pre + "update_net_2.0.weight"], by_name[pre + "update_net_2.0.bias"] = g.w4, g.b4
    by_name["embedding_mlp.0.weight"], by_name["embedding_mlp.0.bias"] = G.e0[:, :nin], G.be0
    by_name["embedding_mlp.2.weight"], by_name["embedding_mlp.2.bias"] = G.e1, G.be1
    c = G.conv
    last = len(m.output_mlp) - 1
    by_name["output_mlp.0.weight"] = c[:8 * k1].view(8, 1, k1)
    by_name["output_mlp.0.bias"] = c[8 * k1:8 * k1 + 8]
    by_name[f"output_mlp.{last}.weight"] = c[8 * k1 + 8:8 * k1 + 8 + 8 * k2].view(1, 8, k2)
    by_name[f"output_mlp.{last}.bias"] = c[8 * k1 + 8 + 8 * k2:]
    # one flat buffer in parameter order (what FusedAdam and the gradient all-reduce consume without a gather)
    from ..optim import FlatGrads, flat_gradient
    params = list(m.parameters())
    from ..autograd import takes_inplace_grad
    if inplace and flat_gradient(params) is not None and all(takes_inplace_grad(p_) for p_ in params):
        # FusedAdam.zero_grad has left every p.grad as a (zeroed, or partly accumulated) view of ONE flat buffer: add this call's
        # share in place and hand autograd nothing - the same ~60 launches as the copies below, and no AccumulateGrad adds behind them.
        # Nothing that hangs on AccumulateGrad fires on this path, so it is taken only for a ``training_step`` outside a DDP forward
        # whose parameters carry no hooks (autograd.accumulate_param_grads); ``torch.autograd.grad`` on that loss would get Nones.
        for (n, p_) in m.named_parameters():
            p_.grad.add_(by_name[n].reshape(p_.grad.shape))
        return [None] * len(params)
    fg = FlatGrads(params)
    for (n, _), view in zip(m.named_parameters(), fg.views):
        view.copy_(by_name[n].reshape(view.shape))
    return fg.views


class _Rollout(torch.autograd.Function):
    """u_hat = rollout(batch); backward = the HIP backward over every forward call of the rollout."""

    @staticmethod
    def forward(ctx, m, batch, teacher_forcing, *params):
        tapes = []
        with torch.no_grad():
            u_hat, target = m.rollout(batch, teacher_forcing, tapes=tapes)
        from ..graphs import capturing
        why = None if capturing() else m._guard.tripped()      # (a captured step - graphs.TrainStepGraph - reads the word behind its replay)
        if why:                                # out of the fp16 pieces' range: the same forward (and its tapes) in the full-range split
            m._widen(why)
            tapes = []
            with torch.no_grad():
                u_hat, target = m.rollout(batch, teacher_forcing, tapes=tapes)
        ctx.m, ctx.tapes, ctx.tf = m, tapes, teacher_forcing
        from ..autograd import ACCUM_PARAM_GRADS, in_ddp_forward
        ctx.inplace = ACCUM_PARAM_GRADS and not in_ddp_forward()    # decided where DDP's forward is still on the stack
        m._last_target = target
        return u_hat.contiguous()

    @staticmethod
    def backward(ctx, d_u_hat):
        m, tapes = ctx.m, ctx.tapes
        tw = m.time_window
        B, _, Ns = d_u_hat.shape
        dev = d_u_hat.device
        with torch.no_grad():
            G = _grad_buffers(m, dev)
            E = tapes[0].topo.csr.E
            ws = ops.wgrad_workspace(max(B * Ns, E), dev)
            carry = None
            for i in reversed(range(len(tapes))):
                # d loss / d out_i in the node layout [B*N, tw] (out_i -> y_hat = reshape(B, N, tw).permute(0, 2, 1))
                dout = d_u_hat[:, i * tw:(i + 1) * tw].permute(0, 2, 1).reshape(B * Ns, tw).float().contiguous()
                if carry is not None:
                    ops.axpy(dout, carry, 1.0, out=dout)
                du = backward_call(m, tapes[i], dout, G, ws)
                carry = None if ctx.tf else du
            grads = parameter_grads(m, G, inplace=ctx.inplace)
        ctx.tapes = None
        return (None, None, None, *grads)


def rollout_with_grad(m, batch, teacher_forcing=False):
    """-> (u_hat with a grad_fn, target)"""
    u_hat = _Rollout.apply(m, batch, teacher_forcing, *m.parameters())
    target, m._last_target = m._last_target, None
    return u_hat, target


class _Forward(torch.autograd.Function):
    """out = forward(data, L, tmax, dt) with a grad_fn: the backward of ONE forward call (the reference's forward is an ordinary
    differentiable torch function, models/mpnn.py:175-203; Lightning never calls it with gradients on, a custom loop may)."""

    @staticmethod
    def forward(ctx, m, data, L_, tmax, dt, x, *params):
        tape = types.SimpleNamespace()
        with torch.no_grad():
            out = m._forward(data, L_, tmax, dt, tape)
        why = m._guard.tripped()
        if why:                                # out of the fp16 pieces' range: the same forward (and its tape) in the full-range split
            m._widen(why)
            tape = types.SimpleNamespace()
            with torch.no_grad():
                out = m._forward(data, L_, tmax, dt, tape)
        ctx.m, ctx.tape = m, tape
        return out

    @staticmethod
    def backward(ctx, dout):
        m, tape = ctx.m, ctx.tape
        dev = dout.device
        with torch.no_grad():
            G = _grad_buffers(m, dev)
            csr = tape.topo.csr
            ws = ops.wgrad_workspace(max(csr.N, csr.E), dev)
            du = backward_call(m, tape, dout.float().contiguous(), G, ws)
            grads = parameter_grads(m, G)
        ctx.tape = None
        return (None, None, None, None, None, du if ctx.needs_input_grad[5] else None, *grads)


def forward_with_grad(m, data, L_, tmax, dt):
    return _Forward.apply(m, data, L_, tmax, dt, data.x, *m.parameters())
