#!/usr/bin/env python3
"""Which launch of the mpnn_2d forward is not deterministic?  (after scripts/stress_precision_leg.py showed that
eager rollouts deviate as well: ~10 % of the 48 x 48 rollouts, errors of 4e-6 .. 9e-6, any forward call.)

One process: the bench's weights, one W x W trajectory, the SAME forward call R times, kernel by kernel.  Every ops.* launch of the
forward is wrapped; its output arrays are cloned (device-side copies, no host sync) and compared bitwise with the first run's.  Per run
the FIRST deviating launch is reported (name, index inside the forward, how many elements / which rows differ), then the summary.

    python scripts/dev/find_flaky_kernel.py [--mesh 48] [--traj 1] [--runs 400] [--gap-every 25 --gap 0.3] [--env K=V ...]
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=48)
    ap.add_argument("--traj", type=int, default=1)
    ap.add_argument("--runs", type=int, default=400)
    ap.add_argument("--gap-every", type=int, default=25)
    ap.add_argument("--gap", type=float, default=0.3, help="seconds of idle every --gap-every runs (cold clocks)")
    ap.add_argument("--hparam", action="append", default=[], help="extra model hparam K=V (python literal)")
    ap.add_argument("--no-record", action="store_true", help="compare the forward's OUTPUT only (no clones between the launches)")
    ap.add_argument("--rollout", action="store_true", help="the unit is a whole validation_step rollout (4 forward calls + the glue)")
    ap.add_argument("--prior", type=int, default=0, help="N rollouts of a 4 x 256^2 batch on another model first (the bench's state)")
    args = ap.parse_args()
    import torch

    import bench
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    extra = {}
    for kv in args.hparam:
        k, v = kv.split("=", 1)
        try:
            extra[k] = eval(v)
        except Exception:  # noqa: BLE001
            extra[k] = v
    if args.prior:
        big = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph="rebuild")).to(dev).eval()
        bb = bench.synthetic_batch(256, 4, 1000, dev)
        for _ in range(args.prior):
            big.validation_step(bb, 0)
        torch.cuda.synchronize()
        torch.manual_seed(0)
    model = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph=False, **extra)).to(dev).eval()
    batch = bench.synthetic_batch(args.mesh, args.traj, 5, dev)
    u, x, t = model._unpack(batch)
    tw = model.time_window
    graph = model._build_graph(u[:, :tw], t, x, [model._first_step()] * args.traj)
    dt = t[0][1] - t[0][0]
    fw = lambda: model.forward(graph, x[0, -1], t[0, -1], dt)   # noqa: E731
    if args.rollout:
        fw = lambda: model.validation_step(batch, 0)             # noqa: E731

    names = ["mpnn_node_input", "mlp", "mpnn_node_halves", "mpnn_edge16", "mpnn_edge", "mpnn_node_update", "instance_norm_stats",
             "instance_norm", "mpnn_readout"]
    log = []

    def outputs_of(name, a, kw, ret):
        if name == "mpnn_node_input":
            return [ret]
        if name == "mlp":
            return [kw["y"]]
        if name == "mpnn_node_halves":
            return [a[4], a[5]]
        if name in ("mpnn_edge16", "mpnn_edge"):
            return [a[5]]
        if name == "mpnn_node_update":
            return [a[5]]
        if name == "instance_norm_stats":
            return list(ret)
        if name == "instance_norm":
            return [kw.get("out", ret)]
        if name == "mpnn_readout":
            return [ret] if ret is not None else []
        return []

    def wrap(name, fn):
        def w(*a, **kw):
            ret = fn(*a, **kw)
            log.append((name, [o.clone() for o in outputs_of(name, a, kw, ret)]))
            return ret
        return w

    if not args.no_record:
        for n in names:
            setattr(ops, n, wrap(n, getattr(ops, n)))
    ref_out = fw().clone()
    ref_log, log = log, []
    torch.cuda.synchronize()
    print(f"forward = {[n for n, _ in ref_log]}", flush=True)
    first_bad = collections.Counter()
    events = []
    for r in range(args.runs):
        if args.gap_every and r and r % args.gap_every == 0:
            torch.cuda.synchronize()
            time.sleep(args.gap)
        log.clear()
        out = fw()
        cur = list(log)
        if args.no_record:
            if not torch.equal(out, ref_out):
                ne = (out != ref_out)
                events.append({"run": r, "n_diff": int(ne.sum()), "rows": int(ne.reshape(-1, ne.shape[-1]).any(1).sum()),
                               "max_abs": float((out - ref_out).abs().max())})
                print(json.dumps(events[-1]), flush=True)
            continue
        bad = None
        for k, ((n0, o0), (n1, o1)) in enumerate(zip(ref_log, cur)):
            assert n0 == n1
            for j, (p, q) in enumerate(zip(o0, o1)):
                if not torch.equal(p, q):
                    ne = p != q
                    p2, q2 = p.reshape(-1, p.shape[-1]), q.reshape(-1, q.shape[-1])
                    rows = (p2 != q2).any(1).nonzero().flatten()
                    cols = (p2 != q2).any(0).nonzero().flatten()
                    bad = {"run": r, "launch": k, "name": n0, "output": j, "n_diff": int(ne.sum()), "n_total": ne.numel(),
                           "rows": int(rows.numel()), "rows_head": rows[:24].tolist(), "cols": int(cols.numel()), "cols_head": cols[:16].tolist(),
                           "max_abs": float((p.double() - q.double()).abs().max()), "max_ref": float(p.abs().max())}
                    break
            if bad:
                break
        if bad:
            first_bad[(bad["launch"], bad["name"])] += 1
            events.append(bad)
            print(json.dumps(bad), flush=True)
    torch.cuda.synchronize()
    print(f"SUMMARY mesh={args.mesh} traj={args.traj} runs={args.runs} deviating_runs={len(events)} first_deviating_launch={dict((f'{k[0]}:{k[1]}', v) for k, v in first_bad.items())}",
          flush=True)


if __name__ == "__main__":
    main()
