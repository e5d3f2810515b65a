#!/usr/bin/env python3
"""Reproducer for the intermittent deviation of bench.py's `precision_modes.fp16x3` leg.

bench.py's sequence, one FRESH process per trial: the 256 x 256 headline model (`hip_graph="rebuild"`) timed as the bench times it,
the two roofline passes, N seconds of GPU idle (the bench's CPU oracle), then ONE 48 x 48 `validation_step` on the SAME model
(first capture of a new shape on a model that holds another capture, first replay).  The 48 x 48 prediction is compared BITWISE with
an eager (`hip_graph=False`) run of a fresh model holding the same weights, and against the oracle's rollout (computed once by the
parent on the CPU and handed over as a file).  The leg is then repeated (second / third replay of the capture) - a wrong value that
persists across replays is baked into the capture, one that goes away is a race inside the replay.

    python scripts/stress_precision_leg.py --trials 200 [--idle 3] [--prior bench|light|none] [--hip-graph rebuild|true|false|auto]
                                           [--poison] [--out gpurun_out/stress]

The parent never touches the GPU (a process that has initialised HIP must not start GPU children by exec; these are plain
subprocesses of a CPU-only parent).  One JSON line per trial on stdout and in <out>/trials.jsonl; a mismatching trial also leaves
<out>/trial_<k>.npz (graph / eager predictions) behind.  Exit code 1 when any trial differs.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def oracle_rollout(path):
    """the bench's weights (torch.manual_seed(0), default init of the mpnn_2d hparams) through the oracle on the 48 x 48 batch"""
    import torch

    import bench
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    torch.manual_seed(0)
    model = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph="rebuild")).eval()
    oracle = O.build("mpnn_2d", types.SimpleNamespace(**bench.HP)).eval()
    oracle.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    small = bench.synthetic_batch(48, 1, 5, "cpu")
    with torch.no_grad():
        want = oracle.validation_step(small)
    torch.save(want, path)


def diff_summary(got, ref, tw=10):
    """where two [B, T, N] predictions differ: counts, the forward call (block of tw steps) that deviates first, the rows"""
    import torch
    d = (got.double() - ref.double()).abs()
    ne = got != ref
    # (NaN != NaN: a poisoned run counts every NaN as a difference, which is what it should do)
    out = {"n_diff": int(ne.sum()), "n_total": ne.numel(), "max_abs": float(torch.nan_to_num(d, nan=float("inf")).max()),
           "nan_in_got": int(torch.isnan(got).sum()), "rel_l2": float((got.double() - ref.double()).norm() / ref.double().norm())}
    if out["n_diff"]:
        T = got.shape[1]
        out["diff_by_call"] = [int(ne[:, c * tw:(c + 1) * tw].sum()) for c in range(T // tw)]
        out["max_abs_by_call"] = [float(torch.nan_to_num(d[:, c * tw:(c + 1) * tw], nan=float("inf")).max()) for c in range(T // tw)]
        rows = ne[:, :tw].any(1)[0].nonzero().flatten()                   # rows of trajectory 0 that differ inside the FIRST call
        out["rows_diff_first_call"] = int(rows.numel())
        out["rows_first_call_head"] = rows[:48].tolist()
        allrows = ne.any(1)[0].nonzero().flatten()
        out["rows_diff_any_call"] = int(allrows.numel())
    return out


def trial(args):
    import torch

    if args.poison:
        os.environ["MAGNET_POISON"] = "nan"
    import bench
    from magnet_amd import debug_poison
    from magnet_amd.models.factory import FACTORY
    debug_poison.install_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rec = {"trial": args.trial, "prior": args.prior, "idle": args.idle, "hip_graph": args.hip_graph, "poison": bool(args.poison)}
    torch.manual_seed(0)
    hg = {"rebuild": "rebuild", "true": True, "false": False, "auto": "auto"}[args.hip_graph]
    model = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph=hg)).to(dev).eval()
    if args.prior != "none":
        W, B = 256, 4
        batch = bench.synthetic_batch(W, B, 1000, dev)
        t_warm = time.perf_counter() + 1.5
        while time.perf_counter() < t_warm:
            model.validation_step(batch, 0)
            torch.cuda.synchronize()
        K = 20 if args.prior == "bench" else 3
        for _ in range(2):
            model.validation_step(batch, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            model.validation_step(batch, 0)
        torch.cuda.synchronize()
        rec["headline_ms"] = round((time.perf_counter() - t0) / K * 1e3, 3)
        if args.prior == "bench":
            bench.edge_roofline(model, batch, dev, W, B)
            bench.scatter_roofline(model, dev, W, B)
    small = bench.synthetic_batch(48, 1, 5, "cpu")
    want = torch.load(args.want) if args.want else None
    if args.idle > 0:
        time.sleep(args.idle)
    small_dev = {k: v.to(dev) for k, v in small.items()}
    preds = []
    for _ in range(3):                                   # first call captures + replays once; the next two replay
        preds.append(model.validation_step(small_dev, 0).cpu())
    torch.cuda.synchronize()
    # the eager twin: a fresh model with the same weights, kernel by kernel
    twin = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph=False)).to(dev).eval()
    twin.load_state_dict(model.state_dict())
    ref = twin.validation_step(small_dev, 0).cpu()
    ref2 = twin.validation_step(small_dev, 0).cpu()
    rec["eager_repeat_equal"] = bool(torch.equal(ref, ref2))
    rec["equal"] = [bool(torch.equal(p, ref)) for p in preds]
    if want is not None:
        rec["rel_l2_vs_oracle"] = [float((p.double() - want.double()).norm() / want.double().norm()) for p in preds]
        rec["eager_rel_l2_vs_oracle"] = float((ref.double() - want.double()).norm() / want.double().norm())
    bad = [k for k, ok in enumerate(rec["equal"]) if not ok]
    if bad or torch.isnan(ref).any():
        rec["diff"] = {f"replay{k}": diff_summary(preds[k], ref) for k in bad}
        rec["nan_in_eager"] = int(torch.isnan(ref).sum())
        import numpy as np
        os.makedirs(args.out, exist_ok=True)
        np.savez_compressed(os.path.join(args.out, f"trial_{args.trial}.npz"), ref=ref.numpy(),
                            **{f"replay{k}": preds[k].numpy() for k in range(3)})
    from magnet_amd.packing import wide
    rec["widened"] = bool(wide())
    print("TRIAL " + json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--idle", type=float, default=3.0, help="seconds of GPU idle before the 48 x 48 leg (the bench: its CPU oracle)")
    ap.add_argument("--prior", default="bench", choices=["bench", "light", "none"],
                    help="what the model did before: the bench's 256^2 legs + roofline passes / a few 256^2 rollouts / nothing")
    ap.add_argument("--hip-graph", default="rebuild", choices=["rebuild", "true", "false", "auto"])
    ap.add_argument("--poison", action="store_true", help="fill every torch.empty of the package with NaN (magnet_amd/debug_poison.py)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress"))
    ap.add_argument("--trial", type=int, default=-1, help="(internal) run ONE trial in this process")
    ap.add_argument("--want", default=None, help="(internal) the oracle's rollout, torch.save'd by the parent")
    ap.add_argument("--timeout", type=float, default=300.0, help="seconds per trial")
    args = ap.parse_args()
    if args.trial >= 0:
        return trial(args)

    os.makedirs(args.out, exist_ok=True)
    want = os.path.join(args.out, "oracle_48x48.pt")
    if not os.path.exists(want):
        oracle_rollout(want)
    log = open(os.path.join(args.out, "trials.jsonl"), "a")
    n_bad = n_ok = 0
    t_start = time.time()
    for k in range(args.trials):
        cmd = [sys.executable, os.path.abspath(__file__), "--trial", str(k), "--want", want, "--idle", str(args.idle), "--prior", args.prior,
               "--hip-graph", args.hip_graph, "--out", args.out] + (["--poison"] if args.poison else [])
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=args.timeout)
            lines = [ln[6:] for ln in p.stdout.splitlines() if ln.startswith("TRIAL ")]
            if p.returncode != 0 or not lines:
                rec = {"trial": k, "error": f"rc {p.returncode}", "stderr_tail": p.stderr[-1500:]}
            else:
                rec = json.loads(lines[-1])
        except subprocess.TimeoutExpired:
            rec = {"trial": k, "error": "timeout"}
        ok = rec.get("equal") == [True, True, True] and rec.get("eager_repeat_equal")
        n_ok += bool(ok)
        n_bad += not ok
        log.write(json.dumps(rec) + "\n")
        log.flush()
        print(json.dumps(rec), flush=True)
    print(f"SUMMARY trials={args.trials} bitwise_equal={n_ok} different_or_failed={n_bad} prior={args.prior} idle={args.idle} "
          f"hip_graph={args.hip_graph} poison={bool(args.poison)} wall_s={time.time() - t_start:.0f}", flush=True)
    log.write(f"# SUMMARY trials={args.trials} bitwise_equal={n_ok} different_or_failed={n_bad} prior={args.prior} idle={args.idle} "
              f"hip_graph={args.hip_graph} poison={bool(args.poison)}\n")
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
