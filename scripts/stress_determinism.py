#!/usr/bin/env python3
"""Repeat-launch determinism of EVERY model path, under timing perturbation (round 5, after DESIGN.md 14.1).

The missing barrier of 14.1 was a timing race that hit one 32-row block in ~2 000 launches and stayed under every tolerance.  No
tolerance test can see that class of bug; a bitwise repeat test can.  This script runs, per case,

    * R training steps' forward + backward on the SAME batch with the SAME seed (no optimizer step: the weights stay) and compares the
      loss and every parameter gradient BITWISE with the first repetition;
    * R free-running validation rollouts, eager (kernel by kernel), compared bitwise with the first;

while a second stream keeps the GPU busy with GEMMs of varying size (other workgroups resident on the CUs, different clocks), and every
few repetitions the device idles (cold clocks).  All kernels of the library are run-to-run deterministic by construction (no atomics on
any data path), so ANY difference is a bug.  Exit code 1 if anything differs.

    python scripts/stress_determinism.py [--reps 100] [--cases c1,c4a,gnn_64,gnn_512,cnn_64,cnn_1d,gnn_1d,c4b,c5,mpnn_256] [--no-perturb]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


ALL_CASES = "c1,cnn_1d,gnn_1d,c4a,gnn_64,gnn_512,cnn_64,c4b,c5,mpnn_256"


def run(cases=ALL_CASES, reps=100, perturbed=True, gap_every=20, gap=0.25, log=print):
    """-> (summary dict, failed: bool); tests/test_regressions_gpu.py runs a short version of this"""
    args = types.SimpleNamespace(cases=cases, reps=reps, no_perturb=not perturbed, gap_every=gap_every, gap=gap)
    print = log                     # noqa: A001
    import torch

    import bench
    from magnet_amd import packing
    from magnet_amd.models.factory import FACTORY
    dev = torch.device("cuda", 0)
    NT = bench.NT

    def batch_1d(name, B, L, nq, seed):
        g = torch.Generator().manual_seed(seed)
        nt = 250
        t = torch.linspace(0, 1, nt)
        if name == "mpnn":
            x = torch.linspace(0, 16, L)
            return {"u": torch.randn(B, L, nt, generator=g), "x": x[None, :, None].repeat(B, 1, 1), "t": (4 * t)[None].repeat(B, 1)}
        full = torch.linspace(-1, 1, L + nq)
        lr_c, hr_c = full[::2][:L], full[1::2][:nq]
        f = lambda c: torch.sin(3 * c[None, :, None] + 4 * t[None, None, :] + torch.rand(B, 1, 1, generator=g) * 6)  # noqa: E731
        return {"t": t[None].repeat(B, 1), "lr_frames": f(lr_c).permute(0, 2, 1)[:, :, None, :].contiguous(),
                "hr_points": f(hr_c).permute(0, 2, 1)[..., None].contiguous(), "coords_hr": hr_c[None, :, None].repeat(B, 1, 1),
                "coords_lr": lr_c[None, :, None].repeat(B, 1, 1)}

    def golden_case(tag, B):
        """inputs + hparams of a committed golden case (data, tests/golden/*.npz), replicated to B different trajectories"""
        import numpy as np
        d = np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz"), allow_pickle=False)
        hp = json.loads(str(d["hparams"]))
        g = torch.Generator().manual_seed(9)
        batch = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("in_")}
        big = {k: v[:1].repeat(B, *([1] * (v.dim() - 1))).clone() for k, v in batch.items()}
        for k in ("lr_frames", "hr_points"):
            big[k] += 0.05 * torch.randn(big[k].shape, generator=g)
        return str(d["name"]), hp, big

    ref = bench.ref_shape_cases()
    cnn1 = golden_case("magnet_cnn_1d", 32)
    gnn1 = dict(bench.MAGNET_HP["magnet_gnn"], time_slice=25, coord_dim=1, noise=0.01)
    cases = {
        "c1": ("mpnn", dict(bench.HP, time_window=25, neighbors=3), lambda train: batch_1d("mpnn", 2, 50, 0, 3)),
        "cnn_1d": (cnn1[0], cnn1[1], lambda train: cnn1[2]),
        "gnn_1d": ("magnet_gnn", gnn1, lambda train: batch_1d("magnet_gnn", 32, 25, 16 if train else 25, 6)),
        "c4a": (ref["c4a"][0], ref["c4a"][1], lambda train: ref["c4a"][2](train)[0]),
        "gnn_64": (ref["gnn_64"][0], ref["gnn_64"][1], lambda train: ref["gnn_64"][2](train)[0]),
        "gnn_512": (ref["gnn_512"][0], ref["gnn_512"][1], lambda train: ref["gnn_512"][2](train)[0]),
        "cnn_64": (ref["cnn_64"][0], ref["cnn_64"][1], lambda train: ref["cnn_64"][2](train)[0]),
        "c4b": ("magnet_cnn_2d", dict(bench.MAGNET_HP["magnet_cnn_2d"]), lambda train: bench.synthetic_magnet_batch("magnet_cnn_2d", 1, 128, 7, "cpu")[0]),
        "c5": ("magnet_gnn", dict(bench.MAGNET_HP["magnet_gnn"]), lambda train: bench.synthetic_magnet_batch("magnet_gnn", 1, 128, 8, "cpu")[0]),
        "mpnn_256": ("mpnn_2d", dict(bench.HP), lambda train: bench.synthetic_batch(256, 4, 1000, "cpu")),
    }

    side = torch.cuda.Stream(dev)
    junk = [torch.randn(n, n, device=dev) for n in (512, 1024, 2048, 3072)]
    tick = [0]

    def perturb():
        if args.no_perturb:
            return
        tick[0] += 1
        a = junk[tick[0] % len(junk)]
        with torch.cuda.stream(side):
            for _ in range(1 + tick[0] % 3):
                torch.mm(a, a)
        if args.gap_every and tick[0] % args.gap_every == 0:
            torch.cuda.synchronize()
            time.sleep(args.gap)

    def flat(ts):
        return torch.cat([t.detach().reshape(-1).float() for t in ts])

    def describe(a, b):
        ne = a != b
        idx = ne.nonzero().flatten()
        return {"n_diff": int(ne.sum()), "n_total": a.numel(), "first": int(idx[0]), "last": int(idx[-1]),
                "max_abs": float((a.double() - b.double()).abs().max()), "max_ref": float(a.abs().max())}

    summary, failed = {}, False
    for key in args.cases.split(","):
        name, hp, mk = cases[key]
        packing.narrow()
        res = {"model": name}
        # ---- training: loss + every gradient
        torch.manual_seed(0)
        tm = FACTORY[name](types.SimpleNamespace(**dict(hp, lr=1e-6))).to(dev).train()
        batch = {k: v.to(dev) for k, v in mk(True).items()}
        params = [p for p in tm.parameters() if p.requires_grad]
        reps_t = max(args.reps // (4 if key in ("mpnn_256", "c5", "c4b") else 1), 5)

        def grads():
            for p in params:
                p.grad = None
            torch.manual_seed(1234)                   # (noise, random start times: the same draw every repetition)
            loss = tm.training_step(batch, 0)
            loss.backward()
            return flat([loss] + [p.grad if p.grad is not None else torch.zeros_like(p) for p in params])

        grads()
        g0 = grads().clone()
        bad = []
        t0 = time.perf_counter()
        for r in range(reps_t):
            perturb()
            g = grads()
            if not torch.equal(g, g0):
                bad.append(dict(describe(g0, g), rep=r))
                print(f"{key} TRAIN DIFF {json.dumps(bad[-1])}", flush=True)
        torch.cuda.synchronize()
        res["train"] = {"reps": reps_t, "differ": len(bad), "grad_elements": g0.numel(), "finite": bool(torch.isfinite(g0).all()),
                        "s": round(time.perf_counter() - t0, 1), "widened": packing.wide(), "events": bad[:5]}
        failed |= bool(bad) or not res["train"]["finite"]
        del tm, params, g0
        packing.narrow()
        # ---- validation rollouts, eager
        torch.manual_seed(0)
        vm = FACTORY[name](types.SimpleNamespace(**hp, hip_graph=False)).to(dev).eval()
        batch = {k: v.to(dev) for k, v in mk(False).items()}
        y0 = vm.validation_step(batch, 0)
        y0 = (y0[0] if isinstance(y0, (tuple, list)) else y0).clone()
        bad = []
        reps_v = max(args.reps // (2 if key in ("mpnn_256", "c5", "c4b") else 1), 5)
        t0 = time.perf_counter()
        for r in range(reps_v):
            perturb()
            y = vm.validation_step(batch, 0)
            y = y[0] if isinstance(y, (tuple, list)) else y
            if not torch.equal(y, y0):
                bad.append(dict(describe(y0.reshape(-1), y.reshape(-1)), rep=r))
                print(f"{key} ROLLOUT DIFF {json.dumps(bad[-1])}", flush=True)
        torch.cuda.synchronize()
        res["rollout"] = {"reps": reps_v, "differ": len(bad), "elements": y0.numel(), "finite": bool(torch.isfinite(y0).all()),
                          "s": round(time.perf_counter() - t0, 1), "widened": packing.wide(), "events": bad[:5]}
        failed |= bool(bad) or not res["rollout"]["finite"]
        del vm, y0
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        summary[key] = res
        print(f"CASE {key} {json.dumps(res)}", flush=True)
    print("SUMMARY " + json.dumps({k: {"train": f"{v['train']['differ']}/{v['train']['reps']}", "rollout": f"{v['rollout']['differ']}/{v['rollout']['reps']}"}
                                    for k, v in summary.items()}) + (" FAILED" if failed else " all bitwise equal"), flush=True)
    return summary, failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--cases", default=ALL_CASES)
    ap.add_argument("--no-perturb", action="store_true")
    ap.add_argument("--gap-every", type=int, default=20)
    ap.add_argument("--gap", type=float, default=0.25)
    a = ap.parse_args()
    _, failed = run(a.cases, a.reps, not a.no_perturb, a.gap_every, a.gap)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
