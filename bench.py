"""bench.py - headline benchmark: mesh-node-updates/s of the free-running 2D Burgers 256^2 rollout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--traj B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric, SURVEY.md 8(d) "metric case"): ``mpnn_2d`` on a regular 256 x 256
mesh (N = 65 536 nodes / trajectory, E ~= 2.16 M edges after the reference's 32(+1)-neighbour cap),
nt = 50, time_window = 10 -> 4 forward calls = 40 predicted time steps per rollout, B independent
trajectories per GPU (weak scaling: per-GPU work is fixed).  One "step" = one complete free-running
rollout (validation_step semantics, no grad) of the B trajectories, INCLUDING one radius-graph + CSR
build (the topology cache is dropped before every step; inside a rollout the graph is reused because
the mesh does not move - the reference rebuilds an identical graph every call).  Inputs are synthetic
and already resident in HBM when the clock starts; weights are random-init (torch default init,
seed 0).  value = nodes x predicted steps x trajectories (all ranks) / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline      - dominant kernel (the fused MPNN edge kernel; `roofline.kernel` names the generation that ran): algorithmic
                  flops per launch (2*E*128*128, the message_net_2 GEMM; message_net_1 is factorised per node, see
                  DESIGN.md) / mean launch time measured with HIP events on the launch stream, in an untimed pass that runs
                  BEFORE the training leg.  `clock_mhz` = the shader clock sampled by a one-wave probe kernel on a second
                  stream while that pass runs (the part is power limited: ~2.1 GHz, not the 2.4 GHz `peak` assumes);
                  `peak_at_clock` / `frac_at_clock` rescale to it.  The kernel runs at the near-additive sum of its matrix-pipe
                  time, its LDS weight-fragment reads, its gathers and its transcendentals, so the fraction is
                  the matrix pipe's share of that sum.  The
                  kernel evaluates every fp32 product through a split into 16-bit pieces on the matrix cores
                  (default fp16x3: 2 pieces, 3 fp16 MFMA products, ~2^-22 relative; bf16x6: 3 pieces, 6 products), so its
                  ceiling is the dense 16-bit MFMA peak / 3 = 833 TFLOP/s of fp32-equivalent work (`peak`);
                  `executed_mfma_tflops` / 2500 is the same fraction.
  roofline_scatter - the STAND-ALONE CSR segment-mean ("scatter mean") kernel on the same graph's
                  [E,128] messages: algorithmic bytes E*H*4 + N*H*4 + 4*(N+1) per launch / mean
                  launch time, against the 8 TB/s HBM3E peak.  No model calls it on the rollout path (the mean is fused
                  into the edge kernels); it is the drop-in for torch_scatter when messages are materialised.
  precision_modes - the same workload and weights in the other GEMM modes (N = 1 only): ms_per_step of `bf16x6`
                  (three bf16 pieces, exact 24-bit operands, full fp32 range) and `fp32` (fp32 MFMA), and every mode's rollout
                  relative L2 against the oracle on a 48 x 48 mesh.  The reference computes in fp32
                  (configs/trainer/default.yaml:5); the headline mode fp16x3 holds ~22-bit operands.
  cpu_baseline  - the oracle (torch-CPU restatement of the reference op sequence) timed on this
                  box's host cores on a bounded sample: `value` = ONE free-running rollout of one 256 x 256 trajectory (the
                  metric's mesh, ~50 s); thread count from a one-forward sweep over {physical cores / 2, physical cores} on a
                  128 x 128 mesh, whose 1 + 3 rollouts are reported beside it (`value_128x128`); forward calls only, graph
                  construction excluded; rank 0, N = 1 only.
  dist          - what the process group saw: backend, world size, per-rank ms_per_step; `training_step.allreduce_ms` =
                  the gradient all-reduce of one step (HIP events).
  traffic (inside both roofline objects) - HBM bytes per launch from the newest committed PMC measurement of this same
                  command (profiles/r*/traffic_v*.json <- scripts/collect_traffic.py <- two separate rocprofv3 --pmc passes,
                  FETCH_SIZE doubled per the gfx950 correction); null when the workload differs from the recorded one.
                  Edge kernel 0.62 GB (0.75 GB algorithmic; 1.58 GB before the XCD-aware tile ranges), scatter-mean 4.57 GB =
                  1.00x algorithmic.
  c4b, c5       - the other two 256^2 cases of SURVEY.md 8(d) as complete objects of the same contract (value, ms_per_step, roofline of
                  their dominant kernel - the InteractionNetwork edge chain of the joint graph -, training_step): `magnet_cnn_2d` 64^2 LR
                  grid -> 256^2 queries and `magnet_gnn` 256 irregular LR nodes -> 256^2 queries, one trajectory per GPU.
                  `ms_per_step` replays the rollout from one hipGraph with the topology of the first call; `ms_per_step_rebuild` is the
                  eager loop with radius search / CSR / k-NN rebuilt every step (what the reference does and what the headline's
                  figure includes), `ms_per_step_eager` the same loop with the cached topology.
  c1            - BASELINE.json configs[0] on the GPU: `mpnn` 1-D, 2 x 50 nodes, 9-forward rollout replayed from one hipGraph (ms per
                  rollout, and the kernel-by-kernel time beside it); rank 0 only.
  ref_shapes    - the shapes the reference's own run scripts train and validate on (64 x 64 meshes at batch 4 / 8 / 32, 512-point
                  irregular meshes at batch 32; bench_ref_shapes): training step ms / C-ABI calls / peak memory, validation rollout
                  enqueued kernel by kernel and replayed from one hipGraph; rank 0, N = 1 only.
  training_step - the same workload trained: rollout with tape, HIP backward
                  kernels, ONE flat gradient all-reduce over RCCL, fused Adam; ms per step and the node-updates/s it
                  corresponds to (all ranks), `--train-steps` timed steps after two warm-up steps (mpnn_2d) / three (`c4b`, `c5`:
                  `warmup_step_ms`; lr = 1e-6 there as in `ref_shapes`; a timed region during which torch's allocator grew its pool
                  by >= 256 MB is timed again, `discarded_ms_per_step`).  Never part of `value`.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HP = dict(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1", hidden_features=128, hidden_layer=5,
          time_window=10, teacher_forcing=False, neighbors=4)     # configs/model/mpnn_2d.yaml
NT = 50
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0
SPLIT_PRODUCTS = {"fp16x3": 3, "bf16x6": 6, "bf16": 1}   # MFMA products per multiply in k_mpnn_edge7 / k_mpnn_edge6 (bf16: reduced precision)
HBM_PEAK_GBS = 8000.0
PEAK_CLOCK_MHZ = 2400.0                                     # the clock the 2.5 PFLOP/s dense peak is quoted at


def synthetic_batch(W, B, seed, device):
    """u = sum of 4 low-frequency sinusoids + 0.01 N(0,1) on linspace(0,1,W)^2 ('ij'), t = linspace(0,1,50)."""
    g = torch.Generator().manual_seed(seed)
    lin = torch.linspace(0, 1, W)
    xy = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)
    t = torch.linspace(0, 1, NT)
    u = torch.zeros(B, W * W, NT)
    for b in range(B):
        for _ in range(4):
            k = torch.rand(2, generator=g) * 2.5 + 0.5
            w, ph, a = torch.rand(1, generator=g) * 1.5 + 0.5, torch.rand(1, generator=g) * 6.28, torch.rand(1, generator=g) * 0.7 + 0.3
            u[b] += a * torch.sin((xy @ k)[:, None] * 3.0 + w * t[None, :] * 4.0 + ph)
    u += 0.01 * torch.randn(u.shape, generator=g)
    return {"u": u.to(device), "x": xy[None].repeat(B, 1, 1).to(device), "t": t[None].repeat(B, 1).to(device)}


OPT = dict(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1")
MAGNET_HP = {   # configs/model/magnet_cnn_2d.yaml (C4b) and configs/model/magnet_gnn.yaml with the 2-D data of C5
    "magnet_cnn_2d": dict(OPT, time_slice=10, latent_dim=32, num_message_passing_steps=10, mlp_layers=4, mlp_hidden=64, radius=0.1,
                          scales=1, n_chan=128, kernel_size=3, res_scale=1, res_layers=16, teacher_forcing=False,
                          interpolation="area"),
    "magnet_gnn": dict(OPT, time_slice=10, latent_dim=128, num_message_passing_steps=5, mlp_layers=4, mlp_hidden=128, radius=0.08,
                       n_chan=128, teacher_forcing=False, codec_neighbors=4, noise=0, interpolation="area"),
}


def synthetic_magnet_batch(name, B, Wq, seed, device):
    """SURVEY.md 8(d) C4b / C5 inputs: LR frames + HR query values over nt = 50 steps, queries = make_coord([Wq, Wq])."""
    from magnet_amd.utils import make_coord
    g = torch.Generator().manual_seed(seed)
    hr_c = make_coord([Wq, Wq])[None].repeat(B, 1, 1)
    t = torch.linspace(0, 1, NT)[None].repeat(B, 1)
    hr = torch.randn(B, NT, Wq * Wq, 1, generator=g)
    if name == "magnet_cnn_2d":
        W = 64
        batch = {"t": t, "lr_frames": torch.randn(B, NT, 1, W, W, generator=g), "hr_points": hr, "coords": hr_c,
                 "cells": torch.full((B, Wq * Wq, 2), 2.0 / Wq)}
        n_lr = W * W
    else:
        pts = torch.clamp(torch.randn(512, 2, generator=g) * 0.25, -1, 1)
        pts = 2 * (pts - pts.min(0).values) / (pts.max(0).values - pts.min(0).values) - 1
        lr_c = pts[::2][None].repeat(B, 1, 1)
        n_lr = lr_c.shape[1]
        batch = {"t": t, "lr_frames": torch.randn(B, NT, 1, n_lr, generator=g), "hr_points": hr, "coords_hr": hr_c,
                 "coords_lr": lr_c}
    return {k: v.to(device) for k, v in batch.items()}, n_lr + Wq * Wq


DTYPE_NOTE = {"fp16x3": "f32-class, full fp32 range (GEMM operands as 2 fp16 pieces, 3 MFMA products, ~2^-22 relative, fp32 accumulate; a "
                        "rollout whose values or weights reach 65504 is repeated in the bf16x6 split in the same call - "
                        "`precision_modes.bf16x6` is what such data runs at)",
              "bf16x6": "f32-class (GEMM operands as 3 bf16 pieces, 6 MFMA products, ~2^-24 relative, fp32 accumulate; full fp32 range)",
              "fp32": "f32 (fp32 MFMA)",
              "bf16": "bf16 - REDUCED precision (inputs and weights rounded to one bf16 piece, fp32 accumulate): not the headline, "
                      "see deviation_vs_f32_path"}


def measured_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the newest committed PMC measurement of this same bench command
    (profiles/r*/traffic_v*.json, written by scripts/collect_traffic.py from two separate rocprofv3 --pmc passes -
    FETCH_SIZE and WRITE_SIZE cannot share a pass - with the gfx950 FETCH_SIZE correction).  bench.py cannot run the
    profiler on itself, so this is a recorded measurement, not a live one; None when no file is present."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_v*.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(os.path.dirname(f)) + os.path.basename(f))])
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
    except Exception:  # noqa: BLE001
        return None, None
    return (k["traffic"], os.path.relpath(files[-1], ROOT)) if k else (None, None)


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """distinct (physical id, core id) pairs in /proc/cpuinfo, restricted to this process's affinity mask"""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
        return max(1, len(seen))
    except Exception:  # noqa: BLE001
        return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(W=128, timed=3, metric_mesh=256):
    """the oracle's free-running rollout of ONE trajectory on a W x W mesh on the host cores, fp32,
    no_grad, 1 warm-up rollout + ``timed`` timed rollouts, median; the radius-graph build is timed separately (the oracle
    builds it by brute force) and excluded from the rollout time, as the reference's own CPU numbers in BASELINE.md are.
    Thread count: the faster of {physical / 2, physical} on one forward call (torch's default = logical CPUs oversubscribes an
    SMT part by an order of magnitude); `cores` is the count the timed rollouts used."""
    from oracle import models as O
    torch.manual_seed(0)
    hp = types.SimpleNamespace(**HP)
    model = O.build("mpnn_2d", hp).eval()
    batch = synthetic_batch(W, 1, 0, "cpu")
    u = batch["u"].permute(0, 2, 1)
    x, t = batch["x"], batch["t"]
    tw = hp.time_window
    t0 = time.perf_counter()
    graph0 = model._build_graph(u[:, :tw], t, x, [tw - 1])
    t_graph = time.perf_counter() - t0
    n_calls = (NT - tw) // tw

    def rollout(calls=n_calls):
        g = copy.copy(graph0)                                             # same edge list, fresh x / pos per rollout
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(calls):
                y = model(g, x[0, -1], t[0, -1], t[0][1] - t[0][0])
                g.x = y                                                   # same mesh: reuse the edge list
                g.pos = torch.cat([torch.full((W * W, 1), float(t[0, (i + 2) * tw - 1])), g.pos[:, 1:]], 1)
        return time.perf_counter() - t0

    logical, phys = len(os.sched_getaffinity(0)), physical_cores()
    default_threads = torch.get_num_threads()
    sweep = {}
    # (torch's default, one thread per LOGICAL cpu, is not a candidate on an SMT box: measured 44 s per forward at 256 threads against 3.2 s at
    # 64 on the pool's EPYC 9575F - trying it would cost the bench a minute and a half)
    for n in sorted({max(1, phys // 2), phys} | ({logical} if logical <= phys else set())):
        torch.set_num_threads(n)
        rollout(1)                                                        # thread pool / allocator warm-up at this size
        sweep[n] = round(rollout(1), 2)                                   # one forward call
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = []
    for k in range(1 + timed):
        dt = rollout()
        if k > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    out = {"value": W * W * tw * n_calls / med, "unit": "mesh-node-updates/s", "cores": best, "physical_cores": phys,
           "logical_cpus": logical, "one_forward_s_by_threads": {str(k): v for k, v in sweep.items()},
           "cpu_model": cpu_model_string(), "kind": "port", "rollouts_s": [round(v, 2) for v in times], "graph_build_s": round(t_graph, 2),
           "sample": f"oracle mpnn_2d, {W}x{W} mesh, 1 trajectory, 1 warm-up + {timed} timed free-running rollouts of {n_calls} forward "
                     f"calls each, median {med:.1f} s (radius-graph build {t_graph:.1f} s timed separately, excluded), fp32, no_grad, "
                     f"torch {torch.__version__} CPU, {best} threads (best of {sorted(sweep)} on one forward call; {phys} physical "
                     f"cores, {logical} logical CPUs)"}
    if metric_mesh and metric_mesh != W:
        # the metric's own mesh (BASELINE.md section 3 promised it): ONE timed rollout of one 256 x 256 trajectory at the thread count
        # found above (~50 s; the thread pool and the allocator are warm from the smaller mesh), the graph built by the oracle's own
        # brute-force search (timed, excluded as above).  `value` is this figure; the smaller mesh's stays beside it.
        Wm = metric_mesh
        bm = synthetic_batch(Wm, 1, 0, "cpu")
        um, xm, tm_ = bm["u"].permute(0, 2, 1), bm["x"], bm["t"]
        t0 = time.perf_counter()
        gm = model._build_graph(um[:, :tw], tm_, xm, [tw - 1])
        tg = time.perf_counter() - t0
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(n_calls):
                y = model(gm, xm[0, -1], tm_[0, -1], tm_[0][1] - tm_[0][0])
                gm.x = y
                gm.pos = torch.cat([torch.full((Wm * Wm, 1), float(tm_[0, (i + 2) * tw - 1])), gm.pos[:, 1:]], 1)
        dtm = time.perf_counter() - t0
        out[f"value_{W}x{W}"], out["value"] = out["value"], Wm * Wm * tw * n_calls / dtm
        out["metric_mesh_rollout_s"], out["metric_mesh_graph_build_s"] = round(dtm, 2), round(tg, 2)
        out["sample"] = (f"oracle mpnn_2d, ONE free-running rollout ({n_calls} forward calls) of one {Wm}x{Wm} trajectory - the metric's mesh - "
                         f"in {dtm:.1f} s at {best} threads (radius-graph build {tg:.1f} s excluded); thread count and a median of {timed} from: "
                         + out["sample"])
    torch.set_num_threads(default_threads)
    return out


class ClockProbe:
    """Shader clock under load: a one-wave kernel on a side stream reads the shader-clock counter and the constant-rate wall
    clock for ``wall_us`` microseconds while the main stream runs the workload (magnet_clock_probe, include/magnet_hip.h).
    (The driver's sysfs figure that used to be reported beside it named another device on some boxes of the pool - 107 MHz in the
    round-3 record - and is gone.)"""

    def __init__(self, dev):
        import ctypes as C
        self.C, self.dev = C, dev
        self.out = torch.zeros(2, dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.khz = C.c_int(0)
        self.samples = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def sample(self, wall_us):
        """enqueue one probe now (call right before the work it should overlap); read it with ``collect`` after a sync"""
        from magnet_amd import ops
        self.stream.wait_stream(torch.cuda.current_stream())
        rc = ops.L.lib().magnet_clock_probe(self.C.c_void_p(self.out.data_ptr()), int(wall_us), self.C.byref(self.khz),
                                            self.C.c_void_p(self.stream.cuda_stream))
        return rc == 0

    def collect(self):
        torch.cuda.synchronize()
        c, w = (int(v) for v in self.out.tolist())
        if w > 0 and self.khz.value > 0:
            mhz = c / w * self.khz.value / 1e3
            if 300.0 < mhz < 4000.0:                       # a counter that is not the shader clock would read as the wall clock rate
                self.samples.append(mhz)

    def result(self):
        return sum(self.samples) / len(self.samples) if self.samples else None


def edge_roofline(model, batch, dev, W, B, passes=3):
    """HIP-event time of every launch of the fused MPNN edge kernel over ``passes`` untimed rollouts (events are recorded on the
    stream the kernel is launched on: the C entry points take torch's current stream), with the shader clock sampled
    alongside.  Returns the `roofline` object."""
    from magnet_amd import ops
    csr = model._topo.csr
    E, N, H = csr.E, csr.N, 128
    ev = []
    lyr0 = model._packed().layers[0]
    wdt = lyr0.msg2[0].dtype
    entries = [e for e in (ops.EDGE_ENTRY[wdt], ops.EDGE_ENTRY_L2E.get(wdt)) if e]   # the model picks one of them
    entries += [e for e in ops.EDGE_ENTRY_EXTRA if hasattr(ops.L.lib(), e)] if wdt == torch.float16 else []
    origs = {e: getattr(ops.L.lib(), e) for e in entries}
    used = set()

    def timed(name, orig):
        def timed_edge(*a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = orig(*a)
            e.record()
            ev.append((s, e))
            used.add(name)
            return rc
        return timed_edge

    for e_, o_ in origs.items():
        setattr(ops.L._lib, e_, timed(e_, o_))
    probe = ClockProbe(dev)
    mode_was = model.hip_graph
    model.hip_graph = False                             # kernel by kernel: a replayed hipGraph does not pass through the timed wrappers
    try:
        with probe:
            for _ in range(passes):
                probe.sample(20000)                     # ~20 ms of the ~25 ms rollout
                model.validation_step(batch, 0)
                probe.collect()
    finally:
        model.hip_graph = mode_was
        for e_, o_ in origs.items():
            setattr(ops.L._lib, e_, o_)
    if not ev:
        raise RuntimeError("edge_roofline: the model dispatched its edge op through an entry point that is not wrapped here "
                           f"(wrapped: {sorted(origs)}); add it to ops.EDGE_ENTRY_EXTRA")
    ms = sum(s.elapsed_time(e) for s, e in ev) / len(ev)
    flops = 2.0 * E * H * H
    ach = flops / (ms * 1e-3) / 1e12
    mode = model.edge_gemm
    nprod = SPLIT_PRODUCTS.get(mode, 1)
    peak = MFMA_BF16_PEAK_TFLOPS / nprod if mode in SPLIT_PRODUCTS else MFMA_F32_PEAK_TFLOPS
    kname = ops.EDGE_KERNEL_OF_ENTRY.get(next(iter(used)) if len(used) == 1 else None, "k_mpnn_edge?")
    clk = probe.result()
    roof = {"kernel": kname, "bound": "mfma",
            "achieved": round(ach, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "launch_ms": round(ms, 4), "launches_timed": len(ev), "flops_per_launch": flops, "gemm_mode": mode,
            "algorithmic_hbm_bytes": 3 * N * H * 4 + 2 * E * 4 + 4 * (N + 1),
            "executed_mfma_tflops": round(ach * nprod, 1), "mfma_16bit_dense_peak": MFMA_BF16_PEAK_TFLOPS,
            "fp32_mfma_peak_for_reference": MFMA_F32_PEAK_TFLOPS,
            "clock_mhz": round(clk, 1) if clk else None, "peak_clock_mhz": PEAK_CLOCK_MHZ,
            "note": "fp32-class GEMM: operands split into 16-bit pieces on the matrix cores "
                    f"({nprod} MFMA products per fp32 product, fp32 accumulate); peak = 2500 / {nprod}.  The kernel runs at the near-additive "
                    "sum of its matrix-pipe time, its LDS weight-fragment reads, its gathers and its two Swish per edge-feature "
                    "(ablations: profiles/r4/edge_rows.txt, DESIGN.md 11.1; SQ counters of generation 14: profiles/r5/pmc_stream_sq_counters.txt, "
                    "DESIGN.md 14.12 - a SIMD executes an instruction of one of its two waves 98 % of the time), so the fraction is the matrix "
                    "pipe's share of that sum; "
                    "measured before the training leg"}
    if clk:
        roof["peak_at_clock"] = round(peak * clk / PEAK_CLOCK_MHZ, 1)
        roof["frac_at_clock"] = round(ach / (peak * clk / PEAK_CLOCK_MHZ), 4)
    if W == 256 and B == 4 and mode in SPLIT_PRODUCTS:     # the recorded PMC passes are of this exact workload
        roof["traffic"], roof["traffic_source"] = measured_traffic(kname)
    return roof


def scatter_roofline(model, dev, W, B):
    from magnet_amd import ops
    csr = model._topo.csr
    E, N, H = csr.E, csr.N, 128
    msg = torch.randn(E, H, device=dev)
    out = torch.empty(N, H, device=dev)
    for _ in range(3):
        ops.segment_mean(msg, csr.rowptr, N, out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    s.record()
    for _ in range(reps):
        ops.segment_mean(msg, csr.rowptr, N, out)
    e.record()
    torch.cuda.synchronize()
    ms_sc = s.elapsed_time(e) / reps
    nbytes = E * H * 4 + N * H * 4 + 4 * (N + 1)
    gbs = nbytes / (ms_sc * 1e-3) / 1e9
    roof_sc = {"kernel": "k_segment_mean<128>", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launch_ms": round(ms_sc, 4),
               "bytes_per_launch": nbytes,
               "note": "stand-alone op (magnet_segment_mean_f32, the drop-in for torch_scatter's scatter-mean on materialised messages); "
                       "not on the rollout path - there the mean is fused into the edge kernels"}
    if W == 256 and B == 4:
        roof_sc["traffic"], roof_sc["traffic_source"] = measured_traffic("k_segment_mean<128>")
    return roof_sc


def precision_modes(model, batch, headline_ms, dev, steps=3):
    """the same workload and weights in every full-precision GEMM mode: ms per rollout (graph rebuilt every step, as the headline)
    and the rollout's relative L2 against the oracle on a 48 x 48 mesh (one trajectory, same weights)"""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    oracle = O.build("mpnn_2d", types.SimpleNamespace(**HP)).eval()
    oracle.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    small = synthetic_batch(48, 1, 5, "cpu")
    with torch.no_grad():
        want = oracle.validation_step(small)
    small_dev = {k: v.to(dev) for k, v in small.items()}
    last0 = small["u"][:, :, HP["time_window"] - 1].double().unsqueeze(1)          # [B, 1, N]: the last input step
    out = {}
    for mode in ("fp16x3", "bf16x6", "fp32"):
        if mode == model.edge_gemm:
            m, ms = model, headline_ms
        else:
            m = FACTORY["mpnn_2d"](types.SimpleNamespace(**dict(HP, edge_gemm=mode, gemm=mode), hip_graph=False)).to(dev).eval()
            m.load_state_dict(model.state_dict())
            m.validation_step(batch, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                m.drop_topology_cache()
                m.validation_step(batch, 0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        got = m.validation_step(small_dev, 0).cpu()
        m.drop_topology_cache()
        err = got.double() - want.double()
        out[mode] = {"ms_per_step": round(ms, 3), "dtype": DTYPE_NOTE[mode],
                     "rollout_rel_l2_vs_oracle_48x48": float(err.norm() / want.double().norm()),
                     # the prediction is `last value + dt * network`: the carried-over input dominates its norm, so the change against
                     # the last input step (SURVEY.md 7 "gate on the increment too") and the largest single error are reported as well
                     "increment_rel_l2_vs_oracle_48x48": float(err.norm() / (want.double() - last0).norm()),
                     "max_abs_err_vs_oracle_48x48": float(err.abs().max())}
        if m is not model:
            del m
            torch.cuda.empty_cache()
    # the three modes compute the same fp32-class result (5e-8 each, bit-stable from run to run): a split mode that strays from the
    # widest one is a defect of THIS run, not rounding - say so in the record and let main() exit non-zero
    ref_err = out["bf16x6"]["rollout_rel_l2_vs_oracle_48x48"]
    bad = [mo for mo in out if not out[mo]["rollout_rel_l2_vs_oracle_48x48"] <= max(4.0 * ref_err, 4e-7)]
    out["check"] = {"ok": not bad, "rule": "every mode's 48 x 48 rollout error <= max(4 x the bf16x6 figure, 4e-7)", "failed": bad}
    return out


def bench_magnet(name, B, K, Wm, Wq, train_steps, dev, rank, world, dist, max_over_ranks):
    """C4b (magnet_cnn_2d, 64^2 LR grid -> 256^2 queries) / C5 (magnet_gnn, 256 LR nodes -> 256^2 queries): free-running
    validation_step rollout (4 forwards = 40 predicted steps), nodes = LR nodes + queries; same JSON contract.
    Returns the JSON object on rank 0 (None elsewhere); every rank takes part in the barriers."""
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    torch.manual_seed(0)
    hp = MAGNET_HP[name]
    # the rollout replayed from one hipGraph (the topology is cached across steps either way; same time as the kernel-by-kernel loop
    # on an idle host - 24.0 / 34.9 ms both ways - but independent of the host's enqueue rate).  MAGNET_BENCH_GRAPH=0: eager
    graph_kw = {"hip_graph": True} if os.environ.get("MAGNET_BENCH_GRAPH", "1") == "1" else {}
    model = FACTORY[name](types.SimpleNamespace(**hp, **graph_kw)).to(dev).eval()
    batch, nodes = synthetic_magnet_batch(name, B, Wq, 2000 + rank, dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_warm = time.perf_counter() + float(os.environ.get("MAGNET_BENCH_PREWARM_S", "1.5"))       # device warm-up (see main): untimed
    while time.perf_counter() < t_warm:
        model.validation_step(batch, 0)
        torch.cuda.synchronize()
    for _ in range(Wm):
        model.validation_step(batch, 0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        model.validation_step(batch, 0)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dev)
    T = hp["time_slice"]
    pred_steps = (NT - T) // T * T
    value = world * B * nodes * pred_steps * K / elapsed
    # ---- the same rollout with the graphs REBUILT every step, as the reference does per forward call (models/magnet_gnn.py:293, :247) and
    # as the headline leg does per rollout: radius search over LR nodes and over LR nodes + queries, CSR, k-NN - kernel by kernel (a
    # captured rollout has its topology baked in), with the cached-topology figure of the same eager loop beside it
    rebuild = {}
    mode_was = model.hip_graph
    model.hip_graph = False
    try:
        for label, drop in (("ms_per_step_eager", False), ("ms_per_step_rebuild", True)):
            model.validation_step(batch, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                if drop:
                    model.drop_topology_cache()
                model.validation_step(batch, 0)
            torch.cuda.synchronize()
            rebuild[label] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    finally:
        model.hip_graph = mode_was
    roof = None
    if rank == 0:      # dominant kernel: the InteractionNetwork edge MLP (5 Linears + LayerNorm + CSR mean) of the joint graph
        D, H = hp["latent_dim"], hp["mlp_hidden"]
        mode = __import__("magnet_amd.packing", fromlist=["default_gemm"]).default_gemm()
        entry = {"fp16x3": "magnet_mlp_fp16x3_f32", "bf16x6": "magnet_mlp_bf16x6_f32", "fp32": "magnet_mlp_f32",
                 "bf16": "magnet_mlp_bf16_f32"}[mode]
        # (the 64-wide chains of the fp16x3 mode run the one-accumulator entry, packing.chain_gemm)
        entries = [entry] + (["magnet_mlp_fp16x3s_f32"] if mode == "fp16x3" else [])
        origs = {e: getattr(ops.L.lib(), e) for e in entries}
        ev = []

        def timed_entry(orig):
            def timed(a, nl, ht, ot, act, la, ln, epi, st):
                big = nl == 5 and epi == ops.EPI_SEGMEAN
                if big:
                    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_.record()
                rc = orig(a, nl, ht, ot, act, la, ln, epi, st)
                if big:
                    e_.record()
                    ev.append((s_, e_, a._obj.M))
                return rc
            return timed

        for e, o in origs.items():
            setattr(ops.L._lib, e, timed_entry(o))
        mode_was = model.hip_graph
        model.hip_graph = False                             # kernel by kernel: a replay does not pass through the timed wrapper
        try:
            model.validation_step(batch, 0)
            torch.cuda.synchronize()
        finally:
            model.hip_graph = mode_was
            for e, o in origs.items():
                setattr(ops.L._lib, e, o)
        Emax = max(m for _, _, m in ev)
        sel = [(s_, e_) for s_, e_, m in ev if m == Emax]           # the joint-graph launches (the LR-only graph is tiny)
        ms = sum(s_.elapsed_time(e_) for s_, e_ in sel) / len(sel)
        # flops of the formulation EXECUTED: with hidden <= latent the first Linear is factored per node (its x_i / x_j thirds
        # run once per node, magnet_amd/models/magnet_common.py:pack_interaction_edge), so the per-edge chain has K0 = D
        factored = H <= D
        flops = 2.0 * Emax * ((1 if factored else 3) * D * H + 3 * H * H + H * D)
        flops_reference = 2.0 * Emax * (3 * D * H + 3 * H * H + H * D)
        ach = flops / (ms * 1e-3) / 1e12
        nprod = SPLIT_PRODUCTS.get(mode, 1)
        peak = MFMA_BF16_PEAK_TFLOPS / nprod if mode in SPLIT_PRODUCTS else MFMA_F32_PEAK_TFLOPS
        roof = {"kernel": f"k_mlp_bf16<5 Linears, hidden {H}, LayerNorm, CSR-mean epilogue>", "bound": "mfma",
                "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": None, "launch_ms": round(ms, 4), "launches_timed": len(sel), "flops_per_launch": flops,
                "gemm_mode": __import__("magnet_amd.packing", fromlist=["chain_gemm"]).chain_gemm(5, H, mode), "edges": Emax, "first_layer_factored_per_node": factored,
                "flops_per_launch_reference_formulation": flops_reference}
        if Wq == 256 and B == 1 and mode in SPLIT_PRODUCTS:      # the recorded PMC passes are of this exact workload
            roof["traffic"], roof["traffic_source"] = measured_traffic(f"{name}:edge_mlp")
    # ---- training step (autograd Functions with HIP backward kernels, one gradient all-reduce, fused Adam) - unfused,
    # the training step; reported beside the rollout, never mixed into `value`
    train = None
    if train_steps > 0:
        from magnet_amd.sharding import allreduce_gradients
        # lr = 1e-6 (as in the ref_shapes legs): a few Adam steps at the configs' 1e-3 on RANDOM fields grow EDSR's feature maps past 65504
        #  and the step would be timed in the bf16x6 fallback - which then, being process-wide, would also be what every
        # later leg of this line runs in.  The work of a step does not depend on the learning rate; `widened_to_bf16x6` says what ran.
        tmodel = FACTORY[name](types.SimpleNamespace(**dict(hp, lr=1e-6))).to(dev).train()
        opt = tmodel.configure_optimizers()["optimizer"]

        ar_events = []

        def train_step():
            opt.zero_grad(set_to_none=True)
            loss = tmodel.training_step(batch, 0)
            loss.backward()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            allreduce_gradients(tmodel.parameters())
            e_.record()
            ar_events.append((s_, e_))
            opt.step()
            return loss

        # three warm-up steps: the caching allocator still grows its pool after step 0 - hipMalloc of GBs at 205 GiB in use, 80-120 ms in the
        # first process on a box (profiles/r4/steps_timeline_c5.txt; `warmup_step_ms` keeps what they took) - a training run does the
        # steady-state step thousands of times
        warm_ms = []
        for i in range(3):
            torch.cuda.synchronize()
            tw = time.perf_counter()
            l_ = train_step()
            torch.cuda.synchronize()
            warm_ms.append(round((time.perf_counter() - tw) * 1e3, 1))
            if i == 0:
                first = float(l_.detach())
        # (a full collection of Python's cyclic GC walks the ~10^5 objects of a step's autograd graph: 50-100 ms of host time whenever its
        #  counters say so.  Collect now and freeze what exists, so that the timed steps see young-generation collections only)
        import gc
        gc.collect()
        gc.freeze()
        # a timed region during which torch's allocator had to grow its pool (hipMalloc of GBs with ~210 GB in use: 50-100 ms, erratic in
        # the first process on a box, profiles/r4/c5_leg_repeat.txt) is not the steady state: timed again, at most twice; what was
        # discarded is reported (`discarded_ms_per_step`)
        discarded = []
        for attempt in range(3):
            reserved0 = torch.cuda.memory_reserved(dev)
            barrier()
            ar_events.clear()
            t0 = time.perf_counter()
            for _ in range(train_steps):
                last = train_step()
            barrier()
            t_el = max_over_ranks(time.perf_counter() - t0, dev)
            # the decision is COLLECTIVE (every attempt contains barriers and the gradient all-reduce): the MAX over the ranks of the pool
            # growth decides for all of them, so that no rank leaves the loop while another re-enters it
            grew = max_over_ranks(float(torch.cuda.memory_reserved(dev) - reserved0), dev)
            if grew < (256 << 20) or attempt == 2:      # (small-block pools keep growing by a few MB)
                break
            discarded.append(round(t_el / train_steps * 1e3, 3))
        gc.unfreeze()
        train = {"ms_per_step": round(t_el / train_steps * 1e3, 3), "steps": train_steps, "warmup": 3, "warmup_step_ms": warm_ms,
                 "discarded_ms_per_step": discarded,
                 "value": world * B * nodes * pred_steps * train_steps / t_el, "unit": "mesh-node-updates/s",
                 "what": "rollout through autograd Functions (HIP forward + backward kernels; hidden Linears walked back in one kernel, first Linear at source-row granularity, tape kept in HBM when it fits) + one gradient all-reduce + fused Adam",
                 "loss_first_last": [round(first, 6), round(float(last.detach()), 6)],
                 "widened_to_bf16x6": __import__("magnet_amd.packing", fromlist=["wide"]).wide(),
                 "peak_mem_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "lr": 1e-6,
                 # the one flat gradient all-reduce of the step (10.3 MB for magnet_gnn, 21.5 MB for magnet_cnn_2d), HIP events
                 # around it; and the DDP invariant - the same parameters on every rank after the steps
                 "allreduce_ms": round(sum(s_.elapsed_time(e_) for s_, e_ in ar_events) / max(len(ar_events), 1), 4),
                 "allreduce_bytes": 4 * sum(p_.numel() for p_ in tmodel.parameters()),
                 "param_abs_sum_max_over_ranks": max_over_ranks(float(sum(p_.detach().double().abs().sum() for p_ in tmodel.parameters())), dev),
                 "param_abs_sum_min_over_ranks": -max_over_ranks(-float(sum(p_.detach().double().abs().sum() for p_ in tmodel.parameters())), dev)}
        del tmodel, opt, last, train_step
        __import__("magnet_amd.packing", fromlist=["narrow"]).narrow()   # (a leg that left the fp16 range must not hand the NEXT one the wider split)
    del model, batch
    # the autograd graph of the last training step (and the ~200 GB of backward tape its Functions kept) sits in
    # reference cycles until the cyclic collector runs; the next leg decides how much tape IT may keep from what is allocated now
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    if rank != 0:
        return None
    return {
        "metric": f"mesh-node-updates/s (nodes x steps x batch), {name} 256^2-query rollout", "value": value,
        "unit": "mesh-node-updates/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3, **rebuild,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_NOTE[__import__("magnet_amd.packing", fromlist=["default_gemm"]).default_gemm()], "data": "synthetic",
        "config": {"workload": f"{name} free-running rollout, {nodes - Wq * Wq} LR nodes + {Wq}x{Wq} queries, nt={NT}, "
                               f"time_slice={T} -> {pred_steps} predicted steps, {B} trajectories per GPU",
                   "trajectories_per_gpu": B, "global_trajectories": B * world,
                   "parallelism": f"batch-shard x{world}, no data-path collective"},
        "roofline": roof, "cpu_baseline": None, "training_step": train}


def bench_c1(dev, reps=50):
    """BASELINE.json configs[0] on the GPU (the launch-bound regime): mpnn 1-D, N = 50, B = 2, tw = 25, nt = 250 - a 9-forward
    free-running rollout (validation_step), replayed from ONE hipGraph (the default for small graphs, magnet_amd/graphs.py) and,
    for comparison, enqueued kernel by kernel (hip_graph = False).  Host wall time per rollout, graph topology cached."""
    from magnet_amd.models.factory import FACTORY
    hp = dict(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1", hidden_features=128, hidden_layer=5, time_window=25,
              teacher_forcing=False, neighbors=3)
    B, N, nt = 2, 50, 250
    batch = {"u": torch.randn(B, N, nt, generator=torch.Generator().manual_seed(0)), "x": torch.linspace(0, 16, N)[None, :, None].repeat(B, 1, 1),
             "t": torch.linspace(0, 4, nt)[None].repeat(B, 1)}
    cb = {k: v.to(dev) for k, v in batch.items()}
    out = {}
    for mode, flag in (("graph", True), ("eager", False)):
        torch.manual_seed(0)                               # the same weights in both models
        m = FACTORY["mpnn"](types.SimpleNamespace(**hp, hip_graph=flag)).to(dev).eval()
        for _ in range(3):
            ref = m.validation_step(cb, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.validation_step(cb, 0)
        torch.cuda.synchronize()
        out[mode] = ((time.perf_counter() - t0) / reps, ref)
    same = bool(torch.equal(out["graph"][1], out["eager"][1]))
    steps = (nt - 25) // 25 * 25
    return {"workload": "mpnn 1-D, 50-node mesh, B = 2, time_window = 25, nt = 250: 9-forward free-running rollout",
            "ms_per_rollout": round(out["graph"][0] * 1e3, 3), "ms_per_rollout_eager": round(out["eager"][0] * 1e3, 3),
            "value": B * N * steps / out["graph"][0], "unit": "mesh-node-updates/s", "graph_replay_equals_eager_bitwise": same,
            "how": "whole rollout replayed from one hipGraph (hip_graph=\"auto\": the default up to 200 000 node rows, magnet_amd/graphs.py)"}


def ref_shape_cases():
    """{key: (model name, hparams dict, make(train: bool) -> (cpu batch, nodes per sample))} - the shapes of the reference's run scripts
    (see bench_ref_shapes)"""
    from magnet_amd.utils import make_coord
    g = torch.Generator().manual_seed(4)


    def mesh_pts(n_side):
        lin = torch.linspace(-1, 1, n_side)
        return torch.stack(torch.meshgrid(lin, lin, indexing="xy"), -1).reshape(-1, 2)

    def gnn_batch(B, pts, nq, train):
        N = pts.shape[0]
        lr_c = pts[::2]
        rest = torch.arange(N)[1::2]
        q = rest[torch.randperm(rest.numel(), generator=g)[:nq]].sort().values if train else rest
        u = torch.randn(B, NT, 1, N, generator=g)
        return {"t": torch.linspace(0, 1, NT)[None].repeat(B, 1), "lr_frames": u[..., ::2].contiguous(),
                "hr_points": u[:, :, 0, q].unsqueeze(-1).contiguous(), "coords_hr": pts[q][None].repeat(B, 1, 1),
                "coords_lr": lr_c[None].repeat(B, 1, 1)}, lr_c.shape[0] + q.numel()

    def cnn_batch(B, W, nq, train):
        allc = make_coord([W, W])
        q = torch.randperm(W * W, generator=g)[:nq].sort().values if train else torch.arange(W * W)
        return {"t": torch.linspace(0, 1, NT)[None].repeat(B, 1), "lr_frames": torch.randn(B, NT, 1, W // 2, W // 2, generator=g),
                "hr_points": torch.randn(B, NT, q.numel(), 1, generator=g), "coords": allc[q][None].repeat(B, 1, 1),
                "cells": torch.full((B, q.numel(), 2), 2.0 / W)}, (W // 2) ** 2 + q.numel()

    irr = torch.clamp(torch.randn(512, 2, generator=g) * 0.25, -1, 1)
    irr = 2 * (irr - irr.min(0).values) / (irr.max(0).values - irr.min(0).values) - 1
    cases = {
        "c4a": ("mpnn_2d", dict(HP), lambda train: (synthetic_batch(64, 4, 11, "cpu"), 64 * 64)),
        "gnn_64": ("magnet_gnn", dict(MAGNET_HP["magnet_gnn"]), lambda train: gnn_batch(8, mesh_pts(64), 256, train)),
        "gnn_512": ("magnet_gnn", dict(MAGNET_HP["magnet_gnn"]), lambda train: gnn_batch(32, irr if train else mesh_pts(32), 256, train)),
        "cnn_64": ("magnet_cnn_2d", dict(MAGNET_HP["magnet_cnn_2d"], teacher_forcing=True), lambda train: cnn_batch(32, 64, 256, train)),
    }
    return cases


def bench_ref_shapes(dev):
    """What the reference's own run scripts train and validate on (rank 0, N = 1 only) - the shapes a user of the reference meets,
    next to the 256^2 metric cases that nobody trains:
      c4a          mpnn_2d, 64 x 64 regular mesh, batch 4                      scripts/mpnn_2d/mpnn_2d_b1_64_regular.sh:27-36
      gnn_64       magnet_gnn, 64 x 64 mesh: 2048 LR nodes ([::2]) + 256 sampled queries (validation: + the other 2048), batch 8
                                                                                scripts/magnet_gnn/magnet_gnn_2d_b1_64_regular.sh:26-36, datamodule/dataset_2d.py:107-121
      gnn_512      magnet_gnn, 512 irregular points: 256 LR + 256 queries, batch 32; validation on the 32 x 32 test mesh (512 + 512)
                                                                                scripts/magnet_gnn/magnet_gnn_2d_b1_512_irregular_concentrated.sh:26-36
      cnn_64       magnet_cnn_2d, 64 x 64 frames: 32 x 32 LR grid (bilinear) + 256 sampled queries (validation: all 4096), batch 32,
                   teacher forcing                                             scripts/magnet_cnn_2d/magnet_cnn_2d_b1_64_regular.sh:26-36, configs/datamodule/h5_datamodule_implicit_2d.yaml:15
    Per case: the training step (forward with tape + HIP backward + fused Adam; ms, C-ABI calls, peak memory) and the validation
    rollout enqueued kernel by kernel and replayed from one hipGraph (ms each; `hip_graph="auto"` picks the replay up to
    magnet_amd.graphs.AUTO_MAX_ROWS node rows)."""
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    cases = ref_shape_cases()
    lib = ops.L.lib()
    names = [n for n in ops.L.declared_symbols() if hasattr(lib, n)]
    out = {}
    from magnet_amd import packing
    for key, (name, hp, mk) in cases.items():
        torch.manual_seed(0)
        res = {"model": name}
        packing.narrow()                   # (a leg that left the fp16 range must not hand the NEXT one the wider split)
        # ---- training step.  lr = 1e-6: eight Adam steps at the configs' 1e-3 on RANDOM fields make EDSR's feature maps grow past
        # 65504 (scripts/dev/many_steps.py cnn_64 24) - the step would then be timed in the bf16x6 fallback, which real
        # (smooth, normalised) data never enter.  The work of a step does not depend on the learning rate; `widened` says what ran.
        batch, nodes = mk(True)
        batch = {k: v.to(dev) for k, v in batch.items()}
        B = batch["t"].shape[0]
        tm = FACTORY[name](types.SimpleNamespace(**dict(hp, lr=1e-6))).to(dev).train()
        opt = tm.configure_optimizers()["optimizer"]

        def train_step():
            opt.zero_grad(set_to_none=True)
            loss = tm.training_step(batch, 0)
            loss.backward()
            opt.step()
            return loss

        train_step()
        train_step()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        calls = [0]
        origs = {n: getattr(lib, n) for n in names}

        def counted(fn):
            def f(*a):
                calls[0] += 1
                return fn(*a)
            return f

        for n, fn in origs.items():
            setattr(ops.L._lib, n, counted(fn))
        try:
            train_step()
        finally:
            for n, fn in origs.items():
                setattr(ops.L._lib, n, fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            train_step()
        torch.cuda.synchronize()
        res["train"] = {"batch": B, "nodes_per_sample": nodes, "ms_per_step": round((time.perf_counter() - t0) / 5 * 1e3, 3),
                        "c_abi_calls_per_step": calls[0], "peak_mem_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 3),
                        "lr": 1e-6, "widened_to_bf16x6": packing.wide()}
        del tm, opt, train_step
        packing.narrow()
        # ---- the same step replayed from ONE hipGraph (magnet_amd.graphs.TrainStepGraph: forward + tape + backward + the in-place re-pack
        # captured; gradient zero fill, range words and the Adam launch outside) - what a custom training loop can use; bit-equal losses
        # (tests/test_train_gpu.py).  It removes the host's share of a step, not the kernels': GPU-bound shapes do not move.
        try:
            from magnet_amd.graphs import TrainStepGraph
            torch.manual_seed(0)
            gm = FACTORY[name](types.SimpleNamespace(**dict(hp, lr=1e-6))).to(dev).train()
            gstep = TrainStepGraph(gm, gm.configure_optimizers()["optimizer"])
            for _ in range(3):
                gstep(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                gstep(batch)
            torch.cuda.synchronize()
            res["train"]["ms_per_step_hip_graph"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
            res["train"]["hip_graph_replays"] = gstep.replays
            del gm, gstep
        except Exception as ex:  # noqa: BLE001  (an extra: never costs the line)
            res["train"]["ms_per_step_hip_graph"] = None
            res["train"]["hip_graph_error"] = f"{type(ex).__name__}: {str(ex)[:120]}"
        packing.narrow()
        # ---- validation rollout: kernel by kernel, and replayed from one hipGraph
        batch, nodes = mk(False)
        batch = {k: v.to(dev) for k, v in batch.items()}
        val = {"batch": batch["t"].shape[0], "nodes_per_sample": nodes, "rows": batch["t"].shape[0] * nodes}
        ref = None
        for label, flag in (("eager", False), ("graph", True)):
            torch.manual_seed(0)
            vm = FACTORY[name](types.SimpleNamespace(**hp, hip_graph=flag)).to(dev).eval()
            for _ in range(3):
                y = vm.validation_step(batch, 0)
            torch.cuda.synchronize()
            groups = []                                   # best of three groups of ten (one host hiccup in ten 3-ms rollouts is +40 %:
            for _ in range(3):                            # profiles/r4/bench_v15.json eager 4.78, profiles/r5/bench_v1.json graph 4.34)
                t0 = time.perf_counter()
                for _ in range(10):
                    vm.validation_step(batch, 0)
                torch.cuda.synchronize()
                groups.append((time.perf_counter() - t0) / 10 * 1e3)
            val[f"ms_per_rollout_{label}"] = round(min(groups), 3)
            val[f"ms_per_rollout_{label}_groups"] = [round(g_, 3) for g_ in groups]
            if ref is None:
                ref = y
            else:
                val["graph_replay_equals_eager_bitwise"] = bool(torch.equal(ref, y))
            del vm
        from magnet_amd.graphs import wants_graph
        val["auto_picks"] = "graph" if wants_graph("auto", val["rows"]) else "eager"
        val["widened_to_bf16x6"] = packing.wide()
        res["validation"] = val
        out[key] = res
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a torch.distributed.run environment: start the N ranks as FRESH child processes
    (one per GPU, what the reference's `trainer.gpus=N trainer.strategy=ddp` does, README.md:124-125) before this process
    has made any GPU call - a process that has initialised HIP must never exec or fork GPU work - and forward the exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    if os.environ.get("MAGNET_BENCH_LAUNCH_DRY") == "1":       # test hook: show the command instead of running it
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--traj", type=int, default=4, help="trajectories per GPU")
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=3, help="timed training steps of the extra 'training_step' leg (0: skip)")
    ap.add_argument("--gemm", default=None, choices=["fp16x3", "bf16x6", "fp32", "bf16"],
                    help="arithmetic of the Linears; default fp16x3 = fp32-class.  'bf16' is the REDUCED-precision mode BASELINE.json "
                         "config 2 asks for: reported with its measured deviation from the fp32-class result, never as the headline")
    ap.add_argument("--workload", default="mpnn_2d", choices=["mpnn_2d", "magnet_cnn_2d", "magnet_gnn"],
                    help="mpnn_2d = the headline metric case (its line also carries the two MAgNet 256^2 cases of SURVEY.md 8(d) as "
                         "the objects `c4b` / `c5`); magnet_cnn_2d / magnet_gnn make one of those the headline of the line")
    ap.add_argument("--no-extra-cases", action="store_true", help="skip the c4b / c5 / c1 objects of the default line")
    ap.add_argument("--no-precision-modes", action="store_true", help="skip the bf16x6 / fp32 timings and the 48 x 48 oracle check")
    ap.add_argument("--no-ref-shapes", action="store_true", help="skip the ref_shapes object (the reference's own training / validation "
                                                                 "shapes); the PMC traffic passes do: its launches share kernel names and "
                                                                 "grid sizes with the roofline launches")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:       # N ranks as fresh children, before any GPU call here
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch.distributed as dist
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    from magnet_amd.sharding import env_world, max_over_ranks

    rank, local_rank, world = env_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); start it as "
                         f"`python bench.py --gpus N` or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    # one process per GPU over RCCL ("nccl" on ROCm).  MAGNET_DIST_BACKEND=gloo is a test hook: it lets several ranks
    # share one GPU (RCCL refuses that) so that the N > 1 control flow can be exercised on a 1-GPU box.
    backend = os.environ.get("MAGNET_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1) if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    W, B, K, Wm = args.mesh, args.traj, args.steps, args.warmup
    if args.gemm:
        os.environ["MAGNET_GEMM"] = args.gemm                  # read by packing.default_gemm() when the models are built
    if args.workload != "mpnn_2d":
        line = bench_magnet(args.workload, args.traj, K, Wm, W, args.train_steps, dev, rank, world, dist, max_over_ranks)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(line))
        return

    torch.manual_seed(0)
    # hip_graph = "rebuild": every timed step builds the radius graph + CSR afresh (eagerly: the edge count is read on the host) and
    # then replays the rollout's ~300 kernels from ONE hipGraph whose topology buffers were just overwritten with the fresh arrays
    # (models/mpnn.py: _replay_with_rebuild).  Same GPU work as enqueueing kernel by kernel (MAGNET_BENCH_EAGER=1: the round-2
    # behaviour - hip_graph = False + drop_topology_cache() per step), but the rate no longer depends on how busy the host is: on
    # shared boxes the eager loop was measured between 28 and 40 ms per rollout with identical kernel times.
    eager = os.environ.get("MAGNET_BENCH_EAGER", "0") == "1"
    model = FACTORY["mpnn_2d"](types.SimpleNamespace(**HP, hip_graph=False if eager else "rebuild")).to(dev).eval()
    batch = synthetic_batch(W, B, 1000 + rank, dev)          # every rank owns B different trajectories

    def step():
        if eager:
            model.drop_topology_cache()                          # rebuild radius graph + CSR every step
        return model.validation_step(batch, 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # device warm-up before the W warm-up steps of the contract: a GPU that has idled for a few seconds (process start-up, the import
    # of torch) comes back at a fraction of its boost clock and ramps over several hundred milliseconds - with W = 2 the K timed
    # rollouts of a 0.3 s measurement ran inside that ramp (38 - 40 ms per rollout on cold boxes, every kernel 25 % slow, against
    # 28 - 31 ms once warm; the roofline pass right after it saw the usual 1.05 - 1.18 ms edge launches).  Untimed, like the warm-up.
    t_warm = time.perf_counter() + float(os.environ.get("MAGNET_BENCH_PREWARM_S", "1.5"))
    while time.perf_counter() < t_warm:
        step()
        torch.cuda.synchronize()
    for _ in range(Wm):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(my_elapsed, dev)
    edges_per_traj = model._topo.csr.E // B
    tw = HP["time_window"]
    pred_steps = (NT - tw) // tw * tw
    value = world * B * W * W * pred_steps * K / elapsed
    # what the process group saw (the first multi-GPU lease should yield a curve, not a debugging session)
    def gather_f64(v):
        if world == 1:
            return [float(v)]
        mine_ = torch.tensor([float(v)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        parts_ = [torch.empty_like(mine_) for _ in range(world)]
        dist.all_gather(parts_, mine_)
        return [float(p_.item()) for p_ in parts_]

    per_rank_ms = gather_f64(my_elapsed / K * 1e3)
    dist_info = {"backend": dist.get_backend() if world > 1 else None, "world_size_seen": dist.get_world_size() if world > 1 else 1,
                 "ms_per_step_by_rank": [round(v, 3) for v in per_rank_ms],
                 "trajectory_seed_by_rank": [int(v) for v in gather_f64(1000 + rank)],
                 "input_checksum_by_rank": gather_f64(float(batch["u"].double().sum()))}

    # ---- per-kernel rooflines, measured live with HIP events on the launch stream (untimed extra passes) BEFORE the training
    # leg: the part is power limited and runs cooler / faster before 25 GB of training traffic have gone through it
    roof = roof_sc = None
    if rank == 0:
        roof = edge_roofline(model, batch, dev, W, B)
        roof_sc = scatter_roofline(model, dev, W, B)
    modes = None
    if rank == 0 and world == 1 and not args.no_precision_modes and model.edge_gemm in ("fp16x3", "bf16x6", "fp32"):
        modes = precision_modes(model, batch, elapsed / K * 1e3, dev)

    # ---- training step on the same workload ("plus training-step throughput with the all-reduce"):
    # rollout with tape -> HIP backward -> ONE flat gradient all-reduce (RCCL) -> fused Adam.  Reported beside the
    # headline, never mixed into `value`.
    train = None
    if args.train_steps > 0:
        from magnet_amd.sharding import allreduce_gradients
        barrier()
        tmodel = FACTORY["mpnn_2d"](types.SimpleNamespace(**HP)).to(dev).train()
        opt = tmodel.configure_optimizers()["optimizer"]
        ar_events = []

        def train_step():
            opt.zero_grad(set_to_none=True)
            loss = tmodel.training_step(batch, 0)
            loss.backward()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            allreduce_gradients(tmodel.parameters())
            e_.record()
            ar_events.append((s_, e_))
            opt.step()
            return loss

        first = float(train_step().detach())
        train_step()        # (second warm-up step, as in bench_magnet)
        barrier()
        ar_events.clear()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            last = train_step()
        barrier()
        t_el = max_over_ranks(time.perf_counter() - t0, dev)
        n_grad = sum(p_.numel() for p_ in tmodel.parameters())
        # ---- the dominant kernel of the step - the fused backward of the edge block (GNN_Layer.message + mean walked back, models/mpnn.py:
        # 73-79,46; generation 3: k_mpnn_edge_bwd_co) - timed live with HIP events on its launch stream in ONE more (untimed) step
        ebw = None
        if True:                                   # (every rank: the extra step contains the gradient all-reduce)
            from magnet_amd import ops as _ops
            lib_ = _ops.L.lib()
            ev_, orig_ = [], lib_.magnet_mpnn_edge_bwd_w_flags_f32

            def timed_bwd(*a):
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                rc = orig_(*a)
                e_.record()
                ev_.append((s_, e_))
                return rc

            _ops.L._lib.magnet_mpnn_edge_bwd_w_flags_f32 = timed_bwd
            try:
                train_step()
                torch.cuda.synchronize()
            finally:
                _ops.L._lib.magnet_mpnn_edge_bwd_w_flags_f32 = orig_
            if not ev_:
                print("bench.py: training_step.edge_backward - the timed C-ABI entry was not called by the step", file=sys.stderr)
            if ev_ and getattr(tmodel, "_topo", None) is not None:
                E_ = tmodel._topo.csr.E
                ms_ = sum(s_.elapsed_time(e_) for s_, e_ in ev_) / len(ev_)
                fl_ = 3 * 2.0 * E_ * 128 * 128                    # GEMM 1 (q = W2 s1), GEMM 2 (ds1 = W2^T dq), the weight gradient (dq^T s1)
                gen_ = int(lib_.magnet_mpnn_edge_bwd_w_generation(-1))
                ebw = {"kernel": {2: "k_mpnn_edge_bwd_co", 1: "k_mpnn_edge_bwd_wp", 0: "k_mpnn_edge_bwd_w"}.get(gen_, "?"),
                       "what": "entry magnet_mpnn_edge_bwd_w_flags_f32 = the edge kernel + the reduction of its per-workgroup dW2 / db2 partials + "
                               "the fix-up of the rows cut by the workgroups' tile ranges",
                       "launch_ms": round(ms_, 4), "launches_timed": len(ev_), "flops_per_launch": fl_, "bound": "mfma",
                       "achieved": round(fl_ / (ms_ * 1e-3) / 1e12, 2), "peak": round(MFMA_BF16_PEAK_TFLOPS / 3, 1), "unit": "TFLOP/s",
                       "frac": round(fl_ / (ms_ * 1e-3) / 1e12 / (MFMA_BF16_PEAK_TFLOPS / 3), 4),
                       "algorithmic_hbm_bytes": E_ * 128 * 4 + 4 * tmodel._topo.csr.N * 128 * 4 + 2 * E_ * 4,
                       "share_of_step": round(len(ev_) * ms_ / (t_el / args.train_steps * 1e3), 3)}
        # DDP invariant: identical parameters on every rank after the steps (same init, same averaged gradient, same Adam)
        psum = float(sum(p_.detach().double().abs().sum() for p_ in tmodel.parameters()))
        dist_info["param_abs_sum_by_rank_after_training"] = gather_f64(psum)
        train = {"ms_per_step": round(t_el / args.train_steps * 1e3, 3), "steps": args.train_steps, "warmup": 2,
                 "value": world * B * W * W * pred_steps * args.train_steps / t_el, "unit": "mesh-node-updates/s",
                 "what": "free-running rollout with tape + HIP backward + one flat RCCL gradient all-reduce + fused Adam",
                 "allreduce_ms": round(sum(s_.elapsed_time(e_) for s_, e_ in ar_events) / len(ar_events), 4),
                 "allreduce_bytes": 4 * n_grad,
                 "loss_first_last": [round(first, 6), round(float(last.detach()), 6)],
                 "widened_to_bf16x6": __import__("magnet_amd.packing", fromlist=["wide"]).wide(),
                 "peak_mem_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "edge_backward": ebw}
        del tmodel, opt, last, train_step
        __import__("magnet_amd.packing", fromlist=["narrow"]).narrow()   # (a leg that left the fp16 range must not hand the NEXT one the wider split)
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(128)
    deviation = None
    if model.edge_gemm == "bf16" and rank == 0:               # reduced precision: measure it against the fp32-class path
        ref_model = FACTORY["mpnn_2d"](types.SimpleNamespace(**dict(HP, edge_gemm="fp16x3", gemm="fp16x3"))).to(dev).eval()
        ref_model.load_state_dict(model.state_dict())
        a, b = model.validation_step(batch, 0), ref_model.validation_step(batch, 0)
        last = batch["u"][:, :, HP["time_window"] - 1].unsqueeze(1)
        deviation = {"rollout_rel_l2": float((a - b).norm() / b.norm()),
                     "increment_rel_l2": float(((a - last) - (b - last)).norm() / (b - last).norm())}
        del ref_model

    # ---- the other two 256^2 cases of SURVEY.md 8(d) (C4b magnet_cnn_2d, C5 magnet_gnn; 1 trajectory per GPU), each a
    # complete object of the same contract (value, ms_per_step, roofline, training_step) beside the headline
    extras = {}
    dtype_note = DTYPE_NOTE[model.edge_gemm]
    if not args.no_extra_cases and W == 256:
        # the headline model is done: its captured rollout keeps a pool of intermediates alive, and the MAgNet training legs below
        # keep as much of their backward tape in HBM as fits
        del model
        torch.cuda.empty_cache()
        order = (("c4b", "magnet_cnn_2d"), ("c5", "magnet_gnn"))
        if os.environ.get("MAGNET_BENCH_C5_FIRST", "0") == "1":
            order = order[::-1]
        for key, name in order:
            extras[key] = bench_magnet(name, 1, max(1, min(K, 10)), max(1, min(Wm, 2)), 256, min(args.train_steps, 2), dev, rank, world,
                                       dist, max_over_ranks)
        if rank == 0:
            extras["c1"] = bench_c1(dev)
            if world == 1 and not args.no_ref_shapes:
                extras["ref_shapes"] = bench_ref_shapes(dev)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = {"metric": "mesh-node-updates/s (nodes x steps x batch), 2D Burgers 256^2 rollout",
                "value": value, "unit": "mesh-node-updates/s", "n_gpus": world, "steps": K, "warmup": Wm,
                "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": dtype_note, "data": "synthetic", "deviation_vs_f32_path": deviation,
                "config": {"workload": f"mpnn_2d free-running rollout, {W}x{W} regular mesh (N={W * W}, E={edges_per_traj} per "
                                       f"trajectory), nt={NT}, time_window={tw} -> {pred_steps} predicted steps, {B} trajectories per GPU, "
                                       "radius graph + CSR rebuilt every step, the rollout's kernels replayed from one hipGraph",
                           "trajectories_per_gpu": B, "global_trajectories": B * world, "parallelism": f"batch-shard x{world}, no data-path collective"},
                "roofline": roof, "roofline_scatter": roof_sc, "cpu_baseline": cpu, "precision_modes": modes, "dist": dist_info,
                "training_step": train, **extras}
        print(json.dumps(line), flush=True)
        if modes and not modes["check"]["ok"]:
            sys.stderr.write(f"bench.py: precision_modes check FAILED for {modes['check']['failed']}: {modes['check']['rule']}\n")
            sys.exit(3)


if __name__ == "__main__":
    main()
