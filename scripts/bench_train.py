"""Timing of one MPNN_2d training step (forward rollout with tape + HIP backward + Adam) on the bench workload.

    python scripts/bench_train.py [--mesh 256] [--traj 4] [--steps 5] [--tf]

Under torch.distributed.run every rank trains on its own trajectories and the gradients are all-reduced once per
step (magnet_amd.sharding.allreduce_gradients) - the only collective of the training step.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from magnet_amd import sharding  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--traj", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tf", action="store_true", help="teacher forcing")
    a = ap.parse_args()
    rank, local, world = sharding.env_world()
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    hp = types.SimpleNamespace(**dict(bench.HP, teacher_forcing=a.tf))
    torch.manual_seed(0)
    model = FACTORY["mpnn_2d"](hp).cuda().train()
    batch = bench.synthetic_batch(a.mesh, a.traj, 1234 + rank, "cuda")
    opt = model.configure_optimizers()["optimizer"]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, 0)
        loss.backward()
        sharding.allreduce_gradients(model.parameters())
        opt.step()
        return loss

    phases = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        phases[name] = phases.get(name, 0.0) + time.perf_counter() - t0
        return r

    losses = [float(step()) for _ in range(a.warmup)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(float(step()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    dt = sharding.max_over_ranks(dt, "cuda")
    # phase split (separately synchronised, a little slower than the pipelined step)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = timed("forward_rollout", lambda: model.training_step(batch, 0))
        timed("backward", loss.backward)
        timed("allreduce", lambda: sharding.allreduce_gradients(model.parameters()))
        timed("adam", opt.step)
    if rank == 0:
        n_upd = world * a.traj * a.mesh * a.mesh * (bench.NT - hp.time_window)
        print(json.dumps({"what": "mpnn_2d training step", "mesh": a.mesh, "traj_per_gpu": a.traj, "n_gpus": world,
                          "teacher_forcing": a.tf, "ms_per_step": round(dt * 1e3, 2),
                          "node_updates_per_s": round(n_upd / dt, 1),
                          "phases_ms": {k: round(v / 2 * 1e3, 2) for k, v in phases.items()},
                          "loss_first_last": [round(losses[0], 6), round(losses[-1], 6)],
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
