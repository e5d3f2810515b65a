"""Two-rank check of the training step's only collective (run under torch.distributed.run).

Every rank trains on its shard of a small batch: rollout + HIP backward, then magnet_amd.sharding.allreduce_gradients
(the flat-buffer all-reduce), then one FusedAdam step.  Checked on every rank: the reduced gradients equal the
gradient of the same loss on the concatenated batch computed in-process (mean loss, equal shards), and the updated
parameters are identical on all ranks.  MAGNET_DIST_BACKEND=gloo lets both ranks share ONE GPU (RCCL refuses two
ranks on one device); on a multi-GPU node the default is nccl (= RCCL).
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from magnet_amd import sharding  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402
from magnet_amd.optim import flat_gradient  # noqa: E402


def main():
    rank, local, world = sharding.env_world()
    backend = os.environ.get("MAGNET_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group(backend)
    hp = types.SimpleNamespace(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                               hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
    W, B, nt = 12, 2 * world, 30
    g = torch.Generator().manual_seed(0)
    lin = torch.linspace(0, 1, W)
    xy = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)
    full = {"u": torch.randn(B, W * W, nt, generator=g).to(dev), "x": xy[None].repeat(B, 1, 1).to(dev),
            "t": torch.linspace(0, 1, nt)[None].repeat(B, 1).to(dev)}

    def make():
        torch.manual_seed(1)
        return FACTORY["mpnn_2d"](hp).to(dev).train()

    ref = make()                                              # full batch, no collective
    ref.training_step(full, 0).backward()
    want = [p.grad.clone() for p in ref.parameters()]

    model = make()
    opt = model.configure_optimizers()["optimizer"]
    mine = sharding.shard_batch(full, rank, world)
    model.training_step(mine, 0).backward()
    assert flat_gradient(list(model.parameters())) is not None
    sharding.allreduce_gradients(model.parameters())
    worst = max(float((p.grad - w).norm() / w.norm().clamp_min(1e-30)) for p, w in zip(model.parameters(), want))
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    same = all(torch.equal(parts[0], q) for q in parts)
    ok = worst < 1e-4 and same
    print(f"[rank {rank}/{world} {backend} {dev}] allreduced-vs-full-batch gradient rel err {worst:.2e}, "
          f"parameters identical across ranks: {same} -> {'DDP_OK' if ok else 'DDP_FAIL'}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
