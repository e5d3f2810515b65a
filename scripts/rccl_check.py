"""One-rank RCCL check (a 1-GPU box cannot host two RCCL ranks): a process group with backend "nccl" (= RCCL on ROCm) of world size 1
on cuda:0, and through it exactly the collectives the N > 1 paths issue (bench.py, magnet_amd/sharding.py): the in-place SUM
all-reduce of the training step's flat gradient buffer (after a real rollout + HIP backward), the MAX / SUM all-reduces of float64
scalars on the device, the all-gather of predictions and of per-rank scalars, barriers, destroy.  With one rank every collective is
the identity - what is exercised is the communicator set-up on this driver / image (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0), RCCL's
launches on the stream the HIP kernels use, and the order of calls.  Prints RCCL_OK, or RCCL_INIT_FAILED <reason> when the
communicator cannot be created at all (environment, not code under test)."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from magnet_amd.models.factory import FACTORY  # noqa: E402
from magnet_amd.optim import flat_gradient  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        probe = torch.ones(4, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("RCCL_INIT_FAILED", type(e).__name__, str(e)[:300])
        return 0
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and torch.equal(probe, torch.ones(4, device=dev))
    hp = types.SimpleNamespace(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                               hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
    W, B, nt = 12, 2, 30
    g = torch.Generator().manual_seed(0)
    lin = torch.linspace(0, 1, W)
    xy = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)
    batch = {"u": torch.randn(B, W * W, nt, generator=g).to(dev), "x": xy[None].repeat(B, 1, 1).to(dev),
             "t": torch.linspace(0, 1, nt)[None].repeat(B, 1).to(dev)}
    torch.manual_seed(1)
    model = FACTORY["mpnn_2d"](hp).to(dev).train()
    opt = model.configure_optimizers()["optimizer"]
    model.training_step(batch, 0).backward()
    flat = flat_gradient(list(model.parameters()))
    assert flat is not None
    before = flat.clone()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)                     # what sharding.allreduce_gradients issues, in place
    flat /= dist.get_world_size()
    dist.barrier()
    assert torch.equal(flat, before), "a one-rank all-reduce changed the gradient buffer"
    opt.step()
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)       # sharding.max_over_ranks / sum_over_ranks
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert float(t.item()) == 3.25
    with torch.no_grad():
        pred = model.eval().validation_step(batch, 0)
    pred = pred[0] if isinstance(pred, (tuple, list)) else pred
    pred = pred if torch.is_tensor(pred) else torch.ones(8, device=dev)
    parts = [torch.empty_like(pred)]
    dist.all_gather(parts, pred.contiguous())                       # sharding.gather_predictions
    assert torch.equal(parts[0], pred)
    dist.barrier()
    torch.cuda.synchronize()
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    dist.destroy_process_group()
    print(f"RCCL_OK backend=nccl world=1 rccl={ver} flat_gradient_floats={flat.numel()}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
