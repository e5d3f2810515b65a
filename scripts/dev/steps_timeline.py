"""Dev: wall time and allocator state of the first N training steps of one workload (same construction as one_train_step.py):
usage: python scripts/dev/steps_timeline.py magnet_gnn [N]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY

name = sys.argv[1] if len(sys.argv) > 1 else "magnet_gnn"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
if name in ("gnn_64", "gnn_512", "cnn_64", "c4a"):
    mname, hp, mk = bench.ref_shape_cases()[name]
    model = FACTORY[mname](types.SimpleNamespace(**hp)).cuda().train()
    batch, nodes = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
else:
    model = FACTORY[name](types.SimpleNamespace(**bench.MAGNET_HP[name])).cuda().train()
    batch, nodes = bench.synthetic_magnet_batch(name, 1, 256, 2000, "cuda")
opt = model.configure_optimizers()["optimizer"]
for i in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"step {i}: {1e3 * (time.perf_counter() - t0):7.1f} ms  allocated peak {torch.cuda.max_memory_allocated() / 2**30:6.1f} GiB  reserved "
          f"{torch.cuda.memory_reserved() / 2**30:6.1f} GiB  hipMalloc calls {st['num_device_alloc']}  frees {st['num_device_free']}  "
          f"retries {st['num_alloc_retries']}", flush=True)
