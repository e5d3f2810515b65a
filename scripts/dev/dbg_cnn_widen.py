import sys, types, torch, warnings
sys.path.insert(0, "/root/repo")
import bench
from magnet_amd.models.factory import FACTORY
from magnet_amd import packing
name, hp, mk = bench.ref_shape_cases()["cnn_64"]
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
opt = model.configure_optimizers()["optimizer"]
batch, nodes = mk(True)
batch = {k: v.cuda() for k, v in batch.items()}
for i in range(6):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    gmax = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
    opt.step()
    wmax = max(float(p.detach().abs().max()) for p in model.parameters())
    fin = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    print(i, "loss", float(loss), "max|grad|", gmax, "max|w|", wmax, "finite", fin, "wide", packing.wide(), "status", {str(k): int(v) for k, v in packing._status.items()})
