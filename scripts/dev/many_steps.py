"""Dev: N optimizer steps at one of the reference's shapes (bench.ref_shape_cases), reporting the loss and every parameter whose
gradient is not finite.  usage: python scripts/dev/many_steps.py cnn_64 24"""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from magnet_amd.models.factory import FACTORY
name, hp, mk = bench.ref_shape_cases()[sys.argv[1]]
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
opt = model.configure_optimizers()["optimizer"]
batch, _ = mk(True)
batch = {k: v.cuda() for k, v in batch.items()}
for i in range(int(sys.argv[2])):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in model.named_parameters()
           if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    gmax = max(float(p.grad.abs().nan_to_num(0, 0, 0).max()) for p in model.parameters() if p.grad is not None)
    print(i, round(float(loss.detach()), 4), "max|g|", f"{gmax:.3e}", "non-finite grads:", bad[:6], len(bad), flush=True)
    if bad:
        break
    opt.step()
