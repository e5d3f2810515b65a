"""Dev: cProfile of autograd._Mlp.backward (it runs on the autograd engine's thread, which a main-thread profiler does not see)"""
import sys, types, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
from magnet_amd import autograd as AG
name = sys.argv[1] if len(sys.argv) > 1 else "magnet_cnn_2d"
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**bench.MAGNET_HP[name])).cuda().train()
opt = model.configure_optimizers()["optimizer"]
batch, nodes = bench.synthetic_magnet_batch(name, 1, 256, 2000, "cuda")


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
orig = AG._Mlp.backward
calls = [0, 0.0]


def wrapped(ctx, *a):
    t0 = time.perf_counter()
    pr.enable()
    try:
        return orig(ctx, *a)
    finally:
        pr.disable()
        calls[0] += 1
        calls[1] += time.perf_counter() - t0


AG._Mlp.backward = staticmethod(wrapped)
t0 = time.perf_counter()
step()
torch.cuda.synchronize()
print(f"{name}: step {1e3 * (time.perf_counter() - t0):.1f} ms; _Mlp.backward: {calls[0]} calls, {1e3 * calls[1]:.1f} ms in total")
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
print("\n".join(l for l in buf.getvalue().splitlines() if l.strip())[:6000])
