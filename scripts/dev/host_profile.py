"""Dev: cProfile of the HOST side of one warm training step at one of the reference's own shapes (bench.ref_shape_cases).
usage: python scripts/dev/host_profile.py [gnn_64|gnn_512|cnn_64|c4a] [out.txt]"""
import cProfile, io, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
key = sys.argv[1] if len(sys.argv) > 1 else "gnn_64"
name, hp, mk = bench.ref_shape_cases()[key]
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
opt = model.configure_optimizers()["optimizer"]
batch, nodes = mk(True)
batch = {k: v.cuda() for k, v in batch.items()}


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{key}: host enqueue {1e3 * (t1 - t0) / 5:.1f} ms per step, drain {1e3 * (t2 - t1):.1f} ms after 5")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
# the backward functions run on the autograd engine's own thread, which cProfile (per thread) does not see: profile inside them
from magnet_amd import autograd as AG
bpr = cProfile.Profile()
for cls in (AG._Mlp, AG._NodeFeatures, AG._KnnBlend, AG._Euler, AG._GridBlend):
    orig = cls.backward

    def wrap(ctx, *a, _o=orig):
        bpr.enable()
        try:
            return _o(ctx, *a)
        finally:
            bpr.disable()
    cls.backward = staticmethod(wrap)
for _ in range(5):
    step()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(bpr, stream=s).strip_dirs().sort_stats("tottime").print_stats(40)
print("==== inside the backward functions (autograd thread), 5 steps\n", s.getvalue()[:8000])
s = io.StringIO()
pstats.Stats(bpr, stream=s).strip_dirs().sort_stats("cumtime").print_stats(30)
print(s.getvalue()[:6000])
for sort in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(sort).print_stats(45)
    print(s.getvalue()[:9000])
