"""Dev: which Python lines issue the small torch kernels (adds / fills / copies / zeros) inside a WARM MAgNet TRAINING step?
usage: python scripts/dev/find_train_glue.py [magnet_cnn_2d|magnet_gnn|gnn_64|gnn_512|cnn_64]"""
import sys, types, collections, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
from magnet_amd import _lib as L
name = sys.argv[1] if len(sys.argv) > 1 else "magnet_cnn_2d"
torch.manual_seed(0)
if name in ("gnn_64", "gnn_512", "cnn_64", "c4a"):                 # the reference's own shapes (bench.ref_shape_cases)
    mname, hp, mk = bench.ref_shape_cases()[name]
    model = FACTORY[mname](types.SimpleNamespace(**hp)).cuda().train()
    batch, nodes = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
else:
    model = FACTORY[name](types.SimpleNamespace(**bench.MAGNET_HP[name])).cuda().train()
    batch, nodes = bench.synthetic_magnet_batch(name, 1, 256, 2000, "cuda")
opt = model.configure_optimizers()["optimizer"]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
import time
for _ in range(2):                 # host enqueue time vs wall time: is the step launch / Python bound?
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0):.1f} ms, + drain {1e3 * (t2 - t1):.1f} ms = {1e3 * (t2 - t0):.1f} ms per step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
import io
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(18)
print("\n".join(l for l in buf.getvalue().splitlines() if l.strip())[:3500])
# count our own C-ABI launches per entry point
calls = collections.Counter()
lib = L.lib()
orig = {}
for n in L.declared_symbols():
    f = getattr(lib, n)
    orig[n] = f

    def mk(n_, f_):
        def w(*a):
            calls[n_] += 1
            return f_(*a)
        return w
    setattr(L._lib, n, mk(n, f))
import time
t0 = time.perf_counter()
step()
torch.cuda.synchronize()
print(f"{name}: {1e3 * (time.perf_counter() - t0):.1f} ms per step (with call counting)")
for n, f in orig.items():
    setattr(L._lib, n, f)
print("C-ABI launches:", sum(calls.values()), dict(calls.most_common(14)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
tot = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_strided", "aten::view", "aten::reshape", "aten::as_strided",
                                                        "aten::slice", "aten::select", "aten::_unsafe_view", "aten::t", "aten::transpose",
                                                        "aten::permute", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::detach",
                                                        "aten::alias", "aten::empty_like", "aten::result_type", "aten::to", "aten::narrow"):
        st = [s for s in ev.stack if "magnet_amd" in s or "bench" in s][:2]
        cnt[(ev.name, " | ".join(s.split("/")[-1] for s in st))] += 1
        tot[ev.name] += 1
print("torch ops:", dict(tot.most_common(16)))
for (n, st), v in cnt.most_common(40):
    print(v, n, st)
