"""Dev: where does a training step at one of the REFERENCE's own shapes (bench.ref_shape_cases) spend its time - host enqueue vs GPU
drain, C-ABI launches by entry point, the torch ops issued from our code.  usage: python scripts/dev/ref_shape_glue.py [gnn_64|gnn_512|cnn_64|c4a]"""
import collections, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
from magnet_amd import _lib as L
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


for _ in range(3):
    step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{key}: host enqueue {1e3 * (t1 - t0):.1f} ms, + drain {1e3 * (t2 - t1):.1f} ms = {1e3 * (t2 - t0):.1f} ms per step")
# GPU time of one step by events around it (after a sync: the stream is empty when the first kernel arrives)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
step()
e.record()
torch.cuda.synchronize()
print(f"{key}: first kernel to last kernel {s.elapsed_time(e):.1f} ms")
calls = collections.Counter()
lib = L.lib()
orig = {}
for n in L.declared_symbols():
    orig[n] = getattr(lib, n)

    def mkw(n_, f_):
        def w(*a):
            calls[n_] += 1
            return f_(*a)
        return w
    setattr(L._lib, n, mkw(n, orig[n]))
step()
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(L._lib, n, f)
print("C-ABI launches:", sum(calls.values()), dict(calls.most_common(16)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt, kern = collections.Counter(), collections.Counter()
ktime = 0.0
for ev in prof.events():
    if ev.device_type.name == "CUDA" or (hasattr(ev, "device_time") and ev.device_time and not ev.name.startswith("aten::")):
        pass
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_strided", "aten::view", "aten::reshape", "aten::as_strided",
                                                        "aten::slice", "aten::select", "aten::_unsafe_view", "aten::t", "aten::transpose",
                                                        "aten::permute", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::detach",
                                                        "aten::alias", "aten::empty_like", "aten::result_type", "aten::to", "aten::narrow"):
        st = [s_ for s_ in ev.stack if "magnet_amd" in s_ or "bench" in s_][:2]
        cnt[(ev.name, " | ".join(s_.split("/")[-1] for s_ in st))] += 1
print("torch ops issued per step:", sum(cnt.values()))
for (n, where), c in cnt.most_common(28):
    print(f"  {c:5d}  {n:28s} {where}")
try:
    tot = sum(e_.device_time_total for e_ in prof.key_averages() if e_.device_time_total)
    print(f"sum of device kernel time (profiler): {tot / 1e3:.1f} ms")
    for e_ in sorted(prof.key_averages(), key=lambda x: -(x.device_time_total or 0))[:14]:
        if e_.device_time_total:
            print(f"  {e_.device_time_total / 1e3:8.2f} ms {e_.count:6d} x  {e_.key[:90]}")
except Exception as ex:  # noqa: BLE001
    print("no device times from the profiler:", ex)
if os.environ.get("GLUE_CPROFILE", "1") == "1":
    import cProfile, io, pstats
    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):        # the engine then runs the Functions' backward in THIS thread: cProfile sees them
        step()
        pr.enable()
        step()
        pr.disable()
    torch.cuda.synchronize()
    for sort in ("tottime", "cumulative"):
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats(sort).print_stats(45)
        print(f"---- cProfile by {sort}")
        print("\n".join(l for l in buf.getvalue().splitlines() if l.strip())[:9000])
