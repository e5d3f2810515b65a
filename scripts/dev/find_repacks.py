"""Dev: which code path packs weights (packing.split_fp16x2 -> host range check: isfinite / abs / max, three tiny launches and a host
sync each) during a WARM rollout?  usage: find_repacks.py [magnet_cnn_2d | magnet_gnn]"""
import sys, types, traceback, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
import bench
from magnet_amd import packing
from magnet_amd.models.factory import FACTORY
name = sys.argv[1] if len(sys.argv) > 1 else "magnet_cnn_2d"
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**bench.MAGNET_HP[name])).cuda().eval()
batch, nodes = bench.synthetic_magnet_batch(name, 1, 256, 2000, "cuda")
for _ in range(2):
    model.validation_step(batch, 0)
sites = collections.Counter()
orig = packing.split_fp16x2


def spy(w):
    fr = traceback.extract_stack(limit=8)
    sites[" <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(fr[:-1]))] += 1
    return orig(w)


packing.split_fp16x2 = spy
abs_sites = collections.Counter()
for opname in ("abs", "equal", "isfinite", "max", "all"):
    o = getattr(torch.Tensor, opname, None) or getattr(torch, opname)
    def mk(o=o, opname=opname):
        def f(*a, **k):
            fr = traceback.extract_stack(limit=6)
            abs_sites[opname + " @ " + " <- ".join(f"{x.filename.split('/')[-1]}:{x.lineno}" for x in reversed(fr[:-1]))] += 1
            return o(*a, **k)
        return f
    if hasattr(torch.Tensor, opname): setattr(torch.Tensor, opname, mk())
    if hasattr(torch, opname): setattr(torch, opname, mk(getattr(torch, opname)))
model.validation_step(batch, 0)
torch.cuda.synchronize()
print("split_fp16x2 calls in one warm rollout:", sum(sites.values()))
for k, v in sites.most_common(6):
    print(v, k)
for k, v in abs_sites.most_common(10):
    print(v, k)
