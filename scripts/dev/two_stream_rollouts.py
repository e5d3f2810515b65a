"""Dev: do two half-batches on two streams beat one full batch?  (The edge kernel is issue bound, the node kernels are HBM bound: if the
GPU ran one stream's node kernels beside the other stream's edge kernel, the rollout would approach max instead of the sum.)
Two mpnn_2d models (same weights), B = 2 trajectories each, hipGraph replays on two streams, against one model with B = 4."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY

dev = torch.device("cuda", 0)
mode = True


def make(B, seed):
    torch.manual_seed(0)
    m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph=mode)).to(dev).eval()
    return m, bench.synthetic_batch(256, B, seed, dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


m4, b4 = make(4, 1000)
t4 = timeit(lambda: m4.validation_step(b4, 0))
print(f"one stream, B = 4: {t4:.2f} ms per rollout of 4 trajectories")
ma, ba = make(2, 1001)
mb, bb = make(2, 1002)
t2 = timeit(lambda: ma.validation_step(ba, 0))
print(f"one stream, B = 2: {t2:.2f} ms per rollout of 2 trajectories ({2 * t2:.2f} for 4)")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(sa):
        ma.validation_step(ba, 0)
    with torch.cuda.stream(sb):
        mb.validation_step(bb, 0)


for s_ in (sa, sb):                 # captures must be made on the stream they will replay on
    with torch.cuda.stream(s_):
        (ma if s_ is sa else mb)._rollout_graph.clear()
t22 = timeit(both)
print(f"two streams, B = 2 + 2: {t22:.2f} ms per pair of rollouts (4 trajectories)")
