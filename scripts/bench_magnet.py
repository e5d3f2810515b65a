"""Dev timing of the full-size MAgNet configs of BASELINE.json (C4b magnet_cnn_2d, C5 magnet_gnn), 1 GPU."""
import sys, time, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from magnet_amd.models.factory import FACTORY
from magnet_amd.utils import make_coord

OPT = dict(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Wq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
torch.manual_seed(0)
dev = "cuda"


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps


if which in ("cnn", "both"):
    hp = types.SimpleNamespace(**OPT, time_slice=10, latent_dim=32, num_message_passing_steps=10, mlp_layers=4, mlp_hidden=64,
                               radius=0.1, scales=1, n_chan=128, kernel_size=3, res_scale=1, res_layers=16,
                               teacher_forcing=False, interpolation="area")
    m = FACTORY["magnet_cnn_2d"](hp).to(dev).eval()
    W, T = 64, 10
    x_t = torch.randn(B, T, 1, W, W, device=dev)
    coords = make_coord([Wq, Wq], device=dev)[None].repeat(B, 1, 1)
    cell = torch.full((B, Wq * Wq, 2), 2.0 / Wq, device=dev)
    t = torch.linspace(0, 1, 2 * T, device=dev)[None].repeat(B, 1)
    hr_last = torch.randn(B, Wq * Wq, 1, device=dev)
    dt = timeit(lambda: m(x_t, coords, cell, t, hr_last))
    n = W * W + Wq * Wq
    topo = m._topo_cache.topologies()[0][1]
    deg = (topo.csr.rowptr[1:] - topo.csr.rowptr[:-1]).float()
    print(f"magnet_cnn_2d C4b B={B}: forward {dt*1e3:.1f} ms, nodes {n}, E {topo.csr.E}, in-degree min/mean/max {deg.min():.0f}/{deg.mean():.1f}/{deg.max():.0f}"
          f" -> {B*n*T/dt:.3e} node-updates/s (per forward of {T} steps), peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
    del m
if which in ("gnn", "both"):
    torch.cuda.reset_peak_memory_stats()
    hp = types.SimpleNamespace(**OPT, time_slice=10, latent_dim=128, num_message_passing_steps=5, mlp_layers=4, mlp_hidden=128,
                               radius=0.08, n_chan=128, teacher_forcing=False, codec_neighbors=4, noise=0, interpolation="area")
    m = FACTORY["magnet_gnn"](hp).to(dev).eval()
    T, L = 10, 256
    pts = torch.clamp(torch.randn(512, 2) * 0.25, -1, 1)
    pts = 2 * (pts - pts.min(0).values) / (pts.max(0).values - pts.min(0).values) - 1
    lr_c = pts[::2][None].repeat(B, 1, 1).to(dev)
    hr_c = make_coord([Wq, Wq], device=dev)[None].repeat(B, 1, 1)
    x_lr = torch.randn(B, T, 1, L, device=dev)
    t = torch.linspace(0, 1, 2 * T, device=dev)[None].repeat(B, 1)
    hr_last = torch.randn(B, Wq * Wq, 1, device=dev)
    dt = timeit(lambda: m(x_lr, lr_c, hr_c, t, hr_last))
    n = L + Wq * Wq
    topo = [v for k, v in m._topo_cache.topologies() if k[0] == "all"][0]
    deg = (topo.csr.rowptr[1:] - topo.csr.rowptr[:-1]).float()
    print(f"magnet_gnn C5 B={B}: forward {dt*1e3:.1f} ms, nodes {n}, E {topo.csr.E}, in-degree min/mean/max {deg.min():.0f}/{deg.mean():.1f}/{deg.max():.0f}"
          f" -> {B*n*T/dt:.3e} node-updates/s (per forward of {T} steps), peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
