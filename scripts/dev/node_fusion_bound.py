"""Dev: what could ONE persistent node kernel per hop boundary (update of hop l, grid-wide InstanceNorm statistics, halves of hop l + 1 on
resident rows - VERDICT r5 item 6) save at most?  Measured on the metric graph (4 x 256^2 nodes) with the product library:
  * the hop's node-level sequence as the rollout launches it (update2 -> statistics -> halves), back to back;
  * each kernel alone;
  * halves right behind the update that wrote its input (warm: what the separate kernels get today) against halves behind 2 GB of other
    traffic (cold: its input comes from HBM) - the difference bounds what keeping h on the chip is worth;
  * the same sequence with the statistics formed in the update's epilogue (fused_node_stats, the measured variant of DESIGN.md 14.7)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from magnet_amd import ops  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402

B, W = 4, 256
Ns, N = W * W, B * W * W
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
lyr = m._packed().layers[0]
H, kin = m.hidden_features, m.kin
h, agg, g = (torch.randn(N, H, device="cuda") for _ in range(3))
node_in = torch.randn(N, kin, device="cuda")
A, Bm = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
junk = torch.empty(512 * 1024 * 1024, device="cuda")          # 2 GB
st0 = ops.instance_norm_stats(h, B, Ns)
u1, u2 = (lyr.upd1s, lyr.upd2s) if hasattr(lyr, "upd1s") else (lyr.upd1, lyr.upd2)


def update():
    ops.mpnn_node_update(h, agg, node_in, u1, u2, g, stats=st0, rows_per_graph=Ns)


def update_with_stats():
    return ops.mpnn_node_update(h, agg, node_in, u1, u2, g, stats=st0, rows_per_graph=Ns, stats_out_graphs=B)


def stats():
    return ops.instance_norm_stats(g, B, Ns)


def halves(st):
    ops.mpnn_node_halves(g, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, A, Bm, stats=st, rows_per_graph=Ns)


def timeit(fn, reps=20, pre=None):
    tot = 0.0
    for i in range(reps + 3):
        if pre is not None:
            pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if i >= 3:
            tot += s.elapsed_time(e)
    return tot / reps * 1e3


st = stats()
t_u, t_s, t_h_warm = timeit(update), timeit(stats), timeit(lambda: halves(st), pre=update)
t_h_cold = timeit(lambda: halves(st), pre=lambda: junk.fill_(1.0))
t_seq = timeit(lambda: (update(), halves(stats())))
t_seq_f = timeit(lambda: halves(update_with_stats()))
print(f"update2 {t_u:.1f} us   statistics {t_s:.1f} us   halves behind the update (warm) {t_h_warm:.1f} us   halves behind 2 GB of other traffic (cold) {t_h_cold:.1f} us")
print(f"sequence update -> statistics -> halves: {t_seq:.1f} us (sum of the parts {t_u + t_s + t_h_warm:.1f});  with the statistics in the update's epilogue: {t_seq_f:.1f} us")
print(f"upper bound of a persistent fused kernel's gain per hop: statistics {t_s:.1f} + what the warm caches do not already give ({t_h_cold - t_h_warm:+.1f} is the whole "
      f"HBM-vs-cache difference of the input) + two launch boundaries - a grid-wide arrive / wait (4 - 7 us at 256 workgroups, MI355X_MICROARCH.md barrier-xcd)")
