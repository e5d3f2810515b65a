"""Dev microbench: the two node-level launches of one MPNN_2d hop at the metric size (4 x 256^2 nodes) with the run-time
ablations of the split-GEMM MLP kernel (needs the development library: python -m magnet_amd.build --dev)."""
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")
import sys, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 256 * 256
hp = types.SimpleNamespace(**bench.HP)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
P = m._packed()
lyr = P.layers[0]
H, kin = m.hidden_features, m.kin
h, agg, g = (torch.randn(N, H, device="cuda") for _ in range(3))
node_in = torch.randn(N, kin, device="cuda")
AB = torch.empty(2, N, H, device="cuda")
T = ops.Term


def halves():
    ops.mlp(N, ops.ktiles(T(h, H), T(node_in, kin)), [lyr.pre_a_l2e, lyr.pre_b_l2e], nl=1, ht=4, ot=4, act=ops.ACT_NONE, y=AB, ldy=H,
            colgroups=2, groups_in_block=True, y_group_stride=N * H)


def update():
    ops.mlp(N, ops.ktiles(T(h, H), T(agg, H), T(node_in, kin)), [lyr.upd1, lyr.upd2], nl=2, ht=4, ot=4, act=ops.ACT_SWISH,
            last_act=True, epi=ops.EPI_STORE_RES, res=h, y=g)


def norm():
    ops.instance_norm(g, 4, N // 4, out=h)


def timeit(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


hn = torch.empty(N, H, device="cuda")


def halves_fused():
    st = ops.instance_norm_stats(g, 4, N // 4)
    ops.mpnn_node_halves(g, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, AB[0], AB[1], stats=st, rows_per_graph=N // 4, h_norm=hn)


def halves_resident_only():
    ops.mpnn_node_halves(h, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, AB[0], AB[1])


print(f"N={N} kin={kin}")
st0 = ops.instance_norm_stats(h, 4, N // 4)
print(f"persistent update kernel: {timeit(lambda: ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, g))*1e3:.1f} us;  with norm on load: "
      f"{timeit(lambda: ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, g, stats=st0, rows_per_graph=N // 4))*1e3:.1f} us;  generic update: {timeit(update)*1e3:.1f} us")
print(f"resident halves with norm on load, no h_norm: {timeit(lambda: ops.mpnn_node_halves(h, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, AB[0], AB[1], stats=st0, rows_per_graph=N // 4))*1e3:.1f} us")
for abl in [0, 1, 4, 16, 5, 17, 20, 21]:
    _lib.lib().magnet_debug_set_mlp_ablation(abl)
    print(f"resident halves abl {abl:2d}: {timeit(halves_resident_only)*1e3:7.1f} us    with norm + h_norm: {timeit(halves_fused)*1e3:7.1f} us (incl. stats)")
_lib.lib().magnet_debug_set_mlp_ablation(0)
print(f"stats + resident halves with norm on load + h_norm: {timeit(halves_fused)*1e3:.1f} us;  resident halves alone: {timeit(halves_resident_only)*1e3:.1f} us"
      f";  stats alone: {timeit(lambda: ops.instance_norm_stats(g, 4, N // 4))*1e3:.1f} us")
for name, fn, byts in (("halves", halves, N * (H + kin + 2 * H) * 4), ("update", update, N * (3 * H + kin) * 4)):
    for abl in [0, 1, 2, 4, 8, 16, 1 | 2, 1 | 2 | 8, 1 | 2 | 16, 31]:
        _lib.lib().magnet_debug_set_mlp_ablation(abl)
        ms = timeit(fn)
        print(f"{name} abl {abl:2d}: {ms*1e3:7.1f} us   ({byts/ms/1e6:.0f} GB/s algorithmic)")
_lib.lib().magnet_debug_set_mlp_ablation(0)
print(f"instance_norm: {timeit(norm)*1e3:.1f} us")
