"""Dev: cycle trace (s_memtime) of wave 0 of one workgroup of k_node_halves (needs python -m magnet_amd.build --dev)."""
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")
import sys, types, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY
import bench
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 37
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = 4 * 256 * 256
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
lyr = m._packed().layers[0]
H, kin = 128, m.kin
g, hn = torch.randn(N, H, device="cuda"), torch.empty(N, H, device="cuda")
node_in = torch.randn(N, kin, device="cuda"); AB = torch.empty(2, N, H, device="cuda")
st = ops.instance_norm_stats(g, 4, N // 4)
run = (lambda: ops.mpnn_node_halves(g, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, AB[0], AB[1], stats=st, rows_per_graph=N // 4, h_norm=hn)) if fused else \
      (lambda: ops.mpnn_node_halves(g, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, AB[0], AB[1]))
for _ in range(3): run()
buf = torch.zeros(1000, dtype=torch.int64, device="cuda")
_lib.lib().magnet_debug_set_mlp_trace(C.c_void_p(buf.data_ptr()), wg)
run(); torch.cuda.synchronize()
_lib.lib().magnet_debug_set_mlp_trace(C.c_void_p(0), 0)
b = buf.cpu().view(-1, 2); b = b[b[:, 1] > 0]
names = {40: "kernel start", 41: "weights in LDS", 42: "tile top", 43: "x tile ready", 44: "mma done (+h store)", 45: "fold done", 46: "stores issued"}
t0 = int(b[0, 0]); prev = t0; tot = {}
for t, tag in b.tolist():
    tot[tag] = tot.get(tag, 0) + (t - prev); prev = t
T = int(b[-1, 0]) - t0
print(f"workgroup {wg} wave 0, fused={fused}: {T} ticks, {len(b)} stamps")
for tag, dt in sorted(tot.items()):
    print(f"  time ending at [{names.get(tag, tag):22s}]: {dt:8d} ({100 * dt / T:5.1f} %)")
print("first two tiles:", [(int(tag), int(t - t0)) for t, tag in b.tolist()[:32]])
