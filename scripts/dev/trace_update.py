"""Dev: cycle trace of wave 0 of one workgroup of k_node_update (needs python -m magnet_amd.build --dev)."""
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")
import sys, types, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY
import bench
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 37
norm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = 4 * 256 * 256
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
lyr = m._packed().layers[0]
H, kin = 128, m.kin
h, agg, g = (torch.randn(N, H, device="cuda") for _ in range(3))
node_in = torch.randn(N, kin, device="cuda")
st = ops.instance_norm_stats(h, 4, N // 4) if norm else None
run = lambda: ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, g, stats=st, rows_per_graph=N // 4)
for _ in range(3): run()
buf = torch.zeros(1000, dtype=torch.int64, device="cuda")
_lib.lib().magnet_debug_set_mlp_trace(C.c_void_p(buf.data_ptr()), wg)
run(); torch.cuda.synchronize()
_lib.lib().magnet_debug_set_mlp_trace(C.c_void_p(0), 0)
b = buf.cpu().view(-1, 2); b = b[b[:, 1] > 0]
names = {50: "start", 51: "step top", 52: "barrier done", 53: "chunk moved + x ready", 54: "mma done", 55: "fold 0 done", 56: "layer 1 done", 57: "epilogue done"}
t0 = int(b[0, 0]); prev = t0; tot = {}
for t, tag in b.tolist():
    tot[tag] = tot.get(tag, 0) + (t - prev); prev = t
T = int(b[-1, 0]) - t0
print(f"workgroup {wg} wave 0, norm={norm}: {T} ticks, {len(b)} stamps")
for tag, dt in sorted(tot.items()):
    print(f"  time ending at [{names.get(tag, tag):22s}]: {dt:8d} ({100 * dt / T:5.1f} %)")
print("second tile:", [(int(tag), int(t - t0)) for t, tag in b.tolist()[41:82]])
