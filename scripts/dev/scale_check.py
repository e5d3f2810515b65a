"""Dev: larger-than-benchmark sizes through the large-graph path (index arithmetic beyond 2^32 bytes): 8 trajectories at 256^2 and 2 at
512^2, fused node kernels vs the generic launches (MAGNET_FUSED_HALVES=0 semantics through the hparam)."""
import sys, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
from tests.helpers import rel_l2
for mesh, B in ((256, 8), (512, 2)):
    outs = []
    for fused in (True, False):
        torch.manual_seed(0)
        m = FACTORY["mpnn_2d"](types.SimpleNamespace(**dict(bench.HP, fused_node_halves=fused, hip_graph=False))).cuda().eval()
        batch = bench.synthetic_batch(mesh, B, 77, "cuda")
        outs.append(m.validation_step(batch, 0))
        torch.cuda.synchronize()
    print(mesh, B, "finite", bool(torch.isfinite(outs[0]).all()), "fused vs generic rel L2", rel_l2(outs[0], outs[1]),
          "peak GB", torch.cuda.max_memory_allocated() / 1e9)
