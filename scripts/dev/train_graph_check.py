"""Dev: graphs.TrainStepGraph against the eager training step at the reference's shapes (bench.ref_shape_cases): eight-step loss
trajectories bit for bit, ms per step of both.  usage: train_graph_check.py [case ...]   (default: gnn_64 gnn_512 cnn_64 c4a)"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from magnet_amd import packing  # noqa: E402
from magnet_amd.graphs import TrainStepGraph  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402

dev = torch.device("cuda", 0)
cases = bench.ref_shape_cases()
for key in sys.argv[1:] or ["gnn_64", "gnn_512", "cnn_64", "c4a"]:
    name, hp, mk = cases[key]
    torch.manual_seed(0)
    batch, nodes = mk(True)
    batch = {k: v.to(dev) for k, v in batch.items()}
    res = {}
    for mode in ("eager", "graph"):
        packing.narrow()
        torch.manual_seed(0)
        m = FACTORY[name](types.SimpleNamespace(**dict(hp, lr=1e-6))).to(dev).train()
        opt = m.configure_optimizers()["optimizer"]
        if mode == "graph":
            step = TrainStepGraph(m, opt)
        else:
            def step(b, m=m, opt=opt):
                opt.zero_grad()
                loss = m.training_step(b, 0)
                loss.backward()
                opt.step()
                return loss
        torch.manual_seed(7)
        losses = [float(step(batch).detach()) for _ in range(8)]
        torch.cuda.synchronize()
        groups = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                step(batch)
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / 5 * 1e3)
        res[mode] = (losses, min(groups), torch.get_rng_state().sum().item())
        if mode == "graph":
            print(f"  [{key}] replays {step.replays}, eager steps {step.eager_steps}, widened {packing.wide()}")
        del m, opt, step
    same = res["eager"][0] == res["graph"][0]
    print(f"{key:8s} {name:14s} eager {res['eager'][1]:8.2f} ms   graph {res['graph'][1]:8.2f} ms   losses bit-equal over 8 steps: {same}"
          f"   host generator in step: {res['eager'][2] == res['graph'][2]}")
    if not same:
        print("   eager", res["eager"][0], "\n   graph", res["graph"][0])
