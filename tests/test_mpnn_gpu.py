"""GPU parity: HIP MPNN / MPNN_2d vs the oracle and the golden vectors of the imported reference."""
import pytest
import torch

from tests.helpers import load_case, rel_l2

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-5      # relative L2 on the forward output (BASELINE.json north_star)
INC_TOL = 1e-4      # relative L2 on the increment out - u_last ("hard parts")


def _models(tag, edge_gemm="bf16x6"):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict

    name, hp, salt, batch, d = load_case(tag)
    hp.edge_gemm = edge_gemm
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    return name, hp, batch, d, oracle, model.cuda()


@pytest.mark.parametrize("edge_gemm", ["bf16x6", "fp32"])
@pytest.mark.parametrize("tag", ["mpnn_1d", "mpnn_2d"])
def test_forward_and_rollout_match(tag, edge_gemm):
    name, hp, batch, d, oracle, model = _models(tag, edge_gemm)
    cb = {k: v.cuda() for k, v in batch.items()}
    u, x, t = model._unpack(cb)
    tw = hp.time_window
    B = u.shape[0]
    g = model._build_graph(u[:, :tw], t, x, [model._first_step()] * B)
    # graph: identical edge list (and order) to the reference's radius_graph
    assert torch.equal(g.edge_index.cpu(), torch.from_numpy(d["edge_index"]))
    y = model(g, x[0, -1], t[0, -1], t[0][1] - t[0][0])
    ref = torch.from_numpy(d["fwd_out"])
    assert rel_l2(y, ref) < OUT_TOL
    last = g.x[:, -1:].cpu()
    assert rel_l2(y.cpu() - last, ref - last) < INC_TOL
    # forward accepts a foreign PyG-style Data (edge_index only, arbitrary edge order)
    import types
    perm = torch.randperm(g.edge_index.shape[1], device="cuda")
    foreign = types.SimpleNamespace(x=g.x, pos=g.pos, batch=g.batch, edge_index=g.edge_index[:, perm])
    y2 = model(foreign, x[0, -1], t[0, -1], t[0][1] - t[0][0])
    assert rel_l2(y2, ref) < OUT_TOL
    # free-running rollout (validation_step semantics)
    pred = model.validation_step(cb, 0)
    assert rel_l2(pred, d["val_pred"]) < 5 * OUT_TOL
    assert abs(float(model.logged["val_mae_loss"]) - float(d["val_mae_loss"])) < 1e-5
    # and against the oracle run here
    with torch.no_grad():
        assert rel_l2(pred, oracle.validation_step(batch)) < 5 * OUT_TOL


def test_forward_is_deterministic():
    name, hp, batch, d, oracle, model = _models("mpnn_2d")
    cb = {k: v.cuda() for k, v in batch.items()}
    a = model.validation_step(cb, 0).clone()
    b = model.validation_step(cb, 0)
    assert torch.equal(a, b)
