"""GPU parity: HIP MAgNet[GNN] / MAgNet[CNN] 1D+2D vs the golden vectors of the imported reference
and vs the oracle run on the host."""
import pytest
import torch

from tests.helpers import forward_args, load_case, rel_l2

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-5
INC_TOL = 2e-4      # increment (out - last) is ~1e-2 of the output norm; fp32 roundoff through 5-10 hops + norms
TAGS = ["magnet_gnn_full", "magnet_gnn_tiny", "magnet_cnn_1d", "magnet_cnn_2d"]


def _models(tag):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict

    name, hp, salt, batch, d = load_case(tag)
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    return name, hp, batch, d, oracle, model.cuda()


@pytest.mark.parametrize("tag", TAGS)
def test_forward_matches_reference(tag):
    name, hp, batch, d, oracle, model = _models(tag)
    args = forward_args(name, hp, batch)
    out = model(*[a.cuda() for a in args])
    hr_last = args[-1]
    for got, key in zip(out, ("fwd_out_hr", "fwd_out_lr", "fwd_hr_points")):
        assert tuple(got.shape) == d[key].shape, key
        assert rel_l2(got, d[key]) < OUT_TOL, (key, rel_l2(got, d[key]))
    # increment of the query nodes: out_hr - hr_last
    ref_inc = torch.from_numpy(d["fwd_out_hr"]) - hr_last[:, None]
    assert rel_l2(out[0].cpu() - hr_last[:, None], ref_inc) < INC_TOL


@pytest.mark.parametrize("tag", TAGS)
def test_rollout_matches_reference(tag):
    name, hp, batch, d, oracle, model = _models(tag)
    cb = {k: v.cuda() for k, v in batch.items()}
    pred = model.validation_step(cb, 0)
    nq = d["val_out_hr"].shape[2]
    assert rel_l2(pred[:, :, :nq], d["val_out_hr"]) < 5 * OUT_TOL
    assert abs(float(model.logged["val_mae_loss"]) - float(d["val_mae_loss"])) < 2e-5
    assert abs(float(model.logged["val_loss"]) - float(d["val_loss"])) < 2e-5
    with torch.no_grad():
        want = oracle.validation_step(batch)
    assert rel_l2(pred, want) < 5 * OUT_TOL
    again = model.validation_step(cb, 0)
    assert torch.equal(pred, again)        # deterministic (no atomics in the value path)
