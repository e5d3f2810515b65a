"""CPU: the oracle restatement reproduces the golden vectors produced by the imported reference."""
import pytest
import torch

from oracle import models as O
from oracle.weightgen import checksum, fill_state_dict
from tests.helpers import CASES, forward_args, load_case, rel_l2

TOL = 2e-6   # same torch CPU kernels, different batching of the GEMMs


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_golden(tag):
    name, hp, salt, batch, d = load_case(tag)
    model = O.build(name, hp).eval()
    fill_state_dict(model, salt)
    assert abs(checksum(model) - float(d["weight_checksum"])) <= 1e-6 * float(d["weight_checksum"])
    with torch.no_grad():
        if name.startswith("mpnn"):
            u = batch["u"].permute(0, 2, 1)
            x = batch["x"].squeeze(-1) if name == "mpnn" else batch["x"]
            t = batch["t"]
            tw = hp.time_window
            g = model._build_graph(u[:, :tw], t, x, [model._step_index(0)] * u.shape[0])
            assert torch.equal(g.edge_index, torch.from_numpy(d["edge_index"]))
            y = model(g, x[0, -1], t[0, -1], t[0][1] - t[0][0])
            assert rel_l2(y, d["fwd_out"]) < TOL
            inc = y - g.x[:, -1:]
            inc_ref = torch.from_numpy(d["fwd_out"]) - g.x[:, -1:]
            assert rel_l2(inc, inc_ref) < 50 * TOL
            pred = model.validation_step(batch)
            assert rel_l2(pred, d["val_pred"]) < 5 * TOL
        else:
            o = model(*forward_args(name, hp, batch))
            for got, key in zip(o, ("fwd_out_hr", "fwd_out_lr", "fwd_hr_points")):
                assert got.shape == d[key].shape
                assert rel_l2(got, d[key]) < TOL, key
            model.validation_step(batch)
        assert abs(float(model.logged["val_mae_loss"]) - float(d["val_mae_loss"])) < 1e-5
        assert abs(float(model.logged["val_loss"]) - float(d["val_loss"])) < 1e-5


@pytest.mark.parametrize("teacher_forcing", [False, True])
@pytest.mark.parametrize("tag", ["mpnn_1d", "mpnn_2d"])
def test_oracle_gradients_match_reference_training_step(tag, teacher_forcing):
    """torch autograd through the oracle's rollout vs the gradients of the reference's own training_step
    (tests/golden/*_grad.npz) - pins the gradient oracle the GPU training tests compare against."""
    from tests.helpers import load_grads, oracle_grads
    name, hp, salt, batch, d = load_case(tag)
    model = O.build(name, hp)
    fill_state_dict(model, salt)
    gold_loss, gold = load_grads(tag, teacher_forcing)
    for dtype, tol in ((torch.float32, 1e-4), (torch.float64, 1e-4)):
        loss, grads = oracle_grads(model, batch, teacher_forcing, hp.loss, dtype=dtype)
        assert abs(loss - gold_loss) < 1e-5 * max(1.0, abs(gold_loss))
        assert set(grads) == set(gold)
        for n, (norm, sample, stride) in gold.items():
            assert abs(float(grads[n].norm()) - norm) < tol * norm, (n, dtype)
            assert rel_l2(grads[n].flatten()[::stride], sample) < tol, (n, dtype)


@pytest.mark.parametrize("teacher_forcing", [False, True])
def test_oracle_magnet_gnn_gradients_match_reference_training_step(teacher_forcing):
    """same pin for MAgNet[GNN] (models/magnet_gnn.py:388-440): the oracle's autograd gradients vs the reference's own"""
    from tests.helpers import load_grads, oracle_magnet_gnn_grads
    name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
    model = O.build(name, hp)
    fill_state_dict(model, salt)
    gold_loss, gold = load_grads("magnet_gnn_tiny", teacher_forcing)
    loss, grads = oracle_magnet_gnn_grads(model, batch, teacher_forcing, hp.loss, dtype=torch.float32)
    assert abs(loss - gold_loss) < 1e-5 * max(1.0, abs(gold_loss))
    assert set(grads) == set(gold)
    for n, (norm, sample, stride) in gold.items():
        assert abs(float(grads[n].norm()) - norm) < 2e-4 * max(norm, 1e-12), n
        assert rel_l2(grads[n].flatten()[::stride], sample) < 2e-4, n


@pytest.mark.parametrize("teacher_forcing", [False, True])
@pytest.mark.parametrize("tag", ["magnet_cnn_1d", "magnet_cnn_2d"])
def test_oracle_magnet_cnn_gradients_match_reference_training_step(tag, teacher_forcing):
    """and for MAgNet[CNN] 1-D / 2-D (models/magnet_cnn.py:393-437, models/magnet_cnn_2d.py:399-443)"""
    from tests.helpers import load_grads, oracle_magnet_cnn_grads
    name, hp, salt, batch, d = load_case(tag)
    model = O.build(name, hp)
    fill_state_dict(model, salt)
    gold_loss, gold = load_grads(tag, teacher_forcing)
    loss, grads = oracle_magnet_cnn_grads(model, batch, teacher_forcing, hp.loss, dtype=torch.float32)
    assert abs(loss - gold_loss) < 1e-5 * max(1.0, abs(gold_loss))
    assert set(grads) == set(gold)
    for n, (norm, sample, stride) in gold.items():
        assert abs(float(grads[n].norm()) - norm) < 5e-4 * max(norm, 1e-12), n
        assert rel_l2(grads[n].flatten()[::stride], sample) < 5e-4, n
