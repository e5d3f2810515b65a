"""Packed weights re-packed IN PLACE after an optimizer step (magnet_amd.packing.PackRecorder) must be exactly what a fresh build
of the packs gives: after several training steps the model's loss and gradients on a new batch equal, bit for bit, those of a twin that
rebuilds every pack from scratch (MAGNET_PACK_REPLAY=0 semantics) from the same state dict."""
import copy
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _twin_run(tag, steps=3):
    import bench
    from magnet_amd.models.factory import FACTORY
    name, hp, mk = bench.ref_shape_cases()[tag]
    hp = dict(hp)
    torch.manual_seed(0)
    model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
    opt = model.configure_optimizers()["optimizer"]
    batch, _ = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        model.training_step(batch, 0).backward()
        opt.step()
    rec = model.__dict__.get("_pack_rec")
    assert rec is not None and rec.usable() and len(rec.launches) > 10, "the packs of this model are not replayed"
    # the twin: same weights, packs built from scratch
    twin = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
    twin.load_state_dict(copy.deepcopy(model.state_dict()))
    twin._pack_replay = False
    out = []
    for m in (model, twin):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(123)                                  # (the training noise of the MAgNet models)
        loss = m.training_step(batch, 0)
        loss.backward()
        out.append((loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    # ... and the inference path (prebuilt EDSR launch plans, hipGraph replay) after one more optimizer step on `model`
    opt.step()
    twin.load_state_dict(copy.deepcopy(model.state_dict()))
    twin._weights_changed()
    vb, _ = mk(False)
    vb = {k: v.cuda() for k, v in vb.items()}
    val = []
    for m in (model, twin):
        m.eval()
        with torch.no_grad():
            val.append(m.validation_step(vb, 0).clone())
    return out, val


@pytest.mark.parametrize("tag", ["gnn_512", "cnn_64"])
def test_replayed_packs_equal_fresh_packs_after_optimizer_steps(tag):
    ((la, ga), (lb, gb)), (va, vb) = _twin_run(tag)
    assert torch.equal(la, lb), (float(la), float(lb))
    assert ga.keys() == gb.keys()
    bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
    assert not bad, bad[:5]
    assert torch.equal(va, vb)


def test_a_weight_beyond_the_fp16_range_is_still_caught_when_the_check_is_deferred():
    """the packers' status word is read at the END of a training step now (packing.deferred_weight_checks): a weight >= 65504 that
    appears between two steps must still send the step to the full-range split, with the result of a model that was there all along"""
    import bench
    from magnet_amd import packing
    from magnet_amd.models.factory import FACTORY
    name, hp, mk = bench.ref_shape_cases()["gnn_512"]
    torch.manual_seed(0)
    model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
    batch, _ = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
    try:
        torch.manual_seed(5)
        model.training_step(batch, 0).backward()               # packs made, recorder armed
        with torch.no_grad():
            model.proj_head.weight[0, 0] = 1.0e5
        model._weights_changed()
        model.zero_grad(set_to_none=True)
        with pytest.warns(RuntimeWarning):
            torch.manual_seed(5)
            loss = model.training_step(batch, 0)
        assert packing.wide() and bool(torch.isfinite(loss))
        twin = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
        twin.load_state_dict(copy.deepcopy(model.state_dict()))
        torch.manual_seed(5)
        assert torch.equal(loss.detach(), twin.training_step(batch, 0).detach())
    finally:
        packing.narrow()


def test_a_weight_the_scaled_backward_planes_cannot_hold_widens_the_step_before_its_backward_runs():
    """31.98 <= |w| < 65504 fits the forward's planes but not the 2^11-scaled ones of linear_bwd128 / the edge backward: the pack
    kernel notes it, the training step reads the note at the end of its forward and repeats itself in bf16x6 - the gradients are
    those of a model that was in bf16x6 all along (before: NaN gradients, reported one optimizer step later)"""
    import bench
    from magnet_amd import packing
    from magnet_amd.models.factory import FACTORY
    name, hp, mk = bench.ref_shape_cases()["gnn_512"]
    torch.manual_seed(0)
    model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
    batch, _ = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
    try:
        with torch.no_grad():
            model.encoder.node_fn[0].linears()[2].weight[3, 5] = 40.0
        with pytest.warns(RuntimeWarning):
            torch.manual_seed(5)
            loss = model.training_step(batch, 0)
        assert packing.wide()
        loss.backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        assert all(bool(torch.isfinite(g).all()) for g in grads.values())
        twin = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
        twin.load_state_dict(copy.deepcopy(model.state_dict()))
        torch.manual_seed(5)
        twin.training_step(batch, 0).backward()
        assert all(torch.equal(g, dict(twin.named_parameters())[n].grad) for n, g in grads.items())
    finally:
        packing.narrow()
