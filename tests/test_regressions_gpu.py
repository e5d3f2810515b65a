"""GPU regression tests for the round-1 review findings: stale topology / launch-plan caches, optimizer checkpoint
interchange with torch.optim.Adam, the fp16x3 range guard, training-time input noise, the 'knn' / 'sph' INR blends and
the multi-rank launcher of bench.py."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

from tests.helpers import forward_args, load_case, rel_l2

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(tag, **over):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case(tag)
    for k, v in over.items():
        setattr(hp, k, v)
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    return name, hp, batch, d, oracle, model.cuda()


# ------------------------------------------------------------------------------------------- topology cache
def test_new_query_coordinates_rebuild_the_graph():
    """MAgNet training draws new query coordinates every step (reference datamodule/dataset.py:95, dataset_2d.py:111).  In
    round 1 the cache key was (address, version, shape): the previous step's tensor was freed, the caching allocator handed
    its address to the next one and the stale joint graph / k-NN table was used.  Now an entry keeps its key tensor alive
    (so the address cannot come back) - same-shape new coordinates, fresh or written in place, must give the oracle's
    result of THOSE coordinates."""
    name, hp, batch, d, oracle, model = _pair("magnet_gnn_tiny")
    T = hp.time_slice
    g = torch.Generator().manual_seed(5)
    outs, wants, ptrs = [], [], []
    for step in range(3):
        b = dict(batch)
        b["coords_hr"] = (torch.rand(batch["coords_hr"].shape, generator=g) * 2 - 1)
        args = forward_args(name, hp, b)
        dev_args = [a.cuda() for a in args]           # fresh device tensors; the previous step's are freed here
        ptrs.append(dev_args[2].data_ptr())
        outs.append([o.cpu() for o in model(*dev_args)])
        with torch.no_grad():
            wants.append(oracle(*args))
        del dev_args
    for got, want in zip(outs, wants):
        for g_, w_ in zip(got, want):
            assert rel_l2(g_, w_) < 1e-5
    # the same storage rewritten in place (same address, new version)
    args = [a.cuda() for a in forward_args(name, hp, batch)]
    model(*args)
    new_hr = torch.rand(batch["coords_hr"].shape, generator=g) * 2 - 1
    args[2].copy_(new_hr)
    got = model(*args)
    b = dict(batch, coords_hr=new_hr)
    with torch.no_grad():
        want = oracle(*forward_args(name, hp, b))
    for g_, w_ in zip(got, want):
        assert rel_l2(g_, w_) < 1e-5


def test_mpnn_mesh_change_rebuilds_the_graph():
    """irregular MPNN meshes change from batch to batch (datamodule/dataset_2d.py:38-59)"""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_1d")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    ptrs = []
    for scale in (1.0, 0.7, 1.3):
        b = {k: v.clone() for k, v in batch.items()}
        b["x"] = batch["x"] * scale                                    # same shape, different spacing -> different radius graph
        cb = {k: v.cuda() for k, v in b.items()}
        ptrs.append(cb["x"].data_ptr())
        pred = model.validation_step(cb, 0).cpu()
        del cb
        with torch.no_grad():
            want = oracle.validation_step(b)
        assert rel_l2(pred, want) < 5e-5, scale


def test_equal_coordinates_in_a_new_tensor_reuse_the_graph():
    name, hp, batch, d, oracle, model = _pair("magnet_gnn_tiny")
    args = forward_args(name, hp, batch)
    a1 = [a.cuda() for a in args]
    a2 = [a.cuda() for a in args]                                      # equal content, different storage (both alive)
    o1 = model(*a1)
    topo_before = [t for _, t in model._topo_cache.topologies()]
    o2 = model(*a2)
    topo_after = [t for _, t in model._topo_cache.topologies()]
    assert len(topo_before) == len(topo_after) and all(any(x is y for y in topo_before) for x in topo_after)
    assert all(torch.equal(x, y) for x, y in zip(o1, o2))


# ------------------------------------------------------------------------------------------- captured rollouts (round-2 review)
def test_rollout_graph_keeps_the_topology_tensors_it_compares_by_identity():
    """RolloutGraph skips the content compare of the coordinates when the caller passes "the same tensor" again.  Round 2 stored
    only (address, shape, strides, version): a validation loop that frees batch k before batch k + 1 arrives gets the same
    address back with version 0, the compare is skipped and the graph replays the OLD mesh.  The entry now keeps the tensor
    (the address cannot be recycled while it can still match) - every batch must give the oracle's result for ITS mesh."""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_1d")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](types.SimpleNamespace(**vars(hp), hip_graph=True)).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    for scale in (1.0, 0.7, 0.7, 1.3, 1.0):
        b = {k: v.clone() for k, v in batch.items()}
        b["x"] = batch["x"] * scale                                    # same shape, different spacing -> different radius graph
        cb = {k: v.cuda() for k, v in b.items()}
        pred = model.validation_step(cb, 0).cpu()
        kept = [e["seen"]["x"][0] for e in model._rollout_graph._entries.values()]
        assert any(k is cb["x"] for k in kept), "the entry must hold the tensor whose identity it trusts"
        del cb, kept                                                   # the only other reference is the entry's
        with torch.no_grad():
            want = oracle.validation_step(b)
        assert rel_l2(pred, want) < 5e-5, scale


def test_a_mesh_that_changes_every_batch_is_not_captured_again_and_again():
    """irregular data brings new coordinates with every batch (datamodule/dataset_2d.py:99-101): a capture is three rollouts and would be
    thrown away by the next batch.  After two captures in a row that a different mesh replaced before they paid off, that batch shape
    is enqueued kernel by kernel (RolloutGraph.volatile); every batch still gets the oracle's result for ITS mesh, and a shape whose
    mesh stays is still replayed."""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_1d")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](types.SimpleNamespace(**vars(hp), hip_graph=True)).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    captures = []
    orig = model._rollout_graph._capture
    model._rollout_graph._capture = lambda fn, b: (captures.append(1), orig(fn, b))[1]
    for scale in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5):
        b = {k: v.clone() for k, v in batch.items()}
        b["x"] = batch["x"] * scale
        pred = model.validation_step({k: v.cuda() for k, v in b.items()}, 0).cpu()
        with torch.no_grad():
            want = oracle.validation_step(b)
        assert rel_l2(pred, want) < 5e-5, scale
    assert len(captures) == 3, captures            # 1.0, 0.9, 0.8 were captured; from then on the shape counts as volatile
    half = {k: v[:1].clone().cuda() for k, v in batch.items()}         # another batch shape, one mesh: captured once, replayed after
    for _ in range(3):
        model.validation_step(half, 0)
    assert len(captures) == 4


@pytest.mark.parametrize("tag", ["magnet_cnn_1d", "magnet_cnn_2d", "magnet_gnn_tiny"])
def test_captured_rollouts_outlive_the_caches_they_were_built_from(tag):
    """A captured rollout bakes in pointers to buffers owned by model-side caches: EDSR keeps ONE launch plan (feature maps with
    their zero-padding rows), TopologyCache evicts its oldest CSR arrays.  Validate geometry A, then others (the caches move
    on), let other tensors take over whatever was freed, validate A again: the replay of A's graph must still be exact."""
    name, hp, batch, d, oracle, model = _pair(tag, hip_graph=True)
    A = {k: v.cuda() for k, v in batch.items()}
    first = model.validation_step(A, 0).clone()
    assert len(model._rollout_graph._entries) == 1
    for sl in (slice(0, 1), slice(1, 2)):                              # other geometries: new EDSR plan, new topologies
        model.validation_step({k: v[sl].contiguous() for k, v in A.items()}, 0)
    model._topo_cache.clear()                                          # ... and every cached topology evicted
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (64, 257, 1024, 4096, 16385, 65536, 300000) for _ in range(6)]
    torch.cuda.synchronize()
    again = model.validation_step(A, 0)
    assert len(model._rollout_graph._entries) == 2                     # A was replayed, not re-captured (the two slices share a key)
    assert torch.equal(first, again)
    del junk
    with torch.no_grad():
        want = oracle.validation_step(batch)
    assert rel_l2(again, want) < 5e-5


# ------------------------------------------------------------------------------------------- EDSR launch plans
@pytest.mark.parametrize("tag", ["magnet_cnn_1d", "magnet_cnn_2d"])
def test_validation_after_fused_adam_steps_uses_the_new_weights(tag):
    """Lightning's order: sanity validation -> train -> validation.  The EDSR launch plans built by the first validation
    hold pointers to the packed weights; FusedAdam rewrites the parameters behind torch's version counters."""
    from magnet_amd.models.factory import FACTORY
    name, hp, batch, d, oracle, model = _pair(tag)
    cb = {k: v.cuda() for k, v in batch.items()}
    before = model.validation_step(cb, 0).clone()
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        model.training_step(cb, 0).backward()
        opt.step()
    model.eval()
    after = model.validation_step(cb, 0)
    fresh = FACTORY[name](hp).eval()
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    want = fresh.cuda().validation_step(cb, 0)
    assert rel_l2(after, before) > 1e-4, "three Adam steps at lr 1e-3 must move the prediction"
    assert torch.equal(after, want)


# ------------------------------------------------------------------------------------------- optimizer checkpoints
def test_fused_adam_and_torch_adam_exchange_checkpoints():
    """state_dict() is torch.optim.Adam's layout (what the reference's Lightning checkpoints hold): a FusedAdam checkpoint
    resumes under torch.optim.Adam and the other way round, and both continue like an uninterrupted torch.optim.Adam."""
    from magnet_amd.optim import FusedAdam
    shapes = [(64, 33), (7,), (5, 3, 2)]

    def make():
        torch.manual_seed(3)
        return [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]

    grads = [[torch.randn(*s, device="cuda") for s in shapes] for _ in range(6)]

    def run(opt, params, gs):
        for g in gs:
            for p, gg in zip(params, g):
                p.grad = gg.clone()
            opt.step()

    ref = make()
    oref = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-3)
    run(oref, ref, grads)
    for first, second in ((FusedAdam, torch.optim.Adam), (torch.optim.Adam, FusedAdam)):
        a = make()
        oa = first(a, lr=1e-2, weight_decay=1e-3)
        run(oa, a, grads[:3])
        sd = oa.state_dict()
        assert sorted(sd["state"].keys()) == [0, 1, 2] and set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}
        assert "magnet_flat" not in sd
        b = make()
        with torch.no_grad():
            for p, q in zip(b, a):
                p.copy_(q)
        ob = second(b, lr=1e-2, weight_decay=1e-3)
        ob.load_state_dict(sd)
        run(ob, b, grads[3:])
        for p, q in zip(b, ref):
            assert rel_l2(p, q) < 1e-6, (first.__name__, second.__name__)
    # state for a parameter this optimizer does not train is refused, not silently dropped
    c = make()
    oc = FusedAdam(c, lr=1e-2)
    bad = oref.state_dict()
    bad["state"] = {**bad["state"], 7: bad["state"][0]}
    with pytest.raises(ValueError):
        oc.load_state_dict(bad)


def test_fused_adam_loads_reference_checkpoints_with_frozen_and_unused_parameters():
    """The reference builds Adam(self.parameters()) (models/mpnn.py:206-214): its checkpoint indexes EVERY parameter, has no
    state entry for a parameter that never received a gradient, and counts frozen parameters.  FusedAdam trains only the
    requires_grad ones, so the indices have to be mapped, and a missing entry means zero moments at the common step."""
    from magnet_amd.optim import FusedAdam
    shapes = [(16, 9), (5,), (4, 3), (6,)]
    FROZEN, UNUSED = 1, 2

    def make():
        torch.manual_seed(11)
        ps = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
        ps[FROZEN].requires_grad_(False)
        return ps

    grads = [[torch.randn(*s, device="cuda") for s in shapes] for _ in range(5)]

    def run(opt, params, gs, skip=()):
        for g in gs:
            for i, (p, gg) in enumerate(zip(params, g)):
                p.grad = None if (i in skip or not p.requires_grad) else gg.clone()
            opt.step()

    # uninterrupted torch run: parameter UNUSED gets no gradient in the first three steps, then gradients
    ref = make()
    oref = torch.optim.Adam(ref, lr=1e-2)
    run(oref, ref, grads[:3], skip=(UNUSED,))
    sd = oref.state_dict()
    assert sorted(sd["state"].keys()) == [0, 3] and sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    # ... resumed under FusedAdam (which sees zero gradients where torch saw None: Adam's update of a zero-moment parameter is zero)
    b = make()
    with torch.no_grad():
        for p, q in zip(b, ref):
            p.copy_(q)
    ob = FusedAdam(b, lr=1e-2)
    ob.load_state_dict(sd)
    assert ob.steps == 3
    run(ob, b, grads[3:], skip=(UNUSED,))
    run(oref, ref, grads[3:], skip=(UNUSED,))
    for i, (p, q) in enumerate(zip(b, ref)):
        assert rel_l2(p, q) < 1e-6, i
    # and back: FusedAdam's checkpoint counts all four parameters, no entry for the frozen one, and torch.optim.Adam loads it
    sd2 = ob.state_dict()
    assert sd2["param_groups"][0]["params"] == [0, 1, 2, 3] and FROZEN not in sd2["state"] and 0 in sd2["state"]
    c = make()
    torch.optim.Adam(c, lr=1e-2).load_state_dict(sd2)


# ------------------------------------------------------------------------------------------- the whole fp32 range by default
@pytest.fixture
def narrow_after():
    """packing.widen() is process-wide and sticky: tests that trip it hand the process back in the configured modes"""
    from magnet_amd import packing
    packing.narrow()
    yield packing
    packing.narrow()


_VALUE_KEYS = {"mpnn": ("u",), "mpnn_2d": ("u",), "magnet_gnn": ("lr_frames", "hr_points"), "magnet_cnn": ("lr_frames", "hr_points"),
               "magnet_cnn_2d": ("lr_frames", "hr_points")}


def _scaled_pair(tag, scale):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case(tag)
    batch = {k: (v * scale if k in _VALUE_KEYS[name] else v.clone()) for k, v in batch.items()}
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp).eval()                      # default hparams: nobody selects a GEMM mode
    model.load_state_dict(oracle.state_dict())
    with torch.no_grad():
        want = oracle.validation_step(batch)
    return name, batch, want, model.cuda()


@pytest.mark.parametrize("scale", [1e-4, 1e4])
@pytest.mark.parametrize("tag", ["mpnn_1d", "mpnn_2d", "magnet_gnn_tiny", "magnet_cnn_1d", "magnet_cnn_2d"])
def test_inputs_scaled_by_1e4_either_way_meet_the_parity_bar_in_the_default_mode(tag, scale, narrow_after):
    """configs/trainer/default.yaml:5: the reference computes in fp32, whatever the magnitude of the (un-normalised) fields.  The default
    fp16x3 split holds a value as two fp16 pieces - 22 bits while the value is a normal fp16, an ABSOLUTE floor of ~3e-11 below that -
    so 1e-4-sized and 1e4-sized fields are inside its range; the rollout must meet the same bar as the unit-sized fixtures and must
    not have needed the full-range fallback."""
    name, batch, want, model = _scaled_pair(tag, scale)
    pred = model.validation_step({k: v.cuda() for k, v in batch.items()}, 0)
    assert torch.isfinite(pred).all()
    assert rel_l2(pred, want) < 5e-5, rel_l2(pred, want)
    if name.startswith("mpnn"):                           # forward-level bar of north_star on the first call's output
        tw = model.time_window
        assert rel_l2(pred[:, :tw], want[:, :tw]) < 1e-5
    assert not narrow_after.wide()


@pytest.mark.parametrize("tag", ["mpnn_2d", "magnet_gnn_tiny", "magnet_cnn_2d"])
def test_fields_beyond_the_fp16_range_get_the_reference_result_not_an_error(tag, narrow_after):
    """|u| ~ 1e5 >= 65504: the two-fp16-piece representation overflows.  The range guard notices (one status word per rollout), the
    process switches to the full-range bf16x6 split with a warning and the SAME call returns the reference's result; the next call
    runs in that mode without another repeat, packing.narrow() goes back."""
    name, batch, want, model = _scaled_pair(tag, 1.0e5)
    cb = {k: v.cuda() for k, v in batch.items()}
    assert max(float(cb[k].abs().max()) for k in _VALUE_KEYS[name]) > 65504
    with pytest.warns(RuntimeWarning, match="bf16x6"):
        pred = model.validation_step(cb, 0)
    assert narrow_after.wide()
    assert rel_l2(pred, want) < 5e-5, rel_l2(pred, want)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                    # no second switch, no second warning
        again = model.validation_step(cb, 0)
    assert torch.equal(again, pred)
    narrow_after.narrow()
    _, small, want_small, _ = _scaled_pair(tag, 1.0)
    ok = model.validation_step({k: v.cuda() for k, v in small.items()}, 0)            # back in fp16x3, on unit-sized data
    assert rel_l2(ok, want_small) < 5e-5 and not narrow_after.wide()


def test_training_step_on_fields_beyond_the_fp16_range(narrow_after):
    """the same fallback under autograd: the forward with its tapes is repeated in the full-range split, loss and gradients are the
    oracle's"""
    from tests.helpers import oracle_grads
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_2d")
    batch = {k: (v * 1.0e5 if k == "u" else v.clone()) for k, v in batch.items()}
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    want_loss, want = oracle_grads(oracle, batch, hp.teacher_forcing, hp.loss)
    model = FACTORY[name](hp)
    fill_state_dict(model, salt)
    model = model.cuda().train()
    with pytest.warns(RuntimeWarning, match="bf16x6"):
        loss = model.training_step({k: v.cuda() for k, v in batch.items()}, 0)
    loss.backward()
    assert abs(float(loss.detach()) - want_loss) < 1e-4 * abs(want_loss)
    for n, p in model.named_parameters():
        g = want[n].float()
        assert rel_l2(p.grad, g) < 2e-3 or float(g.norm()) < 1e-12 * float(abs(want_loss)), (n, rel_l2(p.grad, g))


def test_overflow_inside_the_chain_is_repeated_in_the_full_range_split(narrow_after):
    """inputs below 65504 whose hidden activations leave the fp16 range: the fp16x3 prediction is non-finite from finite inputs; the
    guard sees that at the output and the rollout is repeated in bf16x6, where the same weights give the oracle's (finite) result"""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_1d")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    with torch.no_grad():
        for p in oracle.embedding_mlp.parameters():
            p.mul_(1000.0)                                             # node_in ~ 1 -> embedding output ~ 1e6 after two Linears
        want = oracle.validation_step(batch)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    cb = {k: v.cuda() for k, v in batch.items()}
    assert float(cb["u"].abs().max()) < 100
    with pytest.warns(RuntimeWarning, match="non-finite"):
        pred = model.validation_step(cb, 0)
    assert torch.isfinite(pred).all() and rel_l2(pred, want) < 5e-5


def test_overflow_between_the_linears_of_a_relu_chain_is_caught_although_the_output_is_finite(narrow_after):
    """MAgNet's chains are Linear / ReLU: a hidden activation above 65504 becomes +-inf in the next layer, and ReLU turns -inf (and NaN)
    into 0 - the prediction is finite and wrong, the output check sees nothing.  The chain kernels report the largest activation they
    hand from one Linear to the next into the armed guard's status word (magnet_mlp_args_t.status), and the rollout is repeated in
    bf16x6.  (Found at the reference's magnet_cnn_2d batch size: a diverging run whose masked overflow surfaced as NaN gradients.)"""
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    with torch.no_grad():
        lins = [m for m in oracle.encoder.node_fn[0].modules() if isinstance(m, torch.nn.Linear)]
        for lin in lins[:3]:                                           # inputs ~ 1, weights < 31.98: the third hidden layer is ~ 1e6
            lin.weight.mul_(60.0)
        assert max(float(l.weight.abs().max()) for l in lins) < 31.0
        want = oracle.validation_step(batch)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    cb = {k: v.cuda() for k, v in batch.items()}
    with pytest.warns(RuntimeWarning, match="inside an fp16x3 MLP chain|non-finite"):
        pred = model.validation_step(cb, 0)
    assert narrow_after.wide() and torch.isfinite(pred).all() and rel_l2(pred, want) < 5e-5
    # ... and the kernel-level statement: the flag is raised by the chain launch itself, only while a guard is armed
    narrow_after.narrow()
    from magnet_amd.models.magnet_common import pack_mlp_ln, run_mlp
    seq = model.encoder.node_fn
    pk = pack_mlp_ln(seq, k_pad=32 * ((seq[0].in_dim + 31) // 32))
    x = torch.randn(4096, 32 * ((seq[0].in_dim + 31) // 32), device="cuda")
    y = torch.empty(4096, 32 * pk.ot, device="cuda")
    g = ops.RangeGuard(True)
    run_mlp(4096, ops.ktiles(ops.Term(x, x.shape[1])), pk, y)
    assert g.tripped() is None                                         # (not armed: nothing reported)
    with g.armed():
        run_mlp(4096, ops.ktiles(ops.Term(x, x.shape[1])), pk, y)
    assert "inside an fp16x3 MLP chain" in (g.tripped() or "")


def test_feature_maps_beyond_the_fp16_range_between_two_convolutions_are_caught(narrow_after):
    """EDSR's convolutions are separate launches with a ReLU in between: a feature map that reaches 65504 saturates the next
    convolution's fp16 pieces, and the ReLU after it can hide the resulting -inf.  Every fp16 launch reports the largest value it hands
    on (no LayerNorm behind it) into the armed guard; the rollout is repeated in bf16x6 and meets the oracle."""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("magnet_cnn_2d")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    with torch.no_grad():
        oracle.encoder.head_conv.weight.mul_(2.0e5)                    # inputs ~ 1: the head feature map is ~ 1e5
        assert float(oracle.encoder.head_conv.weight.abs().max()) < 60000
        want = oracle.validation_step(batch)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    cb = {k: v.cuda() for k, v in batch.items()}
    with pytest.warns(RuntimeWarning, match="reached 65504|non-finite"):
        pred = model.validation_step(cb, 0)
    assert narrow_after.wide() and torch.isfinite(pred).all() and rel_l2(pred, want) < 5e-5


def test_weights_beyond_the_fp16_planes_switch_the_split_instead_of_raising(narrow_after):
    """a weight >= 65504 cannot be held as two fp16 pieces: the packs are rebuilt in the full-range split"""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_1d")
    oracle = O.build(name, hp).eval()
    fill_state_dict(oracle, salt)
    with torch.no_grad():
        oracle.gnn_layers[0].update_net_1[0].weight[3, 5] = 7.0e4
        want = oracle.validation_step(batch)
    model = FACTORY[name](hp).eval()
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    with pytest.warns(RuntimeWarning, match="weight"):
        pred = model.validation_step({k: v.cuda() for k, v in batch.items()}, 0)
    assert rel_l2(pred, want) < 5e-5


# ------------------------------------------------------------------------------------------- noise, interpolation modes
def test_training_noise_matches_the_reference_draw_order():
    """magnet_gnn.py:402-407, 424-428: noise * torch.randn(shape) from the host generator, input first, then hr_last, again
    after every forward call.  Same seed -> same loss as the oracle's restatement of that loop."""
    from tests.helpers import oracle_magnet_gnn_grads
    name, hp, batch, d, oracle, model = _pair("magnet_gnn_tiny", noise=0.05)
    model.train()
    torch.manual_seed(77)
    want_loss, _ = oracle_magnet_gnn_grads(oracle, batch, hp.teacher_forcing, hp.loss, dtype=torch.float32, noise=hp.noise)
    torch.manual_seed(77)
    loss = model.training_step({k: v.cuda() for k, v in batch.items()}, 0)
    loss.backward()
    assert abs(float(loss.detach()) - want_loss) < 2e-5 * max(1.0, abs(want_loss))
    torch.manual_seed(78)
    other = model.training_step({k: v.cuda() for k, v in batch.items()}, 0)
    assert abs(float(other.detach()) - want_loss) > 1e-5                 # the noise really enters


@pytest.mark.parametrize("interp", ["knn", "sph"])
def test_inr_interpolation_modes_match_the_oracle(interp):
    """magnet_gnn.py:269-272, 279: 'knn' (inverse squared distance) and 'sph' ((1 - L d^2)^3) weights of the two nearest"""
    name, hp, batch, d, oracle, model = _pair("magnet_gnn_tiny", interpolation=interp)
    args = forward_args(name, hp, batch)
    with torch.no_grad():
        want = oracle(*args)
    got = model(*[a.cuda() for a in args])
    for g_, w_, key in zip(got, want, ("out_hr", "out_lr", "hr_points")):
        assert rel_l2(g_, w_) < 1e-5, (interp, key, rel_l2(g_, w_))


# ------------------------------------------------------------------------------------------- bench.py launcher
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run environment starts two fresh ranks itself; gloo lets them
    share this box's single GPU.  The line must say n_gpus = 2."""
    env = dict(os.environ, MAGNET_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mesh", "64", "--traj", "1", "--steps", "2",
                        "--warmup", "1", "--train-steps", "1", "--no-cpu-baseline", "--no-extra-cases"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_trajectories"] == 2 and line["value"] > 0
    assert line["training_step"]["ms_per_step"] > 0


def test_bench_eight_ranks_on_one_gpu_over_gloo():
    """Multi-GPU readiness without the hardware (README.md:124-125 `trainer.gpus=N trainer.strategy=ddp`): the complete
    `--gpus 8` control flow - eight fresh ranks, rendezvous, batch sharding, barriers, max-over-ranks timing, the flat gradient
    all-reduce, fused Adam - on this box's single GPU over gloo.  Every rank must own different trajectories, the process group
    must have seen 8 ranks, and after two training steps all ranks must hold identical parameters (DDP's invariant)."""
    env = dict(os.environ, MAGNET_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mesh", "64", "--traj", "1", "--steps", "2",
                        "--warmup", "1", "--train-steps", "2", "--no-cpu-baseline", "--no-extra-cases", "--no-precision-modes"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["global_trajectories"] == 8 and line["value"] > 0
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size_seen"] == 8 and len(d["ms_per_step_by_rank"]) == 8
    assert sorted(d["trajectory_seed_by_rank"]) == list(range(1000, 1008))
    assert len(set(d["input_checksum_by_rank"])) == 8                  # eight different sets of trajectories
    ps = d["param_abs_sum_by_rank_after_training"]
    assert len(ps) == 8 and all(p == ps[0] for p in ps), ps            # bitwise identical parameters on every rank
    t = line["training_step"]
    assert t["ms_per_step"] > 0 and t["allreduce_ms"] > 0 and t["allreduce_bytes"] > 2_000_000
    assert t["loss_first_last"][1] != t["loss_first_last"][0]


# ------------------------------------------------------------------------------------------- copies of a warm model
@pytest.mark.parametrize("tag", ["mpnn_2d", "magnet_gnn_tiny", "magnet_cnn_2d"])
def test_deepcopy_and_pickle_of_a_model_after_its_first_forward(tag):
    """EMA / best-model snapshots and spawn-style launchers copy a model that has already run: the packed-weight caches hold ctypes
    function pointers and argument blocks (packing.PackRecorder), captured rollouts, launch plans.  A copy must carry the parameters
    only, rebuild its caches on first use and give the same prediction bit for bit; the original keeps working."""
    import copy
    import pickle
    name, hp, batch, d, oracle, model = _pair(tag)
    cb = {k: v.cuda() for k, v in batch.items()}
    first = model.validation_step(cb, 0)
    first = [t.clone() for t in (first if isinstance(first, (tuple, list)) else (first,))]
    twin = copy.deepcopy(model)
    revived = pickle.loads(pickle.dumps(model))
    for other in (twin, revived, model):
        got = other.validation_step(cb, 0)
        got = got if isinstance(got, (tuple, list)) else (got,)
        assert all(torch.equal(a, b) for a, b in zip(first, got))


# ------------------------------------------------------------------------------------------- run-to-run determinism
def test_training_gradients_and_rollouts_repeat_bitwise_under_a_busy_second_stream():
    """A timing race (one wave reading LDS before the barrier) stayed under every tolerance for two rounds.  Every
    kernel of the library is deterministic by construction, so repeated training steps (loss + all gradients) and eager rollouts
    on the same inputs must be BITWISE equal - here for four small cases with GEMMs running on a second stream and idle gaps in
    between; scripts/stress_determinism.py runs all ten cases 1 000 times (profiles/r5/stress_determinism_v2.txt)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import stress_determinism
    lines = []
    summary, failed = stress_determinism.run("c1,cnn_1d,gnn_1d,c4a", reps=12, gap_every=5, gap=0.1, log=lambda *a, **k: lines.append(" ".join(map(str, a))))
    assert not failed, "\n".join(lines)
    assert all(v["train"]["reps"] == 12 and v["rollout"]["reps"] == 12 and v["train"]["grad_elements"] > 1000 for v in summary.values())


@pytest.mark.gpu
def test_third_party_restatement_is_pinned_wherever_the_wheels_exist():
    """any box that ever has torch_cluster / torch_scatter / torch_geometric 2.0.3 pins oracle/pyg_semantics.py (SURVEY 8
    row a22) without anyone remembering to: scripts/pin_third_party.py --device cuda must print PINNED (or SKIPPED where the wheels are
    absent - this image) and exit 0; a mismatch (exit 1, "NOT PINNED") fails the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "pin_third_party.py"), "--device", "cuda"], capture_output=True,
                       text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and ("PINNED" in r.stdout or "SKIPPED" in r.stdout) and "NOT PINNED" not in r.stdout, out[-3000:]
    if "SKIPPED" in r.stdout:
        pytest.skip("torch_geometric / torch_cluster / torch_scatter are not installed here: a22 stays parity-unpinned (" +
                    r.stdout.strip().splitlines()[-1][:200] + ")")
