"""CPU-only checks: the C-ABI library loads and exports every declared symbol, host-side model
logic (state-dict keys, weight packing, factory) - no kernels are launched."""
import ctypes

import pytest
import torch

from magnet_amd import _lib
from magnet_amd.packing import pack_linear
from tests.helpers import CASES, load_case


def test_library_exports_every_declared_symbol():
    from magnet_amd.build import build
    build()
    names = _lib.declared_symbols()
    assert len(names) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert _lib.lib() is not None


def test_mlp_struct_layout_matches_header():
    # sizes follow from the C declarations in include/magnet_hip.h
    import re
    hdr = open(_lib.HEADER).read()
    assert int(re.search(r"#define MAGNET_MAX_KT (\d+)", hdr).group(1)) == _lib.MAX_KT
    assert int(re.search(r"#define MAGNET_MAX_LAYERS (\d+)", hdr).group(1)) == _lib.MAX_LAYERS
    assert ctypes.sizeof(_lib.Src) == 32
    assert ctypes.sizeof(_lib.KTile) == 80
    assert ctypes.sizeof(_lib.Layer) == 16
    assert ctypes.sizeof(_lib.MlpArgs) == 8 + 80 * _lib.MAX_KT + 16 * 6 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 2 * 32 + 8 + 8 * 6 + 8 + 8 + 8   # ..., ncolgroups|groups_in_block, y_group_stride, pre[2], x0_out, hid[6], x0_skip, status, status_bit|pad
    # ... and against the C compiler's own reading of the header: size and the offset of every field the kernels were given later
    import os, subprocess, tempfile
    fields = ["pre", "x0_out", "hid", "x0_skip", "status", "status_bit"]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "magnet_hip.h"\nint main(void) { printf("%zu", sizeof(magnet_mlp_args_t));' + \
        "".join(f' printf(" %zu", offsetof(magnet_mlp_args_t, {f}));' for f in fields) + " return 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.dirname(_lib.HEADER), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        got = [int(v) for v in subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_lib.MlpArgs)
    assert got[1:] == [getattr(_lib.MlpArgs, f).offset for f in fields]


def test_supported_mlp_variants():
    lib = _lib.lib()
    assert lib.magnet_mlp_supported(2, 4, 4, _lib.ACT_SWISH, 1, 0, _lib.EPI_STORE_RES) == 1
    assert lib.magnet_mlp_supported(5, 4, 4, _lib.ACT_RELU, 0, 1, _lib.EPI_SEGMEAN) == 1
    assert lib.magnet_mlp_supported(3, 4, 4, _lib.ACT_RELU, 0, 1, _lib.EPI_STORE) == 0


def test_pack_linear_layout():
    w = torch.arange(40 * 70, dtype=torch.float32).reshape(40, 70)
    b = torch.arange(40, dtype=torch.float32)
    wp, bp = pack_linear(w, b)
    kt, nt = 3, 2
    v = wp.view(kt, nt, 4, 64, 4)
    for (k_t, n_t, g, lane, c) in [(0, 0, 0, 0, 0), (2, 1, 3, 63, 3), (1, 0, 2, 37, 1), (2, 1, 0, 5, 2)]:
        n, k = n_t * 32 + (lane & 31), k_t * 32 + 8 * g + 4 * (lane >> 5) + c
        want = float(w[n, k]) if (n < 40 and k < 70) else 0.0
        assert float(v[k_t, n_t, g, lane, c]) == want
    assert bp.shape[0] == 64 and float(bp[39]) == 39 and float(bp[40]) == 0


@pytest.mark.parametrize("tag", CASES)
def test_state_dict_keys_match_reference_layout(tag):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    name, hp, salt, batch, d = load_case(tag)
    if name not in FACTORY:
        pytest.skip(f"{name} not built yet")
    ours = FACTORY[name](hp)
    ref = O.build(name, hp)        # oracle key layout == reference key layout (test_oracle_golden pins it)
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in ref.state_dict().items()}


def test_no_cpu_fallback():
    from magnet_amd import ops
    with pytest.raises(RuntimeError):
        ops.segment_mean(torch.zeros(4, 32), torch.zeros(3, dtype=torch.int32), 2)


def test_split_bf16x3_is_exact():
    """hi + mid + lo reproduces every fp32 value bit for bit (the basis of the bf16x6 edge kernel)."""
    from magnet_amd.packing import split_bf16x3
    x = torch.randn(10000) * torch.logspace(-6, 6, 10000)
    hi, mid, lo = split_bf16x3(x)
    back = sum((p.to(torch.int32) << 16).view(torch.float32).double() for p in (hi, mid, lo))
    assert torch.equal(back.float(), x)


def test_split_fp16x2_reconstructs_to_fp32_class():
    """h + l/2048 reproduces fp32 weights to ~2^-22 of max(|w|, fp16's smallest normal) - i.e. fp32-class for
    everything a dot product can see - and refuses values outside the fp16 range instead of producing inf."""
    from magnet_amd.packing import split_fp16x2
    x = torch.randn(10000) * torch.logspace(-4, 3, 10000)
    h, lo = split_fp16x2(x)
    back = h.double() + lo.double() / 2048.0
    assert float(((back - x.double()).abs() / x.double().abs().clamp_min(6.1e-5)).max()) < 2.5e-7
    with pytest.raises(ValueError):
        split_fp16x2(torch.tensor([1.0, 7e4]))


def test_pack_edge16_fragment_order():
    """pack_edge16_fp16x2: lane l of (k step s, output tile t) holds W[16 t + (l & 15)][32 s + 8 (l >> 4) + 0..7] of each
    fp16 plane - the A-operand order of v_mfma_f32_16x16x32_f16 the 16-edge-tile edge kernel reads from LDS."""
    from magnet_amd.packing import pack_edge16_fp16x2, split_fp16x2
    torch.manual_seed(0)
    w, b = torch.randn(128, 128) * 0.1, torch.randn(128)
    flat, bias = pack_edge16_fp16x2(w, b)
    assert flat.dtype == torch.float16 and flat.numel() == 4 * 8 * 2 * 64 * 8 and torch.equal(bias, b)
    frag = flat.view(4, 8, 2, 64, 8)
    h, lo = split_fp16x2(w)
    for s_, t, lane in ((0, 0, 0), (1, 3, 17), (3, 7, 63), (2, 5, 40)):
        row, k0 = 16 * t + (lane & 15), 32 * s_ + 8 * (lane >> 4)
        assert torch.equal(frag[s_, t, 0, lane], h[row, k0:k0 + 8])
        assert torch.equal(frag[s_, t, 1, lane], lo[row, k0:k0 + 8])
    # the two planes reconstruct the weight to fp32 rounding
    assert float(((h.float() + lo.float() / 2048.0) - w).abs().max()) < 1e-7


def test_training_host_logic_on_cpu():
    """host-side pieces of the training path that need no GPU: flat-gradient detection, the width decomposition of the
    backward GEMMs, and the loud failure of the fused optimizer / differentiable ops without a HIP device."""
    import torch
    from magnet_amd import optim
    from magnet_amd.autograd import _tiles_of
    assert _tiles_of(384) == [(0, 128), (128, 128), (256, 128)] and _tiles_of(160) == [(0, 128), (128, 32)]
    assert _tiles_of(96) == [(0, 64), (64, 32)] and _tiles_of(32) == [(0, 32)]
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    fg = optim.FlatGrads(ps)
    for p, v in zip(ps, fg.views):
        p.grad = v
    flat = optim.flat_gradient(ps)
    assert flat is not None and flat.numel() == 17 and flat.data_ptr() == fg.flat.data_ptr()
    ps[1].grad = torch.zeros(5)                                   # a gradient from elsewhere: not one buffer any more
    assert optim.flat_gradient(ps) is None
    with pytest.raises(RuntimeError, match="HIP device"):
        optim.FusedAdam(ps)


# ---- topology cache (magnet_amd/topo_cache.py): identity with keep-alive, content re-key, in-place writes
def test_topology_cache_follows_the_coordinates_not_the_address():
    from magnet_amd.topo_cache import TopologyCache
    cache, built = TopologyCache(4), []

    def get(x, tag="g"):
        return cache.get(tag, (x,), lambda: built.append(len(built)) or ("topo", len(built)))

    a = torch.arange(12.0).reshape(6, 2)
    t1 = get(a)
    assert get(a) is t1 and get(a[:]) is t1 and len(built) == 1          # the same storage through another view object
    a2 = a.clone()
    assert get(a2) is t1 and len(built) == 1                            # equal content in a new tensor: re-keyed, not rebuilt
    assert get(a2) is t1
    b = a + 1.0                                                         # same shape, different coordinates
    t2 = get(b)
    assert t2 is not t1 and len(built) == 2
    b.add_(1.0)                                                         # in-place write bumps the version counter
    t3 = get(b)
    assert t3 is not t2 and len(built) == 3
    assert get(a2, tag="other") is not t1 and len(built) == 4           # the tag is part of the key
    # the entry keeps its key tensor alive, so a freed-and-recycled address cannot alias it
    ptr = b.data_ptr()
    del b
    c = torch.full((6, 2), 7.0)
    assert c.data_ptr() != ptr or get(c) is not t3


def test_topology_cache_is_bounded():
    from magnet_amd.topo_cache import TopologyCache
    cache = TopologyCache(3)
    keep = [torch.full((2, 2), float(i)) for i in range(5)]
    for i, x in enumerate(keep):
        cache.get("g", (x,), lambda i=i: i)
    assert len(cache) == 3 and [t for _, t in cache.topologies()] == [2, 3, 4]


# ---- bench.py multi-rank launcher (`--gpus N` must be real)
def _bench(*argv, **env_over):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus_n_starts_n_fresh_ranks():
    import json
    r = _bench("--gpus", "4", "--steps", "3", MAGNET_BENCH_LAUNCH_DRY="1")
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_bench_refuses_a_rank_count_mismatch():
    r = _bench("--gpus", "4", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "2 rank" in r.stderr


def test_launcher_runs_two_gloo_ranks_end_to_end(tmp_path):
    """the same torch.distributed.run invocation bench.py builds, on a CPU script: two ranks rendezvous on 127.0.0.1 over gloo"""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    script = tmp_path / "echo_rank.py"
    script.write_text("import os, torch, torch.distributed as dist\n"
                      "dist.init_process_group('gloo')\n"
                      "t = torch.tensor([float(dist.get_rank() + 1)])\n"
                      "dist.all_reduce(t)\n"
                      "print('SUM', int(t.item()), 'WORLD', dist.get_world_size()) if dist.get_rank() == 0 else None\n"
                      "dist.destroy_process_group()\n")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(bench.free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "SUM 3 WORLD 2" in r.stdout


def test_product_library_has_no_development_knobs():
    """kernel-variant / ablation selection exists only in the -DMAGNET_DEV_VARIANTS build (include/magnet_hip_dev.h)"""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(lib, "magnet_debug_set_variant") and not hasattr(lib, "magnet_debug_set_mlp_ablation")
    assert all("debug" not in n for n in _lib.declared_symbols())


def test_factory_lists_all_five_models():
    from magnet_amd.models.factory import FACTORY
    assert sorted(FACTORY) == ["magnet_cnn", "magnet_cnn_2d", "magnet_gnn", "mpnn", "mpnn_2d"]


def test_shim_package_shadows_models_factory():
    """with <repo>/shim first on PYTHONPATH the reference's own import line (run.py:8,
    `from models.factory import FACTORY`) resolves to magnet_amd's factory, and FACTORY[name](params) builds the model the way
    run.py:49-50 does - in a fresh interpreter, so that no `models` package imported earlier in this process can answer instead"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import types\n"
            "from models.factory import FACTORY\n"
            "import models.factory as mf, magnet_amd.models.factory as af\n"
            "assert FACTORY is af.FACTORY and 'shim' in mf.__file__\n"
            "hp = types.SimpleNamespace(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss='l1', hidden_features=128,\n"
            "                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)\n"
            "m = FACTORY['mpnn_2d'](hp)\n"
            "print(type(m).__name__, sorted(FACTORY))\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "shim"), root]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "MPNN_2d", out.stdout


def test_large_graph_node_kernel_gate():
    """host-side gate of the persistent node kernels (models/mpnn.py `fused_nodes`): shapes outside it take the generic launches"""
    from magnet_amd import ops
    assert ops.node_halves_supported(32, 4, 65536) and ops.node_halves_supported(64, 8, 2304)
    assert not ops.node_halves_supported(96, 4, 65536)          # kin: only 32 / 64 have resident-weight kernels
    assert not ops.node_halves_supported(32, 9, 65536)          # statistics of at most 8 graphs fit the kernels' LDS table
    assert not ops.node_halves_supported(32, 4, 2050)           # graph boundaries must fall on 32-row tiles
    assert not ops.node_halves_supported(32, 4, 65536, h=64)
    assert ops.NORM_SMALL_ROWS == 2048


def test_one_accumulator_pack_is_the_fp16x3_pack_with_high_plane_and_bias_times_2048():
    """packing mode "fp16x3s" (torch layout path): same slot map as "fp16x3", plane 0 and the bias scaled by 2^11 (exact), a
    SinglePlanes tensor so that ops.mlp_plan picks magnet_mlp_fp16x3s_f32; |w| >= 31.98 is refused"""
    from magnet_amd import packing
    g = torch.Generator().manual_seed(5)
    w, b = torch.randn(64, 40, generator=g), torch.randn(64, generator=g)
    p1, b1 = packing.pack(w, b, "fp16x3")
    p2, b2 = packing.pack(w, b, "fp16x3s")
    assert isinstance(p2, packing.SinglePlanes) and not isinstance(p1, packing.SinglePlanes) and p2.dtype == torch.float16
    a1, a2 = p1.view(-1, 2, 512).float(), torch.Tensor(p2).view(-1, 2, 512).float()
    assert torch.equal(a2[:, 0], a1[:, 0] * 2048.0) and torch.equal(a2[:, 1], a1[:, 1]) and torch.equal(b2, b1 * 2048.0)
    assert packing.chain_gemm(5, 64, "fp16x3") == "fp16x3s" and packing.chain_gemm(5, 128, "fp16x3") == "fp16x3"
    assert packing.chain_gemm(5, 64, "bf16x6") == "bf16x6" and packing.chain_gemm(3, 64, "fp16x3") == "fp16x3"
    w[1, 2] = 40.0
    with pytest.raises(ValueError):
        packing.pack(w, b, "fp16x3s")


def test_shipped_library_has_no_valu_write_right_in_front_of_an_mfma_that_reads_it():
    """hipcc does not insert the two wait states between an INLINE-ASM VALU write and a v_mfma that reads the register (it does for
    its own instructions); the round-2 "biases from LDS" failure was exactly that.  The sources close their asm
    blocks with the wait states; this disassembles the built library's gfx950 code objects and checks the result."""
    import os
    from magnet_amd import hazard_check
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs the built library")
    hazard_check.objdump()                                   # not optional: FileNotFoundError says where it looked
    kernels = hazard_check.disassemble(_lib.LIB_PATH)
    assert len(kernels) > 100 and sum(any(i.startswith("v_mfma") for i in ins) for _, ins in kernels) > 50
    assert hazard_check.violations(kernels) == [] and hazard_check.mfma_result_reads(kernels) == [] and hazard_check.trans_result_reads(kernels) == []
    assert hazard_check.sgpr_and_lane_hazards(kernels) == []
    seams = hazard_check.loop_seams(kernels)
    assert len(seams) > 100                                  # every loop of every kernel: its end in front of its start
    assert hazard_check.check_kernels(kernels) == []
    # the checker itself: a synthetic kernel with the pattern, with one and with two wait states in between
    mk = lambda mid: [("k", ["v_fma_mixhi_f16 v101, v53, s7, 0", *mid, "v_mfma_f32_32x32x16_f16 v[32:47], v[56:59], v[100:103], v[32:47]"])]  # noqa: E731
    assert len(hazard_check.violations(mk(["s_and_b64 s[4:5], s[0:1], s[2:3]"]))) == 1
    assert hazard_check.violations(mk(["s_nop 1"])) == [] and hazard_check.violations(mk(["s_nop 0", "v_mov_b32 v1, v2"])) == []
    assert hazard_check.violations([("k", ["v_fma_mixhi_f16 v13, v53, s7, 0", "v_mfma_f32_32x32x16_f16 a[0:15], v[56:59], v[100:103], a[0:15]"])]) == []
    early = [("k", ["v_mfma_f32_32x32x16_f16 v[32:47], v[56:59], v[100:103], v[32:47]", "s_nop 7", "v_add_f32 v1, v33, v2"])]
    assert len(hazard_check.mfma_result_reads(early)) == 1
    early[0][1][1] = "s_nop 9"
    assert hazard_check.mfma_result_reads(early) == []
    assert len(hazard_check.trans_result_reads([("k", ["v_rcp_f32_e32 v42, v84", "v_fma_mixlo_f16 v46, v80, v42, 0"])])) == 1
    assert hazard_check.trans_result_reads([("k", ["v_rcp_f32_e32 v42, v84", "s_nop 0", "v_fma_mixlo_f16 v46, v80, v42, 0"])]) == []


def test_hazard_rules_for_sgprs_lane_ops_m0_and_loop_seams():
    """the rules ADVICE r3 / VERDICT r3 asked for, each on a tiny violating snippet and on its repaired form"""
    from magnet_amd import hazard_check as H
    # VALU writes an SGPR, a VMEM instruction reads it: five wait states
    v = [("k", ["v_readfirstlane_b32 s4, v9", "s_nop 3", "global_load_dwordx4 v[0:3], v8, s[4:5]"])]
    assert len(H.sgpr_and_lane_hazards(v)) == 1
    v[0][1][1] = "s_nop 4"
    assert H.sgpr_and_lane_hazards(v) == []
    assert H.sgpr_and_lane_hazards([("k", ["s_mov_b32 s4, s9", "global_load_dwordx4 v[0:3], v8, s[4:5]"])]) == []       # SALU writers are interlocked
    # VALU writes a VGPR, a lane op reads it: two wait states for v_permlane*, one for v_readlane / v_readfirstlane
    assert len(H.sgpr_and_lane_hazards([("k", ["v_add_f32_e32 v5, v1, v2", "s_nop 0", "v_permlane32_swap_b32_e32 v5, v6"])])) == 1
    assert H.sgpr_and_lane_hazards([("k", ["v_add_f32_e32 v5, v1, v2", "s_nop 1", "v_permlane32_swap_b32_e32 v5, v6"])]) == []
    assert len(H.sgpr_and_lane_hazards([("k", ["v_add_f32_e32 v6, v1, v2", "v_permlane32_swap_b32_e32 v5, v6"])])) == 1       # the second operand too
    assert len(H.sgpr_and_lane_hazards([("k", ["v_mov_b32_e32 v7, v1", "v_readlane_b32 s3, v7, 5"])])) == 1
    assert H.sgpr_and_lane_hazards([("k", ["v_mov_b32_e32 v7, v1", "s_nop 0", "v_readlane_b32 s3, v7, 5"])]) == []
    # M0 written, LDS-DMA right behind it: one wait state
    assert len(H.sgpr_and_lane_hazards([("k", ["s_mov_b32 m0, s7", "buffer_load_dwordx4 v1, s[8:11], 0 offen lds"])])) == 1
    assert H.sgpr_and_lane_hazards([("k", ["s_mov_b32 m0, s7", "s_nop 0", "buffer_load_dwordx4 v1, s[8:11], 0 offen lds"])]) == []
    assert len(H.sgpr_and_lane_hazards([("k", ["s_mov_b32 m0, s7", "global_load_lds_dwordx4 v[2:3], off"])])) == 1
    # a loop whose LAST instruction writes an MFMA operand that the FIRST instruction of the next iteration reads: invisible to a
    # linear walk, found at the seam
    body = ["v_mfma_f32_32x32x16_f16 v[32:47], v[56:59], v[100:103], v[32:47]", "v_add_f32_e32 v1, v2, v3", "s_add_i32 s2, s2, 1",
            "v_fma_mixhi_f16 v101, v53, s7, 0", "s_cbranch_scc1 65531"]
    k = H.Kernel(body, addr=[0x100, 0x108, 0x10c, 0x110, 0x118], target=[None, None, None, None, 0x100])
    assert H.violations([("k", k)]) == []
    seams = H.loop_seams([("k", k)])
    assert len(seams) == 1 and seams[0][2] == 5
    hits = [h for h in H.violations([(n, i) for n, i, _ in seams]) if h[1] >= seams[0][2]]
    assert len(hits) == 1 and "v_fma_mixhi_f16 v101" in hits[0][3]
    k2 = H.Kernel(body[:4] + ["s_nop 0", body[4]], addr=[0x100, 0x108, 0x10c, 0x110, 0x118, 0x11c], target=[None] * 5 + [0x100])
    assert [h for n, i, f in H.loop_seams([("k", k2)]) for h in H.violations([(n, i)]) if h[1] >= f] == []     # s_nop 0 + the branch = 2


def test_hazard_rule_for_dpp_sources():
    """round 6: the hand-written v_fmac_f32_dpp scan of k_mpnn_edge_bwd_co - a VALU write of the register a DPP instruction reads from
    other lanes needs two wait states in front of it (path_hazards)"""
    from magnet_amd import hazard_check as H
    dpp = "v_fmac_f32_dpp v10, v10, v3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    mk = lambda mid: [("k", H.Kernel(["v_fmac_f32_e32 v10, v1, v2"] + mid + [dpp]))]      # noqa: E731
    assert len(H.path_hazards(mk([]))) == 1 and len(H.path_hazards(mk(["v_fmac_f32_e32 v11, v1, v2"]))) == 1
    assert H.path_hazards(mk(["v_fmac_f32_e32 v11, v1, v2", "v_fmac_f32_e32 v12, v1, v2"])) == [] and H.path_hazards(mk(["s_nop 1"])) == []
    # the multiplier (second source) is read in place: no distance needed
    assert H.path_hazards([("k", H.Kernel(["v_mov_b32_e32 v3, v1", dpp]))]) == []


def test_hazard_rules_follow_taken_forward_branches():
    """round 5: a FORWARD branch that skips code shortens the executed distance between a writer in front of it and a reader behind its
    target - `plane word written by asm; if (run ends) { ... 30 instructions ... }; MFMA reads the word` passes a linear walk and is
    one wait state short when the branch is taken.  path_hazards() walks the control-flow graph."""
    from magnet_amd import hazard_check as H
    body = ["v_fma_mixhi_f16 v101, v53, s7, 0",                                   # 0x100 writes an MFMA operand
            "s_cbranch_scc0 8",                                                   # 0x108 -> 0x120: skips the two fillers
            "v_add_f32_e32 v1, v2, v3",                                           # 0x10c
            "v_add_f32_e32 v4, v2, v3",                                           # 0x110
            "v_mfma_f32_32x32x16_f16 v[32:47], v[56:59], v[100:103], v[32:47]"]   # 0x120 reads it
    k = H.Kernel(body, addr=[0x100, 0x108, 0x10c, 0x110, 0x120], target=[None, 0x120, None, None, None])
    assert H.violations([("k", k)]) == []                                         # linear: branch + 2 fillers = 3 wait states
    hits = H.path_hazards([("k", k)])
    assert len(hits) == 1 and hits[0][2] == 1 and "v_fma_mixhi_f16 v101" in hits[0][3]      # taken: the branch alone
    assert H.check_kernels([("k", k)])                                            # ... and check() reports it
    k2 = H.Kernel([body[0], "s_nop 0"] + body[1:], addr=[0x100, 0x104, 0x108, 0x10c, 0x110, 0x120], target=[None, None, 0x120, None, None, None])
    assert H.path_hazards([("k", k2)]) == []                                      # s_nop 0 + the branch = 2
    # a load into the register in between ends the VALU-written state on that path
    k3 = H.Kernel([body[0], "ds_read_b32 v101, v9"] + body[1:], addr=[0x100, 0x104, 0x108, 0x10c, 0x110, 0x120],
                  target=[None, None, 0x120, None, None, None])
    assert H.path_hazards([("k", k3)]) == []
    # an MFMA result read by a VALU instruction behind a taken branch: passes + 2 wait states on THAT path
    mf = ["v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]", "s_cbranch_scc0 8"] + ["v_mov_b32_e32 v20, v21"] * 6 + ["v_add_f32_e32 v5, v0, v1"]
    k4 = H.Kernel(mf, addr=[0x100, 0x108] + [0x10c + 4 * i for i in range(6)] + [0x130], target=[None, 0x130] + [None] * 7)
    assert H.mfma_result_reads([("k", k4)]) == [] and len(H.path_hazards([("k", k4)])) == 1


def test_x0_free_tape_decision(monkeypatch):
    """autograd._x0_not_needed (host logic, no GPU): a chain keeps no assembled first-layer input on the tape exactly when every k tile is
    either a gathered run the backward factors (index, far fewer source rows than M) or ONE dense Term of M rows - and only from the
    size threshold on."""
    import types
    import torch
    from magnet_amd import autograd as AG
    M = 100_000
    x = torch.zeros(1000, 64)                      # node latents (gathered)
    e = torch.zeros(M, 32)                         # edge latents (dense)
    idx = torch.zeros(M, dtype=torch.int32)

    def meta(tiles, runs):
        return types.SimpleNamespace(tiles=tiles, runs=runs)

    # [x_i | x_j | 4 e]: two gathered 64-wide runs (k tiles 0-1, 2-3) + one dense 32-wide run (k tile 4)
    runs = [(0, 0, idx, 1.0, 1, 0, 64), (0, 0, idx, 1.0, 1, 2, 64), (1, 0, None, 4.0, 1, 4, 32)]
    tiles = [[(0, 0, idx, 1.0, 1)], [(0, 32, idx, 1.0, 1)], [(0, 0, idx, 1.0, 1)], [(0, 32, idx, 1.0, 1)], [(1, 0, None, 4.0, 1)]]
    m = meta(tiles, runs)
    monkeypatch.setenv("MAGNET_X0_FREE_MIN_MB", "0")
    assert AG._factored_first_layer(m, [x, e], M) == [True, True, True, True, False]
    assert AG._x0_not_needed(m, [x, e], M, 128) == {4: (1, 0, 4.0)}
    monkeypatch.setenv("MAGNET_X0_FREE_MIN_MB", "256")          # x0 would be 100 000 x 160 x 4 B = 64 MB: below the threshold
    assert AG._x0_not_needed(m, [x, e], M, 128) is None
    monkeypatch.setenv("MAGNET_X0_FREE_MIN_MB", "0")
    monkeypatch.setenv("MAGNET_X0_FREE", "0")
    assert AG._x0_not_needed(m, [x, e], M, 128) is None
    monkeypatch.setenv("MAGNET_X0_FREE", "1")
    # a k tile that is the SUM of two Terms (x_i - x_j), a dense source with other than M rows, a divided gather: x0 stays
    two = meta([[(0, 0, idx, 1.0, 1), (0, 0, idx, -1.0, 1)]] + tiles[1:], runs)
    assert AG._x0_not_needed(two, [x, e], M, 128) is None or 0 not in AG._x0_not_needed(two, [x, e], M, 128)
    short = torch.zeros(M // 2, 32)
    assert AG._x0_not_needed(m, [x, short], M, 128) is None
    rd = meta(tiles[:4] + [[(1, 0, None, 1.0, 2)]], runs[:2] + [(1, 0, None, 1.0, 2, 4, 32)])
    assert AG._x0_not_needed(rd, [x, e], M, 128) is None
