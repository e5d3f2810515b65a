"""Dev microbench: the fused MPNN edge kernel k_mpnn_edge16 (16-edge tiles, fp16x3) on the 256^2 bench graph, against the
fp32-MFMA kernel's result.  usage: python scripts/bench_edge16.py [W] [B]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack, pack_edge16_fp16x2

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, B).csr
N, E, H = csr.N, csr.E, 128
A, Bm, agg = (torch.randn(N, H, device="cuda") for _ in range(3))
partials = torch.empty(ops.edge_scratch_floats(E), device="cuda")
m2 = m.gnn_layers[0].message_net_2[0]
w32, w16 = pack(m2.weight, m2.bias, "fp32"), pack_edge16_fp16x2(m2.weight, m2.bias)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


ops.mpnn_edge(A, Bm, csr, w32[0], w32[1], agg, partials)
ref = agg.clone()
A2, B2 = A * ops.LOG2E, Bm * ops.LOG2E
w16s = pack_edge16_fp16x2(m2.weight, m2.bias, scaled=True)
for l2e, single in ((False, False), (True, False), (False, True), (True, True)):
    agg.fill_(float("nan"))
    Ax, Bx = (A2, B2) if l2e else (A, Bm)
    wq = w16s if single else w16
    ms = min(timeit(lambda: ops.mpnn_edge16(Ax, Bx, csr, wq[0], wq[1], agg, partials, l2e=l2e, single=single)) for _ in range(3))
    print(f"k_mpnn_edge16{' single-accumulator' if single else ''} l2e={l2e} : {ms:.4f} ms  {2 * E * H * H / ms / 1e9:.1f} TFLOP/s-eq "
          f"(incl. fix-up), frac of 833 = {2 * E * H * H / ms / 1e9 / 833.3:.3f}, rel-diff vs fp32 kernel {float((agg - ref).norm() / ref.norm()):.2e} "
          f"nan={int(torch.isnan(agg).sum())}")

from magnet_amd.packing import pack as _pack
w7 = _pack(m2.weight, m2.bias, "fp16x3")
for l2e in (False, True):
    agg.fill_(float("nan"))
    Ax, Bx = (A2, B2) if l2e else (A, Bm)
    ms = min(timeit(lambda: ops.mpnn_edge(Ax, Bx, csr, w7[0], w7[1], agg, partials, l2e=l2e)) for _ in range(3))
    print(f"k_mpnn_edge7 (32-edge tiles, 1 wave/SIMD) l2e={l2e} : {ms:.4f} ms  frac of 833 = "
          f"{2 * E * H * H / ms / 1e9 / 833.3:.3f}, rel-diff vs fp32 kernel {float((agg - ref).norm() / ref.norm()):.2e} nan={int(torch.isnan(agg).sum())}")

from magnet_amd.packing import pack_edge32t_fp16x2
w32t = pack_edge32t_fp16x2(m2.weight, m2.bias)
for l2e, paired in ((False, False), (True, False), (False, True), (True, True)):
    agg.fill_(float("nan"))
    Ax, Bx = (A2, B2) if l2e else (A, Bm)
    ms = min(timeit(lambda: ops.mpnn_edge32t(Ax, Bx, csr, w32t[0], w32t[1], agg, partials, l2e=l2e, paired=paired)) for _ in range(3))
    print(f"k_mpnn_edge32{'p (pairs, 2 waves/SIMD)' if paired else 't (generation 9, 1 wave/SIMD)'} l2e={l2e} : {ms:.4f} ms  frac of 833 = "
          f"{2 * E * H * H / ms / 1e9 / 833.3:.3f}, rel-diff vs fp32 kernel {float((agg - ref).norm() / ref.norm()):.2e} nan={int(torch.isnan(agg).sum())}")

for l2e in (False, True):
    agg.fill_(float("nan"))
    Ax, Bx = (A2, B2) if l2e else (A, Bm)
    ms = min(timeit(lambda: ops.mpnn_edge_ws(Ax, Bx, csr, w32t[0], w32t[1], agg, partials, l2e=l2e)) for _ in range(3))
    print(f"k_mpnn_edge_ws (generation 10: wave-specialised, 4 waves/SIMD) l2e={l2e} : {ms:.4f} ms  frac of 833 = "
          f"{2 * E * H * H / ms / 1e9 / 833.3:.3f}, rel-diff vs fp32 kernel {float((agg - ref).norm() / ref.norm()):.2e} nan={int(torch.isnan(agg).sum())}")

# ---- clock / power under a sustained loop of each kernel (is the part power-limited? DESIGN.md 10.2)
if os.environ.get("EDGE_CLOCKS", "0") == "1":
    import ctypes as C, glob, re, subprocess, threading, time

    def sysfs_mhz():
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            try:
                for line in open(f):
                    if "*" in line:
                        return int(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1))
            except Exception:
                pass
        return None

    def power_w():
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) + \
                sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")):
            try:
                return int(open(f).read()) / 1e6
            except Exception:
                pass
        return None

    side = torch.cuda.Stream()
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    khz = C.c_int(0)
    kernels = {"edge16": lambda: ops.mpnn_edge16(A2, B2, csr, w16[0], w16[1], agg, partials, l2e=True),
               "edge7": lambda: ops.mpnn_edge(A2, B2, csr, w7[0], w7[1], agg, partials, l2e=True),
               "edge32t": lambda: ops.mpnn_edge32t(A2, B2, csr, w32t[0], w32t[1], agg, partials, l2e=True),
               "edge32p": lambda: ops.mpnn_edge32t(A2, B2, csr, w32t[0], w32t[1], agg, partials, l2e=True, paired=True)}
    print("pp_dpm_sclk:", [open(f).read().replace("\n", " | ") for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")][:1])
    for name, fn in kernels.items():
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        samples, pw, stop = [], [], threading.Event()

        def loop():
            while not stop.is_set():
                v, p_ = sysfs_mhz(), power_w()
                if v:
                    samples.append(v)
                if p_:
                    pw.append(p_)
                stop.wait(0.01)

        th = threading.Thread(target=loop)
        th.start()
        t0 = time.perf_counter()
        side.wait_stream(torch.cuda.current_stream())
        n = 400
        for i in range(n):
            if i == 100:
                ops.L.lib().magnet_clock_probe(C.c_void_p(out.data_ptr()), 200000, C.byref(khz), C.c_void_p(side.cuda_stream))
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        c, w = (int(v) for v in out.tolist())
        probe = c / w * khz.value / 1e3 if w else None
        print(f"{name}: {dt / n * 1e3:.4f} ms per launch sustained, clock probe {probe and round(probe)} MHz, sysfs sclk median "
              f"{samples and sorted(samples)[len(samples) // 2]} (min {samples and min(samples)}, max {samples and max(samples)}), "
              f"power median {pw and round(sorted(pw)[len(pw) // 2])} W")
