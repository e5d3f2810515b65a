"""Dev: phase timeline of k_mpnn_edge_bwd_co from a -DCO_TRACE build (MAGNET_LIB_PATH=...libmagnet_hip_coT.so): s_memtime stamps of
workgroup 0's eight waves at nine points of 32 consecutive steps -> mean duration of every phase per wave, in s_memtime ticks."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

sys.argv = [sys.argv[0], "2", "1"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_edge_bwd.py")).read())
buf = np.zeros((8, 32, 10), dtype=np.int64)
rc = lib.magnet_debug_co_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
names = ["top->X (bookkeeping, load issue)", "barrier X", "GEMM 1", "dq + writes + load_g", "barrier Y", "matrix part (GEMM 2 + dW2)",
         "dp + scan + stores", "produce(next)"]
d = np.diff(buf[:, :, :9], axis=2).astype(float)            # [wave][step][phase]
step_len = (buf[:, 1:, 0] - buf[:, :-1, 0]).astype(float)
print(f"ticks per step (wave 0): mean {step_len[0].mean():.0f}  min {step_len[0].min():.0f}  max {step_len[0].max():.0f}")
for k, n in enumerate(names):
    print(f"{n:38s} " + "  ".join(f"{d[w, :, k].mean():6.0f}" for w in range(8)) + f"   | mean {d[:, :, k].mean():6.0f} = {100 * d[:, :, k].mean() / step_len.mean():4.1f} %")
gap = (buf[:, 1:, 0] - buf[:, :-1, 8]).astype(float)
print(f"{'end -> next top':38s} " + "  ".join(f"{gap[w].mean():6.0f}" for w in range(8)))
