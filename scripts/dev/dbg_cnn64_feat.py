"""Dev: magnitude of the EDSR features at the cnn_64 reference shape, HIP (rows / rows_grad) against torch conv2d on the same weights."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import torch.nn.functional as F
from magnet_amd.models.factory import FACTORY
name, hp, mk = bench.ref_shape_cases()["cnn_64"]
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
batch, _ = mk(True)
batch = {k: v.cuda() for k, v in batch.items()}
u = batch["lr_frames"].float()
T = model.time_slice
x = u[:, :T].reshape(u.shape[0], T * u.shape[2], *u.shape[3:])
print("input", tuple(x.shape), "max|x|", float(x.abs().max()))
enc = model.encoder


def torch_edsr(x):
    h = F.conv2d(x, enc.head_conv.weight, enc.head_conv.bias, padding=1)
    r = h
    for blk in enc.res_layers:
        r = r + F.conv2d(F.relu(F.conv2d(r, blk.conv_1.weight, blk.conv_1.bias, padding=1)), blk.conv_2.weight, blk.conv_2.bias, padding=1)
    return F.conv2d(r, enc.tail_conv.weight, enc.tail_conv.bias, padding=1) + h


with torch.no_grad():
    ref = torch_edsr(x)
    rows = enc.rows(x).clone()
    B, C = ref.shape[:2]
    ref_rows = ref.reshape(B, C, -1).permute(0, 2, 1).reshape(-1, C)
    print("torch conv2d features: max", float(ref_rows.abs().max()), "rms", float(ref_rows.pow(2).mean().sqrt()))
    print("HIP rows          : max", float(rows.abs().max()), "rel L2 vs torch", float((rows - ref_rows).norm() / ref_rows.norm()))
rg = enc.rows_grad(x)
print("HIP rows_grad     : max", float(rg.abs().max()), "rel L2 vs torch", float((rg.detach() - ref_rows).norm() / ref_rows.norm()))
