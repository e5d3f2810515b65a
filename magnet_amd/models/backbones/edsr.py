"""EDSR feature encoder (reference models/backbones/edsr.py:32-60; effective res_scale is always 1
because the reference passes it positionally into ``bias``).  A dense convolution stack:
it runs through torch's MIOpen convolutions (SURVEY.md 8(f) rank 3 keeps a custom implicit-GEMM kernel
as a later row); everything downstream of it is hand-written HIP."""
from torch import nn


class ResBlock(nn.Module):
    def __init__(self, n_chan, kernel_size, mode="1d"):
        super().__init__()
        conv = nn.Conv2d if mode == "2d" else nn.Conv1d
        self.conv_1 = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.act = nn.ReLU(True)
        self.conv_2 = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)

    def forward(self, x):
        return self.conv_2(self.act(self.conv_1(x))) + x


class EDSR(nn.Module):
    def __init__(self, in_chan, n_chan=64, res_layers=16, kernel_size=3, res_scale=1, mode="1d"):
        super().__init__()
        conv = nn.Conv2d if mode == "2d" else nn.Conv1d
        self.head_conv = conv(in_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.res_layers = nn.Sequential(*[ResBlock(n_chan, kernel_size, mode) for _ in range(res_layers)])
        self.tail_conv = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.out_dim = n_chan

    def forward(self, x):
        x = self.head_conv(x)
        return self.tail_conv(self.res_layers(x)) + x
