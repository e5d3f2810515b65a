"""MAgNetCNN_2d - drop-in for the reference's ``models/magnet_cnn_2d.py``: 2-D EDSR, four shifted
nearest-cell lookups and the rectangle-area blend (magnet_cnn_2d.py:225-292); parametrised in
``MAgNetCNN`` by ``DIM``."""
from .magnet_cnn import MAgNetCNN


class MAgNetCNN_2d(MAgNetCNN):
    DIM = 2
