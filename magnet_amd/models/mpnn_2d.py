"""MPNN_2d - drop-in for the reference's ``models/mpnn_2d.py``.  Differs from the 1D model only in
the widths (+2 position features, models/mpnn_2d.py:51,131), the diagonal-spacing radius (:240-243),
the read-out Swish at tw=10 (:138-143) and the time index fed as the 'variables' feature (:265,287);
all of that is parametrised in ``MPNN`` by ``DIM``."""
from .mpnn import MPNN


class MPNN_2d(MPNN):
    DIM = 2
