"""name -> model class, mirroring the reference's models/factory.py:10-19 for the hot-path models.
The FNO baselines and the no-interaction ablation are out of scope."""
from .magnet_cnn import MAgNetCNN
from .magnet_cnn_2d import MAgNetCNN_2d
from .magnet_gnn import MAgNetGNN
from .mpnn import MPNN
from .mpnn_2d import MPNN_2d

FACTORY = {
    "mpnn": MPNN,
    "mpnn_2d": MPNN_2d,
    "magnet_cnn": MAgNetCNN,
    "magnet_cnn_2d": MAgNetCNN_2d,
    "magnet_gnn": MAgNetGNN,
}
