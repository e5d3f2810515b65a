"""magnet_amd - MI355X (gfx950) native implementation of the MAgNet autoregressive PDE-rollout
hot path (MPNN encoder/processor + INR decoders) behind the reference's own model API.

    from magnet_amd.models.factory import FACTORY
    model = FACTORY["mpnn_2d"](hparams).cuda()

All compute runs in ``magnet_amd/csrc/libmagnet_hip.so`` (C ABI: include/magnet_hip.h); there is
no CPU fallback.  Build with ``python -m magnet_amd.build``."""
__version__ = "0.1.0"

import os as _os

if _os.environ.get("MAGNET_POISON", "") not in ("", "0"):      # debug switch (magnet_amd/debug_poison.py): NaN-filled torch.empty
    from . import debug_poison as _debug_poison
    _debug_poison.install_from_env()
