"""``models.factory`` of the reference (models/factory.py:10-19), served by magnet_amd: same keys for the five hot-path models, same
single ``hparams`` argument (run.py:49-50)."""
from magnet_amd.models.factory import FACTORY  # noqa: F401
