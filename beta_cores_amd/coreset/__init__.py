"""Exports of bayesiancoresets/coreset/__init__.py:1-7 (DiffPrivBatchPSVICoreset does not exist in the reference tree)."""
from .coreset import Coreset
from .hilbert import HilbertCoreset
from .bcores import BetaCoreset
from .sparsevi import SparseVICoreset
from .bpsvi import BatchPSVICoreset
from .sampling import UniformSamplingCoreset
from .projector import (Projector, BlackBoxProjector, BetaBlackBoxProjector, DeviceProjector, DeviceBetaProjector)
