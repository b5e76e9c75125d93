"""Exports of bayesiancoresets/coreset/__init__.py:1-7 that are on the SNNLS / beta path
(BatchPSVI, DiffPrivBatchPSVI and UniformSamplingCoreset are out of scope: SURVEY section 2 #12)."""
from .coreset import Coreset
from .hilbert import HilbertCoreset
from .bcores import BetaCoreset
from .sparsevi import SparseVICoreset
from .projector import (Projector, BlackBoxProjector, BetaBlackBoxProjector, DeviceProjector, DeviceBetaProjector)
