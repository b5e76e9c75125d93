"""beta_cores_amd -- the sparse-NNLS coreset hot path of dionman/beta-cores, MI355X-native.

Drop-in for the `bayesiancoresets` names on this path (bayesiancoresets/__init__.py:1):
    import beta_cores_amd as bc
    bc.HilbertCoreset, bc.BetaCoreset, bc.SparseVICoreset, bc.BatchPSVICoreset, bc.UniformSamplingCoreset,
    bc.BlackBoxProjector, bc.BetaBlackBoxProjector,
    bc.Projector, bc.snnls.{GIGA, FrankWolfe, OrthoPursuit, ImportanceSampling, UniformSampling},
    bc.util.{nn_opt, set_verbosity, TOL, set_tolerance}
plus the device-resident projector bc.DeviceProjector / bc.DeviceBetaProjector (K1 on the GPU).
Importing the package does not touch the GPU; constructing a solver/projector does and
raises if libbeta_cores.so or a gfx950 device is missing (there is no CPU fallback).
"""
from . import util
from . import snnls
from .util.errors import NumericalPrecisionError
from .device import Context, DeviceData, DevicePhi, default_context, set_default_context
from .coreset import (Coreset, HilbertCoreset, BetaCoreset, SparseVICoreset, BatchPSVICoreset, UniformSamplingCoreset,
                      Projector, BlackBoxProjector,
                      BetaBlackBoxProjector, DeviceProjector, DeviceBetaProjector)
from . import likelihoods
from . import samplers
from .posterior import weighted_gram, weighted_post, weighted_post_corrected, gaussian_weighted_post
from .dist import ShardComm, shard_bounds

__all__ = ['util', 'snnls', 'likelihoods', 'samplers', 'NumericalPrecisionError', 'Context', 'DeviceData', 'DevicePhi',
           'default_context', 'set_default_context', 'Coreset', 'HilbertCoreset', 'BetaCoreset', 'SparseVICoreset',
           'BatchPSVICoreset', 'UniformSamplingCoreset',
           'Projector', 'BlackBoxProjector', 'BetaBlackBoxProjector', 'DeviceProjector', 'DeviceBetaProjector',
           'ShardComm', 'shard_bounds', 'weighted_gram', 'weighted_post', 'weighted_post_corrected',
           'gaussian_weighted_post']
