"""mhx -- Python host mirror of the AdvancedMH.jl API over libmhx.so (MI355X / gfx950 HIP kernels)."""
from ._lib import (ArgumentError, Context, MhxError, PosDefException, FLAG_GENERIC, FLAG_NO_JIT, FLAG_EMCEE_SEQUENTIAL, FLAG_ZIGGURAT, FLAG_DENSE_FACTOR, FLAG_RAM_DEFERRED, LIB_PATH,
                   EXPORTS, MHX_EINVAL, MHX_ESTATE, Schedule, check, host_array, lib, get_default_dtype, set_default_dtype, use_library, TOOLS_LIB_PATH)
from .dist import Group
from .api import (I, Banana, Chains, CorrGaussian, DensityModel, Ensemble, Funnel, HipLogDensity, IIDNormal,
                  InverseGamma, IsoGaussian, LogDensityModel, MALA, MCMCDistributed, MCMCHIP, MCMCSerial, MCMCThreads, MetropolisHastings, MvNormal, Normal, RandomWalkProposal,
                  RobustAdaptiveMetropolis, Run, RWMH, StaticMH, StaticProposal, StructArray, combine_diagnostics, StretchProposal,
                  SymmetricRandomWalkProposal, Transition, bundle_samples,
                  logdensity, pack_lower, sample, unpack_lower, zeros)
from . import trace


def set_option(name, value):
    """An explicit engine option (mhx_ctx_set_option; include/mhx.h lists them) on the default contexts; None unsets."""
    Context.set_default_option(name, value)


def clear_options():
    Context.clear_default_options()


__all__ = [n for n in dir() if not n.startswith("_")]
