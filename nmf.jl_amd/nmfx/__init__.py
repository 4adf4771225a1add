"""nmfx -- host-side mirror of the NMF.jl interface over libnmfx.so (MI355X, hand-written HIP)."""
from .api import (ALSPGrad, ArgumentError, Context, CoordinateDescent, GreedyCD, DimensionMismatch, LocalGroup, MultUpdate, NMFXError, PosDefException,
                  ProjectedALS, Result, SPA, alspgrad_updateh, alspgrad_updatew, comm_unique_id, make_opts, nmf_checksize,
                  nndsvd, nnmf, randinit, rsvd, solve, spa, truncated_svd)
from . import _lib, dist

__all__ = ["ALSPGrad", "ArgumentError", "Context", "CoordinateDescent", "GreedyCD", "DimensionMismatch", "LocalGroup", "MultUpdate", "NMFXError", "PosDefException",
           "ProjectedALS", "Result", "SPA", "alspgrad_updateh", "alspgrad_updatew", "comm_unique_id", "make_opts",
           "nmf_checksize", "nndsvd", "nnmf", "randinit", "rsvd", "solve", "spa", "truncated_svd", "_lib", "dist"]
