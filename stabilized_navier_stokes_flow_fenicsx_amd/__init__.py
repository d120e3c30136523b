"""MI355X-native stabilised Stokes / Navier-Stokes hot path (P1-P1 tets):
HIP assembly + BSR4 SpMV + Krylov/AMG + Newton behind the C-ABI of include/sns.h.

Host-side (numpy) pieces: ``mesh`` (box mesher, .msh reader), ``bcs``.
Device path: ``solver.FlowProblem`` (requires the built libsns.so and a GPU).
"""
from . import bcs, mesh  # noqa: F401

__all__ = ["mesh", "bcs"]
