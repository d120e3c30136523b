"""CPU oracle for the stabilised P1-P1 Stokes / Navier-Stokes hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the
product package ``stabilized_navier_stokes_flow_fenicsx_amd``; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and there only as the checker / the timed CPU baseline.

PARITY PINNED BY THE REFERENCE'S ONLY CONSTANTS (round 2): the reference
(mungerct/Stabilized_Navier_Stokes_Flow_FEniCSx) ships no fixtures, golden
vectors or tests, and its arithmetic lives in un-vendored third-party
libraries that are absent here (fenics-dolfinx/basix/ffcx 0.9.0, fenics-ufl
2024.2.0, petsc 3.23.4, environment.yml:37-44,171-177), so it cannot be
executed.  What it does hold are the DFG 2D-1 benchmark values C_d =
5.57953523384, C_l = 0.010618948146 (Validation_Flow/DFG_2D_Validation.py:202-203):
``forms2d.py`` (the 2-D UGN form of that script) is checked against them
directly, and the 3-D G-metric forms restated below converge to the same
C_d on a one-cell slab of tets (tests/test_gpu_2d.py) --
a pin of the converged functional, not an element-by-element fixture.
Element by element the oracle is a
restatement of the weak-form text itself:

  * ``forms_literal.py``  term-by-term evaluation of the UFL expressions of
    NavierStokes/NavierStokesChannelFlow.py:160-172 (Stokes) and :220-251 (NS)
    with explicit test functions; the Jacobian is the automatic (torch fp64
    autograd) Gateaux derivative, as ``ufl.derivative`` at :46 prescribes.
  * ``element.py``        closed-form, vectorised numpy element kernels
    (SURVEY.md Appendix A), checked against ``forms_literal`` and finite
    differences in tests/test_oracle_*.py.
  * ``assemble.py``       global CSR assembly + Dirichlet semantics of
    :51-75 (rows AND columns zeroed, unit diagonal, lifting, F_B = x_B - g).
  * ``solve.py``          LU / Krylov linear solves, Newton + backtracking
    (settings of :198-202, :274-283).
  * ``c/sns_oracle.c``    the same element kernels + BSR BiCGStab in plain C
    with OpenMP: the ``cpu_baseline`` ("port") timed by bench.py.

Further known-answer pins (not from the reference):
finite differences of F vs J, patch tests, and the analytic fully developed
square-duct profile.
"""
