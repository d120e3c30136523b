#!/usr/bin/env python
"""Same command line as the reference's NavierStokes/Validation_Flow/DFG_2D_Validation.py (<msh file>, or
``builtin[:level]`` for the gmsh-free mesh of dfg_pillar_2D.geo's geometry); runs on the MI355X hot path
(see stabilized_navier_stokes_flow_fenicsx_amd/drivers.py:dfg_2d_main)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stabilized_navier_stokes_flow_fenicsx_amd.drivers import dfg_2d_main  # noqa: E402

if __name__ == "__main__":
    dfg_2d_main(sys.argv)
