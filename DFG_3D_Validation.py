#!/usr/bin/env python
"""Same command line as the reference's NavierStokes/Validation_Flow/DFG_3D_Validation.py (reads dfg_pillar_3D.msh from
the working directory; ``builtin[:n]`` meshes the same geometry without gmsh); runs on the MI355X hot path
(see stabilized_navier_stokes_flow_fenicsx_amd/drivers.py:dfg_3d_main)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stabilized_navier_stokes_flow_fenicsx_amd.drivers import dfg_3d_main  # noqa: E402

if __name__ == "__main__":
    dfg_3d_main(sys.argv)
