#!/usr/bin/env python
"""Same command line as the reference's LidDrivenStokesFlow.py (no arguments; an optional cell count); runs on the MI355X
hot path (see stabilized_navier_stokes_flow_fenicsx_amd/drivers.py for what is kept and what differs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stabilized_navier_stokes_flow_fenicsx_amd.drivers import lid_driven_stokes_main  # noqa: E402

if __name__ == "__main__":
    lid_driven_stokes_main(sys.argv)
