#!/usr/bin/env python
"""Same command line as the reference's NavierStokes/streamtrace.py (<img_fname> <solname> <funcname>): GPU particle
tracing + the reference's post-processing (see stabilized_navier_stokes_flow_fenicsx_amd/drivers.py:streamtrace_main)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stabilized_navier_stokes_flow_fenicsx_amd.drivers import streamtrace_main  # noqa: E402

if __name__ == "__main__":
    streamtrace_main(sys.argv)
