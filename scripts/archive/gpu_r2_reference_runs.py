"""round 2: the reference's production parameter range (run_all_RE.sh: Re 40..70, lc 0.04; run_all_images.sh: Re 10) through
the three-stage driver on the synthetic two-stream channel: does every stage converge?"""
import os, sys, tempfile
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import drivers as D
os.chdir(tempfile.mkdtemp())
for Re, lc in ((10, 0.04), (40, 0.04), (70, 0.04), (70, 0.1), (100, 0.05), (200, 0.05)):
    try:
        r = D.solve_NS_flow(["NavierStokesChannelFlow.py", str(Re), "./InletImages/Synthetic.png", "0.5", str(lc)])
        n = r["newton"]
        print(f"== Re {Re} lc {lc}: fine stage newton its {n.its} reason {n.reason} ksp {n.ksp_its} |F| {n.fnorms[-1]:.2e}", flush=True)
    except Exception as e:
        print(f"== Re {Re} lc {lc}: EXC {e}", flush=True)
