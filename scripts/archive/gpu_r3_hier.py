"""Round 3: hierarchies of the structured duct and the Delaunay channel side by side (rows, blocks per row, damping per level)."""
import sys
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cases = (("4u bcc 1/47", M.delaunay_channel_mesh(47, lattice="bcc"), 50.0), ("bcc 1/36", M.delaunay_channel_mesh(36, lattice="bcc"), 50.0),
         ("channel 240x60x60", M.channel_mesh((240, 60, 60)), 50.0), ("duct 300x75x75", M.duct_mesh((300, 75, 75), 4.0), 200.0))
for name, m, Re in cases:
    bcs = B.duct_bcs(m) if name.startswith("duct") else B.channel_bcs(m, *B.two_stream_profiles(0.5))
    P = FlowProblem(m, bcs, reynolds=Re)
    U, r = P.stokes_solve()
    w, n = P.newton_solve(U.clone())
    print(f"{name} {m.num_tets} tets: stokes {r.its}, ksp/step {n.ksp_its / n.its:.1f}")
    for l, L in enumerate(P.hierarchy()):
        print(f"   level {l}: rows {L['rows']:9d} blocks/row {L['blocks'] / max(1, L['rows']):6.2f} sweeps {L['sweeps']} omega {L['omega']:.3f}")
    P.close()
