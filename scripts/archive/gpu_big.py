"""Largest-problem check on ONE GPU: duct with ~100 M tets (no element-matrix scratch any more)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = tuple(int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "640,160,160").split(","))
stop = False
def beat():
    t0 = time.time()
    while not stop:
        time.sleep(30); print(f"  ... {time.time() - t0:.0f}s", flush=True)
threading.Thread(target=beat, daemon=True).start()
t0 = time.time()
part = PT.duct_slab_part(cells, 4.0, 0, 1)                 # slab builder: no boundary-facet sort over the whole mesh
print(f"mesh {cells}: {part.mesh.num_tets/1e6:.1f} M tets, {part.mesh.num_nodes/1e6:.2f} M nodes, built in {time.time()-t0:.0f}s", flush=True)
t0 = time.time()
P = FlowProblem(part.mesh, (part.bc_mask, part.bc_val), reynolds=200.0, snes_max_it=1)
print(f"problem created in {time.time()-t0:.0f}s", flush=True)
free, total = torch.cuda.mem_get_info(); print(f"device memory used after create: {(total-free)/2**30:.1f} GiB", flush=True)
t0 = time.time(); U, r = P.stokes_solve(); torch.cuda.synchronize()
print(f"stokes: its {r.its} reason {r.reason} {time.time()-t0:.1f}s", flush=True)
t0 = time.time(); w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); t1 = time.time()
w, n2 = P.newton_solve(w); torch.cuda.synchronize(); t2 = time.time()
free, total = torch.cuda.mem_get_info()
print(f"newton steps: ksp its {n.ksp_its},{n2.ksp_its} fnorm {n.fnorms[-1]:.2e},{n2.fnorms[-1]:.2e} {t1-t0:.2f}s,{t2-t1:.2f}s -> {part.mesh.num_dofs/(t2-t1)/1e6:.1f} M-DOF/s; "
      f"device memory used {(total-free)/2**30:.1f} GiB, levels {P.timings().amg_levels}", flush=True)
stop = True
P.close()
