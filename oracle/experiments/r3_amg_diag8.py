"""K-cycle (Notay: two GCR steps on the coarse system, preconditioned by the next level's cycle) needs a FLEXIBLE outer Krylov method --
BiCGStab with a variable preconditioner (what oracle/proto_amg.py measured) is not meaningful.  Here: FGMRES (no restart) around V-, W- and
K-cycles; counted are preconditioner applications (= fine-level cycles) and visits per level."""
import sys, time, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.proto_amg import problem, setup, cycle

def fgmres(A, b, M, rtol=1e-8, maxit=200):
    n = len(b); bn = np.linalg.norm(b)
    V = [b / bn]; Z = []; H = np.zeros((maxit + 1, maxit)); g = np.zeros(maxit + 1); g[0] = bn
    cs, sn = np.zeros(maxit), np.zeros(maxit)
    for j in range(maxit):
        z = M(V[j]); Z.append(z); w = A @ z
        for _ in range(2):
            for i in range(j + 1):
                h = V[i] @ w; H[i, j] += h; w = w - h * V[i]
        H[j + 1, j] = np.linalg.norm(w); V.append(w / H[j + 1, j])
        for i in range(j):
            t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]; H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]; H[i, j] = t
        d = np.hypot(H[j, j], H[j + 1, j]); cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
        H[j, j] = d; H[j + 1, j] = 0; g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j]
        if abs(g[j + 1]) <= rtol * bn:
            y = np.linalg.solve(np.triu(H[:j + 1, :j + 1]), g[:j + 1])
            return sum(yi * zi for yi, zi in zip(y, Z)), j + 1
    return None, maxit

cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4]); length = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
A, b, free = problem(cells, Re, length)
lv = setup(A, free)
print("dofs", A.shape[0], "levels", [L.n for L in lv], flush=True)
def run(label, **kw):
    cnt = {}
    t0 = time.time()
    x, its = fgmres(A, b, lambda v: cycle(lv, 0, v, counter=cnt, **kw))
    rel = np.linalg.norm(b - A @ x) / np.linalg.norm(b) if x is not None else float("nan")
    work = sum(cnt.get(l, 0) * lv[l].A.nnz for l in range(1, len(lv))) / lv[0].A.nnz
    print(f"{label:34s} FGMRES its (= fine cycles) {its:4d} rel {rel:.1e} visits/level {[cnt.get(l, 0) for l in range(len(lv))]} coarse work/fine nnz {work:6.1f}  {time.time() - t0:.0f}s", flush=True)
# BiCGStab + V for reference (2 cycles per iteration)
its = [0]; cnt = {}
M_ = spla.LinearOperator(A.shape, matvec=lambda v: cycle(lv, 0, v, counter=cnt))
x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=M_, maxiter=400, callback=lambda xk: its.__setitem__(0, its[0] + 1))
print(f"BiCGStab + V(1,4,6,2): its {its[0]} = {cnt[0]} fine cycles", flush=True)
run("V (1,4,6,2)")
run("V (1,2,2,2)", sched=(1, 2, 2, 2))
run("W at 1,2 (1,4,6,2)", kind="W")
run("K at 1 (1,4,6,2)", kind="K", klevels=(1,))
run("K at 1,2 (1,4,6,2)", kind="K", klevels=(1, 2))
run("K at 1,2 (1,2,2,2)", kind="K", klevels=(1, 2), sched=(1, 2, 2, 2))
run("K at 1,2,3 (1,2,2,2)", kind="K", klevels=(1, 2, 3), sched=(1, 2, 2, 2))
run("K at 1..5 (1,1,1,1)", kind="K", klevels=(1, 2, 3, 4, 5), sched=(1, 1, 1, 1))
run("K at 1..5 (1,2,2,2)", kind="K", klevels=(1, 2, 3, 4, 5), sched=(1, 2, 2, 2))
