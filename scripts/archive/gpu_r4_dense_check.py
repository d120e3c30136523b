"""Round 4: the blocked Gauss-Jordan dense inverse (csrc/sns_dense.hip) against numpy on the GPU box + its timing."""
import ctypes as C
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
for N in (1, 5, 63, 64, 65, 128, 200, 777, 1900, 2048):
    S = rng.normal(size=(N, N))
    A = (S - S.T) * 3.0 + np.diag(1.0 + rng.random(N)) * np.sqrt(N) + 0.1 * (S + S.T)      # PD symmetric part, strong skew part
    Ad = torch.from_numpy(A).cuda()
    Xd = torch.empty_like(Ad)
    rc = lib.sns_dense_inverse(0, N, Ad.data_ptr(), Xd.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rc = lib.sns_dense_inverse(0, N, Ad.data_ptr(), Xd.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    X = Xd.cpu().numpy()
    err = np.abs(X @ A - np.eye(N)).max()
    ref = np.linalg.inv(A)
    print(f"N {N:5d} rc {rc} |XA - I|max {err:.2e} rel diff to numpy inv {np.abs(X - ref).max() / np.abs(ref).max():.2e} "
          f"cond {np.linalg.cond(A):.1e} wall incl. alloc/copies {dt * 1e3:.2f} ms", flush=True)
