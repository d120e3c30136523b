"""Host-side mirror of the reference's solve seam over the HIP C-ABI.

Names, argument meaning and error behaviour follow
NavierStokes/NavierStokesChannelFlow.py:
  * ``NonlinearPDE_SNESProblem`` (:40-75)  -> F / J callbacks on device vectors
  * ``solve_stokes_problem``     (:197-218)
  * ``solve_navier_stokes``      (:268-312): returns ``(w, u, p)``, prints
    iterations / reason / seconds (:297-299); non-convergence is NOT an error.
All arithmetic happens in libsns.so (HIP); torch only owns device buffers and
the stream.  There is no CPU fallback: constructing a ``FlowProblem`` without
a GPU and the built library raises.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import FORM_NS, FORM_STOKES, SnsError, SnsOptions, SnsTimings, check, default_options
from .bcs import DirichletSet
from .mesh import TetMesh

_FORMS = {"stokes": FORM_STOKES, "ns": FORM_NS, FORM_STOKES: FORM_STOKES, FORM_NS: FORM_NS}


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


@dataclass
class KrylovResult:
    its: int
    reason: int
    rnorm: float


@dataclass
class NewtonResult:
    its: int
    reason: int
    ksp_its: int
    fnorms: list
    seconds: float


class FlowProblem:
    """Mesh + Dirichlet data + operator hierarchy on one GPU (one per rank).  ``mesh`` is a ``TetMesh`` (3-D
    G-metric forms) or a ``mesh2d.TriMesh`` (2-D UGN forms of the lid-driven / DFG-2D scripts; "stokes" and "ns"
    then name the 2-D forms, see sns_create_2d in include/sns.h).

    Replaces what ``functionspace`` / ``dirichletbc`` / ``create_matrix`` /
    ``fem.form`` build for the reference (:127-147, :45-46, :271-272).
    """

    def __init__(self, mesh: TetMesh, bcs, *, device="cuda:0", options: SnsOptions | None = None, part=None,
                 group=None, **opt_kw):
        if not torch.cuda.is_available():
            raise RuntimeError("FlowProblem needs a HIP device; the hot path has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.mesh = mesh
        if isinstance(bcs, DirichletSet):
            mask, g = bcs.flatten()
        else:
            mask, g = bcs
        self.bc_mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self.bc_val = np.ascontiguousarray(g, dtype=np.float64)
        self.options = options if options is not None else default_options(**opt_kw)
        self.dim = int(getattr(mesh, "dim", 3))
        pts = np.ascontiguousarray(mesh.points, dtype=np.float64)
        tets = np.ascontiguousarray(mesh.tris if self.dim == 2 else mesh.tets, dtype=np.int32)
        if self.bc_mask.shape != (4 * len(pts),) or self.bc_val.shape != (4 * len(pts),):
            raise ValueError("bc arrays must have 4*num_nodes entries")
        if pts.shape[1] != self.dim or tets.shape[1] != self.dim + 1:
            raise ValueError("mesh arrays do not match the mesh dimension")
        if self.dim == 2:
            if part is not None:
                raise ValueError("2-D problems run on a single GPU (no partition)")
            # the unused z component is a homogeneous Dirichlet dof (libsns.so imposes the same)
            self.bc_mask = self.bc_mask.copy()
            self.bc_val = self.bc_val.copy()
            self.bc_mask[2::4] = 1
            self.bc_val[2::4] = 0.0
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        create = self.lib.sns_create_2d if self.dim == 2 else self.lib.sns_create
        check(create(C.byref(h), len(pts), len(tets), pts.ctypes.data, tets.ctypes.data,
                     self.bc_mask.ctypes.data, self.bc_val.ctypes.data, idx, C.byref(self.options)))
        self.h = h
        self.n_local = len(pts)
        self.n_owned = len(pts)
        with torch.cuda.device(self.device):
            check(self.lib.sns_set_stream(self.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self.g_dev = torch.from_numpy(self.bc_val).to(self.device)
        self.part = part
        self.group = group
        if part is not None:
            self._attach_comm(part, group)

    @classmethod
    def distributed(cls, mesh: TetMesh, bcs, *, group=None, device=None, **kw):
        """One rank's problem of an element-partitioned run (one process per GPU).

        Every rank passes the same global mesh / BC data; RCB partition, local
        renumbering and the halo plan are computed here (partition.py); the RCCL
        communicator of the C-ABI is bootstrapped through torch.distributed."""
        import torch.distributed as dist
        from . import partition as PT
        if isinstance(group, PeerGroup):
            rank, world = group.rank, group.nranks
        else:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        mask, g = bcs.flatten() if isinstance(bcs, DirichletSet) else bcs
        owner = PT.rcb_partition(mesh.points, world)
        part = PT.build_local_part(mesh, mask, g, owner, rank, world)
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}"
        self = cls(part.mesh, (part.bc_mask, part.bc_val), device=device, part=part, group=group, **kw)
        self.global_mesh = mesh
        return self

    @classmethod
    def from_part(cls, part, *, group=None, device=None, **kw):
        """One rank's problem from a ready-made LocalPart (e.g. partition.duct_slab_part, which meshes only
        this rank's slab); same communicator bootstrap as ``distributed``."""
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}"
        self = cls(part.mesh, (part.bc_mask, part.bc_val), device=device, part=part, group=group, **kw)
        self.global_mesh = None
        self.n_global_nodes = int(part.mesh.meta.get("global_num_nodes", 0))
        return self

    def _attach_comm(self, part, group):
        nb = np.ascontiguousarray(part.neighbors, dtype=np.int32)
        sp_, si = np.ascontiguousarray(part.send_ptr, np.int32), np.ascontiguousarray(part.send_idx, np.int32)
        rp, ri = np.ascontiguousarray(part.recv_ptr, np.int32), np.ascontiguousarray(part.recv_idx, np.int32)
        if isinstance(group, Team):                    # tests: N ranks = N threads of this process
            with torch.cuda.device(self.device):
                check(self.lib.sns_attach_team(self.h, group.ptr, part.rank, part.nranks, part.n_owned, len(nb),
                                               nb.ctypes.data, sp_.ctypes.data, si.ctypes.data, rp.ctypes.data,
                                               ri.ctypes.data))
            self.n_owned = part.n_owned
            return
        if isinstance(group, PeerGroup):               # one process per GPU of one node, peer windows over xGMI
            if (group.rank, group.nranks) != (part.rank, part.nranks):
                raise ValueError("partition and peer communicator disagree about rank / ranks")
            with torch.cuda.device(self.device):
                check(self.lib.sns_attach_peer(self.h, group.ptr, part.n_owned, len(nb), nb.ctypes.data, sp_.ctypes.data,
                                               si.ctypes.data, rp.ctypes.data, ri.ctypes.data))
            self.n_owned = part.n_owned
            return
        if group == "local-only":                      # tests: owned/ghost split without a communicator
            box = [None]
        else:
            import torch.distributed as dist
            uid = C.create_string_buffer(128)
            if part.rank == 0:
                check(self.lib.sns_comm_unique_id(uid))
            box = [bytes(uid.raw)]
            if part.nranks > 1:
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                           group=group)
        with torch.cuda.device(self.device):
            check(self.lib.sns_attach_comm(self.h, part.rank, part.nranks, box[0], part.n_owned, len(nb),
                                           nb.ctypes.data, sp_.ctypes.data, si.ctypes.data, rp.ctypes.data,
                                           ri.ctypes.data))
        self.n_owned = part.n_owned

    def scatter(self, x_global) -> torch.Tensor:
        """Local (owned + ghost) device copy of a global host dof vector."""
        from . import partition as PT
        xg = np.asarray(x_global, dtype=np.float64)
        return torch.from_numpy(PT.scatter_global(self.part, xg) if self.part is not None else xg.copy()).to(self.device)

    def gather(self, x_local) -> torch.Tensor:
        """Global dof vector on every rank from the owned parts (setup / output only)."""
        from . import partition as PT
        if self.part is None or self.part.nranks == 1:
            if self.part is None:
                return x_local.clone()
            out = torch.zeros(4 * len(self.part.l2g), dtype=x_local.dtype, device=x_local.device)
            out.view(-1, 4)[torch.as_tensor(self.part.l2g, device=x_local.device)] = x_local.view(-1, 4)
            return out
        ng = self.global_mesh.num_nodes if self.global_mesh is not None else self.n_global_nodes
        return PT.gather_owned(self.part, x_local, ng, self.group.dist_group if isinstance(self.group, PeerGroup) else self.group)

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.sns_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------
    @property
    def ndof(self) -> int:
        return 4 * self.n_local

    def zeros(self) -> torch.Tensor:
        return torch.zeros(self.ndof, dtype=torch.float64, device=self.device)

    def _vec(self, x) -> torch.Tensor:
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)
        if x.dtype != torch.float64 or x.device.type != "cuda" or not x.is_contiguous() or x.numel() != self.ndof:
            raise ValueError(f"expected a contiguous float64 device vector of {self.ndof} entries")
        return x

    def set_options(self, **kw):
        for k, v in kw.items():
            if k == "ksp_type" and isinstance(v, str):
                v = _lib.KSP_NAMES[v]
            if k == "pc_type" and isinstance(v, str):
                v = _lib.PC_NAMES[v]
            if not hasattr(self.options, k):
                raise TypeError(f"unknown option {k}")
            setattr(self.options, k, v)
        check(self.lib.sns_set_options(self.h, C.byref(self.options)))

    def set_form_variant(self, c_inverse=36.0, lsic_scale=1.0, pspg_sign=1.0, one_point_quadrature=False):
        """DIAGNOSTIC: perturb the 3-D NS form (sns_set_form_variant); the defaults restore the reference's form."""
        check(self.lib.sns_set_form_variant(self.h, float(c_inverse), float(lsic_scale), float(pspg_sign), int(bool(one_point_quadrature))))

    # -- hot path -------------------------------------------------------------
    def residual(self, w, form="ns", out=None) -> torch.Tensor:
        """F(w) as the reference's .F callback leaves it (:51-67)."""
        w = None if w is None else self._vec(w)
        out = self.zeros() if out is None else self._vec(out)
        check(self.lib.sns_residual(self.h, _FORMS[form], _ptr(w), _ptr(out)))
        return out

    def jacobian(self, w, form="ns", residual_out=None):
        """Assemble J(w) into the handle (:69-75); optionally the fused residual."""
        w = None if w is None else self._vec(w)
        F = None if residual_out is None else self._vec(residual_out)
        check(self.lib.sns_jacobian(self.h, _FORMS[form], _ptr(w), _ptr(F)))
        return F

    def spmv(self, x, out=None) -> torch.Tensor:
        x = self._vec(x)
        out = self.zeros() if out is None else self._vec(out)
        check(self.lib.sns_spmv(self.h, _ptr(x), _ptr(out)))
        return out

    def pc_setup(self):
        check(self.lib.sns_pc_setup(self.h))

    def pc_apply(self, r, out=None) -> torch.Tensor:
        r = self._vec(r)
        out = self.zeros() if out is None else self._vec(out)
        check(self.lib.sns_pc_apply(self.h, _ptr(r), _ptr(out)))
        return out

    def krylov_solve(self, b, x0=None):
        b = self._vec(b)
        x = self.zeros() if x0 is None else self._vec(x0).clone()
        its, reason, rn = C.c_int(), C.c_int(), C.c_double()
        check(self.lib.sns_krylov_solve(self.h, _ptr(b), _ptr(x), C.byref(its), C.byref(reason), C.byref(rn)))
        return x, KrylovResult(its.value, reason.value, rn.value)

    def stokes_solve(self):
        U = self.zeros()
        its, reason, rn = C.c_int(), C.c_int(), C.c_double()
        check(self.lib.sns_stokes_solve(self.h, _ptr(U), C.byref(its), C.byref(reason), C.byref(rn)))
        return U, KrylovResult(its.value, reason.value, rn.value)

    def newton_solve(self, w):
        w = self._vec(w)
        its, reason, kits = C.c_int(), C.c_int(), C.c_int()
        cap = self.options.snes_max_it + 2
        hist = (C.c_double * cap)()
        t0 = time.time()
        check(self.lib.sns_newton_solve(self.h, _ptr(w), C.byref(its), C.byref(reason), C.byref(kits), hist, cap))
        dt = time.time() - t0
        return w, NewtonResult(its.value, reason.value, kits.value, list(hist[:its.value + 1]), dt)

    # -- introspection ----------------------------------------------------------
    def sizes(self):
        nl, no, nt, nz = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        check(self.lib.sns_get_sizes(self.h, C.byref(nl), C.byref(no), C.byref(nt), C.byref(nz)))
        return dict(n_local=nl.value, n_owned=no.value, n_tets=nt.value, nnzb=nz.value)

    def export(self, what: int, dtype, count: int) -> torch.Tensor:
        t = torch.empty(count, dtype=dtype, device=self.device)
        check(self.lib.sns_export(self.h, what, _ptr(t), t.numel() * t.element_size()))
        return t

    def bsr(self):
        """(rowptr, colind, vals[nnzb,4,4]) copies of the assembled operator."""
        s = self.sizes()
        rp = self.export(_lib.EXPORT_ROWPTR, torch.int32, s["n_local"] + 1)
        ci = self.export(_lib.EXPORT_COLIND, torch.int32, s["nnzb"])
        va = self.export(_lib.EXPORT_VALS, torch.float64, s["nnzb"] * 16)
        return rp, ci, va.view(-1, 4, 4)

    def to_scipy(self):
        import scipy.sparse as sp
        rp, ci, va = self.bsr()
        n = self.n_local
        return sp.bsr_matrix((va.cpu().numpy(), ci.cpu().numpy(), rp.cpu().numpy()), shape=(4 * n, 4 * n)).tocsr()

    def element_matrices(self) -> torch.Tensor:
        """Ke [n_tets, a, b, c, d] of the last jacobian() call (element-kernel output)."""
        s = self.sizes()
        return self.export(_lib.EXPORT_KE, torch.float64, s["n_tets"] * 256).view(-1, 4, 4, 4, 4)

    def time_kernels(self, on=True):
        check(self.lib.sns_time_kernels(self.h, 1 if on else 0))

    def kernel_times(self):
        """{mode: (total_ms, calls)} of the level-0 k_spmv family since reset_timings()."""
        ms = (C.c_double * 8)()
        calls = (C.c_int64 * 8)()
        check(self.lib.sns_get_kernel_times(self.h, ms, calls))
        names = ("ax", "b_minus_ax", "jacobi", "ax_dot", "post_m")
        return {names[i]: (ms[i], calls[i]) for i in range(5)}

    def timings(self) -> SnsTimings:
        t = SnsTimings()
        check(self.lib.sns_get_timings(self.h, C.byref(t)))
        return t

    def counters(self):
        """Debug counters of the last Krylov solve: host syncs, all-reduces, halo exchanges (+ ksp its since reset)."""
        c = (C.c_int64 * 8)()
        check(self.lib.sns_get_counters(self.h, c))
        return dict(host_syncs=c[0], allreduces=c[1], exchanges=c[2], ksp_its=c[3], damping_retries=c[4],
                    damping_factor=c[5] * 1e-6, first_attempt_reason=c[6], ap_blocks=c[7])

    def comm_info(self):
        """Transport / rank / ranks of the handle's communicator; ``rccl_ranks`` is what ncclCommCount reports."""
        c = (C.c_int32 * 4)()
        check(self.lib.sns_comm_info(self.h, c))
        return dict(transport={0: "none", 1: "rccl", 2: "team", 3: "peer"}[c[0]], rank=c[1], nranks=c[2], rccl_ranks=c[3])

    def hierarchy(self):
        """The AMG hierarchy as built: one dict per level (rows, 4x4 blocks, sweeps per half cycle, block-Jacobi damping)."""
        nl = C.c_int32()
        rows, blocks, nu, om = (C.c_int64 * 16)(), (C.c_int64 * 16)(), (C.c_int32 * 16)(), (C.c_double * 16)()
        check(self.lib.sns_get_hierarchy(self.h, C.byref(nl), rows, blocks, nu, om))
        return [dict(rows=rows[l], blocks=blocks[l], sweeps=nu[l], omega=om[l]) for l in range(nl.value)]

    def cycle(self):
        """The V-cycle as run: per level dict(kind, pre, post) -- kind 0 nodal-block Jacobi, 1 aggregate-block Jacobi, 2 / 3 dense
        direct solve (one-workgroup / blocked Gauss-Jordan), 4 sweeps only (sns_get_cycle)."""
        nl = C.c_int32()
        kind, pre, post = (C.c_int32 * 16)(), (C.c_int32 * 16)(), (C.c_int32 * 16)()
        check(self.lib.sns_get_cycle(self.h, C.byref(nl), kind, pre, post))
        return [dict(kind=kind[l], pre=pre[l], post=post[l]) for l in range(nl.value)]

    def reset_timings(self):
        check(self.lib.sns_reset_timings(self.h))

    def bench_spmv(self, reps=20) -> float:
        x = torch.randn(self.ndof, dtype=torch.float64, device=self.device)
        y = self.zeros()
        ms = C.c_double()
        check(self.lib.sns_bench_spmv(self.h, _ptr(x), _ptr(y), reps, C.byref(ms)))
        return ms.value

    def bench_assemble(self, w, form="ns", reps=5) -> float:
        w = self._vec(w)
        F = self.zeros()
        ms = C.c_double()
        check(self.lib.sns_bench_assemble(self.h, _FORMS[form], _ptr(w), _ptr(F), reps, C.byref(ms)))
        return ms.value


def _bench_collective(self, which: str, count: int = 5, reps: int = 200) -> float:
    """ms per collective of the attached communicator, back to back (collective call; measurement hook)."""
    ms = C.c_double()
    check(self.lib.sns_bench_collective(self.h, {"exchange": 0, "allreduce": 1, "allgather": 2}[which], int(count), int(reps),
                                        C.byref(ms)))
    return ms.value


FlowProblem.bench_collective = _bench_collective


class PeerGroup:
    """Peer-window communicator of the C-ABI (sns_peer_*): one process per GPU of one node, the collectives of the solver are
    stores into the other ranks' IPC-mapped windows -- no RCCL in the data path.  The 64-byte IPC handles are exchanged through
    ``torch.distributed`` (any backend: gloo is enough, nothing but this bootstrap and ``close`` goes through it).

        dist.init_process_group(...)
        peers = PeerGroup(device="cuda:0")                 # collective
        P = FlowProblem.distributed(mesh, bcs, group=peers)
        ...
        P.close(); peers.close()                           # collective
    """

    def __init__(self, device=None, *, group=None, window_bytes: int = 0, check_rounds: int = 50):
        import torch.distributed as dist
        self.lib = _lib.load()
        self.dist_group = group
        self.rank, self.nranks = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}"
        self.device = torch.device(device)
        p = C.c_void_p()
        hd = C.create_string_buffer(64)
        check(self.lib.sns_peer_create(self.device.index or 0, self.rank, self.nranks, int(window_bytes), C.byref(p), hd))
        self.ptr = p
        box = [None] * self.nranks
        dist.all_gather_object(box, bytes(hd.raw), group=group)          # (implies: every window exists and is zeroed)
        check(self.lib.sns_peer_connect(self.ptr, b"".join(box)))
        dist.barrier(group=group)                                        # every rank has mapped every window
        if check_rounds > 0:
            # verified all-reduces / all-gathers between the real ranks before anything relies on the links (sns_peer_check_links);
            # every rank learns the common verdict, so a failure raises everywhere instead of stranding the healthy ranks
            rc = self.lib.sns_peer_check_links(self.ptr, int(check_rounds))
            msg = self.lib.sns_last_error().decode() if rc != 0 else ""
            box2 = [None] * self.nranks
            dist.all_gather_object(box2, (rc, msg), group=group)
            bad = [(r, m) for r, (c, m) in enumerate(box2) if c != 0]
            if bad:
                raise RuntimeError("peer-window link check failed on rank(s) " + "; ".join(f"{r}: {m}" for r, m in bad))

    def close(self):
        """Collective; call after the problems attached to this communicator are closed."""
        if self.ptr:
            import torch.distributed as dist
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.dist_group)                          # nobody stores into a window that is about to go
            self.lib.sns_peer_disconnect(self.ptr)                       # unmap the others' windows ...
            dist.barrier(group=self.dist_group)                          # ... and free the own one only when nobody has it mapped
            self.lib.sns_peer_destroy(self.ptr)
            self.ptr = None


class Team:
    """In-process test transport (sns_team_*): N ranks = N threads sharing one GPU."""

    def __init__(self, nranks: int):
        self.lib = _lib.load()
        self.n = nranks
        p = C.c_void_p()
        check(self.lib.sns_team_create(nranks, C.byref(p)))
        self.ptr = p

    def close(self):
        if self.ptr:
            self.lib.sns_team_destroy(self.ptr)
            self.ptr = None

    def run(self, fn):
        """fn(rank, team) on N concurrent threads; returns the list of results, re-raises the first error."""
        import threading
        out, err = [None] * self.n, [None] * self.n

        def work(r):
            try:
                out[r] = fn(r, self)
            except BaseException as e:        # noqa: BLE001
                err[r] = e

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return out


class NonlinearPDE_SNESProblem:
    """Drop-in for the reference class of the same name (:40-75): ``F(x, F)``
    assembles the residual, ``J(x)`` the Jacobian, both on device vectors.
    (The reference's first positional ``snes`` argument has no counterpart.)"""

    def __init__(self, problem: FlowProblem, u: torch.Tensor):
        self.problem = problem
        self.u = u

    def F(self, snes, x, F):
        self.u.copy_(x)
        self.problem.residual(x, "ns", out=F)

    def J(self, snes, x, J=None, P=None):
        self.problem.jacobian(x, "ns")


def solve_stokes_problem(problem: FlowProblem, rank: int = 0):
    """``solve_stokes_problem(a, L, bcs, W)`` of the reference (:197-218): returns U."""
    if rank == 0:
        print("Starting Linear Solve", flush=True)
    U, res = problem.stokes_solve()
    if rank == 0:
        print(f"Finished Linear Solve (its {res.its}, reason {res.reason}, |r| {res.rnorm:.3e})", flush=True)
    return U


def newton_with_reynolds_continuation(problem: FlowProblem, w: torch.Tensor, max_halvings: int = 6, verbose: bool = False):
    """Newton at the problem's Reynolds number; if it fails (typically the first Jacobian at a guess far from the
    solution, at a cell Reynolds number the preconditioner cannot handle), Re is halved until a solve converges and
    then doubled back up, each stage starting from the previous solution.  NOT in the reference (which reports the
    failed reason and carries on, :297-298): opt-in robustness.  Returns (w, result of the last stage)."""
    Re = float(problem.options.reynolds)
    w_try, res = problem.newton_solve(w.clone())
    if res.reason > 0:
        w.copy_(w_try)
        return w, res
    k = 0
    cur = w.clone()
    while True:                                         # go down until a stage converges from the given guess
        k += 1
        if k > max_halvings:
            problem.set_options(reynolds=Re)
            return w, res
        problem.set_options(reynolds=Re / 2 ** k)
        w_try, r = problem.newton_solve(cur.clone())
        if verbose:
            print(f"  continuation: Re {Re / 2 ** k:g}: SNES reason {r.reason}, {r.its} its, {r.ksp_its} ksp its", flush=True)
        if r.reason > 0:
            cur = w_try
            break
    total_ksp = res.ksp_its + r.ksp_its
    while k > 0:                                        # ... and back up
        k -= 1
        problem.set_options(reynolds=Re / 2 ** k)
        w_try, r = problem.newton_solve(cur.clone())
        total_ksp += r.ksp_its
        if verbose:
            print(f"  continuation: Re {Re / 2 ** k:g}: SNES reason {r.reason}, {r.its} its, {r.ksp_its} ksp its", flush=True)
        if r.reason <= 0:
            problem.set_options(reynolds=Re)
            return w, r
        cur = w_try
    w.copy_(cur)
    r.ksp_its = total_ksp
    return w, r


def solve_navier_stokes(problem: FlowProblem, w: torch.Tensor, rank: int = 0, continuation: bool = False):
    """``solve_navier_stokes(a, w, dF, bcs, W, ksp_type, comm, rank)`` (:268-312).

    ``w`` is updated in place (``snes.solve(None, w)`` :293); returns
    ``(w, u, p)`` with u (n,3) and p (n,) the collapsed sub-functions (:310-312).
    ``continuation=True`` (not in the reference) retries a failed solve with Reynolds-number continuation.
    """
    if rank == 0:
        print("Running SNES solver", flush=True)
        print("Start Nonlinear Solve", flush=True)
    if continuation:
        w, res = newton_with_reynolds_continuation(problem, w, verbose=(rank == 0))
    else:
        w, res = problem.newton_solve(w)
    if rank == 0:
        print(f"Num SNES iterations: {res.its}", flush=True)
        print(f"SNES termination reason: {res.reason}", flush=True)
        print(f"Navier-Stokes solve time: {res.seconds:.2f} sec", flush=True)
        print("Finished Nonlinear Solve", flush=True)
    problem.last_newton = res
    W = w.view(-1, 4)
    return w, W[:, :3].contiguous(), W[:, 3].contiguous()
