"""Element partition across GPUs: owned nodes + one layer of ghost nodes + all
tets touching an owned node (host side, numpy).

This is the role DOLFINx's cell partitioner + ghost index maps play for the
reference under ``mpirun -n 6`` (run_all_images.sh:6; ghost scatters at
NavierStokesChannelFlow.py:57-66).  Differences by design:
  * every rank assembles its ghost tets redundantly, so assembly needs NO
    communication (replaces ``F.ghostUpdate(ADD, REVERSE)`` :66 and the
    MatAssembly stash exchange :75);
  * one neighbour exchange of ghost x-values per SpMV and one small all-reduce
    per group of dot products are the only data-path collectives.
Partitioner: recursive coordinate bisection (no METIS/SCOTCH offline); for the
duct it degenerates to slabs in x (<= 2 neighbours per GPU).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .mesh import TetMesh


def rcb_partition(points: np.ndarray, nparts: int) -> np.ndarray:
    """owner[node] in [0, nparts): recursive bisection along the longest extent."""
    owner = np.zeros(len(points), dtype=np.int32)

    def rec(idx, lo, k):
        if k == 1:
            owner[idx] = lo
            return
        ext = points[idx].max(axis=0) - points[idx].min(axis=0)
        ax = int(np.argmax(ext))
        kl = k // 2
        nl = (len(idx) * kl) // k
        # stable order: coordinate, then node id => deterministic, contiguous slabs on structured boxes
        order = np.lexsort((idx, points[idx, ax]))
        rec(idx[order[:nl]], lo, kl)
        rec(idx[order[nl:]], lo + kl, k - kl)

    rec(np.arange(len(points)), 0, int(nparts))
    return owner


def slab_bounds(n_nodes: int, nparts: int) -> np.ndarray:
    """Node-id ranges [b[r], b[r+1]) that rcb_partition produces when every bisection cuts the SAME axis
    and node ids ascend along it (box meshes: x slowest): the recursion on counts alone."""
    b = np.zeros(int(nparts) + 1, dtype=np.int64)

    def rec(a, e, lo, k):
        if k == 1:
            b[lo], b[lo + 1] = a, e
            return
        kl = k // 2
        nl = ((e - a) * kl) // k
        rec(a, a + nl, lo, kl)
        rec(a + nl, e, lo + kl, k - kl)

    rec(0, int(n_nodes), 0, int(nparts))
    return b


def slab_owner(n_nodes: int, nparts: int, ids=None) -> np.ndarray:
    """owner[node] for the x-slab partition (equal node counts, contiguous id ranges)."""
    ids = np.arange(n_nodes, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
    return (np.searchsorted(slab_bounds(n_nodes, nparts), ids, side="right") - 1).astype(np.int32)


def duct_slab_part(cells, x_outlet: float, rank: int, nranks: int, make_bcs=None) -> "LocalPart":
    """One rank's LocalPart of the structured duct WITHOUT building the global mesh (weak-scaling runs:
    the global mesh of an 8-GPU job has 81 M tets, every rank would otherwise mesh and partition all of it).

    Identical (ids, ordering, plans, coordinates, Dirichlet data) to
    ``build_local_part(duct_mesh(cells, x_outlet), ..., slab_owner(n_nodes, nranks), rank, nranks)``;
    slab_owner equals rcb_partition while every slab is longer than the duct is wide (RCB then cuts x only)."""
    from . import bcs as B, mesh as M
    nx, ny, nz = (int(c) for c in cells)
    sx = (ny + 1) * (nz + 1)
    n_nodes = (nx + 1) * sx
    bounds = slab_bounds(n_nodes, nranks)
    if np.any(np.diff(bounds) < 2 * sx):
        raise ValueError("duct_slab_part: fewer than two node planes per rank")
    b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
    if b1 <= b0:
        raise ValueError("duct_slab_part: empty slab")
    p0, p1 = b0 // sx, (b1 - 1) // sx                   # first / last node plane holding an owned node
    c0, c1 = max(0, p0 - 1), min(nx, p1 + 1)            # cells touching those planes
    win = M.duct_mesh((nx, ny, nz), x_outlet, x_window=(c0, c1))
    off = c0 * sx
    gid = np.arange(win.num_nodes, dtype=np.int64) + off
    owner = slab_owner(n_nodes, nranks, gid)
    mask, g = (make_bcs or B.duct_bcs)(win).flatten()
    part = build_local_part(win, mask, g, owner, rank, nranks)
    part.l2g = part.l2g + off
    part.tet_ids = part.tet_ids + 6 * c0 * ny * nz
    part.mesh.meta["global_cells"] = (nx, ny, nz)
    part.mesh.meta["global_num_nodes"] = n_nodes
    return part


@dataclass
class LocalPart:
    """One rank's share.  Local node numbering: owned nodes first (ascending
    global id), then ghosts grouped by owner rank (ascending), ascending global id."""
    rank: int
    nranks: int
    mesh: TetMesh                 # local mesh (renumbered)
    bc_mask: np.ndarray           # 4*n_local
    bc_val: np.ndarray
    n_owned: int
    l2g: np.ndarray               # local -> global node id
    tet_ids: np.ndarray           # global ids of the local tets
    neighbors: np.ndarray         # neighbour ranks (ascending)
    send_ptr: np.ndarray          # per neighbour: owned LOCAL ids to send
    send_idx: np.ndarray
    recv_ptr: np.ndarray          # per neighbour: ghost LOCAL ids to fill
    recv_idx: np.ndarray

    @property
    def n_local(self) -> int:
        return len(self.l2g)


def build_local_part(mesh: TetMesh, bc_mask, bc_val, owner: np.ndarray, rank: int, nranks: int) -> LocalPart:
    tets = mesh.tets
    town = owner[tets]                                            # (E,4)
    mine = (town == rank)
    keep = mine.any(axis=1)
    tet_ids = np.nonzero(keep)[0]
    lt = tets[keep]
    nodes = np.unique(lt.ravel())
    is_owned = owner[nodes] == rank
    owned = nodes[is_owned]
    ghosts = nodes[~is_owned]
    ghosts = ghosts[np.lexsort((ghosts, owner[ghosts]))]
    l2g = np.concatenate([owned, ghosts]).astype(np.int64)
    # nodes owned by this rank that touch none of its kept tets cannot exist (every node is in a tet),
    # but an owned node may be isolated in a degenerate input: keep it
    stray = np.setdiff1d(np.nonzero(owner == rank)[0], owned)
    if len(stray):
        owned = np.sort(np.concatenate([owned, stray]))
        l2g = np.concatenate([owned, ghosts]).astype(np.int64)
    g2l = -np.ones(mesh.num_nodes, dtype=np.int64)
    g2l[l2g] = np.arange(len(l2g))
    ltets = g2l[lt].astype(np.int32)
    n_owned = len(owned)
    # receive lists: ghosts grouped by owner
    gown = owner[ghosts]
    nbr_recv = np.unique(gown)
    # send lists: my owned nodes that sit in a tet together with a node owned by r
    mixed = keep & ~mine.all(axis=1)
    mt, mo = tets[mixed], town[mixed]
    pairs = []
    for a in range(4):
        sel_a = mo[:, a] == rank
        for b in range(4):
            if a == b:
                continue
            sel = sel_a & (mo[:, b] != rank)
            pairs.append(np.stack([mo[sel, b].astype(np.int64), mt[sel, a].astype(np.int64)], axis=1))
    pairs = np.unique(np.concatenate(pairs), axis=0) if pairs else np.zeros((0, 2), np.int64)
    nbr_send = np.unique(pairs[:, 0]).astype(np.int32)
    neighbors = np.union1d(nbr_recv, nbr_send).astype(np.int32)
    send_ptr, recv_ptr = [0], [0]
    send_idx, recv_idx = [], []
    for r in neighbors:
        s = pairs[pairs[:, 0] == r, 1]                             # ascending global id (np.unique sorted rows)
        send_idx.append(g2l[s])
        send_ptr.append(send_ptr[-1] + len(s))
        gl = np.nonzero(gown == r)[0] + n_owned
        recv_idx.append(gl)
        recv_ptr.append(recv_ptr[-1] + len(gl))
    cat = (lambda L: np.concatenate(L).astype(np.int32) if L else np.zeros(0, np.int32))
    lmesh = TetMesh(np.ascontiguousarray(mesh.points[l2g]), np.ascontiguousarray(ltets), np.zeros((0, 3), np.int32),
                    np.zeros(0, np.int32), name=f"{mesh.name}[{rank}/{nranks}]", meta=dict(mesh.meta))
    dof = (4 * l2g[:, None] + np.arange(4)[None]).ravel()
    return LocalPart(rank, nranks, lmesh, np.ascontiguousarray(np.asarray(bc_mask)[dof], dtype=np.uint8),
                     np.ascontiguousarray(np.asarray(bc_val)[dof], dtype=np.float64), n_owned, l2g, tet_ids,
                     neighbors, np.array(send_ptr, np.int32), cat(send_idx), np.array(recv_ptr, np.int32),
                     cat(recv_idx))


def scatter_global(part: LocalPart, xg: np.ndarray) -> np.ndarray:
    """Local (owned + ghost) copy of a global dof vector."""
    dof = (4 * part.l2g[:, None] + np.arange(4)[None]).ravel()
    return np.ascontiguousarray(xg[dof])


def halo_exchange_torch(part: LocalPart, x, group=None):
    """Fill the ghost tail of the local dof vector ``x`` (torch, CPU or GPU) from
    the owning ranks with torch.distributed point-to-point ops.  Same plan the
    C-ABI consumes through ``sns_attach_comm``; used by the gloo tests and for
    one-off exchanges outside the Krylov loop."""
    import torch
    import torch.distributed as dist
    xv = x.view(-1, 4)
    ops, bufs = [], []
    for k, r in enumerate(part.neighbors):
        s0, s1 = part.send_ptr[k], part.send_ptr[k + 1]
        r0, r1 = part.recv_ptr[k], part.recv_ptr[k + 1]
        if s1 > s0:
            sb = xv[torch.as_tensor(part.send_idx[s0:s1], dtype=torch.long, device=x.device)].contiguous()
            ops.append(dist.P2POp(dist.isend, sb, int(r), group))
        if r1 > r0:
            rb = torch.empty((r1 - r0, 4), dtype=x.dtype, device=x.device)
            bufs.append((rb, r0, r1))
            ops.append(dist.P2POp(dist.irecv, rb, int(r), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for rb, r0, r1 in bufs:
        xv[torch.as_tensor(part.recv_idx[r0:r1], dtype=torch.long, device=x.device)] = rb
    return x


def gather_owned(part: LocalPart, x_local, n_global_nodes: int, group=None):
    """All ranks obtain the global dof vector from the owned parts (all_gather; setup/IO only)."""
    import torch
    import torch.distributed as dist
    out_device = x_local.device
    if dist.get_backend(group) == "gloo":          # (peer-window runs bootstrap over gloo: host tensors for this setup / IO gather)
        x_local = x_local.cpu()
    own = x_local.view(-1, 4)[: part.n_owned].contiguous()
    ids = torch.as_tensor(part.l2g[: part.n_owned], dtype=torch.long, device=x_local.device)
    counts = [torch.zeros(1, dtype=torch.long, device=x_local.device) for _ in range(part.nranks)]
    dist.all_gather(counts, torch.tensor([part.n_owned], dtype=torch.long, device=x_local.device), group=group)
    mx = int(max(int(c) for c in counts))
    pad_v = torch.zeros((mx, 4), dtype=x_local.dtype, device=x_local.device)
    pad_i = torch.zeros(mx, dtype=torch.long, device=x_local.device)
    pad_v[: part.n_owned] = own
    pad_i[: part.n_owned] = ids
    vs = [torch.empty_like(pad_v) for _ in range(part.nranks)]
    is_ = [torch.empty_like(pad_i) for _ in range(part.nranks)]
    dist.all_gather(vs, pad_v, group=group)
    dist.all_gather(is_, pad_i, group=group)
    out = torch.zeros((n_global_nodes, 4), dtype=x_local.dtype, device=x_local.device)
    for c, v, i in zip(counts, vs, is_):
        out[i[: int(c)]] = v[: int(c)]
    return out.view(-1).to(out_device)
