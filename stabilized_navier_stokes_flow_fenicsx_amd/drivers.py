"""Entry points with the reference's command lines, on the HIP hot path.

  NavierStokesChannelFlow.py <Re> <img_fname> <flowrate_ratio> [<channel_mesh_size>=0.1]
        (NavierStokes/NavierStokesChannelFlow.py:81-93, flow :468-580)
  StokesChannelFlow.py       <img_fname> <flowrate_ratio> [<mesh_size>=0.25]
        (StokesFlow/StokesChannelFlow.py:34-41)
  DuctStokesFlow.py          <gmsh_fname> <mesh_lc> <x_outlet>
        (StokesFlow/DuctStokesFlow.py:18-20)
  LidDrivenNavierStokesFlow.py <Re> [<NumCells>=64]
        (LidDrivenFlow/LidDrivenNavierStokesFlow.py:17-23)

What is kept: argv, the Stokes -> coarse NS -> fine NS continuation (:513-530),
boundary-condition sets, solver settings, printed diagnostics, output folder /
file names.  What differs, because gmsh / skimage / dolfinx do not exist offline
(SURVEY 8f, next-row 2): an existing inlet PNG is processed on its pixel grid
(inlet_image.py: regions, Poisson profiles, flow-ratio scaling as image2inlet.py:
240-339, nozzle walls as no-slip nodes) and drives a structured 4x1x1 box channel;
``<img_fname>`` may also be a gmsh ``.msh`` file (tags inlet_1=1, inlet_2=2,
outlet=3, wall=4); with neither, a centred square inner stream with analytic
profiles of the same normalisation is used.  DuctStokesFlow keeps
the geometry/BCs/CLI of the reference file but solves the P1-P1 stabilised form
(the north-star discretisation), not its P2-P1 + MUMPS one.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from . import bcs as B, mesh as M
from .interpolate import interpolate_initial_guess

snes_ksp_type = "bicgstab"          # reference: 'tfqmr' (:77); both are short-recurrence Lanczos-type methods


def _continuation() -> bool:
    """SNS_CONTINUATION=1: retry a failed Newton solve with Reynolds-number continuation (the command lines stay
    those of the reference scripts, so this is an environment switch)."""
    return bool(os.environ.get("SNS_CONTINUATION"))


def _rank():
    try:
        import torch.distributed as dist
        return dist.get_rank() if dist.is_initialized() else 0
    except Exception:
        return 0


# --------------------------------------------------------------------------- #
# argv
# --------------------------------------------------------------------------- #
def parse_arguments(argv=None):
    """Same contract and error as the reference (:81-93)."""
    argv = sys.argv if argv is None else argv
    if len(argv) not in [4, 5]:
        raise ValueError("Usage: script.py <Re> <img_fname> <flowrate_ratio> [<channel_mesh_size>]")
    Re = int(argv[1])
    given = argv[2]
    img_fname = given[1:] if given.startswith(".") else given         # str.removeprefix(".")  (:87)
    img_fname = os.getcwd() + img_fname                                # (:88-89)
    if os.path.isabs(given) and os.path.exists(given):                 # leniency: an existing absolute path is used as is
        img_fname = given
    flowrate_ratio = float(argv[3])
    channel_mesh_size = float(argv[4]) if len(argv) == 5 else 0.1
    return Re, img_fname, flowrate_ratio, channel_mesh_size


def lc_to_cells(lc: float, length: float = 4.0):
    """Structured stand-in for gmsh's characteristic length: h = lc on a length x 1 x 1 box."""
    n = max(2, int(round(1.0 / lc)))
    return (max(2, int(round(length / lc))), n, n)


# --------------------------------------------------------------------------- #
# mesh + BCs  (generate_mesh :107-116, create_boundary_conditions :127-147)
# --------------------------------------------------------------------------- #
def generate_mesh(img_fname: str, channel_mesh_size: float):
    if _rank() == 0:
        print("Meshing", flush=True)
    if img_fname.endswith(".msh") and os.path.exists(img_fname):
        msh = M.read_msh(img_fname)
        msh.meta.update(kind="channel", tags=dict(M.CHANNEL_TAGS))
    else:
        msh = M.channel_mesh(lc_to_cells(channel_mesh_size))
    if _rank() == 0:
        print(f"Num elem: {msh.num_tets}", flush=True)
    return msh


def create_boundary_conditions(msh, flowrate_ratio):
    p1, p2 = B.two_stream_profiles(flowrate_ratio, msh.meta.get("inner_half_width", 0.25))
    return B.channel_bcs(msh, p1, p2)


def channel_problem_inputs(img_fname: str, flowrate_ratio: float, channel_mesh_size: float):
    """(mesh, bcs) of one continuation stage (generate_mesh + create_boundary_conditions, :107-147).
    An existing image drives the inlet as in image2inlet.py (pixel-grid restatement, inlet_image.py);
    a ``.msh`` file is read as is; anything else falls back to the synthetic centred-square inlet."""
    is_img = img_fname.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")) and os.path.exists(img_fname)
    if is_img:
        from .inlet_image import channel_from_image
        if _rank() == 0:
            print("Meshing", flush=True)
        msh, bcs, _ = channel_from_image(img_fname, flowrate_ratio, lc_to_cells(channel_mesh_size))
        if _rank() == 0:
            print(f"Num elem: {msh.num_tets}", flush=True)
        return msh, bcs
    msh = generate_mesh(img_fname, channel_mesh_size)
    return msh, create_boundary_conditions(msh, flowrate_ratio)


# --------------------------------------------------------------------------- #
# output (make_output_folder :416-465, write_run_metadata :384-413,
#         save_navier_stokes_solution :316-346)
# --------------------------------------------------------------------------- #
def make_output_folder(Re, img_fname, channel_mesh_size):
    cwd = os.getcwd()
    img_name = img_fname[:-4] if img_fname.endswith((".png", ".msh")) else img_fname
    if img_name.startswith(cwd):
        img_name = img_name[len(cwd):]
    if img_name.startswith("/InletImages/"):
        img_name = img_name[len("/InletImages/"):]
    img_name = img_name.strip("/").replace("/", "_")
    mesh_str = str(channel_mesh_size).replace(".", "")
    noether = os.path.join(cwd, "noether_data")
    folder = os.path.join(noether, f"NSChannelFlow_RE{Re}_MeshLC{mesh_str}_{img_name}")
    if _rank() == 0:
        os.makedirs(folder, exist_ok=True)
    return folder, img_name


def write_run_metadata(folder, Re, img_fname, flowrate_ratio, channel_mesh_size, msh, nranks=1):
    if _rank() != 0:
        return
    with open(os.path.join(folder, "RunParameters.txt"), "w") as fh:
        fh.write(f"Re={Re}\n")
        fh.write(f"img_filename={img_fname}\n")
        fh.write(f"Flowrate Ratio={flowrate_ratio}\n")
        fh.write(f"Channel Mesh Size={channel_mesh_size}\n")
        fh.write(f"Pressure DOFs: {msh.num_nodes}\n")
        fh.write(f"Velocity DOFs: {msh.num_nodes}\n")        # dolfinx counts blocked dofs: one per node
        fh.write(f"{nranks} Cores Used\n")


def write_xdmf(path_noext: str, msh, name: str, values: np.ndarray):
    """P1 nodal field as XDMF + raw little-endian binaries (``write_mesh`` +
    ``write_function`` of :333-341).  The reference stores heavy data in HDF5;
    h5py is not available offline, so the heavy data are ``Format="Binary"`` items
    -- same XDMF structure, same Grid/Attribute names (Velocity / Pressure)."""
    base = os.path.basename(path_noext)
    values = np.ascontiguousarray(values, dtype="<f8")
    geo, topo, dat = path_noext + "_geometry.bin", path_noext + "_topology.bin", path_noext + f"_{name}.bin"
    np.ascontiguousarray(msh.points, dtype="<f8").tofile(geo)
    np.ascontiguousarray(msh.tets, dtype="<i4").tofile(topo)
    values.tofile(dat)
    ncomp = 1 if values.ndim == 1 else values.shape[1]
    atype = "Scalar" if ncomp == 1 else "Vector"
    dims = f"{msh.num_nodes} {ncomp}" if ncomp > 1 else f"{msh.num_nodes}"
    xml = f"""<?xml version="1.0"?>
<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>
<Xdmf Version="3.0" xmlns:xi="https://www.w3.org/2001/XInclude">
  <Domain>
    <Grid Name="mesh" GridType="Uniform">
      <Topology TopologyType="Tetrahedron" NumberOfElements="{msh.num_tets}" NodesPerElement="4">
        <DataItem Dimensions="{msh.num_tets} 4" NumberType="Int" Precision="4" Format="Binary" Endian="Little">{base}_topology.bin</DataItem>
      </Topology>
      <Geometry GeometryType="XYZ">
        <DataItem Dimensions="{msh.num_nodes} 3" NumberType="Float" Precision="8" Format="Binary" Endian="Little">{base}_geometry.bin</DataItem>
      </Geometry>
      <Attribute Name="{name}" AttributeType="{atype}" Center="Node">
        <DataItem Dimensions="{dims}" NumberType="Float" Precision="8" Format="Binary" Endian="Little">{base}_{name}.bin</DataItem>
      </Attribute>
    </Grid>
  </Domain>
</Xdmf>
"""
    with open(path_noext + ".xdmf", "w") as fh:
        fh.write(xml)


def save_navier_stokes_solution(u, p, msh, FolderName, Re):
    if _rank() != 0:
        return
    print("[Rank 0] Starting save_navier_stokes_solution()", flush=True)
    write_xdmf(os.path.join(FolderName, f"Re{Re}ChannelPressure"), msh, "Pressure", np.asarray(p))
    write_xdmf(os.path.join(FolderName, f"Re{Re}ChannelVelocity"), msh, "Velocity", np.asarray(u))
    print("[Rank 0] Solution writing complete.", flush=True)


# --------------------------------------------------------------------------- #
# drivers
# --------------------------------------------------------------------------- #
def _init_distributed():
    """Under ``torchrun`` / ``torch.distributed.run`` (the counterpart of the reference's
    ``mpirun -n 6``, run_all_images.sh:6): one rank per GPU over RCCL."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{lr}"))
    return world


def _problem(msh, bcs, **opt):
    """One FlowProblem per rank: element-partitioned when launched on several GPUs."""
    import torch
    from .solver import FlowProblem
    if _init_distributed() > 1:
        opt.pop("device", None)
        return FlowProblem.distributed(msh, bcs, device=f"cuda:{torch.cuda.current_device()}", **opt)
    return FlowProblem(msh, bcs, **opt)


def _to_global_host(P, x):
    """Global nodal vector on the host (identical on every rank)."""
    return (P.gather(x) if getattr(P, "part", None) is not None else x).cpu().numpy()


def _from_global_host(P, xg):
    import torch
    return P.scatter(xg) if getattr(P, "part", None) is not None else torch.from_numpy(np.ascontiguousarray(xg)).to(P.device)


def solve_NS_flow(argv=None, *, coarse_mesh_size: float = 0.1, device="cuda:0"):
    """The reference's three-stage continuation (:468-549): Stokes on the 0.1 mesh ->
    Navier-Stokes on the 0.1 mesh -> Navier-Stokes on the user mesh, each stage
    warm-started from the previous one."""
    import torch
    from .solver import solve_navier_stokes, solve_stokes_problem
    Re, img_fname, flowrate_ratio, channel_mesh_size = parse_arguments(argv)
    rank = _rank()
    if rank == 0:
        print("Accepted Inputs", flush=True)
    # Solve Stokes Flow
    msh, bcs = channel_problem_inputs(img_fname, flowrate_ratio, coarse_mesh_size)
    P = _problem(msh, bcs, reynolds=float(Re), ksp_type=snes_ksp_type, device=device)
    U_stokes = solve_stokes_problem(P, rank)
    # Solve Coarse Navier Stokes
    if rank == 0:
        print("Interpolating Stokes Flow", flush=True)
    w_coarse, u, p = solve_navier_stokes(P, U_stokes.clone(), rank, continuation=_continuation())
    w_coarse_host = _to_global_host(P, w_coarse)
    P.close()
    # Solve Navier Stokes With User Defined Mesh
    msh_f, bcs_f = channel_problem_inputs(img_fname, flowrate_ratio, channel_mesh_size)
    Pf = _problem(msh_f, bcs_f, reynolds=float(Re), ksp_type=snes_ksp_type, device=device)
    if rank == 0:
        print("Interpolating Coarse NS Flow", flush=True)
    w0 = interpolate_initial_guess(msh, w_coarse_host, msh_f)
    w, u, p = solve_navier_stokes(Pf, _from_global_host(Pf, w0), rank, continuation=_continuation())
    wg = _to_global_host(Pf, w)
    out = dict(msh=msh_f, w=wg, u=wg.reshape(-1, 4)[:, :3].copy(), p=wg.reshape(-1, 4)[:, 3].copy(), Re=Re, img_fname=img_fname,
               channel_mesh_size=channel_mesh_size, flowrate_ratio=flowrate_ratio, newton=Pf.last_newton)
    Pf.close()
    return out


def navier_stokes_channel_main(argv=None):
    t0 = time.time()
    r = solve_NS_flow(argv)
    folder, _ = make_output_folder(r["Re"], r["img_fname"], r["channel_mesh_size"])
    save_navier_stokes_solution(r["u"], r["p"], r["msh"], folder, r["Re"])
    write_run_metadata(folder, r["Re"], r["img_fname"], r["flowrate_ratio"], r["channel_mesh_size"], r["msh"])
    if _rank() == 0:
        print(f"Run Time = {time.time() - t0:.2f} sec; output in {folder}", flush=True)
    return r


def stokes_channel_main(argv=None):
    """StokesChannelFlow.py: linear P1-P1 pressure-stabilised Stokes, bcgs rtol=atol=1e-10 (:166)."""
    argv = sys.argv if argv is None else argv
    if len(argv) not in [3, 4]:
        raise ValueError("Usage: script.py <img_fname> <flowrate_ratio> [<mesh_size>]")
    img_fname, ratio = argv[1], float(argv[2])
    mesh_size = float(argv[3]) if len(argv) == 4 else 0.25
    t0 = time.perf_counter()
    msh, bcs = channel_problem_inputs(os.path.abspath(img_fname), ratio, mesh_size)
    P = _problem(msh, bcs, ksp_type="bicgstab", ksp_rtol=1e-10, ksp_atol=1e-10)
    print("\nStart Assembling Stiffness Matrix and Forcing Vector", flush=True)
    U, res = P.stokes_solve()
    print(f"Solve finished: {res.its} iterations, reason {res.reason}, {time.perf_counter() - t0:.2f} s", flush=True)
    W = _to_global_host(P, U).reshape(-1, 4)
    if _rank() == 0:
        write_xdmf("StokesChannelPressure", msh, "Pressure", W[:, 3])
        write_xdmf("StokesChannelVelocity", msh, "Velocity", W[:, :3])
    P.close()
    return msh, W


def duct_stokes_main(argv=None):
    """DuctStokesFlow.py <gmsh_fname> <mesh_lc> <x_outlet> (:18-20): geometry :36-124, BCs :156-183,
    printed norms :234-241 (labels as in the reference: 'L1' is the l2 norm of the coefficient vector)."""
    argv = sys.argv if argv is None else argv
    if len(argv) != 4:
        raise ValueError("Usage: DuctStokesFlow.py <gmsh_fname> <mesh_lc> <x_outlet>")
    gmsh_fname, mesh_lc, x_outlet = argv[1], float(argv[2]), float(argv[3])
    msh = M.duct_mesh(lc_to_cells(mesh_lc, x_outlet), x_outlet)
    M.write_msh2(msh, f"{gmsh_fname}.msh")                        # gmsh.write(f'{gmsh_fname}.msh') :141
    P = _problem(msh, B.duct_bcs(msh))
    U, res = P.stokes_solve()
    W = _to_global_host(P, U).reshape(-1, 4)
    u, p = W[:, :3], W[:, 3]
    if _rank() == 0:
        print(f"L1 norm of velocity coefficient vector: {np.linalg.norm(u.ravel())}")
        print(f"L1 norm of pressure coefficient vector: {np.linalg.norm(p)}")
        print(f"Linf norm of pressure coefficient vector: {np.abs(u).max()}")
        print(f"Linf norm of pressure coefficient vector: {np.abs(p).max()}")
        write_xdmf("StokesDuctPressure", msh, "f", p)
        write_xdmf("StokesDuctVelcoity", msh, "f", u)             # (sic) file name of the reference :255
    P.close()
    return msh, W, res


def lid_driven_main(argv=None):
    """LidDrivenNavierStokesFlow.py <Re> [<NumCells>=64] (:17-23) on the 3-D unit cube:
    lid y=1 moving with (1,0,0), no-slip elsewhere, p=0 at the origin (:57-77); Stokes then Newton."""
    import torch
    from .solver import solve_navier_stokes
    argv = sys.argv if argv is None else argv
    if len(argv) not in [2, 3]:
        raise ValueError("Usage: LidDrivenNavierStokesFlow.py <Re> [<NumCells>]")
    Re = int(argv[1])
    n = int(argv[2]) if len(argv) == 3 else 64
    t0 = time.time()
    msh = M.cavity_mesh(n)
    print(f"Pressure Degress of Freedom: {msh.num_nodes}")
    print(f"Velocity Degress of Freedom: {msh.num_nodes}")
    P = _problem(msh, B.cavity_bcs(msh), reynolds=float(Re))
    U, res = P.stokes_solve()
    print("Solved Stokes Flow")
    w, u, p = solve_navier_stokes(P, U.clone(), _rank(), continuation=_continuation())
    wg = _to_global_host(P, w)
    if _rank() == 0:
        print(f"run time = {time.time() - t0: 0.2f} sec")
        write_xdmf(f"NavierStokesLidDrivenPressureLinear{Re}", msh, "Pressure", wg.reshape(-1, 4)[:, 3].copy())
        write_xdmf(f"NavierStokesLidDrivenPressureVelocity{Re}", msh, "Velocity", wg.reshape(-1, 4)[:, :3].copy())
    r = P.last_newton
    P.close()
    return msh, wg, r
