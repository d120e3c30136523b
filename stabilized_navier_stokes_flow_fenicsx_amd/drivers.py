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
(SURVEY 8f, next-row 2): an existing inlet PNG goes through the repository's own
restatement of the reference's pipeline (inlet_contours.py: marching-squares contours,
FFT low-pass, RDP, P1 Poisson profiles with the flow-ratio scaling of image2inlet.py:
240-339) and -- since round 5 -- drives the BODY-FITTED channel of nozzle_mesh.py: the box
4 x 1 x 1 minus the nozzle wall extruded over x in [0, 0.5], tags inlet_1 / inlet_2 /
outlet / wall = 1-4, sizes along x after the reference's Box fields
(image2gmsh3D.py:164-486).  SNS_CHANNEL_MESH=structured selects rounds 2-4's
structured box with the nozzle wall as staircase no-slip nodes (inlet_image.py).
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
    An existing image drives the inlet as in image2inlet.py + image2gmsh3D.py: contour pipeline, Poisson profiles and the
    body-fitted nozzle channel (nozzle_mesh.py; SNS_CHANNEL_MESH=structured: the structured box with staircase walls);
    a ``.msh`` file is read as is; anything else falls back to the synthetic centred-square inlet."""
    is_img = img_fname.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")) and os.path.exists(img_fname)
    if is_img:
        if _rank() == 0:
            print("Meshing", flush=True)
        if os.environ.get("SNS_CHANNEL_MESH", "bodyfitted") == "structured":
            from .inlet_image import channel_from_image
            msh, bcs, _ = channel_from_image(img_fname, flowrate_ratio, lc_to_cells(channel_mesh_size))
        else:
            from .nozzle_mesh import channel_from_image_bodyfitted
            msh, bcs, _ = channel_from_image_bodyfitted(img_fname, flowrate_ratio, channel_mesh_size)
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
    """P1 nodal field as ``<path>.xdmf`` + ``<path>.h5`` -- ``XDMFFile.write_mesh`` + ``write_function`` of
    :333-341.  Same container layout as dolfinx 0.9 writes: ``/Mesh/mesh/geometry`` (N x gdim float64),
    ``/Mesh/mesh/topology`` (E x nodes-per-cell int64), ``/Function/<name>/0`` (N x ncomp float64; a scalar is
    N x 1, a 2-D vector is padded to N x 3 with uz = 0 as dolfinx does), the light data pointing into it with ``Format="HDF"`` items -- so the reference's consumer
    (streamtrace.py:87-96: ``h5f["Function"][name]["0"]``) and ParaView open it unchanged.  h5py does not exist
    offline; the container is written by ``h5lite`` (file-format spec, v1 objects)."""
    from .h5lite import H5Writer
    base = os.path.basename(path_noext)
    dim = int(getattr(msh, "dim", 3))
    cells = msh.tris if dim == 2 else msh.tets
    values = np.asarray(values, dtype=np.float64)
    vals2 = values.reshape(len(values), -1)
    if vals2.shape[1] == 2:
        # dolfinx' XDMF write_function pads a 2-component vector to 3 (XDMF has no 2-vector attribute): N x 3, uz == 0
        vals2 = np.concatenate([vals2, np.zeros((len(vals2), 1))], axis=1)
    ncomp = vals2.shape[1]
    w = H5Writer()
    w.dataset("/Mesh/mesh/geometry", np.asarray(msh.points, dtype=np.float64))
    w.dataset("/Mesh/mesh/topology", np.asarray(cells, dtype=np.int64))
    w.dataset(f"/Function/{name}/0", vals2)
    w.write(path_noext + ".h5")
    atype = "Scalar" if ncomp == 1 else "Vector"
    ttype, npe, gtype = ("Triangle", 3, "XY") if dim == 2 else ("Tetrahedron", 4, "XYZ")
    ncell, nn = len(cells), msh.num_nodes
    xml = f"""<?xml version="1.0"?>
<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>
<Xdmf Version="3.0" xmlns:xi="https://www.w3.org/2001/XInclude">
  <Domain>
    <Grid Name="mesh" GridType="Uniform">
      <Topology TopologyType="{ttype}" NumberOfElements="{ncell}" NodesPerElement="{npe}">
        <DataItem Dimensions="{ncell} {npe}" NumberType="Int" Format="HDF">{base}.h5:/Mesh/mesh/topology</DataItem>
      </Topology>
      <Geometry GeometryType="{gtype}">
        <DataItem Dimensions="{nn} {dim}" Format="HDF">{base}.h5:/Mesh/mesh/geometry</DataItem>
      </Geometry>
    </Grid>
    <Grid Name="{name}" GridType="Collection" CollectionType="Temporal">
      <Grid Name="{name}" GridType="Uniform">
        <xi:include xpointer="xpointer(/Xdmf/Domain/Grid[@GridType='Uniform'][1]/*[self::Topology or self::Geometry])" />
        <Time Value="0" />
        <Attribute Name="{name}" AttributeType="{atype}" Center="Node">
          <DataItem Dimensions="{nn} {ncomp}" Format="HDF">{base}.h5:/Function/{name}/0</DataItem>
        </Attribute>
      </Grid>
    </Grid>
  </Domain>
</Xdmf>
"""
    with open(path_noext + ".xdmf", "w") as fh:
        fh.write(xml)


def read_xdmf_function(path_noext: str, name: str):
    """(points, cells, values) back from a ``write_xdmf`` / dolfinx file pair, the access pattern of
    ``read_mesh_and_function`` (streamtrace.py:58-130): mesh from ``/Mesh/mesh``, data from ``/Function/<name>/0``."""
    from .h5lite import read_datasets
    d = read_datasets(path_noext + ".h5")
    return d["/Mesh/mesh/geometry"], d["/Mesh/mesh/topology"], d[f"/Function/{name}/0"]


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
    """LidDrivenNavierStokesFlow.py <Re> [<NumCells>=64] (:17-23).

    Default = the script as written: the 2-D unit square, ``create_rectangle`` triangles (:29-30), lid y=1 moving
    with (1,0), no-slip elsewhere, p=0 at the origin (:57-77); Stokes with viscosity nu and mu_T = a0 h^2/(4 nu)
    (:96-109) as initial guess, then the UGN-stabilised NS form (:123-143).  A trailing ``3d`` argument (or
    SNS_CAVITY_DIM=3) runs the 3-D unit-cube extension with the G-metric form instead."""
    import torch
    from .solver import solve_navier_stokes
    argv = list(sys.argv if argv is None else argv)
    three_d = os.environ.get("SNS_CAVITY_DIM", "2") == "3"
    if argv and argv[-1].lower() == "3d":
        three_d = True
        argv = argv[:-1]
    if len(argv) not in [2, 3]:
        raise ValueError("Usage: LidDrivenNavierStokesFlow.py <Re> [<NumCells>] [3d]")
    Re = int(argv[1])
    n = int(argv[2]) if len(argv) == 3 else 64
    t0 = time.time()
    if three_d:
        msh = M.cavity_mesh(n)
        bcs, opt = B.cavity_bcs(msh), {}
    else:
        from . import mesh2d as M2
        nu = 1.0 / Re
        msh = M2.rectangle_mesh(n)
        bcs, opt = M2.cavity2d_bcs(msh), dict(stokes_viscosity=nu, stokes_beta=(1.0 / 3.0) / (4.0 * nu))   # :98-100
    print(f"Pressure Degress of Freedom: {msh.num_nodes}")
    print(f"Velocity Degress of Freedom: {msh.num_nodes}")
    P = _problem(msh, bcs, reynolds=float(Re), **opt)
    U, res = P.stokes_solve()
    print("Solved Stokes Flow")
    w, u, p = solve_navier_stokes(P, U.clone(), _rank(), continuation=_continuation())
    wg = _to_global_host(P, w)
    nd = 3 if three_d else 2
    if _rank() == 0:
        print(f"run time = {time.time() - t0: 0.2f} sec")
        write_xdmf(f"NavierStokesLidDrivenPressureLinear{Re}", msh, "Pressure", wg.reshape(-1, 4)[:, 3].copy())
        write_xdmf(f"NavierStokesLidDrivenPressureVelocity{Re}", msh, "Velocity", wg.reshape(-1, 4)[:, :nd].copy())
    r = P.last_newton
    P.close()
    return msh, wg, r


def lid_driven_stokes_main(argv=None):
    """LidDrivenStokesFlow.py (no arguments): Stokes flow in the 2-D unit square, 64 x 64 ``create_rectangle`` triangles
    (:16-17), lid y = 1 moving with (1, 0), no slip on the other three sides (:20-56), nu = 0.01 and the pressure
    stabilisation mu_T = a0 h^2 / (4 nu), a0 = 1/3 (:67-75), bcgs to 1e-10 (:81), norms of the coefficient vectors (:87-93),
    XDMF + HDF5 output (:100-121).  The script's active element is Taylor-Hood P2-P1 with its P1-P1 line commented out
    (:33-34); here it is the P1-P1 pair with exactly that stabilised form, as for DuctStokesFlow.py.
    The script pins no pressure level; the hot path keeps the cavity's p = 0 at the origin of the NS script (the constant
    is otherwise free)."""
    from . import mesh2d as M2
    argv = list(sys.argv if argv is None else argv)
    n = int(argv[1]) if len(argv) > 1 else 64
    nu = 0.01
    msh = M2.rectangle_mesh(n)
    print(f"There are this Many Degrees of Freedom in the Pressure Nodes: {msh.num_nodes}")
    P = _problem(msh, M2.cavity2d_bcs(msh), reynolds=1.0 / nu, stokes_viscosity=nu, stokes_beta=(1.0 / 3.0) / (4.0 * nu),
                 ksp_type="bicgstab", ksp_rtol=1e-10, ksp_atol=1e-10)
    U, res = P.stokes_solve()
    Ug = _to_global_host(P, U).reshape(-1, 4)
    u, p = Ug[:, :2], Ug[:, 3]
    if _rank() == 0:
        print(f"\nL2 Norm of velocity coefficient vector: {np.linalg.norm(u)}")
        print(f"Infinite Norm of velocity coefficient vector: {np.abs(u).max()}")
        print(f"L2 Norm of pressure coefficient vector: {np.linalg.norm(p)}")
        print(f"Infinite Norm of pressure coefficient vector: {np.abs(p).max()}")
        print("\nFinished Solving, Saving Solution Field")
        write_xdmf("StokesLidDrivenPressureHighRe", msh, "Pressure", p.copy())
        write_xdmf("StokesLidDrivenVelocityHighRe", msh, "Velocity", u.copy())
    P.close()
    return msh, Ug.ravel(), res


def dfg_2d_main(argv=None):
    """DFG_2D_Validation.py <msh file> (:22-28): Stokes with unit viscosity and mu_T = 0.2 h^2 (:101-125), NS with
    nu = 1e-3 and the UGN stabilisation from the Stokes field (:141-187), drag / lift and their relative errors
    against the script's constants (:195-214), XDMF output (:216-238).

    ``<msh file>`` is a gmsh ASCII file of dfg_pillar_2D.geo, or ``builtin[:level]`` for the gmsh-free mesh of the
    same geometry (mesh2d.dfg_2d_mesh).  Newton starts from the Stokes field exactly as the script passes it on
    (:141-187).  Only if that solve does NOT converge is it repeated with the Stokes pressure rescaled by nu (the unit-
    viscosity pressure is 1/nu times too large for the NS problem; on meshes coarser than the .geo's the full Newton
    step from the unscaled field diverges, in the oracle's LU-Newton as well) -- a documented fallback, announced on
    stdout; SNS_DFG_RESCALED_GUESS=1 goes straight to the rescaled guess."""
    from . import mesh2d as M2
    from .solver import solve_navier_stokes
    argv = sys.argv if argv is None else argv
    if len(argv) != 2:
        raise ValueError("Usage: DFG_2D_Validation.py <msh file | builtin[:level]>")
    src = argv[1]
    if src.startswith("builtin"):
        msh = M2.dfg_2d_mesh(float(src.split(":")[1]) if ":" in src else 4.0)
    else:
        msh = M2.read_msh_2d(src)
    nu = 1e-3                                                          # :148
    if _rank() == 0:
        print(f"Pressure Degrees of Freedom: {msh.num_nodes}", flush=True)
        print(f"Velocity Degrees of Freedom: {msh.num_nodes}", flush=True)
    P = _problem(msh, M2.dfg2d_bcs(msh), reynolds=1.0 / nu, stokes_viscosity=1.0, stokes_beta=0.2,
                 snes_rtol=1e-9, snes_stol=1e-9)
    U, res = P.stokes_solve()
    if _rank() == 0:
        print("Solved Stokes Flow", flush=True)
    rescale_first = os.environ.get("SNS_DFG_RESCALED_GUESS", "0") == "1"
    U0 = U.clone()
    if rescale_first:
        U0.view(-1, 4)[:, 3] *= nu
    w, u, p = solve_navier_stokes(P, U0.clone(), _rank(), continuation=_continuation())
    if P.last_newton.reason <= 0 and not rescale_first:
        if _rank() == 0:
            print(f"Newton from the unit-viscosity Stokes field did not converge (reason {P.last_newton.reason}): "
                  "repeating with the Stokes pressure rescaled by nu", flush=True)
        U0 = U.clone()
        U0.view(-1, 4)[:, 3] *= nu
        w, u, p = solve_navier_stokes(P, U0, _rank(), continuation=_continuation())
    wg = _to_global_host(P, w)
    cd, cl = M2.drag_lift_2d(msh, wg, nu)
    if _rank() == 0:
        print(f"Coefficient of Lift: {[cl]}", flush=True)
        print(f"Cl Percent Error: {[(cl - M2.DFG2D_CL_REF) / M2.DFG2D_CL_REF]}", flush=True)
        print(f"Coefficient of Drag: {[cd]}", flush=True)
        print(f"Cd Percent Error: {[(cd - M2.DFG2D_CD_REF) / M2.DFG2D_CD_REF]}", flush=True)
        write_xdmf("DFG2DValidationPressure", msh, "Pressure", wg.reshape(-1, 4)[:, 3].copy())
        write_xdmf("DFG2DValidationVelocity", msh, "Velocity", wg.reshape(-1, 4)[:, :2].copy())
    r = P.last_newton
    P.close()
    return msh, wg, (cd, cl), r


def dfg_3d_main(argv=None):
    """DFG_3D_Validation.py (no arguments: reads ``dfg_pillar_3D.msh`` from the working directory, :73-79): Stokes,
    NS with nu = 1e-3 (:193) from the Stokes field, traction integral over the obstacle (tag 5) and
    C = 2 F / (rho Uc^2 Lc), Uc = 0.2, Lc = 0.1 * 0.41 (:344-367), XDMF output (:381-396).  Without the .msh (gmsh is
    absent offline) the same geometry is meshed by Delaunay (``mesh.dfg_pillar_mesh``; ``builtin[:n]`` as argument picks
    the resolution h = W/n, default 32)."""
    from . import functionals as Fn
    from .solver import solve_navier_stokes
    argv = sys.argv if argv is None else argv
    src = argv[1] if len(argv) > 1 else "dfg_pillar_3D.msh"
    if src.startswith("builtin") or not os.path.exists(src):
        n = int(src.split(":")[1]) if src.startswith("builtin") and ":" in src else 32
        msh = M.reorder_for_locality(M.dfg_pillar_mesh(n, lattice="bcc"))[0]
    else:
        msh = M.read_msh(src)
        msh.meta.setdefault("tags", {"inlet": 2, "outlet": 3, "wall": 4, "obstacle": 5})      # :104-109
        msh.meta.setdefault("width", 0.41)
    nu = 0.001
    if _rank() == 0:
        print(f"Pressure Degrees of Freedom: {msh.num_nodes}", flush=True)
        print(f"Velocity Degrees of Freedom: {msh.num_nodes}", flush=True)
    P = _problem(msh, B.dfg_bcs(msh), reynolds=1.0 / nu, ksp_type=snes_ksp_type)
    U, res = P.stokes_solve()
    if _rank() == 0:
        print("Solved Stokes Flow", flush=True)
    w, u, p = solve_navier_stokes(P, U.clone(), _rank(), continuation=_continuation())
    wg = _to_global_host(P, w)
    force = Fn.boundary_traction_force(msh, wg, nu, msh.meta["tags"]["obstacle"])
    cd, cl = Fn.drag_lift_coefficients(force)
    if _rank() == 0:
        print(f"Coefficient of Lift: {cl}", flush=True)
        print(f"Coefficient of Drag: {cd}", flush=True)
        write_xdmf("DFGValidationPressureNavierStokes", msh, "Pressure", wg.reshape(-1, 4)[:, 3].copy())
        write_xdmf("DFGValidationVelocityNavierStokes", msh, "Velocity", wg.reshape(-1, 4)[:, :3].copy())
    r = P.last_newton
    P.close()
    return msh, wg, (cd, cl), r


def streamtrace_main(argv=None):
    """streamtrace.py <img_fname> <solname> <funcname> (:475-491, main :667-689): velocity from ``<solname>.xdmf/.h5``,
    forward trace of the inner inlet mesh, alpha-shape bound, 50 x 50 reverse seeds, contour filter; writes
    ``rev_seeds.csv`` and ``final_output.csv`` next to the image (save_figs :497-520; the SVG figures are not drawn)."""
    from . import streamtrace as ST
    argv = sys.argv if argv is None else argv
    if len(argv) != 4:
        raise ValueError("Usage: script.py <img_fname> <solname> <funcname>")
    img_fname, solname, funcname = argv[1], argv[2], argv[3]
    folder, base = os.path.dirname(os.path.abspath(solname)), os.path.basename(solname)
    if not (base.startswith("Re") and base.endswith("ChannelVelocity")) or funcname != "Velocity":
        raise ValueError("expected <solname> = .../Re<Re>ChannelVelocity and <funcname> = Velocity")
    Re = base[len("Re"):-len("ChannelVelocity")]
    out = ST.for_and_rev_streamtrace_files(50, img_fname, Re, folder, out_dir=os.path.dirname(os.path.abspath(img_fname)))
    print(f"{len(out['final_output'])} of {len(out['rev_seeds'])} reverse seeds end inside the inner inlet contour", flush=True)
    return out
