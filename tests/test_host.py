"""Host logic and C-ABI surface (CPU, no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M


def test_box_mesh_counts_and_orientation():
    m = M.duct_mesh((4, 3, 2), 4.0)
    assert m.num_nodes == 5 * 4 * 3 and m.num_tets == 6 * 4 * 3 * 2
    X = m.points[m.tets]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6
    assert np.isclose(vol.sum(), 4.0) and vol.min() > 0
    # boundary triangles: 2 per cell face on the box surface
    assert len(m.facets) == 2 * 2 * (4 * 3 + 4 * 2 + 3 * 2)
    t = m.meta["tags"]
    assert len(m.find(t["inlet"])) == 2 * 3 * 2 and len(m.find(t["outlet"])) == 2 * 3 * 2
    assert np.allclose(m.points[m.facet_nodes(t["inlet"]), 0], 0.0)
    assert np.allclose(m.points[m.facet_nodes(t["outlet"]), 0], 4.0)


def test_jitter_keeps_boundary_and_validity():
    m0, m1 = M.duct_mesh((5, 4, 4), 2.0), M.duct_mesh((5, 4, 4), 2.0, jitter=0.2)
    bnd = np.unique(m0.facets.ravel())
    assert np.array_equal(m0.points[bnd], m1.points[bnd]) and not np.array_equal(m0.points, m1.points)
    X = m1.points[m1.tets]
    assert (np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) > 0).all()


def test_msh_roundtrip_preserves_cell_local_order(tmp_path):
    m = M.channel_mesh((3, 2, 2))
    p = str(tmp_path / "c.msh")
    M.write_msh2(m, p)
    r = M.read_msh(p, reorder=False)
    assert np.array_equal(r.tets, m.tets) and np.allclose(r.points, m.points)
    assert sorted(map(tuple, np.sort(r.facets, 1))) == sorted(map(tuple, np.sort(m.facets, 1)))
    assert set(r.facet_tags) == {1, 2, 3, 4}


def test_msh41_reader(tmp_path):
    txt = """$MeshFormat
4.1 0 8
$EndMeshFormat
$Entities
0 0 1 1
1 0 0 0 1 1 0 1 7 0
1 0 0 0 1 1 1 1 9 0
$EndEntities
$Nodes
1 4 1 4
3 1 0 4
1
2
3
4
0 0 0
1 0 0
0 1 0
0 0 1
$EndNodes
$Elements
2 2 1 2
2 1 2 1
1 1 2 3
3 1 4 1
2 2 1 3 4
$EndElements
"""
    p = tmp_path / "t.msh"
    p.write_text(txt)
    r = M.read_msh(str(p), reorder=False)
    assert r.num_tets == 1 and list(r.tets[0]) == [1, 0, 2, 3]          # file order kept
    assert list(r.facet_tags) == [7] and list(r.facets[0]) == [0, 1, 2]


def test_cavity_and_channel_bcs():
    m = M.cavity_mesh(3)
    mask, g = B.cavity_bcs(m).flatten()
    lid = m.facet_nodes(m.meta["tags"]["lid"])
    assert np.all(g[4 * lid] == 1.0) and np.all(g[4 * lid + 1] == 0.0)
    assert mask[3] == 1 and mask.reshape(-1, 4)[:, 3].sum() == 1           # p pinned at the origin only
    mc = M.channel_mesh((4, 4, 4))
    p1, p2 = B.two_stream_profiles(0.5)
    mk, gg = B.channel_bcs(mc, p1, p2).flatten()
    inl = np.union1d(mc.facet_nodes(1), mc.facet_nodes(2))
    assert np.all(gg[4 * inl + 1] == 0) and np.all(gg[4 * inl + 2] == 0) and gg[4 * inl].max() > 0


def test_header_symbols_match_library(built_lib):
    """Every function include/sns.h declares is exported by libsns.so and bound in _lib."""
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sns.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sns_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported"
    # ... and NOTHING else (-fvisibility=hidden + the version script csrc/sns_exports.map -- no C++ internals, no
    # kernel handles in the dynamic symbol table): nm -D of the library == the header's declarations
    import subprocess
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode().split("\n")
    exported = {ln.split()[-1] for ln in nm if ln.strip() and ln.split()[-2] in "TtWw"}
    assert exported == declared, (sorted(exported - declared)[:5], sorted(declared - exported)[:5])
    o = _lib.default_options()
    assert o.ksp_rtol == 1e-8 and o.snes_max_it == 30 and o.snes_rtol == 1e-8     # reference's settings :281-283


def test_struct_layouts_match_the_header(tmp_path):
    """sns_options / sns_timings are restated field by field in _lib.py (and in INTEGRATION.md's stub): compile a
    C program against include/sns.h that prints offsetof/sizeof of every field and compare with ctypes."""
    import subprocess
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sns.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(sns_options|sns_timings)\s*;", hdr_nc, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1]
            fields += [n.strip() for n in names.split(",")]
        structs[name] = fields
    assert set(structs) == {"sns_options", "sns_timings"}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "sns.h"', 'int main(void) {']
    for sname, fields in structs.items():
        src.append(f'  printf("{sname} sizeof %zu\\n", sizeof({sname}));')
        for f in fields:
            src.append(f'  printf("{sname} {f} %zu %zu\\n", offsetof({sname}, {f}), sizeof((({sname}*)0)->{f}));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    mirror = {"sns_options": _lib.SnsOptions, "sns_timings": _lib.SnsTimings}
    seen = {k: [] for k in mirror}
    for ln in out:
        t = ln.split()
        if not t:
            continue
        cls = mirror[t[0]]
        if t[1] == "sizeof":
            assert ctypes.sizeof(cls) == int(t[2]), t[0]
            continue
        fld = getattr(cls, t[1])
        assert (fld.offset, fld.size) == (int(t[2]), int(t[3])), (t[0], t[1])
        seen[t[0]].append(t[1])
    for k, cls in mirror.items():
        assert seen[k] == [f[0] for f in cls._fields_], k                      # same fields, same order
    # the ctypes stub a maintainer copies from INTEGRATION.md names the same fields in the same order
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = re.search(r"class SnsOptions\(C\.Structure\):(.*?)\n\n", integ, flags=re.S)
    assert stub, "INTEGRATION.md lost its SnsOptions stub"
    assert re.findall(r'\("([a-z0-9_]+)",', stub.group(1)) == structs["sns_options"]
    # ... and guards the ABI version of the header it was written against (the stub once refused the shipped library)
    abi_hdr = int(re.search(r"#define SNS_ABI_VERSION (\d+)", hdr).group(1))
    abi_stub = re.search(r"sns\.sns_abi_version\(\) == (\d+)", integ)
    assert abi_stub and int(abi_stub.group(1)) == abi_hdr == _lib.ABI_VERSION


def test_host_pattern_matches_scipy(built_lib):
    import scipy.sparse as sp
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    m = M.duct_mesh((5, 3, 4), 2.0)
    rp, ci, cp, cx = _lib.host_pattern(m.num_nodes, m.tets)
    rows = np.repeat(m.tets, 4, axis=1).ravel()
    cols = np.tile(m.tets, (1, 4)).ravel()
    A = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(m.num_nodes,) * 2).tocsr()
    A.sort_indices()
    assert np.array_equal(A.indptr, rp) and np.array_equal(A.indices, ci)
    # gather lists: every element block appears exactly once, under the right slot, in tet order per slot
    assert cp[-1] == 16 * m.num_tets and np.array_equal(np.sort(cx), np.arange(16 * m.num_tets))
    slot_of = np.repeat(np.arange(len(ci)), np.diff(cp))
    t, ab = cx // 16, cx % 16
    a, b = ab // 4, ab % 4
    row_of_slot = np.repeat(np.arange(m.num_nodes), np.diff(rp))
    assert np.array_equal(row_of_slot[slot_of], m.tets[t, a]) and np.array_equal(ci[slot_of], m.tets[t, b])
    assert A.data.sum() == cp[-1]
    for s in (0, len(ci) // 2, len(ci) - 1):
        assert np.all(np.diff(cx[cp[s]:cp[s + 1]] // 16) >= 0)


def test_host_aggregation_covers_and_limits(built_lib):
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    m = M.duct_mesh((6, 4, 4), 2.0)
    rp, ci, _, _ = _lib.host_pattern(m.num_nodes, m.tets)
    agg, nc = _lib.host_aggregate(rp, ci, max_agg=8)
    assert agg.min() == 0 and agg.max() == nc - 1 and nc < m.num_nodes / 3
    # aggregates are connected to their seed: every member is the seed or its neighbour's aggregate
    n_act = m.num_nodes - 10
    agg2, nc2 = _lib.host_aggregate(rp, ci, n_active=n_act, max_agg=4)
    assert np.all(agg2[n_act:] == -1) and agg2[:n_act].min() >= 0
    agg3, nc3 = _lib.host_aggregate(rp, ci, max_agg=8)
    assert np.array_equal(agg, agg3)                                           # deterministic


def test_transition_layer_between_two_cross_sections():
    """nozzle_mesh.transition_slab: the one layer of general tets between two different cross-sections conforms to both (its faces in
    either plane are that plane's triangles), fills the slab, holds no sliver between nested lattices -- and says so (ValueError)
    when a cross-section's triangles are not the Delaunay ones, which is what lets the mesher keep the old cross-section instead."""
    from stabilized_navier_stokes_flow_fenicsx_amd import nozzle_mesh as NM
    pa, ta = NM.lattice_section(8)
    pb, tb = NM.lattice_section(4)
    # every lattice triangle is a right triangle whose right angle sits at its MIDDLE vertex (what makes its prism three Kuhn cells)
    for p, t in ((pa, ta), (pb, tb)):
        v = p[np.sort(t, axis=1)]
        assert np.abs(((v[:, 0] - v[:, 1]) * (v[:, 2] - v[:, 1])).sum(axis=1)).max() < 1e-14
        e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        assert abs(0.5 * np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]).sum() - 1.0) < 1e-12
    T, q = NM.transition_slab(pa, ta, 1.0, pb, tb, 1.125, 0.125)
    P = np.zeros((len(pa) + len(pb), 3))
    P[:len(pa), 0], P[:len(pa), 1:], P[len(pa):, 0], P[len(pa):, 1:] = 1.0, pa, 1.125, pb
    vol = np.einsum("ij,ij->i", np.cross(P[T[:, 1]] - P[T[:, 0]], P[T[:, 2]] - P[T[:, 0]]), P[T[:, 3]] - P[T[:, 0]]) / 6.0
    assert vol.min() > 0 and abs(vol.sum() - 0.125) < 1e-12 and q > 0.3
    nbot = (T < len(pa)).sum(axis=1)
    assert set(np.unique(nbot)) <= {1, 2, 3}
    bottom = {tuple(sorted(t[t < len(pa)])) for t in T[nbot == 3]}
    assert bottom == {tuple(sorted(t)) for t in ta}
    # a cross-section whose diagonals are not the ones the shear selects is refused
    bad = ta.copy()
    quad = set(bad[0]) | set(bad[len(bad) // 2])                                  # the two triangles of cell (0, 0)
    shared = sorted(set(bad[0]) & set(bad[len(bad) // 2]))
    others = sorted(quad - set(shared))
    assert len(shared) == 2 and len(others) == 2
    bad[0] = [others[0], others[1], shared[0]]
    bad[len(bad) // 2] = [others[0], others[1], shared[1]]
    with pytest.raises(ValueError):
        NM.transition_slab(pa, bad, 1.0, pb, tb, 1.125, 0.125)


def test_aggregation_does_not_depend_on_the_node_numbering(built_lib):
    """Round 5 (profiles/r5_prism_vs_kuhn.txt): the greedy sweep takes the free neighbours AHEAD of its front.  On a Kuhn lattice
    numbered along the cells' common diagonal those are the seven other corners of a cube; on the same lattice numbered against it
    a sheared box three nodes wide (41 % more Krylov iterations on the GPU).  With coordinates the level's aggregation
    (sns_host_aggregate_pts) keeps the sweep when its aggregates are cube-compact and otherwise compares it with a pairwise
    aggregation (closest centroids, three rounds) and returns the more compact one."""
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    from stabilized_navier_stokes_flow_fenicsx_amd.mesh import TetMesh
    m = M.duct_mesh((31, 15, 15), 31.0 / 15.0)               # 32 x 16 x 16 nodes
    h = 1.0 / 15

    def scatter_per_node(pts, agg, nc):
        c = np.stack([np.bincount(agg, weights=pts[:, k], minlength=nc) for k in range(3)], 1) / np.bincount(agg, minlength=nc)[:, None]
        return ((pts - c[agg]) ** 2).sum() / len(pts)

    def connected(rp, ci, agg, nc):                      # every aggregate is a connected set of the node graph
        rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        same = agg[rows] == agg[ci]
        import scipy.sparse as sp
        import scipy.sparse.csgraph as cg
        g = sp.coo_matrix((np.ones(same.sum()), (rows[same], ci[same])), shape=(len(agg), len(agg)))
        ncomp, _ = cg.connected_components(g, directed=False)
        return ncomp == nc

    rp, ci, _, _ = _lib.host_pattern(m.num_nodes, m.tets)
    agg, nc, which = _lib.host_aggregate(rp, ci, max_agg=8, pts=m.points)
    agg0, nc0 = _lib.host_aggregate(rp, ci, max_agg=8)
    assert which == 0 and nc == nc0 and np.array_equal(agg, agg0)            # aligned numbering: the sweep of rounds 1-4, bit for bit
    s_aligned = scatter_per_node(m.points, agg, nc)
    assert s_aligned <= 0.8 * h * h and connected(rp, ci, agg, nc)
    # the same cells, the nodes numbered against the diagonal (y descending fastest)
    p = m.points
    order = np.lexsort((-p[:, 1].round(9), p[:, 2].round(9), p[:, 0].round(9)))
    new = np.empty(len(p), np.int64)
    new[order] = np.arange(len(p))
    m2 = TetMesh(p[order], new[m.tets].astype(np.int32), m.facets, m.facet_tags)
    rp2, ci2, _, _ = _lib.host_pattern(m2.num_nodes, m2.tets)
    g_agg, g_nc = _lib.host_aggregate(rp2, ci2, max_agg=8)
    agg2, nc2, which2 = _lib.host_aggregate(rp2, ci2, max_agg=8, pts=m2.points)
    s_greedy, s_chosen = scatter_per_node(m2.points, g_agg, g_nc), scatter_per_node(m2.points, agg2, nc2)
    print(f"  aligned: {nc} aggregates, scatter {s_aligned / h / h:.3f} h^2; against the diagonal: sweep {g_nc} aggregates, {s_greedy / h / h:.3f} h^2 -> "
          f"{'pairwise' if which2 else 'sweep'} {nc2} aggregates, {s_chosen / h / h:.3f} h^2")
    assert s_greedy * g_nc ** (2.0 / 3.0) > 1.15 * s_aligned * nc ** (2.0 / 3.0)   # what the numbering costs the sweep (at equal coarsening)
    assert which2 == 1 and agg2.min() == 0 and agg2.max() == nc2 - 1 and np.bincount(agg2).max() <= 9
    assert s_chosen * nc2 ** (2.0 / 3.0) < 0.95 * s_greedy * g_nc ** (2.0 / 3.0) and connected(rp2, ci2, agg2, nc2)
    assert m2.num_nodes / 8.6 < nc2 < m2.num_nodes / 6.0                      # octets, up to the lattice's odd planes
    again = _lib.host_aggregate(rp2, ci2, max_agg=8, pts=m2.points)
    assert np.array_equal(again[0], agg2)                                     # deterministic
    # inactive tail (ghost nodes of a partitioned level) stays out, as in the sweep
    n_act = m2.num_nodes - 40
    agg3, nc3, _ = _lib.host_aggregate(rp2, ci2, n_active=n_act, max_agg=8, pts=m2.points)
    assert np.all(agg3[n_act:] == -1) and agg3[:n_act].min() == 0 and agg3[:n_act].max() == nc3 - 1


def test_bad_mesh_is_rejected_on_host(built_lib):
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    with pytest.raises(_lib.SnsError):
        _lib.host_pattern(4, np.array([[0, 1, 2, 7]], np.int32))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: the product package must not reference it."""
    pkg = os.path.join(ROOT, "stabilized_navier_stokes_flow_fenicsx_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_cli_contracts_and_interpolation(tmp_path, monkeypatch):
    from stabilized_navier_stokes_flow_fenicsx_amd import drivers as D
    from stabilized_navier_stokes_flow_fenicsx_amd.interpolate import interpolate_initial_guess
    monkeypatch.chdir(tmp_path)
    with pytest.raises(ValueError):
        D.parse_arguments(["prog", "10"])                                  # reference raises ValueError (:82-83)
    Re, img, ratio, lc = D.parse_arguments(["prog", "10", "./InletImages/PlusF_final.png", "0.5"])
    assert (Re, ratio, lc) == (10, 0.5, 0.1) and img == str(tmp_path) + "/InletImages/PlusF_final.png"
    folder, name = D.make_output_folder(10, img, 0.04)
    assert folder.endswith("noether_data/NSChannelFlow_RE10_MeshLC004_PlusF_final") and os.path.isdir(folder)
    assert D.lc_to_cells(0.1) == (40, 10, 10)
    c, f = M.duct_mesh((6, 3, 3), 4.0, jitter=0.2), M.duct_mesh((17, 7, 5), 4.0)
    fn = lambda x: np.stack([1 + 2 * x[:, 0] - x[:, 1], 3 * x[:, 2], x[:, 0] + x[:, 1], 5 - x[:, 0]], 1)
    w = interpolate_initial_guess(c, fn(c.points).ravel(), f)
    assert np.abs(w.reshape(-1, 4) - fn(f.points)).max() < 1e-12           # P1 interpolation is exact for linears
    D.write_xdmf(str(tmp_path / "v"), c, "Velocity", fn(c.points)[:, :3])
    assert "v.h5:/Function/Velocity/0" in open(tmp_path / "v.xdmf").read()
    pts, cells, vals = D.read_xdmf_function(str(tmp_path / "v"), "Velocity")
    assert np.array_equal(vals, fn(c.points)[:, :3]) and np.array_equal(pts, c.points) and np.array_equal(cells, c.tets)
    D.write_run_metadata(folder, 10, img, 0.5, 0.04, c)
    assert open(os.path.join(folder, "RunParameters.txt")).readline() == "Re=10\n"


def test_image_inlet_pipeline(tmp_path):
    """Pixel-grid restatement of image2inlet.solve_inlet_profiles (:294-353): two regions separated by a dark
    band, Poisson profiles with unit mean scaled to ratio/area and (1-ratio)/area."""
    from PIL import Image, ImageDraw
    from stabilized_navier_stokes_flow_fenicsx_amd import inlet_image as II
    im = Image.new("RGBA", (200, 200), (255, 255, 255, 255))
    ImageDraw.Draw(im).ellipse([50, 60, 150, 140], outline=(0, 0, 0, 255), width=8)
    f = str(tmp_path / "ring.png")
    im.save(f)
    D = II.solve_inlet_profiles(f, 0.3)
    px = 1.0 / D.gray.size
    assert abs(D.u1.sum() * px - 0.3) < 1e-12 and abs(D.u2.sum() * px - 0.7) < 1e-12     # flow rates = ratio split
    assert D.area_1 < D.area_2 and abs(D.area_1 + D.area_2 + (D.region == 0).mean() - 1.0) < 1e-12
    assert D.region_at(0.0, 0.0) == 1 and D.region_at(-0.45, 0.45) == 2                   # centre inner, corner outer
    assert np.all(D.u1[D.region != 1] == 0) and np.all(D.u2[D.region != 2] == 0) and D.u1.max() > D.u2.max()
    m, (mask, g), _ = II.channel_from_image(f, 0.3, (12, 10, 10))
    t = m.meta["tags"]
    assert len(m.find(t["inlet_1"])) > 0 and len(m.find(t["inlet_2"])) > len(m.find(t["inlet_1"]))
    inl1 = m.facet_nodes(t["inlet_1"])
    assert g[4 * inl1].max() > 0 and np.all(g[4 * inl1 + 1] == 0)                         # x-component only (:150-157)
    # nozzle wall: band nodes with x <= 0.5 are no-slip, none beyond
    band = (D.region_at(m.points[:, 1], m.points[:, 2]) == 0)
    interior = (np.abs(m.points[:, 1]) < 0.49) & (np.abs(m.points[:, 2]) < 0.49)
    near, far = band & interior & (m.points[:, 0] < 0.4), band & interior & (m.points[:, 0] > 0.6) & (m.points[:, 0] < 3.9)
    assert near.any() and np.all(mask[4 * np.nonzero(near)[0]] == 1) and np.all(mask[4 * np.nonzero(far)[0]] == 0)
    with pytest.raises(ValueError):
        Image.new("L", (64, 64), 255).save(str(tmp_path / "blank.png"))
        II.solve_inlet_profiles(str(tmp_path / "blank.png"), 0.5)


def test_locality_reordering_keeps_geometry_and_local_order():
    rng = np.random.default_rng(3)
    m = M.duct_mesh((6, 4, 4), 3.0, jitter=0.2)
    shuf = rng.permutation(m.num_nodes)                       # scramble like a gmsh file
    inv = np.empty_like(shuf); inv[shuf] = np.arange(len(shuf))
    scr = M.TetMesh(m.points[shuf], inv[m.tets][rng.permutation(m.num_tets)].astype(np.int32),
                    inv[m.facets].astype(np.int32), m.facet_tags.copy())
    r, perm = M.reorder_for_locality(scr)
    assert np.allclose(r.points, scr.points[perm])
    key = lambda mm: sorted(map(tuple, np.round(mm.points[mm.tets].reshape(len(mm.tets), -1), 12)))
    assert key(r) == key(scr)                                 # same tets, same cell-local vertex order
    band = lambda mm: np.abs(mm.tets.max(axis=1) - mm.tets.min(axis=1)).mean()
    assert band(r) < 0.6 * band(scr)                           # neighbours end up closer in memory


def test_tet_face_adjacency():
    from stabilized_navier_stokes_flow_fenicsx_amd.streamtrace import make_rev_streamtrace_seeds, tet_face_neighbors
    m = M.duct_mesh((4, 3, 3), 4.0, jitter=0.1)
    nb = tet_face_neighbors(m.tets)
    assert nb.shape == (m.num_tets, 4) and (nb < 0).sum() == len(m.facets)          # boundary faces <-> -1
    for t in range(0, m.num_tets, 7):
        for a in range(4):
            u = nb[t, a]
            if u >= 0:
                assert (set(m.tets[t]) - {m.tets[t, a]}).issubset(set(m.tets[u])) and t in nb[u]
    s = make_rev_streamtrace_seeds(-0.2, 0.3, -0.1, 0.1, 5)
    assert s.shape == (25, 3) and np.all(s[:, 0] == 3.9)                             # streamtrace.py:346-355


def test_boundary_traction_force_known_answers_and_oracle():
    """Drag/lift functional (DFG_3D_Validation.py:344-367): exact for P1 fields; Couette + hydrostatic known
    answers on the duct wall, and the facet-by-facet oracle on a jittered mesh with a random field."""
    from oracle.functionals import traction_force_loops
    from stabilized_navier_stokes_flow_fenicsx_amd import functionals as Fn, mesh as M
    m = M.duct_mesh((6, 3, 3), 4.0)
    x, y, z = m.points.T
    nu = 0.37
    w = np.zeros((m.num_nodes, 4))
    w[:, 0] = y                                   # Couette u = (y,0,0): sym grad u has only xy = 1/2
    w[:, 3] = 2.5                                 # constant pressure
    t = m.meta["tags"]
    # wall = 4 faces y,z = +-0.5, each of area 4; n = -(outward):  y=+.5: n=(0,-1,0) -> traction (-nu, +p, 0)
    # y=-.5: n=(0,1,0) -> (nu, -p, 0); z faces: n=(0,0,-+1) -> (0,0,+-p): sum = 0.  Closed surface pieces cancel:
    f = Fn.boundary_traction_force(m, w.ravel(), nu, t["wall"])
    assert np.allclose(f, 0.0, atol=1e-12)
    # inlet face x = 0 (area 1): outward = (-1,0,0), n = (1,0,0): traction = (-p, nu*1, 0)
    f = Fn.boundary_traction_force(m, w.ravel(), nu, t["inlet"])
    assert np.allclose(f, [-2.5, nu, 0.0], atol=1e-12)
    cd, cl = Fn.drag_lift_coefficients(f)
    assert np.isclose(cd, 2 * -2.5 / (0.2 ** 2 * 0.041)) and np.isclose(cl, 2 * nu / (0.2 ** 2 * 0.041))
    # linear pressure p = 3 - x on the outlet x = 4: n = (-1,0,0): traction_x = +p = -1
    w2 = np.zeros((m.num_nodes, 4)); w2[:, 3] = 3.0 - x
    assert np.allclose(Fn.boundary_traction_force(m, w2.ravel(), nu, t["outlet"]), [-1.0, 0, 0], atol=1e-12)
    mj = M.duct_mesh((4, 2, 3), 4.0, jitter=0.2)
    wr = np.random.default_rng(5).normal(size=mj.num_dofs)
    for tag in (t["wall"], t["inlet"]):
        ref = traction_force_loops(mj.points, mj.tets, mj.facets, mj.find(tag), wr, nu)
        assert np.allclose(Fn.boundary_traction_force(mj, wr, nu, tag), ref, rtol=1e-12, atol=1e-12)


def test_bench_helpers_without_gpu():
    """bench.py pieces that need no GPU: the PMC traffic figure comes from the committed profiles (latest round),
    defaults finish within minutes (K=4, W=1, N=1), and the module never touches the oracle at import time."""
    import importlib, sys
    sys.modules.pop("bench", None)
    before = {m for m in sys.modules if m == "oracle" or m.startswith("oracle.")}
    bench = importlib.import_module("bench")
    after = {m for m in sys.modules if m == "oracle" or m.startswith("oracle.")}
    assert after == before                                   # cpu_baseline imports the oracle lazily, only when it runs
    t, src = bench.pmc_traffic("k_spmv_f32<2, 1")
    assert t is not None and 2.0e9 < t < 3.0e9               # 2.14 GB algorithmic, ~2.4 GB measured
    assert src.startswith("profiles/") and "pmc" in src      # the line names the committed CSVs the figure is read from
    t2, _ = bench.pmc_traffic("k_spmv_lp<2, 1, 0, 2")       # round 2: fp16 Jacobi sweep, 1.24 GB algorithmic
    assert t2 is not None and 1.2e9 < t2 < 1.8e9
    t3, _ = bench.pmc_traffic("k_spmv<0, 1, 1, 0>")          # fp64 Krylov operator: 3.51 GB algorithmic
    assert t3 is not None and 3.4e9 < t3 < 3.7e9
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
    assert bench.HBM_PEAK_GBS == 8000.0


def test_host_symbolic_setup_under_sanitizers(tmp_path):
    """The host-side setup code (BSR pattern, gather lists, aggregation, coarse patterns: sns_host.cpp) built with
    AddressSanitizer + UBSan and run on a Kuhn box with shuffled cell-local vertex orders (GPU sanitizers are not
    available on the pool, so the CPU build is where memory errors of the symbolic phase would show)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sanitize_host")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-I", os.path.join(root, "include"),
                            "-I", os.path.join(root, "stabilized_navier_stokes_flow_fenicsx_amd", "csrc"),
                            os.path.join(root, "tests", "sanitize_host_main.cpp"),
                            os.path.join(root, "stabilized_navier_stokes_flow_fenicsx_amd", "csrc", "sns_host.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True,
                         env=dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0, run.stderr[-2000:]
    assert "active aggregation" in run.stdout and "ERROR" not in run.stderr


def test_delaunay_channel_mesh_geometry_and_tags():
    """mesh.delaunay_channel_mesh (bench.py --config 4u; the unstructured stand-in for the reference's gmsh meshes,
    image2gmsh3D.py:445-486): watertight 4 x 1 x 1 box, channel tag set with the inlet split by facet centroid, node
    ids x-slowest (x-slabs stay contiguous id ranges), variable valence, no degenerate tets; both lattices."""
    for lattice, qmin in (("bcc", 0.3), ("cubic", 1e-6)):
        m = M.delaunay_channel_mesh(6, lattice=lattice, seed=1)
        X4 = m.points[m.tets]
        vol = np.abs(np.linalg.det(np.stack([X4[:, 1] - X4[:, 0], X4[:, 2] - X4[:, 0], X4[:, 3] - X4[:, 0]], axis=2))) / 6
        assert abs(vol.sum() - 4.0) < 1e-9 and vol.min() > 0
        e = np.stack([np.linalg.norm(X4[:, a] - X4[:, b], axis=1) for a in range(4) for b in range(a + 1, 4)], axis=1)
        q = 6 * np.sqrt(2) * vol / np.sqrt((e ** 2).mean(axis=1)) ** 3                   # 1 for a regular tet
        assert np.quantile(q, 0.01) > qmin
        t = m.meta["tags"]
        assert set(np.unique(m.facet_tags)) == {t["inlet_1"], t["inlet_2"], t["outlet"], t["wall"]}
        fa = 0.5 * np.linalg.norm(np.cross(m.points[m.facets[:, 1]] - m.points[m.facets[:, 0]],
                                           m.points[m.facets[:, 2]] - m.points[m.facets[:, 0]]), axis=1)
        assert abs(fa[m.facet_tags == t["outlet"]].sum() - 1.0) < 1e-9
        assert abs(fa[(m.facet_tags == t["inlet_1"]) | (m.facet_tags == t["inlet_2"])].sum() - 1.0) < 1e-9
        assert abs(fa[m.facet_tags == t["wall"]].sum() - 16.0) < 1e-9
        assert 0.1 < fa[m.facet_tags == t["inlet_1"]].sum() < 0.5                         # the centred 0.5 x 0.5 square, by centroid
        x = m.points[:, 0]
        assert np.all(np.diff(np.round(x / (0.5 / 6))) >= 0)                              # x-slowest node ids
        deg = np.bincount(m.tets.ravel(), minlength=m.num_nodes)
        assert deg.min() >= 1 and deg.max() > 2 * deg.min()
        mask, g = B.channel_bcs(m, *B.two_stream_profiles(0.5)).flatten()
        assert mask.reshape(-1, 4)[:, 3].sum() == len(m.facet_nodes(t["outlet"]))         # p = 0 on the outlet only


def test_dfg_pillar_mesh_geometry():
    """Delaunay mesh of the DFG pillar channel (dfg_pillar_3D.geo): watertight, fluid volume and pillar surface
    area as the geometry says (the pillar is a polygonal prism: slightly larger volume, smaller area), tags and
    boundary conditions of DFG_3D_Validation.py:100-141 (no pressure condition)."""
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
    m = M.dfg_pillar_mesh(12)
    X4 = m.points[m.tets]
    vol = np.abs(np.linalg.det(np.stack([X4[:, 1] - X4[:, 0], X4[:, 2] - X4[:, 0], X4[:, 3] - X4[:, 0]], axis=2))).sum() / 6
    exact = 2.2 * 0.41 * 0.41 - np.pi * 0.05 ** 2 * 0.41
    assert exact < vol < exact * 1.001
    t = m.meta["tags"]
    Pf = m.points[m.facets[m.facet_tags == t["obstacle"]]]
    area = 0.5 * np.linalg.norm(np.cross(Pf[:, 1] - Pf[:, 0], Pf[:, 2] - Pf[:, 0]), axis=1).sum()
    assert 0.98 * 2 * np.pi * 0.05 * 0.41 < area < 2 * np.pi * 0.05 * 0.41
    assert set(np.unique(m.facet_tags)) == {t["inlet"], t["outlet"], t["wall"], t["obstacle"]}
    mask, g = B.dfg_bcs(m).flatten()
    assert mask.reshape(-1, 4)[:, 3].sum() == 0                      # no pressure Dirichlet dofs
    assert abs(g.max() - 0.45) < 5e-3 and g.min() == 0.0             # inlet peak u_max, no-slip elsewhere
    on_pillar = m.facet_nodes(t["obstacle"])
    assert np.allclose(np.hypot(m.points[on_pillar, 0] - 0.5, m.points[on_pillar, 1] - 0.2), 0.05)


def test_xdmf_output_is_the_hdf5_container_the_reference_reads(tmp_path):
    """save_navier_stokes_solution (:316-346) writes <name>.xdmf + <name>.h5; the reference's own consumer opens the
    .h5 with h5py and reads h5f["Function"][name]["0"] (streamtrace.py:87-96).  An independent minimal reader
    (tests/h5read_min.py: superblock -> symbol tables -> object headers) recovers every dataset BIT-EXACTLY;
    libhdf5 itself is a second reader where it happens to be installed."""
    import h5read_min as R
    from stabilized_navier_stokes_flow_fenicsx_amd import drivers as D, mesh2d as M2
    m = M.duct_mesh((5, 3, 2), 2.0, jitter=0.1)
    rng = np.random.default_rng(4)
    u, p = rng.normal(size=(m.num_nodes, 3)), rng.normal(size=m.num_nodes)
    D.save_navier_stokes_solution(u, p, m, str(tmp_path), 7)
    f = R.H5File(str(tmp_path / "Re7ChannelVelocity.h5"))
    assert f.keys() == ["Function", "Mesh"] and f["Function"].keys() == ["Velocity"]
    data = f["Function"]["Velocity"]["0"]                                     # streamtrace.py:96
    assert data.dtype == np.float64 and data.shape == (m.num_nodes, 3) and np.array_equal(data, u)
    assert np.array_equal(f["Mesh"]["mesh"]["geometry"], m.points)
    topo = f["Mesh/mesh/topology"]
    assert topo.dtype == np.int64 and np.array_equal(topo, m.tets)
    pr = R.H5File(str(tmp_path / "Re7ChannelPressure.h5"))["Function"]["Pressure"]["0"]
    assert pr.shape == (m.num_nodes, 1) and np.array_equal(pr[:, 0], p)
    x = open(tmp_path / "Re7ChannelVelocity.xdmf").read()
    assert 'Format="HDF">Re7ChannelVelocity.h5:/Function/Velocity/0' in x and 'TopologyType="Tetrahedron"' in x
    lib = R.libhdf5_read(str(tmp_path / "Re7ChannelVelocity.h5"), "/Function/Velocity/0", (m.num_nodes, 3))
    if lib is not None:
        assert np.array_equal(lib, u)
        assert np.array_equal(R.libhdf5_read(str(tmp_path / "Re7ChannelVelocity.h5"), "/Mesh/mesh/topology",
                                             (m.num_tets, 4), "i8"), m.tets)
    # 2-D meshes (lid-driven / DFG-2D scripts) and groups with many children
    m2 = M2.rectangle_mesh(3)
    v2 = rng.normal(size=(m2.num_nodes, 2))
    D.write_xdmf(str(tmp_path / "c"), m2, "Velocity", v2)
    assert R.H5File(str(tmp_path / "c.h5"))["Mesh/mesh/topology"].shape == (18, 3)
    # a 2-D vector is padded to three components (uz = 0), as dolfinx' XDMF write_function does
    f2 = R.H5File(str(tmp_path / "c.h5"))["Function"]["Velocity"]["0"]
    assert f2.shape == (m2.num_nodes, 3) and np.array_equal(f2[:, :2], v2) and np.all(f2[:, 2] == 0.0)
    assert f'Dimensions="{m2.num_nodes} 3"' in open(tmp_path / "c.xdmf").read()
    from stabilized_navier_stokes_flow_fenicsx_amd.h5lite import H5Writer, read_datasets
    w = H5Writer()
    for i in range(37):
        w.dataset(f"/G/d{i:02d}", np.arange(i + 1, dtype=np.int32))
    w.dataset("/empty", np.zeros((0, 3)))
    w.write(str(tmp_path / "many.h5"))
    g = R.H5File(str(tmp_path / "many.h5"))
    assert len(g["G"].keys()) == 37 and g["G"]["d36"][-1] == 36 and g["empty"].shape == (0, 3)
    back = read_datasets(str(tmp_path / "many.h5"))
    assert len(back) == 38 and back["/G/d05"].tolist() == list(range(6))


@pytest.mark.parametrize("name", ["PlusF_final", "asym_offset", "Triangle"])
def test_inlet_contour_pipeline_on_reference_images(name):
    """image2inlet.py:58-139,240-353 on the reference's own inlet images (tests/golden/inlet_*.png: box-filtered
    copies of NavierStokes/InletImages/*.png, oracle/make_inlet_fixtures.py): marching squares at 0.5 finds exactly
    the two edges of the nozzle wall, both pass the 5 % area filter; FFT low-pass + RDP leave small polygons; the
    two P1 Poisson profiles carry the flow rates ratio and 1 - ratio; the pixel-grid solver bounds the result."""
    from stabilized_navier_stokes_flow_fenicsx_amd import inlet_contours as IC, inlet_image as II
    f = os.path.join(ROOT, "tests", "golden", f"inlet_{name}.png")
    gray = IC.load_image(f)
    raw = IC.find_contours(gray, 0.5)
    cs = IC.get_contours(gray)
    assert len(raw) == 2 and len(cs) == 2
    for c in raw:
        assert np.allclose(c[0], c[-1]) and len(c) > 500                       # closed, one point per crossed edge
        on_grid = np.isclose(c % 1.0, 0.0) | np.isclose(c % 1.0, 1.0)
        assert np.all(on_grid.any(axis=1))                                      # every point lies on a grid edge
    assert np.abs(cs[0]).max() <= 0.5 and np.abs(cs[1]).max() < np.abs(cs[0]).max()   # contours[0] is the outer one
    ratio = 0.3
    P = IC.solve_inlet_profiles(f, ratio)
    assert 5 <= len(P.contour_inner) <= 200 and 5 <= len(P.contour_outer) <= 200
    assert P.mesh_lc[0] == pytest.approx(0.05 * min(np.ptp(P.contour_inner[:, 0]), np.ptp(P.contour_inner[:, 1])))
    band = 1.0 - P.area_1 - P.area_2
    assert 0.05 < band < 0.3 and P.area_1 < P.area_2
    assert P.inner.integral() == pytest.approx(ratio, rel=1e-12) and P.outer.integral() == pytest.approx(1 - ratio, rel=1e-12)
    # sanity bound: the older pixel-grid restatement (connected components + 5-point Poisson)
    D = II.solve_inlet_profiles(f, ratio)
    assert abs(P.area_1 - D.area_1) < 0.01 * D.area_1 + 2e-3 and abs(P.area_2 - D.area_2) < 0.01 * D.area_2
    assert abs(P.inner.u.max() - D.u1.max()) < 0.05 * D.u1.max() and abs(P.outer.u.max() - D.u2.max()) < 0.06 * D.u2.max()
    # evaluation: zero in the band and outside its own region, region map consistent with the pixel one
    rng = np.random.default_rng(0)
    y, z = rng.uniform(-0.49, 0.49, 400), rng.uniform(-0.49, 0.49, 400)
    reg = P.region_at(y, z)
    x = np.stack([np.zeros(400), y, z], axis=1)
    assert np.all(P.profile_1(x)[reg != 1] == 0) and np.all(P.profile_2(x)[reg != 2] == 0)
    assert (P.profile_1(x)[reg == 1] > 0).mean() > 0.95
    assert (reg == D.region_at(y, z)).mean() > 0.97
    m, (mask, g), data = II.channel_from_image(f, ratio, (12, 10, 10))
    t = m.meta["tags"]
    assert len(m.find(t["inlet_1"])) > 0 and len(m.find(t["inlet_2"])) > len(m.find(t["inlet_1"]))
    assert g[0::4].max() > 1.0 and np.all(g[1::4] == 0)


def test_marching_squares_rdp_and_fft_known_answers():
    from stabilized_navier_stokes_flow_fenicsx_amd import inlet_contours as IC
    # a disc of radius 0.3 sampled on a 200 x 200 image: one closed contour, radius recovered to sub-pixel accuracy
    n = 200
    yy, xx = np.mgrid[0:n, 0:n]
    r = np.hypot(xx - 99.5, yy - 99.5)
    img = np.clip((0.3 * n - r) / 2.0 + 0.5, 0.0, 1.0)                           # linear ramp across the edge
    cs = IC.find_contours(img, 0.5)
    assert len(cs) == 1 and np.allclose(cs[0][0], cs[0][-1])
    rad = np.hypot(cs[0][:, 0] - 99.5, cs[0][:, 1] - 99.5)
    assert np.abs(rad - 0.3 * n).max() < 0.02
    # saddle cell: the two LOW corners are connected (two separate high blobs)
    s = np.zeros((4, 4)); s[1, 1] = s[2, 2] = 1.0
    assert len(IC.find_contours(s, 0.5)) == 2
    # RDP keeps the corners of a densely sampled square, nothing else
    sq = np.concatenate([np.linspace([0, 0], [1, 0], 50, endpoint=False), np.linspace([1, 0], [1, 1], 50, endpoint=False),
                         np.linspace([1, 1], [0, 1], 50, endpoint=False), np.linspace([0, 1], [0, 0], 51)])
    out = IC.rdp(sq, 1e-6)
    assert len(out) == 5 and np.allclose(out[0], out[-1])
    # the low-pass keeps a circle (frequency 1/N) and removes a ripple above the cutoff
    N = 400
    t = 2 * np.pi * np.arange(N + 1) / N
    c = np.stack([0.3 * np.sin(t) + 0.01 * np.sin(80 * t), 0.3 * np.cos(t) + 0.01 * np.cos(80 * t)], axis=1)
    poly, lc = IC.optimize_contour(c, cutoff=0.12, epsilon=0.0005)
    assert np.abs(np.hypot(poly[:, 0], poly[:, 1]) - 0.3).max() < 5e-3 and lc == pytest.approx(0.05 * 0.6, rel=0.02)
    # P1 Poisson on the unit disc: u = p (R^2 - r^2) / 4, mean = p R^2 / 8
    ang = 2 * np.pi * np.arange(120) / 120
    disc = 0.4 * np.stack([np.cos(ang), np.sin(ang)], axis=1)
    pts, tris, bnd = IC.mesh_region(disc, None, 0.03)
    u, area, avg = IC.solve_velocity_field(pts, tris, bnd)
    assert area == pytest.approx(np.pi * 0.16, rel=2e-3) and avg == pytest.approx(10 * 0.16 / 8, rel=0.02)
    assert u.max() == pytest.approx(10 * 0.16 / 4, rel=0.02)


def test_streamtrace_postprocessing_alpha_shape_blur_and_contour_filter():
    """expand_streamtace / find_seed_end / update_contour of NavierStokes/streamtrace.py:132-143,292-343,536-553."""
    from stabilized_navier_stokes_flow_fenicsx_amd import streamtrace as ST
    rng = np.random.default_rng(3)
    # alpha = 0.2 on a unit-scale cloud: all but a few flat hull slivers have circumradius < 5 -> ~ the convex hull
    p = rng.uniform(-0.2, 0.3, size=(400, 2))
    ring = ST.alpha_shape_exterior(p, 0.2)
    from scipy.spatial import ConvexHull
    hull = ConvexHull(p)
    ring_area = 0.5 * abs(np.sum(ring[:-1, 0] * ring[1:, 1] - ring[1:, 0] * ring[:-1, 1]))
    assert np.allclose(ring[0], ring[-1]) and len(ring) - 1 >= len(hull.vertices)
    assert ring_area == pytest.approx(hull.volume, rel=5e-3) and np.allclose(ring.min(0), p.min(0)) and np.allclose(ring.max(0), p.max(0))
    # a large alpha carves the concave corner out of an L-shaped cloud
    g = np.stack(np.meshgrid(np.linspace(0, 1, 21), np.linspace(0, 1, 21)), -1).reshape(-1, 2)
    L = g[~((g[:, 0] > 0.5) & (g[:, 1] > 0.5))]
    r20 = ST.alpha_shape_exterior(L, 20.0 / 1.0)                    # circumradius < 0.05: only the small lattice cells
    area = 0.5 * abs(np.sum(r20[:-1, 0] * r20[1:, 1] - r20[1:, 0] * r20[:-1, 1]))
    assert area == pytest.approx(0.75, abs=0.03)
    # blur: straddling zero -> both ends pushed outwards by 20 %
    lo_y, hi_y, lo_z, hi_z = ST.expand_streamtace(p[:, 0], p[:, 1])
    assert lo_y == pytest.approx(1.2 * p[:, 0].min()) and hi_y == pytest.approx(1.2 * p[:, 0].max())
    assert lo_z == pytest.approx(1.2 * p[:, 1].min()) and hi_z == pytest.approx(1.2 * p[:, 1].max())
    # not straddling zero: min * (1 - 0.2), max * (1 + 0.2), literally as the reference computes it (:316-321)
    q = rng.uniform(0.1, 0.3, size=(300, 2))
    a0, a1, b0, b1 = ST.expand_streamtace(q[:, 0], -q[:, 1])
    assert a0 == pytest.approx(0.8 * q[:, 0].min()) and a1 == pytest.approx(1.2 * q[:, 0].max())
    zs = np.sort(-q[:, 1])
    # all-negative axis: both formulas move the touched vertex TOWARDS zero, so the next vertices take over
    assert zs[0] < b0 <= 0.8 * zs[0] + 1e-12 and 1.2 * zs[-1] - 1e-12 <= b1 < zs[-1]
    # reverse-trace filter: seeds whose end point lies inside the inner contour
    contour = np.array([[0, -0.2, -0.2], [0, 0.2, -0.2], [0, 0.2, 0.2], [0, -0.2, 0.2]], dtype=float)
    seeds = ST.make_rev_streamtrace_seeds(-0.4, 0.4, -0.4, 0.4, 5)
    ends_y, ends_z = seeds[:, 1] * 0.9, seeds[:, 2] * 0.9
    out = ST.find_seed_end(ends_y, ends_z, seeds, contour)
    assert out.shape == (9, 2) and np.abs(out).max() <= 0.2 + 1e-12
    c = ST.update_contour(os.path.join(ROOT, "tests", "golden", "inlet_PlusF_final.png"))
    assert c.shape[1] == 3 and np.all(c[:, 0] == 0) and 5 < len(c) < 200 and np.abs(c[:, 1:]).max() < 0.5


def test_extruded_slab_mesh_is_conforming_and_carries_the_2d_tags():
    """mesh.extrude_tri_mesh / mesh2d.dfg2d_slab_problem: 3 tets per triangle filling the slab exactly, every interior face
    shared by two tets, the boundary faces are exactly the listed facets (extruded boundary edges with their 2-D tags +
    the two z planes), every node constrained in z, the inlet profile of DFG_2D_Validation.py:52."""
    from stabilized_navier_stokes_flow_fenicsx_amd import functionals as Fn, mesh2d as M2
    m3, (mask, g), thick = M2.dfg2d_slab_problem(1)
    m2 = M2.dfg_2d_mesh(1)
    assert m3.num_nodes == 2 * m2.num_nodes and m3.num_tets == 3 * m2.num_cells
    X = m3.points[m3.tets]
    vol = np.abs(np.linalg.det(np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2))) / 6
    e = m2.points[m2.tris]
    a, b = e[:, 1] - e[:, 0], e[:, 2] - e[:, 0]
    area = 0.5 * np.abs(a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0])
    assert abs(vol.sum() - thick * area.sum()) < 1e-12 and vol.min() > 0.2 * thick * area.min() / 3
    t = m3.tets.astype(np.int64)
    faces = np.sort(np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]]), axis=1)
    u, c = np.unique(faces, axis=0, return_counts=True)
    assert set(np.unique(c)) == {1, 2}
    bnd = {tuple(f) for f in u[c == 1].tolist()}
    assert len(bnd) == len(m3.facets) and all(tuple(sorted(f)) in bnd for f in m3.facets.tolist())
    tg = m3.meta["tags"]
    for name in ("inlet", "outlet", "walls", "obstacle"):
        assert len(m3.find(tg[name])) == 2 * len(m2.find(tg[name]))
    assert len(m3.find(tg["zmin"])) == m2.num_cells == len(m3.find(tg["zmax"]))
    assert len(Fn.facet_parent_tets(m3, m3.find(tg["obstacle"]))) == 2 * len(m2.find(tg["obstacle"]))
    assert mask[2::4].all() and not mask[3::4].any()
    nd = m3.facet_nodes(tg["inlet"])
    y = m3.points[nd, 1]
    assert np.allclose(g[4 * nd], 4 * 0.3 * y * (0.41 - y) / 0.41 ** 2) and g[4 * nd].max() > 0.29
    out_only = np.setdiff1d(m3.facet_nodes(tg["outlet"]), m3.facet_nodes(tg["walls"]))
    assert not mask[4 * out_only].any() and not mask[4 * out_only + 1].any()       # natural outlet


def test_hessenberg_eigenvalues_match_numpy():
    """sns_host_hessenberg_eigs (shifted complex QR on a small upper-Hessenberg matrix): the Ritz values behind the damping limit
    of amg_ritz_limit.  Random matrices of every size the Arnoldi process produces, incl. reducible ones; vs numpy.linalg.eigvals."""
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 8, 10, 17, 32):
        for trial in range(10):
            H = np.triu(rng.normal(size=(n, n)), -1)
            if trial % 3 == 0 and n > 2:
                H[n // 2, n // 2 - 1] = 0.0
            re, im = np.zeros(n), np.zeros(n)
            Hc = np.ascontiguousarray(H)
            assert lib.sns_host_hessenberg_eigs(n, Hc.ctypes.data, re.ctypes.data, im.ctypes.data) == 0
            ref = np.linalg.eigvals(H)
            ev = re + 1j * im
            assert max(np.min(np.abs(ref - e)) for e in ev) < 1e-8 * max(1.0, np.abs(ref).max())
            assert max(np.min(np.abs(ev - e)) for e in ref) < 1e-8 * max(1.0, np.abs(ref).max())
    assert lib.sns_host_hessenberg_eigs(0, None, None, None) != 0


@pytest.mark.parametrize("name", ["inlet_PlusF_final.png", "inlet_Triangle.png"])
def test_bodyfitted_nozzle_channel_geometry(name):
    """Row f2 at full fidelity: the channel of image2gmsh3D.py:164-486 -- box 4 x 1 x 1 minus the nozzle wall
    (the band between the two contours of the inlet image) extruded over x in [0, 0.5] (:193-194), tags inlet_1 / inlet_2 /
    outlet / wall = 1-4 (:435-438), node planes after the three Box fields (:445-483) -- as nozzle_mesh.py builds it: the mesh
    is conforming (every interior face shared by two tets), its volume is the box minus the extruded band, the tagged surfaces have
    the areas of the geometry (inlets = the two regions, wall = duct walls + both nozzle surfaces + the band's end face), no node
    lies inside the wall, and the Dirichlet data carry the flow-rate split."""
    from stabilized_navier_stokes_flow_fenicsx_amd import nozzle_mesh as NM
    from stabilized_navier_stokes_flow_fenicsx_amd.inlet_contours import points_in_polygon
    lc = 0.06
    m, (mask, g), data = NM.channel_from_image_bodyfitted(os.path.join(ROOT, "tests", "golden", name), 0.4, lc)
    t = m.meta["tags"]
    assert t == {"inlet_1": 1, "inlet_2": 2, "outlet": 3, "wall": 4}
    X = m.points[m.tets]
    vol = np.abs(np.linalg.det(np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2))) / 6.0
    band = 1.0 - data.area_1 - data.area_2
    assert vol.min() > 1e-9 and abs(vol.sum() - (4.0 - 0.5 * band)) < 2e-3              # (the chains cut the polygon's corners by O(h^2))
    assert len(np.unique(m.tets)) == m.num_nodes                                          # every node belongs to a tet
    # conformity: faces shared by exactly two tets, or boundary facets -- _boundary_facets found every boundary face, and their
    # areas add up to the surface of the geometry
    def area(f):
        P = m.points[f]
        return 0.5 * np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1)
    A = {k: float(area(m.facets[m.facet_tags == v]).sum()) for k, v in t.items()}
    ci, co = data.contour_inner[:, ::-1], data.contour_outer[:, ::-1]
    per = lambda c: float(np.linalg.norm(np.roll(c, -1, axis=0) - c, axis=1).sum())
    assert abs(A["inlet_1"] - data.area_1) < 3e-3 and abs(A["inlet_2"] - data.area_2) < 3e-3 and abs(A["outlet"] - 1.0) < 1e-12
    wall_expected = 4 * 4.0 * 1.0 + 0.5 * (per(ci) + per(co)) + band
    assert abs(A["wall"] - wall_expected) < 0.02 * wall_expected, (A["wall"], wall_expected)
    # nothing inside the wall upstream of the lip; planes at x = 0, 0.5, 4 and the size fields of :445-483 along x
    up = m.points[m.points[:, 0] < 0.5 - 1e-9]
    yz = up[:, 1:]
    inside_band = points_in_polygon(yz, co) & ~points_in_polygon(yz, ci)
    from scipy.spatial import cKDTree
    d = cKDTree(np.concatenate([NM.resample_contour(ci, 0.005), NM.resample_contour(co, 0.005)])).query(yz[inside_band])[0] if inside_band.any() else np.zeros(0)
    assert (d < 0.02).all()                                                               # (only nodes ON the surfaces, up to the corner cuts)
    xs = np.unique(np.round(m.points[:, 0], 12))
    assert xs[0] == 0.0 and xs[-1] == 4.0 and np.any(np.abs(xs - 0.5) < 1e-12)
    dx = np.diff(xs)
    mid = 0.5 * (xs[1:] + xs[:-1])
    # (each stretch is rescaled to end on its plane: up to +30 % on the handful of planes upstream of the lip)
    assert dx[mid < 0.25].max() < 1.0 * lc and dx[(mid > 0.4) & (mid < 0.6)].max() < 0.5 * lc        # 0.75 lc, 0.375 lc
    assert dx[(mid > 0.75) & (mid < 1.0)].max() < 0.65 * lc and 1.5 * lc < dx[mid > 2.5].max() <= 2.2 * lc   # lc / 2, 2 lc
    # behind the lip the cross-section changes twice -- contour-free lattice, then the lattice of twice the spacing where the planes
    # lie far apart --, each time through one layer of general tets that conforms on both sides (the volume and area sums above
    # would show a hole or an overlap) and holds no sliver
    tp, tx = m.meta["transition_planes"], m.meta["transition_x"]
    cs = m.meta["cross_sections"]
    assert len(tp) == 2 and 0.5 + 0.15 - 1e-9 <= tx[0] < 0.5 + 0.15 + lc and tx[1] > tx[0] and len(cs) == 3
    assert cs[2]["nodes"] < 0.3 * cs[1]["nodes"] and abs(cs[1]["nodes"] - cs[0]["nodes"]) < 0.1 * cs[0]["nodes"]
    assert min(m.meta["transition_smallest_cell"]) > 1e-3
    npl = np.array([np.sum(np.abs(m.points[:, 0] - x) < 1e-12) for x in xs])
    assert npl[-1] == cs[2]["nodes"] and npl[np.searchsorted(xs, tx[0]) + 1] == cs[1]["nodes"]
    far = m.points[:, 0] > tx[1] + 2.5 * lc                                           # far field: a lattice, every prism a Kuhn cell
    tets_far = m.tets[far[m.tets].all(axis=1)]
    P4 = m.points[tets_far]
    import itertools
    worst = np.zeros(len(tets_far))
    for (i, j) in itertools.combinations(range(4), 2):
        k, l = [q for q in range(4) if q not in (i, j)]
        e = P4[:, j] - P4[:, i]
        e /= np.linalg.norm(e, axis=1)[:, None]
        a_ = P4[:, k] - P4[:, i]
        a_ -= (a_ * e).sum(1)[:, None] * e
        b_ = P4[:, l] - P4[:, i]
        b_ -= (b_ * e).sum(1)[:, None] * e
        worst = np.maximum(worst, np.degrees(np.arccos(np.clip((a_ * b_).sum(1) / np.linalg.norm(a_, axis=1) / np.linalg.norm(b_, axis=1), -1, 1))))
    assert len(tets_far) > 1000 and worst.max() < 90.0 + 1e-6                        # no obtuse dihedral angle in the far field
    # Dirichlet sets in the reference's order [wall, inlet_1, inlet_2, outlet]; the inlet data carry ratio : 1 - ratio
    q1, q2, _ = NM.inlet_fluxes(m, g)
    # (lc = 0.06, cross-section size 0.045: the P1 interpolant of the Poisson profiles on a few hundred inlet triangles, as coarse as
    # the reference's own non-matching interpolation at that size; 1 % at lc = 0.035, tests/test_gpu_parity.py)
    assert abs(q1 / 0.4 - 1.0) < 0.12 and abs(q2 / 0.6 - 1.0) < 0.12 and abs(q1 / q2 / (0.4 / 0.6) - 1.0) < 0.06, (q1, q2)
    wall = m.facet_nodes(t["wall"])
    G, Mk = g.reshape(-1, 4), mask.reshape(-1, 4)
    only_wall = np.setdiff1d(wall, np.union1d(m.facet_nodes(t["inlet_1"]), m.facet_nodes(t["inlet_2"])))
    assert np.all(Mk[only_wall, :3] == 1) and np.all(G[only_wall, :3] == 0.0)
    out = m.facet_nodes(t["outlet"])
    assert np.all(Mk[out, 3] == 1) and np.all(G[out, 3] == 0.0)
    # the driver takes this mesh for an image argument (and the structured box on request)
    from stabilized_navier_stokes_flow_fenicsx_amd import drivers as D
    msh, _ = D.channel_problem_inputs(os.path.join(ROOT, "tests", "golden", name), 0.4, 0.2 if name.startswith("inlet_Tri") else 0.15)
    assert msh.meta["kind"] == "channel-nozzle"


def test_hierarchy_policy_table(built_lib):
    """the size thresholds and sweep schedules of the AMG hierarchy live in ONE host function
    (csrc/sns_policy.h, exported as sns_host_cycle_policy) -- here its table for the hierarchies on record:
    (kind, sweeps before, sweeps after the coarse-grid correction) per level; kind 0 nodal / 1 aggregate blocks / 2 one-workgroup
    inverse / 3 blocked Gauss-Jordan inverse / 4 sweeps only / -1 the source of the replicated copy."""
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    T = lambda rows, **kw: [(r["kind"], r["pre"], r["post"]) for r in _lib.host_cycle_policy(rows, **kw)]
    # the 10.1 M-tet duct on one GPU (profiles/r4_strong_rehearsal_team.txt, N = 1): nodal fine level, level 1 at 1 + 3 block
    # sweeps, level 2 at 4 + 4, level 3 at 2 + 2, the 475-row level solved by the blocked inverse
    serial = [1738576, 218044, 27436, 3800, 475]
    assert T(serial) == [(0, 1, 1), (1, 1, 3), (1, 4, 4), (1, 2, 2), (3, 0, 0)]
    # its 8-way strong split over RCCL (rank-local sweeps): blocks on the fine level too (217 k rows per rank <= 600 k), level 1 at
    # 4 + 4, level 2 only the source of the replicated tail, which runs 4 + 4 / 2 + 2 / 2 + 2 and ends in the 136-row inverse
    part = [1738576, 218044, 28880, 30027, 4454, 721, 136]
    rccl = T(part, nranks=8, rep_level=3, rows_global_l1=218044)
    assert rccl == [(1, 1, 1), (1, 4, 4), (-1, 0, 0), (1, 4, 4), (1, 2, 2), (1, 2, 2), (3, 0, 0)]
    # ... over a window transport with amg_exact_sweeps (round 5): level 1 runs the single-GPU schedule with exact sweeps
    win = _lib.host_cycle_policy(part, nranks=8, windows=True, rep_level=3, rows_global_l1=218044)
    assert [(r["kind"], r["pre"], r["post"]) for r in win] == [(1, 1, 1), (1, 1, 4), (-1, 0, 0), (1, 4, 4), (1, 2, 2), (1, 2, 2), (3, 0, 0)]
    assert [r["exact"] for r in win] == [0, 1, 0, 0, 0, 0, 0]
    assert T(part, nranks=8, windows=True, rep_level=3, rows_global_l1=218044, amg_exact_sweeps=0) == rccl
    # 2 ranks: 869 k rows per rank keep the nodal fine level
    assert T([1738576, 218044, 27436, 29470, 4193, 597, 110], nranks=2, rep_level=3, rows_global_l1=218044)[0] == (0, 1, 1)
    # large problems get more sweeps below level 1 (amg_nu_scale_with_size): 81 M tets (13.9 M rows): + (4, 6), halved for blocks
    big = [13900000, 1740000, 218000, 27400, 3800, 475]
    assert T(big) == [(0, 1, 1), (1, 1, 3), (1, 6, 6), (1, 5, 5), (1, 5, 5), (3, 0, 0)]
    assert T(big, amg_nu_scale_with_size=0) == [(0, 1, 1), (1, 1, 3), (1, 4, 4), (1, 2, 2), (1, 2, 2), (3, 0, 0)]
    # an unstructured mesh (first coarsening keeps more than one row in six) with a deep hierarchy enters the first tier early
    unstr = [850748, 186650, 29289, 4726, 779, 136]
    assert T(unstr)[2] == (1, 5, 5) and T(unstr, amg_nu_scale_with_size=0)[2] == (1, 4, 4)
    # rounds 1-3's nodal-block cycle on request: level 1 at 1 + 6, level 2 at 6 + 6, deeper 2 + 2, down to the small inverse
    old = [1738576, 218044, 27436, 3800, 475, 60, 8]
    assert T(old, amg_block_smooth=0, amg_dense_rows=0) == [(0, 1, 1), (0, 1, 6), (0, 6, 6), (0, 2, 2), (0, 2, 2), (0, 2, 2), (2, 0, 0)]
    # a last level too large for any inverse is swept; fixed level-1 counts win over the automatic ones
    assert T([100000, 12500, 5000], amg_dense_rows=512)[-1] == (4, 0, 0)
    assert T(serial, amg_nu_l1_pre=2, amg_nu_l1_post=2)[1] == (1, 2, 2)
    # aggregate blocks only where the level is small enough per rank, when limited (amg_block_max_rows)
    assert T(serial, amg_block_max_rows=8192) == [(0, 1, 1), (0, 1, 6), (0, 6, 6), (1, 2, 2), (3, 0, 0)]
