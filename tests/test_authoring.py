"""Mesh -> constraint-graph authoring: Unity-style render meshes and TetGen tet meshes."""
import numpy as np
import pytest

from softbodyunity_amd.mesh import from_tet_mesh, from_triangle_mesh, read_tetgen
from helpers import build_plan, make_oracle


def unity_cube():
    """24 vertices / 12 triangles like Unity's built-in cube: every face has its own 4 vertices."""
    faces = [((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)), ((0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1)),
             ((0, 0, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1)), ((1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)),
             ((0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 0, 0)), ((0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1))]
    V, F = [], []
    for f in faces:
        b = len(V); V += list(f); F += [(b, b + 2, b + 1), (b, b + 3, b + 2)]   # outward winding
    return np.array(V, np.float32), np.array(F, np.int32)


def grid_cloth(n):
    xs, ys = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    V = np.stack([xs.ravel(), np.zeros(n * n), ys.ravel()], axis=1).astype(np.float32)
    idx = lambda i, j: i * n + j
    F = []
    for i in range(n - 1):
        for j in range(n - 1):
            F += [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)), (idx(i, j), idx(i + 1, j + 1), idx(i, j + 1))]
    return V, np.array(F, np.int32)


def test_unity_cube_welds_to_8_particles():
    V, F = unity_cube()
    m, pov = from_triangle_mesh(V, F)
    assert m.n == 8 and len(pov) == 24
    assert np.allclose(m.rest_pos[pov], V)
    assert len(m.dist_rest) == 18                      # 12 cube edges + 6 face diagonals
    assert len(m.bend_rest) == 18                      # every edge is shared by exactly two triangles
    c = m.bend_rest[:, 0]; s = m.bend_rest[:, 1]
    assert np.allclose(c * c + s * s, 1, atol=1e-6)
    assert np.sum(np.isclose(c, 1, atol=1e-6)) == 6    # the six face diagonals are flat hinges
    assert np.sum(np.isclose(np.abs(s), 1, atol=1e-6)) == 12   # cube edges: 90 degree hinges (sign follows the hinge vertex order)


def test_cloth_rest_state_is_a_fixed_point_and_bending_resists_folding(oracle_mod):
    V, F = grid_cloth(10)
    m, _ = from_triangle_mesh(V, F)
    assert m.n == 100 and len(m.bend_rest) == 9 * 9 + 2 * 9 * 8     # one hinge per interior edge
    assert np.allclose(m.bend_rest, [1, 0], atol=1e-6)
    o = make_oracle(oracle_mod, m, build_plan(m, tile_particles=64), gravity=(0, 0, 0))
    o.step(0.02, 10)
    assert np.allclose(o.x, m.rest_pos, atol=1e-6)
    # fold one half up by 40 degrees about the line x = 4.5; bending pulls the sheet back towards flat
    th = np.deg2rad(40.0)
    m.pos = m.rest_pos.copy()
    up = m.rest_pos[:, 0] > 4.5
    dx = m.rest_pos[up, 0] - 4.5
    m.pos[up, 0] = 4.5 + dx * np.cos(th); m.pos[up, 1] = dx * np.sin(th)

    def fold_height(x):      # mean (1 - cos phi) over all hinges: 0 = flat, independent of rigid motion
        x = x.astype(np.float64); h = m.bend_ijkl
        A, B, Cw, D = (x[h[:, k]] for k in range(4))
        n1 = np.cross(A - Cw, B - Cw); n2 = np.cross(B - D, A - D)
        cs = np.einsum("ij,ij->i", n1, n2) / np.linalg.norm(n1, axis=1) / np.linalg.norm(n2, axis=1)
        return float((1 - cs).mean())
    res = {}
    for name, comp in (("springs only", None), ("with bending", 0.0)):
        mm = m if comp is not None else type(m)(rest_pos=m.rest_pos, pos=m.pos, vel=m.vel, inv_mass=m.inv_mass,
                                                 dist_ij=m.dist_ij, dist_rest=m.dist_rest)
        o = make_oracle(oracle_mod, mm, build_plan(mm, tile_particles=64), gravity=(0, 0, 0), damping=20.0)
        for _ in range(5):
            o.step(0.02, 10)
        assert np.isfinite(o.x).all()
        res[name] = fold_height(o.x)
    assert res["with bending"] < 0.5 * res["springs only"], res


def test_tet_mesh_and_tetgen_reader(tmp_path, oracle_mod):
    nodes = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]], float)
    tets = np.array([[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 2, 6, 7], [0, 6, 4, 7]])   # 6-tet cube
    m = from_tet_mesh(nodes, tets)
    assert np.all(m.vol_rest > 0) and abs(m.vol_rest.sum() - 1.0) < 1e-6
    assert len(m.dist_rest) == 12 + 6 + 1 and len(m.bend_rest) == 18
    (tmp_path / "c.node").write_text("8 3 0 0\n" + "\n".join(f"{i + 1} {p[0]} {p[1]} {p[2]}" for i, p in enumerate(nodes)) + "\n# end\n")
    (tmp_path / "c.ele").write_text("6 4 0\n" + "\n".join(f"{i + 1} " + " ".join(str(v + 1) for v in t) for i, t in enumerate(tets)) + "\n")
    m2 = read_tetgen(tmp_path / "c.node", tmp_path / "c.ele")
    assert np.array_equal(m2.vol_ijkl, m.vol_ijkl) and np.array_equal(m2.dist_ij, m.dist_ij)
    # squash the cube: volume constraints restore the volume
    m.pos = m.rest_pos * np.array([1, 0.8, 1], np.float32)
    o = make_oracle(oracle_mod, m, build_plan(m), gravity=(0, 0, 0))

    def vol(x):
        t = m.vol_ijkl
        return (np.einsum("ij,ij->i", x[t[:, 1]] - x[t[:, 0]], np.cross(x[t[:, 2]] - x[t[:, 0]], x[t[:, 3]] - x[t[:, 0]])) / 6).sum()
    for _ in range(10):
        o.step(0.02, 10)
    assert abs(vol(o.x.astype(np.float64)) - 1.0) < 0.02


def _write_msh(path, nodes, tets, version):
    tags = [10 + 3 * i for i in range(len(nodes))]        # non-consecutive node tags
    with open(path, "w") as f:
        if version == 2:
            f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % (len(nodes) + 1))
            f.write("5 9.0 9.0 9.0\n")                     # a geometry point no element uses
            for t, p in zip(tags, nodes):
                f.write("%d %r %r %r\n" % (t, float(p[0]), float(p[1]), float(p[2])))
            f.write("$EndNodes\n$Elements\n%d\n1 15 2 0 1 5\n2 2 2 0 1 %d %d %d\n" % (len(tets) + 2, tags[0], tags[1], tags[2]))
            for k, t in enumerate(tets):
                f.write("%d 4 2 0 1 %d %d %d %d\n" % (k + 3, *[tags[v] for v in t]))
            f.write("$EndElements\n")
        else:
            f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n2 %d 5 %d\n" % (len(nodes) + 1, tags[-1]))
            f.write("0 1 0 1\n5\n9.0 9.0 9.0\n")
            f.write("3 1 0 %d\n" % len(nodes))
            for t in tags:
                f.write("%d\n" % t)
            for p in nodes:
                f.write("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))
            f.write("$EndNodes\n$Elements\n2 %d 1 %d\n2 1 2 1\n1 %d %d %d\n3 1 4 %d\n" % (len(tets) + 1, len(tets) + 1, tags[0], tags[1], tags[2], len(tets)))
            for k, t in enumerate(tets):
                f.write("%d %d %d %d %d\n" % (k + 2, *[tags[v] for v in t]))
            f.write("$EndElements\n")


@pytest.mark.parametrize("version", [2, 4])
def test_gmsh_reader_gives_the_mesh_from_tet_mesh_gives(tmp_path, version):
    # ".node/.ele or .msh reader" -- both exist; the .msh reader (ASCII 2.2 and 4.1) must give exactly the
    # authoring result of from_tet_mesh on the same nodes and tets: non-consecutive node tags, unused nodes and non-tet elements skipped
    from softbodyunity_amd.mesh import read_gmsh
    rng = np.random.default_rng(5)
    nodes = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [2, 0.5, 0.5]], np.float64) + rng.uniform(-0.05, 0.05, (6, 3))
    tets = np.array([[0, 1, 2, 3], [1, 2, 3, 4], [1, 4, 2, 5]])
    path = tmp_path / f"m{version}.msh"
    _write_msh(path, nodes, tets, version)
    a, b = read_gmsh(str(path)), from_tet_mesh(nodes, tets)
    for f in ("rest_pos", "inv_mass", "dist_ij", "dist_rest", "vol_ijkl", "vol_rest", "bend_ijkl", "bend_rest"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.n == 6 and len(a.vol_rest) == 3 and (a.vol_rest > 0).all()


@pytest.mark.gpu
def test_cloth_through_the_plugin(oracle_mod):
    from softbodyunity_amd import Softbody
    V, F = grid_cloth(40)
    m, _ = from_triangle_mesh(V, F)
    m.inv_mass[m.rest_pos[:, 0] == 0] = 0.0           # pin one edge, let it swing under gravity
    sb = Softbody(m, substeps=10, tile_particles=128, distance_compliance=1e-7, bending_compliance=1e-3).Start()
    try:
        o = make_oracle(oracle_mod, m, sb.plan(), compliance=(1e-7, 0, 1e-3))
        for _ in range(10):
            sb.step(); o.step(0.02, 10)
        x = sb.get_positions()
    finally:
        sb.OnDestroy()
    rel, mabs, bit = oracle_mod.parity_error(x, o.x, m.pos)
    assert rel <= 1e-4 and bit
    assert x[:, 1].min() < -0.1                        # it did swing down (0.2 s of fall)


# ---- render normals (SPEC.md §6a) ------------------------------------------------------------

def test_oracle_vertex_normals_known_answers(oracle_mod):
    V, F = grid_cloth(6)                                   # flat sheet in the xz plane... whichever: all normals equal
    nrm = oracle_mod.vertex_normals(V, F)
    n0 = nrm[0]
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-6)
    assert np.all(nrm == n0), "a flat sheet has one exact normal"
    assert np.count_nonzero(n0) == 1 and abs(abs(n0).max() - 1.0) == 0.0
    # reversing the winding flips the sign exactly
    assert np.array_equal(oracle_mod.vertex_normals(V, F[:, ::-1]), -nrm)
    # closed cube: the normal of a corner particle is the normalised sum of the face areas it touches
    Vc, Fc = unity_cube()
    m, pov = from_triangle_mesh(Vc, Fc)
    tri = pov[Fc]
    nc = oracle_mod.vertex_normals(m.rest_pos, tri)
    centre = m.rest_pos.mean(0)
    out = m.rest_pos - centre
    assert np.all(np.einsum("ij,ij->i", nc, out) > 0) and np.allclose(np.linalg.norm(nc, axis=1), 1, atol=1e-6)
    # a particle in no triangle, a degenerate triangle: zero normal, no NaN
    x = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [5, 5, 5]], np.float32)
    z = oracle_mod.vertex_normals(x, np.array([[0, 1, 2]], np.int32))
    assert not z.any()


@pytest.mark.gpu
def test_gpu_render_normals_match_the_oracle(oracle_mod):
    from softbodyunity_amd import Softbody, native
    V, F = grid_cloth(40)
    rng = np.random.default_rng(3)
    V = V + rng.normal(0, 0.05, V.shape).astype(np.float32)
    m, pov = from_triangle_mesh(V, F)
    tri = pov[F]
    m.inv_mass[:40] = 0.0
    sb = Softbody(m, substeps=6, tile_particles=128, bending_compliance=1e-4).Start()
    try:
        with pytest.raises(native.SoftbodyError):          # no triangles yet
            sb.readback_begin(); sb.readback_end(normals=True)
        sb.set_render_triangles(tri)
        for k in range(3):
            sb.step()
            sb.readback_begin()
            sb.step()                                        # the next tick overlaps the copy and the normals kernel
            pos, nrm = sb.readback_end(normals=True)
            ref = oracle_mod.vertex_normals(pos, tri)
            assert np.array_equal(nrm.view(np.uint32), ref.view(np.uint32)), f"snapshot {k}"
            assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-5)
        # fewer triangles: particles that lost all their triangles read zero
        sb.set_render_triangles(tri[: len(tri) // 2])
        sb.readback_begin(); pos, nrm = sb.readback_end(normals=True)
        assert np.array_equal(nrm.view(np.uint32), oracle_mod.vertex_normals(pos, tri[: len(tri) // 2]).view(np.uint32))
        assert (np.linalg.norm(nrm, axis=1) == 0).any()
        with pytest.raises(native.SoftbodyError):
            sb.set_render_triangles(np.array([[0, 1, m.n]], np.int32))
    finally:
        sb.OnDestroy()


@pytest.mark.gpu
def test_gpu_render_set_only_readback_is_the_compact_view_of_the_full_one(oracle_mod):
    # a volumetric body renders its surface: the readback can bring only the particles the render triangles use
    from softbodyunity_amd import Softbody, jelly_cube
    sys_path_tools = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tools")
    __import__("sys").path.insert(0, sys_path_tools)
    from readback_bench import surface_triangles
    n = 20
    mesh = jelly_cube(n)
    tri = surface_triangles(n)
    sb = Softbody(mesh, substeps=5).Start()
    try:
        sb.set_render_triangles(tri)
        sb.step(); sb.readback_begin()
        full_pos, full_nrm = (a.copy() for a in sb.readback_end(normals=True))
        sb.set_readback_render_set_only(True)
        sb.readback_begin()                                  # same state, compact view
        pos, nrm = sb.readback_end(normals=True)
        ids = sb.render_set()
        assert len(ids) == n ** 3 - (n - 2) ** 3 and np.array_equal(ids, np.unique(tri))
        assert pos.shape == (len(ids), 3) and np.array_equal(pos, full_pos[ids]) and np.array_equal(nrm, full_nrm[ids])
        assert np.array_equal(nrm.view(np.uint32), oracle_mod.vertex_normals(full_pos, tri)[ids].view(np.uint32))
        # outward normals on the cube's faces
        c = full_pos.mean(0)
        assert (np.einsum("ij,ij->i", nrm, pos - c) > 0).mean() > 0.99
        sb.set_readback_render_set_only(False)
        sb.readback_begin()
        assert sb.readback_end().shape == (mesh.n, 3)
    finally:
        sb.OnDestroy()


def test_cfg1_cpu_component_path_needs_no_gpu(oracle_mod):
    # BASELINE.json:7 -- 8^3 cube, 10 substeps, CPU FixedUpdate only: the component's CPU branch (csharp/Softbody.cs,
    # useGpu = false) must not create a solver handle (sb_create fails without a gfx950); it takes the schedule from the
    # host-only planner. Same call order through the Python mirror; the tick is the oracle (the C# solver's twin), and the
    # result is the committed golden vector.
    import os
    import numpy as np
    from softbodyunity_amd import Softbody, jelly_cube
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cube8_s10_t5.npz"))
    mesh = jelly_cube(8)
    sb = Softbody(mesh, substeps=10, use_gpu=False).Start()        # works in a container without any GPU
    try:
        assert sb._h is None
        sched = sb.cpu_schedule()
        for parity in (0, 1):
            assert np.array_equal(sched[parity][1], g[f"order_id{parity}"])
        with pytest.raises(RuntimeError):
            sb.FixedUpdate()
    finally:
        sb.OnDestroy()
    from helpers import make_oracle
    o = make_oracle(oracle_mod, mesh, None)
    for parity in (0, 1):
        o.set_order(sched[parity][0], sched[parity][1], parity=parity)
    for _ in range(5):
        o.step(0.02, 10)
    assert np.array_equal(o.x.view(np.uint32), g["x"].view(np.uint32))


def test_csharp_cpu_branch_never_creates_a_solver():
    # source-level check of the uncompiled C# (no toolchain in the image): inside `if (!useGpu) { ... return; }` of Start()
    # only host-only entry points may appear
    import os
    import re
    cs = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "csharp", "Softbody.cs")).read()
    start = cs.index("if (!useGpu)")
    branch = cs[start:cs.index("return;", start)]
    called = set(re.findall(r"SoftbodyNative\.(sb_[a-z_]+)", branch))
    assert called == {"sb_plan_build", "sb_plan_order_count", "sb_plan_get_order", "sb_plan_destroy"}, called
