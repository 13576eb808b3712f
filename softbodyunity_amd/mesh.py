"""Synthetic mesh + constraint-graph generators (SPEC.md §7; BASELINE.json:7-11).

The reference ships no meshes (/root/reference/README.md:1 is the whole tree), so these are the
builder-defined workloads of SURVEY.md §8d: the n^3 "jelly cube" with structural springs and an
irregular tetrahedral surrogate for the Stanford-bunny config (no mesh asset exists offline).
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class SoftbodyMesh:
    """Host-side authoring data, in the layouts the C ABI takes (AoS xyz float32, int32 indices)."""
    rest_pos: np.ndarray            # (N,3) f32 rest positions
    pos: np.ndarray                 # (N,3) f32 initial positions
    vel: np.ndarray                 # (N,3) f32
    inv_mass: np.ndarray            # (N,)  f32
    dist_ij: np.ndarray             # (Md,2) i32
    dist_rest: np.ndarray           # (Md,) f32
    vol_ijkl: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.int32))
    vol_rest: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    bend_ijkl: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.int32))
    bend_rest: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))  # (cos,sin) of rest dihedral
    label: str = ""
    # sharded authoring (include/softbody.h sb_domain): this mesh is one rank's WINDOW of a larger mesh
    domain: object = None           # native.SbDomain of the whole mesh
    global_id: np.ndarray = None    # (N,) i32 ids of the window's particles in the whole mesh, ascending

    @property
    def n(self):
        return self.pos.shape[0]


def jelly_cube(n, spacing=1.0, perturb=0.05, seed=1234, pin_top=False, stencil="structural", heterogeneous=False):
    """n^3 lattice, index (iz*n+iy)*n+ix, structural springs x-dir then y then z (SPEC.md §7).

    heterogeneous=True is the data-layout WORST case beside the benchmark's best case (SPEC.md §7): every particle its own mass
    (m ~ U(0.5, 2) from the seed, so the inverse mass travels as a 4-byte float instead of not at all) and every spring its own
    rest length (the perturbed pose's length x U(0.95, 1.05), so no tile can dictionary-code its slots: 8 bytes instead of 4)."""
    assert n >= 2
    N = n ** 3
    ax = np.arange(n, dtype=np.float32) * np.float32(spacing)
    rest = np.empty((n, n, n, 3), np.float32)                # [iz, iy, ix, xyz]
    rest[..., 0] = ax[None, None, :]; rest[..., 1] = ax[None, :, None]; rest[..., 2] = ax[:, None, None]
    rest = rest.reshape(N, 3)
    rng = np.random.default_rng(seed)
    jitter = rng.uniform(-perturb, perturb, size=(N, 3))
    jitter *= spacing
    jitter += rest                                           # float64: rest + jitter, rounded once to float32
    pos = jitter.astype(np.float32)
    del jitter
    idx = np.arange(N, dtype=np.int32).reshape(n, n, n)
    m1 = n * n * (n - 1)
    # structural springs, x-direction first, then y, then z, each in particle order (SPEC.md 7)
    n_struct = 3 * m1
    ij_struct = np.empty((n_struct, 2), np.int32)
    for k, (lo, off) in enumerate(((idx[:, :, :-1], 1), (idx[:, :-1, :], n), (idx[:-1, :, :], n * n))):
        ij_struct[k * m1:(k + 1) * m1, 0] = lo.reshape(-1)
        np.add(ij_struct[k * m1:(k + 1) * m1, 0], np.int32(off), out=ij_struct[k * m1:(k + 1) * m1, 1])
    edges = [ij_struct]
    if stencil == "full":
        lin = np.arange(N, dtype=np.int64)
        cx = lin % n
        cy = (lin // n) % n
        cz = lin // (n * n)
    if stencil == "full":
        # face + body diagonals (26-neighbour stencil), for colouring stress tests
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (0, 1):
                    if (dx, dy, dz) in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)) or abs(dx) + abs(dy) + abs(dz) < 2:
                        continue
                    if dz == 0 and (dy < 0 or (dy == 0 and dx < 0)):
                        continue
                    m = ((cx + dx >= 0) & (cx + dx < n) & (cy + dy >= 0) & (cy + dy < n) & (cz + dz < n))
                    lo = lin[m]
                    edges.append(np.stack([lo, lo + dx + dy * n + dz * n * n], axis=1))
    ij = ij_struct if len(edges) == 1 else np.concatenate(edges).astype(np.int32)
    if stencil == "full":
        rest_len = np.linalg.norm(rest[ij[:, 0]].astype(np.float64) - rest[ij[:, 1]].astype(np.float64), axis=1).astype(np.float32)
    else:
        rest_len = np.full(ij.shape[0], spacing, np.float32)   # axis springs: L0 = spacing (SPEC.md §7)
    w = np.ones(N, np.float32)
    if heterogeneous:
        hrng = np.random.default_rng(seed + 1)
        w = (1.0 / hrng.uniform(0.5, 2.0, size=N)).astype(np.float32)
        d = pos[ij[:, 0]].astype(np.float64)
        d -= pos[ij[:, 1]]
        rest_len = (np.sqrt(np.einsum("ij,ij->i", d, d)) * hrng.uniform(0.95, 1.05, size=ij.shape[0])).astype(np.float32)
        del d
    if pin_top:
        w.reshape(n, n, n)[:, n - 1, :] = 0.0
    return SoftbodyMesh(rest_pos=rest, pos=pos, vel=np.zeros_like(pos), inv_mass=w, dist_ij=ij,
                        dist_rest=rest_len, label=f"jelly_cube_{n}^3_{stencil}" + ("_heterogeneous" if heterogeneous else ""))


def jelly_cube_window(n, rank, world, part_dims=(0, 0, 0), tile_particles=512, spacing=1.0, perturb=0.05, seed=1234, pin_top=False,
                      heterogeneous=False):
    """The part of jelly_cube(n) rank `rank` of `world` hands over under sharded authoring: the lattice points inside the box
    sb_domain_window gives for it, the structural springs among them (x, then y, then z, each in ascending order of the lower end
    point: the whole cube's order restricted to the window), ids of the whole cube in `global_id`. Positions are the whole cube's
    (the jitter stream is drawn for the window's planes only: O(window) memory, identical values); heterogeneous=True likewise
    reproduces the whole cube's per-particle masses and per-spring rest-length factors by seeking in their stream."""
    from . import native
    N = n ** 3
    hi_c = float(np.float32(n - 1) * np.float32(spacing))
    dom = native.make_domain(N, (0.0, 0.0, 0.0), (hi_c, hi_c, hi_c), float(np.float32(spacing)))
    lo, hi = native.domain_window(dom, rank, world, part_dims, tile_particles)
    ax = (np.arange(n, dtype=np.float32) * np.float32(spacing)).astype(np.float64)
    rng_idx = []
    for a in range(3):
        inside = np.nonzero((ax >= lo[a]) & (ax < hi[a]))[0]
        rng_idx.append((int(inside[0]), int(inside[-1]) + 1))
    (x0, x1), (y0, y1), (z0, z1) = rng_idx
    wx, wy, wz = x1 - x0, y1 - y0, z1 - z0
    gz, gy, gx = np.meshgrid(np.arange(z0, z1), np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
    gid = ((gz.astype(np.int64) * n + gy) * n + gx).reshape(-1)
    axf = np.arange(n, dtype=np.float32) * np.float32(spacing)
    rest = np.stack([axf[gx.reshape(-1)], axf[gy.reshape(-1)], axf[gz.reshape(-1)]], axis=1)
    # the whole cube draws rng.uniform(size=(N, 3)) row by row: plane iz of the cube is rows [iz n^2, (iz+1) n^2) of that stream
    rng = np.random.default_rng(seed)
    jit = np.empty((wz, wy, wx, 3), np.float64)
    bg = rng.bit_generator
    plane_doubles = n * n * 3
    bg.advance(z0 * plane_doubles)                 # PCG64: one 64-bit draw per double
    for k in range(wz):
        pl = rng.uniform(-perturb, perturb, size=(n, n, 3))
        jit[k] = pl[y0:y1, x0:x1]
    jit *= spacing
    pos = (jit.reshape(-1, 3) + rest).astype(np.float32)
    lidx = np.arange(wx * wy * wz, dtype=np.int32).reshape(wz, wy, wx)
    edges = []
    for lo_s, off in ((lidx[:, :, :-1], 1), (lidx[:, :-1, :], wx), (lidx[:-1, :, :], wx * wy)):
        a = lo_s.reshape(-1)
        edges.append(np.stack([a, a + off], axis=1))
    ij = np.concatenate(edges).astype(np.int32)
    w = np.ones(len(gid), np.float32)
    rest_len = np.full(ij.shape[0], spacing, np.float32)
    if heterogeneous:
        # jelly_cube draws, from default_rng(seed + 1): N masses in particle order, then one factor per spring in spring order
        # (x springs by (iz, iy, ix < n-1), y springs by (iz, iy < n-1, ix), z springs by (iz < n-1, iy, ix)): seek plane by plane
        def planes(offset, rows, cols, zs, ys, xs, lo_v, hi_v):
            out = np.empty((len(zs), len(ys), len(xs)), np.float64)
            for k, iz in enumerate(zs):
                h = np.random.default_rng(seed + 1)
                h.bit_generator.advance(offset + int(iz) * rows * cols)
                out[k] = h.uniform(lo_v, hi_v, size=(rows, cols))[np.ix_(ys, xs)]
            return out.reshape(-1)
        zs, ys, xs = np.arange(z0, z1), np.arange(y0, y1), np.arange(x0, x1)
        w = (1.0 / planes(0, n, n, zs, ys, xs, 0.5, 2.0)).astype(np.float32)
        m1 = n * n * (n - 1)
        fac = np.concatenate([planes(N, n, n - 1, zs, ys, xs[:-1], 0.95, 1.05),
                              planes(N + m1, n - 1, n, zs, ys[:-1], xs, 0.95, 1.05),
                              planes(N + 2 * m1, n, n, zs[:-1], ys, xs, 0.95, 1.05)])
        d = pos[ij[:, 0]].astype(np.float64)
        d -= pos[ij[:, 1]]
        rest_len = (np.sqrt(np.einsum("ij,ij->i", d, d)) * fac).astype(np.float32)
    if pin_top:
        w[gy.reshape(-1) == n - 1] = 0.0
    m = SoftbodyMesh(rest_pos=rest, pos=pos, vel=np.zeros_like(pos), inv_mass=w, dist_ij=ij,
                     dist_rest=rest_len, label=f"jelly_cube_{n}^3_window_of_rank_{rank}/{world}" + ("_heterogeneous" if heterogeneous else ""))
    m.domain = dom
    m.global_id = gid.astype(np.int32)
    return m


def _blob_inside(p):
    """Implicit 'bunny-ish' blob: union of a body ellipsoid, a head sphere and two ear ellipsoids."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    body = (x / 1.0) ** 2 + ((y + 0.2) / 0.8) ** 2 + (z / 0.75) ** 2 < 1.0
    head = ((x - 0.85) / 0.5) ** 2 + ((y - 0.55) / 0.5) ** 2 + (z / 0.45) ** 2 < 1.0
    ear1 = ((x - 0.8) / 0.16) ** 2 + ((y - 1.25) / 0.5) ** 2 + ((z - 0.2) / 0.12) ** 2 < 1.0
    ear2 = ((x - 0.8) / 0.16) ** 2 + ((y - 1.25) / 0.5) ** 2 + ((z + 0.2) / 0.12) ** 2 < 1.0
    return body | head | ear1 | ear2


def bunny_surrogate(target_verts=100_000, seed=1234, perturb=0.01):
    """Irregular tet mesh SURROGATE for BASELINE.json:11 (no Stanford-bunny asset is available).

    Jittered-grid points inside an implicit blob -> scipy Delaunay -> keep tets whose centroid is
    inside. Constraints: every tet edge (distance), every tet (volume), every pair of boundary
    triangles sharing an edge (cosine-dihedral bending, SPEC.md §6).
    """
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    lo = np.array([-1.1, -1.1, -0.85]); hi = np.array([1.45, 1.85, 0.85])
    # choose the grid pitch so that roughly target_verts points land inside
    probe = rng.uniform(lo, hi, size=(200_000, 3))
    frac = _blob_inside(probe).mean()
    vol = np.prod(hi - lo) * frac
    pitch = (vol / target_verts) ** (1.0 / 3.0)
    g = [np.arange(lo[a], hi[a], pitch) for a in range(3)]
    P = np.stack(np.meshgrid(*g, indexing="ij"), axis=-1).reshape(-1, 3)
    P = P + rng.uniform(-0.35, 0.35, size=P.shape) * pitch
    P = P[_blob_inside(P)]
    tri = Delaunay(P)
    T = tri.simplices.astype(np.int64)
    cen = P[T].mean(axis=1)
    e = P[T]
    vol6 = np.einsum("ij,ij->i", e[:, 1] - e[:, 0], np.cross(e[:, 2] - e[:, 0], e[:, 3] - e[:, 0]))
    # drop outside + sliver tets
    lens = np.linalg.norm(e[:, [0, 0, 0, 1, 1, 2]] - e[:, [1, 2, 3, 2, 3, 3]], axis=2).max(axis=1)
    keep = _blob_inside(cen) & (np.abs(vol6) > 1e-3 * pitch ** 3) & (lens < 2.5 * pitch)
    T = T[keep]; vol6 = vol6[keep]
    # orient positively
    neg = vol6 < 0
    T[neg] = T[neg][:, [0, 2, 1, 3]]
    vol6 = np.abs(vol6)
    # compact vertices
    used = np.unique(T)
    remap = -np.ones(P.shape[0], np.int64); remap[used] = np.arange(used.size)
    P = P[used]; T = remap[T]
    m = from_tet_mesh(P, T, label="bunny_SURROGATE")
    rest = m.rest_pos; r64 = rest.astype(np.float64)
    jitter = rng.uniform(-perturb, perturb, size=rest.shape) * pitch
    pos = (r64 + jitter).astype(np.float32)
    m.pos = pos
    m.label = f"bunny_SURROGATE_{rest.shape[0]}v"
    return m


# ---- authoring: render / tet meshes -> particles + constraint graph -------------------

def _hinges(r64, tris):
    """Bending hinges (SPEC.md §6) for every manifold edge shared by exactly two triangles of `tris`.
    Returns (ijkl, rest_cos_sin): shared edge (i0,i1), wings i2,i3."""
    he = np.concatenate([tris[:, [0, 1, 2]], tris[:, [1, 2, 0]], tris[:, [2, 0, 1]]])  # (a,b,opp)
    ek = np.sort(he[:, :2], axis=1)
    order = np.lexsort((ek[:, 1], ek[:, 0]))
    ek = ek[order]; he = he[order]
    same = (ek[1:] == ek[:-1]).all(axis=1)
    first = np.nonzero(same)[0]
    ok = np.ones(first.size, bool)
    if first.size > 1:   # keep only edges seen exactly twice
        ok[1:] &= first[1:] != first[:-1] + 1
        ok[:-1] &= first[1:] != first[:-1] + 1
    first = first[ok]
    bend = np.stack([ek[first, 0], ek[first, 1], he[first, 2], he[first + 1, 2]], axis=1).astype(np.int32)
    if bend.shape[0] == 0:
        return bend.reshape(0, 4), np.zeros((0, 2), np.float32)
    A, B, Cw, D = (r64[bend[:, k]] for k in range(4))
    e = B - A
    n1 = np.cross(A - Cw, B - Cw); n2 = np.cross(B - D, A - D)
    l1 = np.linalg.norm(n1, axis=1); l2 = np.linalg.norm(n2, axis=1); le = np.linalg.norm(e, axis=1)
    good = (l1 > 1e-12) & (l2 > 1e-12) & (le > 1e-12)
    bend = bend[good]; n1 = n1[good] / l1[good, None]; n2 = n2[good] / l2[good, None]; e = e[good] / le[good, None]
    cs = np.einsum("ij,ij->i", n1, n2)
    sn = -np.einsum("ij,ij->i", np.cross(n1, n2), e)
    return bend, np.stack([cs, sn], axis=1).astype(np.float32)


def from_tet_mesh(nodes, tets, mass_density=None, label="tet_mesh"):
    """Tetrahedral mesh -> particles, edge springs, tet volume constraints, surface-hinge bending.

    Tets are re-oriented to positive volume. inv_mass = 1 unless mass_density is given (then lumped tet masses)."""
    P = np.asarray(nodes, np.float64).reshape(-1, 3)
    T = np.asarray(tets, np.int64).reshape(-1, 4).copy()
    vol6 = np.einsum("ij,ij->i", P[T[:, 1]] - P[T[:, 0]], np.cross(P[T[:, 2]] - P[T[:, 0]], P[T[:, 3]] - P[T[:, 0]]))
    neg = vol6 < 0
    T[neg] = T[neg][:, [0, 2, 1, 3]]
    rest = P.astype(np.float32)
    r64 = rest.astype(np.float64)
    pairs = np.concatenate([T[:, [a, b]] for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))])
    pairs.sort(axis=1)
    ij = np.unique(pairs, axis=0).astype(np.int32)
    dist_rest = np.linalg.norm(r64[ij[:, 0]] - r64[ij[:, 1]], axis=1).astype(np.float32)
    vol = np.einsum("ij,ij->i", r64[T[:, 1]] - r64[T[:, 0]], np.cross(r64[T[:, 2]] - r64[T[:, 0]], r64[T[:, 3]] - r64[T[:, 0]])) / 6.0
    faces = np.concatenate([T[:, [1, 2, 3]], T[:, [0, 3, 2]], T[:, [0, 1, 3]], T[:, [0, 2, 1]]])
    key = np.sort(faces, axis=1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    bfaces = faces[cnt[inv.ravel()] == 1]
    bend, c0 = _hinges(r64, bfaces)
    w = np.ones(rest.shape[0], np.float32)
    if mass_density is not None:
        mass = np.zeros(rest.shape[0])
        np.add.at(mass, T.ravel(), np.repeat(np.abs(vol) * mass_density / 4.0, 4))
        w = (1.0 / np.maximum(mass, 1e-30)).astype(np.float32)
    return SoftbodyMesh(rest_pos=rest, pos=rest.copy(), vel=np.zeros_like(rest), inv_mass=w, dist_ij=ij,
                        dist_rest=dist_rest, vol_ijkl=T.astype(np.int32), vol_rest=vol.astype(np.float32),
                        bend_ijkl=bend, bend_rest=c0, label=label)


def from_triangle_mesh(vertices, triangles, weld_eps=1e-6, label="surface_mesh"):
    """Render mesh (Unity Mesh.vertices / Mesh.triangles) -> welded particles, edge springs, hinge bending.

    Unity duplicates vertices along UV/normal seams; particles are the welded positions. Returns
    (SoftbodyMesh, particle_of_vertex) so the component can scatter particle positions back to mesh.vertices."""
    V = np.asarray(vertices, np.float64).reshape(-1, 3)
    F = np.asarray(triangles, np.int64).reshape(-1, 3)
    q = np.round(V / max(weld_eps, 1e-30)).astype(np.int64)
    _, first, particle_of_vertex = np.unique(q, axis=0, return_index=True, return_inverse=True)
    particle_of_vertex = particle_of_vertex.ravel()
    rest = V[first].astype(np.float32)
    r64 = rest.astype(np.float64)
    Fp = particle_of_vertex[F]
    Fp = Fp[(Fp[:, 0] != Fp[:, 1]) & (Fp[:, 1] != Fp[:, 2]) & (Fp[:, 0] != Fp[:, 2])]     # drop degenerate triangles
    pairs = np.concatenate([Fp[:, [0, 1]], Fp[:, [1, 2]], Fp[:, [2, 0]]])
    pairs.sort(axis=1)
    ij = np.unique(pairs, axis=0).astype(np.int32)
    dist_rest = np.linalg.norm(r64[ij[:, 0]] - r64[ij[:, 1]], axis=1).astype(np.float32)
    bend, c0 = _hinges(r64, Fp)
    m = SoftbodyMesh(rest_pos=rest, pos=rest.copy(), vel=np.zeros_like(rest), inv_mass=np.ones(rest.shape[0], np.float32),
                     dist_ij=ij, dist_rest=dist_rest, bend_ijkl=bend, bend_rest=c0, label=label)
    return m, particle_of_vertex.astype(np.int32)


def read_tetgen(node_path, ele_path):
    """TetGen .node/.ele pair -> from_tet_mesh (so a real Stanford-bunny tet mesh can replace the surrogate)."""
    def rows(path):
        out = []
        for line in open(path):
            line = line.split("#", 1)[0].strip()
            if line:
                out.append(line.split())
        return out
    nr = rows(node_path); er = rows(ele_path)
    n_nodes, dim = int(nr[0][0]), int(nr[0][1])
    assert dim == 3, "3-D .node file expected"
    ids = np.array([int(r[0]) for r in nr[1:1 + n_nodes]])
    nodes = np.array([[float(c) for c in r[1:4]] for r in nr[1:1 + n_nodes]])
    base = ids.min()      # TetGen files are 0- or 1-based
    assert np.array_equal(ids, np.arange(base, base + n_nodes)), "node ids must be consecutive"
    n_tets, npt = int(er[0][0]), int(er[0][1])
    assert npt >= 4
    tets = np.array([[int(c) - base for c in r[1:5]] for r in er[1:1 + n_tets]])
    return from_tet_mesh(nodes, tets, label=f"tetgen:{node_path}")


def read_gmsh(path):
    """Gmsh .msh, ASCII format 2.2 or 4.1 -> from_tet_mesh (4-node tetrahedra, element type 4; other element types are skipped).

    The second tet-mesh input SURVEY.md 8f names beside TetGen's .node/.ele. Node tags need not be consecutive."""
    lines = [ln.strip() for ln in open(path)]
    sect = {}
    i = 0
    while i < len(lines):
        if lines[i].startswith("$") and not lines[i].startswith("$End"):
            name = lines[i][1:]
            j = i + 1
            while j < len(lines) and lines[j] != "$End" + name:
                j += 1
            sect[name] = [ln for ln in lines[i + 1:j] if ln]
            i = j + 1
        else:
            i += 1
    fmt = sect["MeshFormat"][0].split()
    version, is_binary = float(fmt[0]), int(fmt[1])
    assert is_binary == 0, "binary .msh files are not read: export ASCII (gmsh -format msh2 / msh4 without -bin)"
    tags, coords, tets = [], [], []
    N, E = sect["Nodes"], sect["Elements"]
    if version < 3.0:                                   # 2.2: "n" then "tag x y z"; "m" then "tag type ntags tags... nodes..."
        for ln in N[1:1 + int(N[0])]:
            t = ln.split(); tags.append(int(t[0])); coords.append([float(c) for c in t[1:4]])
        for ln in E[1:1 + int(E[0])]:
            t = [int(c) for c in ln.split()]
            if t[1] == 4:
                tets.append(t[3 + t[2]:3 + t[2] + 4])
    else:                                               # 4.x: entity blocks
        nb = int(N[0].split()[0]); k = 1
        for _ in range(nb):
            cnt = int(N[k].split()[3]); k += 1
            tags += [int(t) for t in N[k:k + cnt]]; k += cnt
            coords += [[float(c) for c in ln.split()[:3]] for ln in N[k:k + cnt]]; k += cnt
        nb = int(E[0].split()[0]); k = 1
        for _ in range(nb):
            h = E[k].split(); etype, cnt = int(h[2]), int(h[3]); k += 1
            if etype == 4:
                tets += [[int(c) for c in ln.split()[1:5]] for ln in E[k:k + cnt]]
            k += cnt
    assert tets, "no 4-node tetrahedra (element type 4) in the file"
    tags = np.asarray(tags, np.int64)
    index_of = np.full(tags.max() + 1, -1, np.int64)
    index_of[tags] = np.arange(len(tags))
    T = index_of[np.asarray(tets, np.int64)]
    assert (T >= 0).all(), "an element refers to a node the file does not define"
    used = np.unique(T)                                 # drop nodes no tet uses (geometry points of the CAD model)
    remap = np.full(len(tags), -1, np.int64); remap[used] = np.arange(len(used))
    return from_tet_mesh(np.asarray(coords, np.float64)[used], remap[T], label=f"gmsh:{path}")
