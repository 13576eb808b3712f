"""Position reads between two ticks PEEK instead of completing the tick (include/softbody.h, sb_step / sb_readback_begin):
tile_kernel<4> runs the held-back last kernel's rounds + collision into a side array and leaves the state alone, so the tick
stays fusable with the next one. What a caller sees must be bit for bit what the flushed path (SB_NO_PEEK=1) and the CPU
oracle give -- on spring tiles, on tiles with tets and hinges, with particles resting on the ground plane, in the compact
render-set mode (a subset of the tiles is launched) and with every width of workgroup the launcher picks."""
import os
import sys

import numpy as np
import pytest

from helpers import make_oracle
from softbodyunity_amd import Softbody, bunny_surrogate, jelly_cube

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _surface_triangles_of_tets(mesh):
    """Boundary faces of a tet mesh (faces that belong to one tet only): what a renderer would draw."""
    t = mesh.vol_ijkl.reshape(-1, 4)
    faces = np.concatenate([t[:, [0, 1, 2]], t[:, [0, 1, 3]], t[:, [0, 2, 3]], t[:, [1, 2, 3]]])
    key = np.sort(faces, axis=1)
    _, idx, cnt = np.unique(key, axis=0, return_index=True, return_counts=True)
    return faces[idx[cnt == 1]].astype(np.int32)


def _session(mesh, tri, peek, monkeypatch, ticks, compact, **kw):
    """ticks x (step, async readback [+ a blocking get_positions on odd ticks]) -> snapshots, final state, stats"""
    monkeypatch.setenv("SB_PEEK_MIN_TILES", "0")        # (by default only launches of >= 2 048 workgroups peek: below that it does not pay)
    if peek:
        monkeypatch.delenv("SB_NO_PEEK", raising=False)
    else:
        monkeypatch.setenv("SB_NO_PEEK", "1")
    sb = Softbody(mesh, **kw).Start()
    try:
        snaps, blocking = [], []
        if tri is not None:
            sb.set_render_triangles(tri)
            if compact:
                sb.set_readback_render_set_only(True)
        for k in range(ticks):
            sb.step()
            sb.readback_begin()
            if k & 1:
                blocking.append(sb.get_positions().copy())       # a second read of the same tick end: peeks again
            got = sb.readback_end(normals=tri is not None)
            snaps.append(tuple(a.copy() for a in got) if isinstance(got, tuple) else (got.copy(),))
        ids = sb.render_set().copy() if (tri is not None and compact) else None
        st = sb.stats()
        return snaps, blocking, sb.get_positions().copy(), sb.get_velocities().copy(), st, ids
    finally:
        sb.OnDestroy()


@pytest.mark.parametrize("case", ["cube_full", "cube_render_set", "cube_heterogeneous_ground", "tets_render_set", "cube_wide_tiles", "cube_large_tiles"])
def test_peeked_reads_equal_flushed_reads_and_the_oracle(case, monkeypatch, oracle_mod):
    from readback_bench import surface_triangles
    kw, ticks, compact, tri = dict(substeps=6), 5, False, None
    if case == "cube_full":
        mesh = jelly_cube(24)
    elif case == "cube_render_set":
        mesh = jelly_cube(24); tri = surface_triangles(24); compact = True
    elif case == "cube_heterogeneous_ground":           # 8-byte slots, per-particle masses; the cube falls onto the plane: collide matters
        mesh = jelly_cube(16, heterogeneous=True); tri = surface_triangles(16); compact = True
        kw = dict(substeps=4, ground_plane=(0, 1, 0, -0.5), damping=0.1); ticks = 30
    elif case == "tets_render_set":                      # tiles with tets + hinges (8-wave workgroups), T2 layers between the tile kernels
        mesh = bunny_surrogate(target_verts=6000, seed=7); tri = _surface_triangles_of_tets(mesh); compact = True
        kw = dict(substeps=4, distance_compliance=1e-7, volume_compliance=1e-7, bending_compliance=1e-4,
                  ground_plane=(0, 1, 0, float(mesh.pos[:, 1].min()) + 0.02))
    elif case == "cube_large_tiles":                     # tiles of up to 1 024 particles: the kernels' second capacity
        mesh = jelly_cube(30); tri = surface_triangles(30); compact = True; kw = dict(substeps=6, tile_particles=1000)
    else:                                                # > 768 tiles: 256-lane workgroups; tile 128 also makes under-full packs
        mesh = jelly_cube(40); kw = dict(substeps=4, tile_particles=128)
    a = _session(mesh, tri, True, monkeypatch, ticks, compact, **kw)
    b = _session(mesh, tri, False, monkeypatch, ticks, compact, **kw)
    assert a[4]["readback_peeks"] >= ticks and b[4]["readback_peeks"] == 0
    for k, (p, q) in enumerate(zip(a[0], b[0])):
        for x, y in zip(p, q):
            assert np.array_equal(_bits(x), _bits(y)), f"snapshot {k}"
    for p, q in zip(a[1], b[1]):
        assert np.array_equal(_bits(p), _bits(q))
    assert np.array_equal(_bits(a[2]), _bits(b[2])) and np.array_equal(_bits(a[3]), _bits(b[3]))      # the state itself was not disturbed
    if compact:
        n_tiles = a[4]["n_tiles"][0]
        assert 0 < a[4]["readback_peek_tiles"] <= n_tiles
        if case == "cube_render_set":
            assert a[4]["readback_peek_tiles"] < n_tiles         # 24^3: 27 T0 tiles, the centre one holds no surface particle
    # ... and against the oracle: the last snapshot is the state after `ticks` ticks
    sbp = Softbody(mesh, **kw).Start()
    try:
        plan = sbp.plan()
        comp = tuple(kw.get(k, 0.0) for k in ("distance_compliance", "volume_compliance", "bending_compliance"))
        o = make_oracle(oracle_mod, mesh, plan, damping=kw.get("damping", 0.0), compliance=comp, ground_plane=kw.get("ground_plane"))
        for _ in range(ticks):
            o.step(0.02, kw["substeps"])
    finally:
        sbp.OnDestroy()
    last = a[0][-1][0]
    want = o.x if not compact else o.x[a[5]]
    assert np.array_equal(_bits(last), _bits(want))


def test_a_peek_keeps_the_tick_fusable(monkeypatch):
    """The point of the peek: with a render readback after every tick, a tick of S substeps stays S launches over the mesh plus one
    over the surface tiles, instead of S + 1 (the held-back last kernel forced out, the next tick starting with an unfused first
    kernel). Checked through the plugin's own counters (sb_stats.ticks_fused), readbacks pipelined one tick behind as a renderer
    would; the HIP-event time per tick is printed, not asserted (192^3, 4 substeps, one box: 0.397 -> 0.349 ms per tick; at 128^3 the
    tick is bound by the ~130 us of host calls either way)."""
    from readback_bench import surface_triangles
    n = 192
    mesh = jelly_cube(n)
    tri = surface_triangles(n)

    def per_tick(peek):
        if peek:
            monkeypatch.delenv("SB_NO_PEEK", raising=False)
        else:
            monkeypatch.setenv("SB_NO_PEEK", "1")
        sb = Softbody(mesh, substeps=4).Start()
        try:
            sb.set_render_triangles(tri); sb.set_readback_render_set_only(True)

            def run(ticks):
                for k in range(ticks):
                    sb.step(); sb.readback_begin()
                    if k:
                        sb.readback_end()
                sb.readback_end()
            run(5)
            best = 1e9
            for _ in range(3):
                sb.profile_begin()
                run(40)
                best = min(best, sb.profile_end() / 40)
            return best, sb.stats()
        finally:
            sb.OnDestroy()
    (t_flush, st_flush), (t_peek, st) = per_tick(False), per_tick(True)
    print(f"192^3, 4 substeps, render-set readback every tick: flushed {t_flush:.4f} ms per tick, peeked {t_peek:.4f}")
    assert st["readback_peek_tiles"] == 24 ** 3 - 22 ** 3 and st["readback_peeks"] == 125
    assert st["ticks_fused"] == 124 and st_flush["ticks_fused"] == 0 and st_flush["readback_peeks"] == 0


def test_small_launches_do_not_peek(monkeypatch):
    """Where every tile of a launch is resident at once a launch is a fixed latency, and a peek (one more launch) costs more than the
    fusion it keeps: by default only tilings of at least 2 048 workgroups peek."""
    monkeypatch.delenv("SB_NO_PEEK", raising=False); monkeypatch.delenv("SB_PEEK_MIN_TILES", raising=False)
    sb = Softbody(jelly_cube(32), substeps=4).Start()
    try:
        for _ in range(3):
            sb.step(); sb.get_positions()
        assert sb.stats()["readback_peeks"] == 0
    finally:
        sb.OnDestroy()
