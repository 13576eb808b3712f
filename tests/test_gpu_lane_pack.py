"""Lane-packed spring slots (csrc/kernel_types.hpp kLanePack*, kWidePack*): where every launch of a tiling runs 128-lane workgroups, a register-resident
spring tile stores its slots as one 16-byte word per lane (six 21-bit fields) instead of 4 bytes per slot. Same constraints, same
order, hence the same bits as the oracle; the validator reads the packed form with the kernel's decoding rules. Small meshes are forced
onto the narrow launch (SB_NARROW_MIN_TILES=1) so that the packed path runs here; the 256^3 / 192^3 / 512^3 tests run it by default."""
import numpy as np
import pytest

from helpers import make_oracle
from softbodyunity_amd import Softbody, jelly_cube

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _three_rest_lengths(n):
    """A lattice whose springs have one rest length per axis (palette of 3: the palette field of the packed slots is used)."""
    mesh = jelly_cube(n)
    d = mesh.rest_pos[mesh.dist_ij[:, 1]] - mesh.rest_pos[mesh.dist_ij[:, 0]]
    axis = np.abs(d).argmax(axis=1)
    mesh.dist_rest = np.array([1.0, 1.07, 0.94], np.float32)[axis]
    return mesh


def _distinct_rest(n):
    mesh = jelly_cube(n)
    rng = np.random.default_rng(5)
    mesh.dist_rest = (1.0 + rng.uniform(-0.05, 0.05, len(mesh.dist_rest))).astype(np.float32)
    return mesh


CASES = {
    "cube24": lambda: (jelly_cube(24), dict(substeps=6), True),
    "cube40_tile128_packs_of_rim_tiles": lambda: (jelly_cube(40), dict(substeps=4, tile_particles=128), True),
    "cube20_three_rest_lengths_ground": lambda: (_three_rest_lengths(20), dict(substeps=6, ground_plane=(0, 1, 0, -3.0), damping=0.05), True),
    # per-particle masses + per-spring rest lengths: the full form (rest lengths behind the index words, 40 bytes per lane)
    "cube16_heterogeneous_full_slots": lambda: (jelly_cube(16, heterogeneous=True), dict(substeps=4, ground_plane=(0, 1, 0, -2.0)), True),
    "cube24_heterogeneous_tile128": lambda: (jelly_cube(24, heterogeneous=True), dict(substeps=4, tile_particles=128), True),
    # per-spring rest lengths but ONE mass: those kernels (inverse masses as palette indices) do not carry the full form
    "cube16_distinct_rest_uniform_mass_not_packable": lambda: (_distinct_rest(16), dict(substeps=4), False),
    "cube30_large_tiles_not_packable": lambda: (jelly_cube(30), dict(substeps=4, tile_particles=1000), False),
    # round 4: the 8-byte form for 256-lane workgroups (kWidePack*: three 21-bit fields per lane) -- tilings of more than 768 and fewer than
    # 10 240 workgroups, i.e. mid-size meshes and the ranks of a partitioned solver; no switch forced here, these meshes get it by default
    # (a tile is packed only where 2 KiB is LESS than its 4-byte slots: the full 512-particle tiles, not the rim packs of the shifted tiling)
    "wide_cube80": lambda: (jelly_cube(80), dict(substeps=4), True),
    "wide_cube80_three_rest_lengths_ground": lambda: (_three_rest_lengths(80), dict(substeps=4, ground_plane=(0, 1, 0, -3.0), damping=0.05), True),
    "wide_cube80_heterogeneous_not_packable": lambda: (jelly_cube(80, heterogeneous=True), dict(substeps=4), False),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_lane_packed_tiles_match_the_oracle_and_validate(case, monkeypatch, oracle_mod):
    mesh, kw, expect_packed = CASES[case]()
    if case.startswith("wide_"):
        monkeypatch.delenv("SB_NARROW_MIN_TILES", raising=False)
    else:
        monkeypatch.setenv("SB_NARROW_MIN_TILES", "1")
    monkeypatch.delenv("SB_NO_WIDE_SLOTS", raising=False)
    monkeypatch.delenv("SB_NO_LANE_PACK", raising=False)
    sb = Softbody(mesh, **kw).Start()
    try:
        st = sb.stats()
        packed = sum(st["lane_packed_tiles"])
        assert (packed > 0) == expect_packed, st["lane_packed_tiles"]
        if case.startswith("wide_"):
            assert 768 < st["n_tiles"][0] < 10240                            # the regime of the 256-lane launches
            if expect_packed:      # the shifted tiling's rim packs hold fewer than 512 slots: 2 KiB would be MORE than their 4-byte slots
                assert st["lane_packed_tiles"][0] == st["n_tiles"][0] and 0 < st["lane_packed_tiles"][1] < st["n_tiles"][1]
        if expect_packed and "tile128" not in case and "tile64" not in case:      # (small tiles: some rim packs zip into more than three rounds)
            assert st["lane_packed_tiles"][0] == st["n_tiles"][0]          # every full T0 tile qualifies
        okw = {k: v for k, v in kw.items() if k in ("ground_plane", "damping")}
        o = make_oracle(oracle_mod, mesh, sb.plan(), **okw)
        for t in range(4):
            sb.step(); o.step(0.02, kw["substeps"])
            if t == 1:
                assert np.array_equal(_bits(sb.get_positions()), _bits(o.x))      # (a read between ticks: flush or peek on the packed tiling)
        assert np.array_equal(_bits(sb.get_positions()), _bits(o.x)) and np.array_equal(_bits(sb.get_velocities()), _bits(o.v))
        rep = sb.validate()
        assert rep["errors"] == [0] * 6 and rep["constraints_checked"] == len(mesh.dist_rest), rep
        if expect_packed:
            dup = sb.validate(inject_fault=1)
            assert dup["errors"][1] >= 1 and dup["first_stage"] == 0, dup
            assert sb.validate(inject_fault=2)["errors"][2] >= 1
    finally:
        sb.OnDestroy()


def test_packed_and_unpacked_builds_of_one_mesh_agree_and_the_packed_one_is_smaller(monkeypatch):
    mesh = jelly_cube(32)
    monkeypatch.setenv("SB_NARROW_MIN_TILES", "1")

    def run(pack):
        if pack:
            monkeypatch.delenv("SB_NO_LANE_PACK", raising=False)
        else:
            monkeypatch.setenv("SB_NO_LANE_PACK", "1")
        sb = Softbody(mesh, substeps=8).Start()
        try:
            for _ in range(5):
                sb.step()
            return sb.get_positions().copy(), sb.get_velocities().copy(), sb.stats()
        finally:
            sb.OnDestroy()
    xa, va, sa = run(True)
    xb, vb, sb_ = run(False)
    assert np.array_equal(_bits(xa), _bits(xb)) and np.array_equal(_bits(va), _bits(vb))
    assert sum(sa["lane_packed_tiles"]) > 0 and sum(sb_["lane_packed_tiles"]) == 0
    # 16 bytes per lane against 4 bytes per slot: a full 512-particle tile 2 KiB instead of 3 KiB
    assert sa["launch_bytes"][0] < sb_["launch_bytes"][0] - 900 * sa["lane_packed_tiles"][0]


def test_wide_packed_and_unpacked_builds_agree_and_the_packed_one_is_smaller(monkeypatch):
    mesh = jelly_cube(80)
    monkeypatch.delenv("SB_NARROW_MIN_TILES", raising=False)

    def run(pack):
        if pack:
            monkeypatch.delenv("SB_NO_WIDE_SLOTS", raising=False)
        else:
            monkeypatch.setenv("SB_NO_WIDE_SLOTS", "1")
        sb = Softbody(mesh, substeps=6).Start()
        try:
            for _ in range(3):
                sb.step()
            return sb.get_positions().copy(), sb.get_velocities().copy(), sb.stats()
        finally:
            sb.OnDestroy()
    xa, va, sa = run(True)
    xb, vb, sb_ = run(False)
    assert np.array_equal(_bits(xa), _bits(xb)) and np.array_equal(_bits(va), _bits(vb))
    assert sa["lane_packed_tiles"][0] == sa["n_tiles"][0] == 1000 and sum(sb_["lane_packed_tiles"]) == 0
    # 8 bytes per lane of a 256-lane workgroup against 4 bytes per slot: a full 512-particle tile 2 KiB instead of 3 KiB
    assert sa["launch_bytes"][0] < sb_["launch_bytes"][0] - 900 * sa["lane_packed_tiles"][0]


@pytest.mark.parametrize("world,narrow", [(2, True), (4, False)])
def test_packed_slots_on_the_ranks_of_a_partitioned_solver(world, narrow, monkeypatch, oracle_mod):
    # round 4: a rank of a partitioned solver packs its slots too -- the 16-byte form where its launches are 128 lanes wide (forced here; 256^3 on
    # 2 ranks by itself), the 8-byte form in the 256-lane regime (a rank's 1 000+ tiles here; 256^3 on 4 or 8 ranks by itself). T1 launches
    # read their ghosts straight from the receive buffer (fused unpack) with the packed tiles in them. 2 / 4 ranks on the one GPU, the host as the wire.
    from hosted import HostedRanks
    if narrow:
        monkeypatch.setenv("SB_NARROW_MIN_TILES", "1")
    else:
        monkeypatch.delenv("SB_NARROW_MIN_TILES", raising=False)
    mesh = jelly_cube(96 if narrow else 128)
    with HostedRanks(mesh, world, 4) as H:
        st = [sb.stats() for sb in H.ranks]
        assert all(sum(q["lane_packed_tiles"]) > 0 for q in st), [q["lane_packed_tiles"] for q in st]
        assert all(q["halo_unpack_fused"] == 1 for q in st)
        if not narrow:
            assert all(768 < q["n_tiles"][0] < 10240 for q in st)
        o = make_oracle(oracle_mod, mesh, H.ranks[0].plan())
        for _ in range(2):
            H.tick(); o.step(0.02, 4, parallel=True)
        x, v, ghosts = H.merged_state()
        assert ghosts > 0 and np.array_equal(_bits(x), _bits(o.x)) and np.array_equal(_bits(v), _bits(o.v))
        for sb in H.ranks:
            rep = sb.validate()
            assert rep["errors"] == [0] * 6, rep


def _random_spring_cloud(seed, n, k, rest_levels, uniform_mass):
    """A jittered cloud with k-nearest-neighbour springs: particle degrees (hence rounds per tile) vary, so that with the narrow launch
    forced some tiles qualify for lane packing (at most three rounds) and others do not -- both kinds inside ONE launch."""
    from scipy.spatial import cKDTree
    from softbodyunity_amd.mesh import SoftbodyMesh
    rng = np.random.default_rng(seed)
    side = int(round(n ** (1 / 3)))
    g = np.stack(np.meshgrid(*[np.arange(side, dtype=np.float64)] * 3, indexing="ij"), -1).reshape(-1, 3)
    rest = (g + rng.uniform(-0.3, 0.3, g.shape)).astype(np.float32)
    _, nb = cKDTree(rest).query(rest, k=k + 1)
    pairs = set()
    for i in range(len(rest)):
        for j in nb[i, 1:]:
            if rng.random() < 0.6:
                pairs.add((min(i, int(j)), max(i, int(j))))
    ij = np.array(sorted(pairs), np.int32)
    L = np.linalg.norm(rest[ij[:, 1]].astype(np.float64) - rest[ij[:, 0]], axis=1)
    if rest_levels:        # a handful of rest lengths: dictionary-coded tiles (palette <= 8 where a tile sees few of them)
        L = np.round(L * rest_levels) / rest_levels
        L[L == 0] = 1.0 / rest_levels
    w = np.ones(len(rest), np.float32) if uniform_mass else rng.uniform(0.5, 2.0, len(rest)).astype(np.float32)
    w[rng.random(len(rest)) < 0.02] = 0.0
    pos = (rest + rng.normal(0, 0.03, rest.shape)).astype(np.float32)
    return SoftbodyMesh(rest_pos=rest, pos=pos, vel=np.zeros_like(pos), inv_mass=w, dist_ij=ij, dist_rest=L.astype(np.float32))


@pytest.mark.parametrize("seed,k,rest_levels,uniform_mass,tile", [(1, 3, 4, True, 64), (2, 4, 6, True, 128), (3, 3, 0, False, 64), (4, 5, 0, False, 256),
                                                                   (5, 2, 3, False, 128)])
def test_packed_and_unpacked_tiles_mixed_in_one_launch(seed, k, rest_levels, uniform_mass, tile, monkeypatch, oracle_mod):
    mesh = _random_spring_cloud(seed, 4000, k, rest_levels, uniform_mass)
    monkeypatch.setenv("SB_NARROW_MIN_TILES", "1")
    monkeypatch.delenv("SB_NO_LANE_PACK", raising=False)
    kw = dict(substeps=5, tile_particles=tile, distance_compliance=1e-6, ground_plane=(0, 1, 0, -1.0), damping=0.1)
    sb = Softbody(mesh, **kw).Start()
    try:
        st = sb.stats()
        o = make_oracle(oracle_mod, mesh, sb.plan(), damping=0.1, compliance=(1e-6, 0.0, 0.0), ground_plane=kw["ground_plane"])
        for _ in range(4):
            sb.step(); o.step(0.02, 5)
        assert np.array_equal(_bits(sb.get_positions()), _bits(o.x)) and np.array_equal(_bits(sb.get_velocities()), _bits(o.v))
        rep = sb.validate()
        assert rep["errors"] == [0] * 6 and rep["constraints_checked"] == len(mesh.dist_rest), rep
        print(f"seed {seed}: tiles {st['n_tiles']} lane-packed {st['lane_packed_tiles']} T2 layers {st['n_t2_layers']} global colours {st['n_global_colours']}")
    finally:
        sb.OnDestroy()


def test_the_mixed_cases_really_mix(monkeypatch):
    # at least one of the clouds above must put packed and unpacked tiles into the same launch (else the test above proves less than it says)
    monkeypatch.setenv("SB_NARROW_MIN_TILES", "1")
    monkeypatch.delenv("SB_NO_LANE_PACK", raising=False)
    mixed = 0
    for seed, k, rest_levels, uniform_mass, tile in [(4, 5, 0, False, 256), (2, 4, 6, True, 128)]:
        sb = Softbody(_random_spring_cloud(seed, 4000, k, rest_levels, uniform_mass), substeps=5, tile_particles=tile).Start()
        try:
            st = sb.stats()
            for tl in (0, 1):
                if 0 < st["lane_packed_tiles"][tl] < st["n_tiles"][tl]:
                    mixed += 1
        finally:
            sb.OnDestroy()
    assert mixed >= 1
