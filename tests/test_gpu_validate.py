"""sb_debug_validate -- the table validator ("race detection"): a GPU kernel re-reads what the tile kernels read (descriptors,
run tables / particle lists, group words, 4- and 8-byte spring slots, four-vertex slots, wave items, after packing, lane dealing and
cost ordering) with their own decoding rules. The only race the design can have is a particle twice in one group or in two tiles of one
launch: every kind of mesh the suite runs must validate clean, every constraint must be seen exactly once, and a planted fault must be
found."""
import numpy as np
import pytest

from softbodyunity_amd import Softbody, bunny_surrogate, jelly_cube, native
from softbodyunity_amd.mesh import from_triangle_mesh

pytestmark = pytest.mark.gpu


def _cloth(n=40):
    xs, ys = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="ij")
    V = np.stack([xs.ravel(), np.zeros(n * n, np.float32), ys.ravel()], axis=1)
    idx = lambda i, j: i * n + j
    F = []
    for i in range(n - 1):
        for j in range(n - 1):
            F += [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)), (idx(i, j), idx(i + 1, j + 1), idx(i, j + 1))]
    return from_triangle_mesh(V, np.array(F, np.int32))[0]


CASES = {
    "cube_dictionary_slots": lambda: (jelly_cube(48), dict()),
    "cube_8_byte_slots": lambda: (jelly_cube(32, heterogeneous=True), dict()),
    "cube_packed_rim_tiles": lambda: (jelly_cube(40), dict(tile_particles=128)),
    "cube_large_tiles": lambda: (jelly_cube(30), dict(tile_particles=1000)),
    "cube_global_colours_only": lambda: (jelly_cube(16), dict(tile_particles=-1)),
    "tets_hinges_wave_items_t2_layers": lambda: (bunny_surrogate(target_verts=8000, seed=3), dict()),
    "cloth_hinges": lambda: (_cloth(), dict(tile_particles=128)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_every_table_validates_clean_and_every_constraint_is_seen_once(case):
    mesh, kw = CASES[case]()
    sb = Softbody(mesh, substeps=4, **kw).Start()
    try:
        rep = sb.validate()
        st = sb.stats()
        assert rep["errors"] == [0] * 6 and rep["first_stage"] == -1, rep
        total = len(mesh.dist_rest) + len(mesh.vol_rest) + len(mesh.bend_rest)
        assert rep["constraints_checked"] == total == sum(st["n_constraints_local"])
        assert rep["tiles_checked"] == st["n_tiles"][0] + st["n_tiles"][1] + st["n_t2_tiles"]
        if kw.get("tile_particles") != -1:
            assert rep["groups_checked"] > 0
        sb.step()                                   # the validator leaves the solver as it found it
        assert np.isfinite(sb.get_positions()).all()
    finally:
        sb.OnDestroy()


@pytest.mark.parametrize("world,dims", [(8, (2, 2, 2)), (3, (3, 1, 1))])
def test_every_rank_of_a_partitioned_solver_validates_clean(world, dims):
    # ghosts, redundant straddling T1 tiles, boundary-first tile order: same rules
    mesh = jelly_cube(32)
    for rank in range(world):
        sb = Softbody(mesh, substeps=4, rank=rank, world=world, part_dims=dims, debug_flags=native.SB_DEBUG_NO_COMM).Start()
        try:
            rep = sb.validate()
            assert rep["errors"] == [0] * 6, (rank, rep)
            assert rep["constraints_checked"] == sum(sb.stats()["n_constraints_local"])
        finally:
            sb.OnDestroy()


def test_partitioned_tet_mesh_with_rcb_validates_clean():
    mesh = bunny_surrogate(target_verts=8000, seed=3)
    for rank in range(4):
        sb = Softbody(mesh, substeps=4, rank=rank, world=4, partition=native.SB_PARTITION_RCB, debug_flags=native.SB_DEBUG_NO_COMM).Start()
        try:
            rep = sb.validate()
            assert rep["errors"] == [0] * 6, (rank, rep)
        finally:
            sb.OnDestroy()


@pytest.mark.parametrize("case", ["cube_dictionary_slots", "cube_8_byte_slots", "tets_hinges_wave_items_t2_layers"])
def test_planted_faults_are_found(case):
    mesh, kw = CASES[case]()
    sb = Softbody(mesh, substeps=4, **kw).Start()
    try:
        dup = sb.validate(inject_fault=1)          # a slot copied over its neighbour: a particle twice in one group
        assert dup["errors"][1] >= 1 and dup["first_stage"] == 0 and dup["first_kind"] in (1, 5) and dup["first_group"] >= 0, dup
        two = sb.validate(inject_fault=2)          # a descriptor copied over its neighbour: two workgroups stage the same particles
        assert two["errors"][2] >= 1 and two["first_stage"] == 0, two
        assert sb.validate()["errors"] == [0] * 6  # the solver's own tables were not touched
        with pytest.raises(native.SoftbodyError):
            sb.validate(inject_fault=7)
    finally:
        sb.OnDestroy()
