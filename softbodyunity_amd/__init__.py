"""softbodyunity_amd — MI355X-native soft-body solver behind the Unity Softbody component API.

Product package: HIP kernels + C-ABI plugin (csrc/, libsoftbody_mi355x.so), the ctypes twin of the C#
P/Invoke layer (native.py), the Softbody component mirror (softbody.py) and the synthetic mesh
generators (mesh.py). The CPU oracle lives in /oracle and is never imported from here.
"""
from .mesh import (SoftbodyMesh, bunny_surrogate, from_tet_mesh, from_triangle_mesh, jelly_cube,  # noqa: F401
                   read_gmsh, read_tetgen)
from .softbody import Softbody, SoftbodyGroup, comm_unique_id  # noqa: F401
