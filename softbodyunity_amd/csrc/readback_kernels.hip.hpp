// readback_kernels.hip.hpp — render readback: snapshots in caller numbering and area-weighted vertex normals (SPEC.md §6a)
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
#pragma once
#include "device_math.hip.hpp"

namespace sbk {

// Render readback: owned positions (device order, float4) -> caller order, packed xyz.
__global__ __launch_bounds__(256) void snapshot_kernel(PosView pos, const int32_t *local_to_old, float *out_xyz, int n_owned) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n_owned) return;
    const float4 p = pv_load(pos, l);
    const size_t o = 3 * (size_t)local_to_old[l];
    out_xyz[o] = p.x; out_xyz[o + 1] = p.y; out_xyz[o + 2] = p.z;
}

// Snapshot of the render set only: entry subset[k] of the caller-numbered snapshot <- particle local_of_subset[k].
__global__ __launch_bounds__(256) void snapshot_subset_kernel(PosView pos, const int32_t *subset, const int32_t *local_of_subset,
                                                             float *out_xyz, int count) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const float4 p = pv_load(pos, local_of_subset[k]);
    const size_t o = 3 * (size_t)subset[k];
    out_xyz[o] = p.x; out_xyz[o + 1] = p.y; out_xyz[o + 2] = p.z;
}

// The render set of a rank of a partitioned solver (no normals there): compact entry k <- particle local_of_subset[k].
__global__ __launch_bounds__(256) void snapshot_compact_kernel(PosView pos, const int32_t *local_of_subset, float *out_xyz, int count) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const float4 p = pv_load(pos, local_of_subset[k]);
    const size_t o = 3 * (size_t)k;
    out_xyz[o] = p.x; out_xyz[o + 1] = p.y; out_xyz[o + 2] = p.z;
}

// SPEC.md §6a: area-weighted vertex normals on a position snapshot in caller numbering. One lane per vertex gathers its
// incident triangles in ascending order (adj lists built on the host), so the additions happen in the oracle's order.
__global__ __launch_bounds__(256) void normals_kernel(const float *snap_xyz, const int32_t *adj_off, const int32_t *adj_tri,
                                                      const int32_t *tri, float *nrm_xyz, int n, const int32_t *subset,
                                                      float *subset_pos_xyz) {
    // subset == nullptr: lane k handles particle k and writes normal k. Otherwise lane k handles particle subset[k] and
    // writes compact entry k of the normals AND of the positions (the render set travels to the host on its own).
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int v = subset ? subset[k] : k;
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    for (int q = adj_off[v]; q < adj_off[v + 1]; ++q) {
        const int t = adj_tri[q];
        const size_t a = 3 * (size_t)tri[3 * t], b = 3 * (size_t)tri[3 * t + 1], c = 3 * (size_t)tri[3 * t + 2];
        const V3 xa = {snap_xyz[a], snap_xyz[a + 1], snap_xyz[a + 2]};
        const V3 e1 = sub3({snap_xyz[b], snap_xyz[b + 1], snap_xyz[b + 2]}, xa);
        const V3 e2 = sub3({snap_xyz[c], snap_xyz[c + 1], snap_xyz[c + 2]}, xa);
        const V3 f = cross3(e1, e2);
        nx = nx + f.x; ny = ny + f.y; nz = nz + f.z;
    }
    float xx = nx * nx, yy = ny * ny, zz = nz * nz;
    float L2 = (xx + yy) + zz;
    if (L2 >= 0x1p-96f) { float L = sqrt_rn_normal(L2); nx = nx / L; ny = ny / L; nz = nz / L; }
    else { nx = 0.0f; ny = 0.0f; nz = 0.0f; }
    const size_t o = 3 * (size_t)k;
    nrm_xyz[o] = nx; nrm_xyz[o + 1] = ny; nrm_xyz[o + 2] = nz;
    if (subset) {
        const size_t sv = 3 * (size_t)v;
        subset_pos_xyz[o] = snap_xyz[sv]; subset_pos_xyz[o + 1] = snap_xyz[sv + 1]; subset_pos_xyz[o + 2] = snap_xyz[sv + 2];
    }
}

}  // namespace sbk
