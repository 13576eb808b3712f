// aux_kernels.hip.hpp — global-colour kernels, kinematic targets, halo pack / unpack and the peer-store transport kernels
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
#pragma once
#include "device_math.hip.hpp"

namespace sbk {


// Global-colour kernels: one constraint per lane, gather/scatter straight on HBM.
__global__ __launch_bounds__(256) void global_distance_kernel(PosView pos, const int2 *ij, const float *rest, int count,
                                                              const TickParams *tpp) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const float at = tpp->at_d;
    const int2 e = ij[k];
    float4 a = pv_load(pos, e.x), b = pv_load(pos, e.y);
    if (project_distance(a, b, rest[k], at)) { pv_store(pos, e.x, a); pv_store(pos, e.y, b); }
}

__global__ __launch_bounds__(256) void global_quad_kernel(PosView pos, const int4 *idx, const float2 *rest, int count,
                                                          int type, const TickParams *tpp) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int4 e = idx[k];
    float4 p0 = pv_load(pos, e.x), p1 = pv_load(pos, e.y), p2 = pv_load(pos, e.z), p3 = pv_load(pos, e.w);
    const float2 r = rest[k];
    bool ok = type == 1 ? project_volume(p0, p1, p2, p3, r.x, tpp->at_v) : project_bending(p0, p1, p2, p3, r, tpp->at_b);
    if (ok) { pv_store(pos, e.x, p0); pv_store(pos, e.y, p1); pv_store(pos, e.z, p2); pv_store(pos, e.w, p3); }
}

// Kinematic particles (SPEC.md 2): entry k moves particle idx[k] (device numbering) to targets[3k..]; the tables live in pinned host
// memory (a few hundred entries per tick: not worth a copy of their own).
__global__ __launch_bounds__(256) void kinematic_scatter_kernel(float *pos_xyz, const int32_t *idx, const float *targets, int count) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const size_t o = 3 * (size_t)idx[k];
    pos_xyz[o] = targets[3 * (size_t)k]; pos_xyz[o + 1] = targets[3 * (size_t)k + 1]; pos_xyz[o + 2] = targets[3 * (size_t)k + 2];
}

// The same targets handed to the fused first kernel of the next tick instead (tile_kernel KIND 5): entry k goes to its particle's slot.
__global__ __launch_bounds__(256) void kinematic_fill_kernel(const int32_t *kin_map, float *kin_target, const int32_t *idx, const float *targets, int count) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const size_t o = 3 * (size_t)kin_map[idx[k]];
    kin_target[o] = targets[3 * (size_t)k]; kin_target[o + 1] = targets[3 * (size_t)k + 1]; kin_target[o + 2] = targets[3 * (size_t)k + 2];
}



// Halo pack / unpack: a ghost travels as 3 floats (position) or, WITH_PREV, 6 floats (position, previous position:
// the T1 kernels run velocity + integrate on ghosts too). The inverse mass of a ghost is static: uploaded once.
template <bool WITH_PREV>
__global__ __launch_bounds__(256) void halo_pack_kernel(PosView pos, const float *prev, const int32_t *idx, float *buf, int count) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const size_t o = 3 * (size_t)idx[k];
    constexpr int F = WITH_PREV ? 6 : 3;
    float *b = buf + (size_t)F * k;
    // 12-byte vector accesses: one load and one store per array instead of three
    const f32x3 x = *reinterpret_cast<const f32x3 *>(pos.xyz + o);
    // (stores through the L2, see store3_through: these kernels are a few hundred workgroups between two dependent launches)
    if (WITH_PREV) {
        const f32x3 p = *reinterpret_cast<const f32x3 *>(prev + o);
        store3_through(b + 3, p.x, p.y, p.z);
    }
    store3_through(b, x.x, x.y, x.z);
}
template <bool WITH_PREV>
__global__ __launch_bounds__(256) void halo_unpack_kernel(PosView pos, float *prev, const int32_t *idx, const float *buf, int count) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const size_t o = 3 * (size_t)idx[k];
    constexpr int F = WITH_PREV ? 6 : 3;
    const float *b = buf + (size_t)F * k;
    const f32x3 x = *reinterpret_cast<const f32x3 *>(b);
    if (WITH_PREV) {
        const f32x3 p = *reinterpret_cast<const f32x3 *>(b + 3);
        store3_through(prev + o, p.x, p.y, p.z);
    }
    store3_through(pos.xyz + o, x.x, x.y, x.z);
}



__device__ __forceinline__ void peer_wait_at_least(const uint32_t *flag, uint32_t want, uint32_t *error) {
    int spins = 0;
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {     // (uncached word: no cache maintenance)
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kPeerSpinLimit) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}

// A segment holds F floats per ghost (x y z [xprev yprev zprev]), ghosts back to back; one lane moves one ghost (a wave
// covers a contiguous stretch of the segment; 16-byte chunks per lane with four gathers each were no faster).
template <bool WITH_PREV>
__global__ __launch_bounds__(256) void peer_push_kernel(PosView pos, const float *prev, const int32_t *idx, int n_chunks, PeerSlot P) {
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    const uint32_t e = __hip_atomic_load(P.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // one ghost per lane and iteration (24 / 12 contiguous bytes of a segment); the grid is capped (solver.hip) because every
    // workgroup ends with an atomic on ONE word: a thousand of them serialise into more time than the copy itself
    for (int k = blockIdx.x * 256 + tid; k < n_chunks; k += gridDim.x * 256) {
        int j = 0;
#pragma unroll
        for (int q = 1; q < kMaxPeers; ++q) j = (q < P.n_send && k >= P.send_off[q]) ? q : j;
        constexpr int F = WITH_PREV ? 6 : 3;
        const int lk = k - P.send_off[j];
        if (lk >= 0 && lk < P.send_cap[j]) {      // (a neighbour this rank packs for but does not push to leaves a gap in k)
            float *b = P.remote_data[j] + (size_t)(e & 1u) * P.remote_stride[j] + (size_t)F * lk;
            const size_t o = 3 * (size_t)idx[k];
            // System-scope stores (write-through: the data must not linger in this XCD's L2 -- the reader may run on another
            // XCD of the same device, or on another device -- and a release FENCE per wave would write the whole L2 back)
            const f32x3 x = *reinterpret_cast<const f32x3 *>(pos.xyz + o);
            if (WITH_PREV) {       // 24 bytes per ghost, 8-byte aligned: three 64-bit stores
                const f32x3 pv = *reinterpret_cast<const f32x3 *>(prev + o);
                unsigned long long *b8 = reinterpret_cast<unsigned long long *>(b);
                auto pack2 = [](float lo, float hi) { return (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32); };
                __hip_atomic_store(b8 + 0, pack2(x.x, x.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(b8 + 1, pack2(x.z, pv.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(b8 + 2, pack2(pv.y, pv.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                __hip_atomic_store(b + 0, x.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(b + 1, x.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(b + 2, x.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // The segment stores above are write-through, so waiting for this wave's stores (vmcnt 0) orders them before the flag;
    // a release FENCE at system scope would write back the whole L2 (the tile kernels' dirty lines), once per wave.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(P.local + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last) {                                          // every workgroup's stores have landed: raise the flags
        if (tid == 0) __hip_atomic_store(P.local + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < P.n_send) __hip_atomic_store(P.remote_data_flag[tid], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // ... and (this ONE workgroup: a launch full of waiting workgroups would starve the neighbours on a shared device)
        // wait for what the next kernels in the stream need: this exchange's segments from every sender, and the receivers'
        // acknowledgement of the PREVIOUS exchange, which frees the buffer the next exchange will write
        if (tid < P.n_recv) peer_wait_at_least(P.my_data_flag[tid], e, P.error);
        if (tid < P.n_send) peer_wait_at_least(P.my_ack_flag[tid], e - 1u, P.error);
    }
}

template <bool WITH_PREV>
__global__ __launch_bounds__(256) void peer_unpack_kernel(PosView pos, float *prev, const int32_t *idx, int n_chunks, PeerSlot P) {
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    const uint32_t e = __hip_atomic_load(P.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    for (int k = blockIdx.x * 256 + tid; k < n_chunks; k += gridDim.x * 256) {      // one ghost per lane and iteration
        int j = 0;
#pragma unroll
        for (int q = 1; q < kMaxPeers; ++q) j = (q < P.n_recv && k >= P.recv_off[q]) ? q : j;
        constexpr int F = WITH_PREV ? 6 : 3;
        const int lk = k - P.recv_off[j];
        if (lk >= 0 && lk < P.recv_cnt[j]) {
            const float *b = P.my_data[j] + (size_t)(e & 1u) * P.my_stride[j] + (size_t)F * lk;
            const size_t o = 3 * (size_t)idx[k];
            // system-scope loads: past this XCD's L2, where an older copy of the segment may sit
            f32x3 x;
            if (WITH_PREV) {
                const unsigned long long *b8 = reinterpret_cast<const unsigned long long *>(b);
                const unsigned long long q0 = __hip_atomic_load(b8 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long q1 = __hip_atomic_load(b8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long q2 = __hip_atomic_load(b8 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                x.x = __uint_as_float((uint32_t)q0); x.y = __uint_as_float((uint32_t)(q0 >> 32)); x.z = __uint_as_float((uint32_t)q1);
                f32x3 pv;
                pv.x = __uint_as_float((uint32_t)(q1 >> 32)); pv.y = __uint_as_float((uint32_t)q2); pv.z = __uint_as_float((uint32_t)(q2 >> 32));
                store3_through(prev + o, pv.x, pv.y, pv.z);
            } else {
                x.x = __hip_atomic_load(b + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                x.y = __hip_atomic_load(b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                x.z = __hip_atomic_load(b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            store3_through(pos.xyz + o, x.x, x.y, x.z);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's mailbox reads are done
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(P.local + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last) {        // every workgroup has read its part of the mailbox: acknowledge, advance the slot's epoch
        if (tid == 0) {
            __hip_atomic_store(P.local + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.local, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < P.n_recv) __hip_atomic_store(P.remote_ack_flag[tid], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace sbk
