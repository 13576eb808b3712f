// tile_kernel.hip.hpp — the fused tile kernel (one launch per substep): rounds -> collide / velocity / integrate -> the same rounds again, tile resident in LDS
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
#pragma once
#include "device_math.hip.hpp"

namespace sbk {

// One workgroup = one tile (or one pack of under-full tiles, solver.hip build_device) of THREADS lanes; a tile owns one
// constraint list, cut into rounds of at most kRoundSlots independent constraints (plan.hpp):
//   KIND 0 (first kernel of a tick)  : MARK: v from the velocity array, integrate; the tile's rounds
//   KIND 1 (every other substep)     : the tile's rounds (finish substep s-1), MARK: v = (x-xprev)/h, integrate
//                                      (start substep s), the same rounds again
//   KIND 2 (after the last substep)  : the tile's rounds, MARK: write v, stop
//   KIND 3 (T2 layer, every substep)  : the tile's rounds once, no MARK; the particles come from an explicit list
//   KIND 5 (a tick's fused first kernel when kinematic targets are pending): KIND 1, and between the velocity of the substep that
//                                      ended and the integrate of the next one a particle with w = 0 that has a target takes it
//   KIND 4 (peek at the tick's end)   : what KIND 2 would leave as positions (the tile's rounds + collide), written to a side
//                                      array; nothing of the state is written, so the deferred last kernel of a tick can still
//                                      be fused with the first kernel of the next one (render readback, solver.hip peek_positions)
// Particles AND the tile's constraint stream are staged in LDS with wide coalesced loads issued together,
// so a tile pays the HBM latency once; rounds then run LDS-to-LDS with one barrier each. Each lane keeps
// ownership of up to PPT particles for the MARK step and projects kRoundSlots / THREADS constraints per round.
// QUADS = the tiling stores 4-vertex rounds.
// Workgroup barrier that orders LDS only (global memory is never exchanged between lanes inside a launch);
// __syncthreads() would also drain vmcnt, i.e. wait for the xprev stores of the MARK step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Register budget (HIP: second launch-bound = waves per SIMD). LDS allows ~14 tiles per CU, so aim for that many waves.
template <bool QUADS, int THREADS> constexpr int kWavesPerSimd = QUADS ? (THREADS == 64 ? 3 : (THREADS == 512 ? 2 : 4)) : (THREADS == 256 ? 8 : (THREADS == 128 ? 6 : 4));   // 128 lanes: 7 waves (72 VGPRs) measured 1 % slower, 8 spill
// WPAL = inverse masses are read as one-byte palette indices (the palette entry comes from another lane by ds_bpermute).
template <int KIND_, bool QUADS, int THREADS, int PPT, bool WPAL, int HALO = kHaloNone>
// (KIND 5 -- one launch per tick, and only when kinematic targets are pending -- gets one wave per SIMD less: its MARK step holds more
// live values, and under the ordinary bound it spilled 4 .. 14 registers)
__global__ __launch_bounds__(THREADS, (kWavesPerSimd<QUADS, THREADS> - (KIND_ == 5 && !QUADS ? 1 : 0))) void tile_kernel(const TileDesc *tiles_at_base, int n_workgroups, TileArgs A) {
    // KIND 5 = KIND 1 whose MARK step also applies pending kinematic targets (see TileArgs::kin_map): an instantiation of its own, so the
    // ordinary mid-tick kernel carries nothing of it
    constexpr int KIND = KIND_ == 5 ? 1 : KIND_;
    constexpr bool KIN = KIND_ == 5;
    constexpr bool GHOSTS = HALO == kHaloGhosts;
    // The first two arguments (3 dwords) are preloaded into SGPRs at dispatch (-mllvm -amdgpu-kernarg-preload-count=3, Makefile):
    // the descriptor fetch starts with the kernel instead of behind the kernel-argument load (one memory round trip less on the
    // latency chain of a small launch). n_workgroups = gridDim.x (reading gridDim would be another kernel-argument load).
    constexpr int kTileThreads = THREADS;
    constexpr int kCPL = THREADS >= kRoundSlots ? 1 : kRoundSlots / THREADS;       // constraints per lane per round (8-wave tiles: QUADS only)
    extern __shared__ uint4 lds_raw[];
    float4 *lds_pos = reinterpret_cast<float4 *>(lds_raw);
    uint32_t *s_rounds = reinterpret_cast<uint32_t *>(lds_pos + A.max_local);
    float *s_pal = reinterpret_cast<float *>(s_rounds + A.rounds_dwords);
    uint32_t *cbuf = s_rounds + A.rounds_dwords + A.pal_dwords;
    f32x3 *lds_spare = reinterpret_cast<f32x3 *>(cbuf + A.win_dwords);   // 16 bytes nobody reads (see the rounds)
    // the tile tables are never written by a kernel: read the descriptor through the constant address space so it
    // stays on the scalar-memory path (s_load), one wide read
    typedef const TileDesc __attribute__((address_space(4))) *ConstTileDescPtr;
    // Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md §Workgroup dispatch; speed only):
    // give each XCD a contiguous range of tiles, so that neighbouring tiles -- whose runs meet inside a 128-byte line
    // wherever a T1 tile's pieces of one T0 tile lie side by side -- read and write those lines through the same L2.
#ifndef SB_NO_XCD_REMAP
    const int nwg = n_workgroups, wg = (int)blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = wg & 7;
    const int tile_index = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wg >> 3);
#else
    const int tile_index = (int)blockIdx.x;
#endif
    const TileDesc __attribute__((address_space(4))) &td = *((ConstTileDescPtr)(uintptr_t)tiles_at_base + tile_index);
    const int n_rounds_all = td.n_rounds;
    const int tid = threadIdx.x;
    const int n_local = td.n_local;
    const int run_count = td.run_count;
    const TickParams tp = *A.tp;
    const uint32_t *tstream = A.stream + td.s_begin;
#if defined(SB_ABLATE) && SB_ABLATE == 4   // timing experiment only: dispatch + descriptor fetch
    if (n_local >= 0) { if (n_rounds_all == 0x7fffffff) A.vel[0] = tp.h; return; }
#endif

    // ---- particle ownership: lane tid owns tile-local particles tid + THREADS*m ------------------
    // Branch-free run lookup: the runs are sorted by their first local index and unused inline entries hold
    // INT_MAX there, so the last run that starts at or before l is found by a compare/select chain on scalars
    // (a loop with a scalar branch per run and slot cost ~400 SALU instructions per wave BEFORE the first load).
    int g[PPT];
    if (KIND == 3) {
#pragma unroll
        for (int m = 0; m < PPT; ++m) {
            const int l = tid + m * kTileThreads;
            g[m] = l < n_local ? A.gather[td.gather_begin + l] : -1;
        }
    } else {
        int run_d[kInlineRuns], run_y[kInlineRuns];
#pragma unroll
        for (int r = 0; r < kInlineRuns; ++r) { run_y[r] = td.runs[r].y; run_d[r] = td.runs[r].x - td.runs[r].y; }
#pragma unroll
        for (int m = 0; m < PPT; ++m) {
            const int l = tid + m * kTileThreads;
            int d = run_d[0];
#pragma unroll
            for (int r = 1; r < kInlineRuns; ++r) d = l >= run_y[r] ? run_d[r] : d;
            int gi = l + d;
            if (run_count > kInlineRuns)
                for (int r = kInlineRuns; r < run_count; ++r) {
                    const int2 rn = A.runs_overflow[td.run_overflow + r - kInlineRuns];
                    if (rn.y <= l) gi = rn.x + (l - rn.y);
                }
            g[m] = l < n_local ? gi : -1;
        }
    }
    // ---- the stretch of the stream this kernel needs, staged through an LDS window -----------------
    // virtual program: rounds 0..R-1 (finish the previous substep), MARK, rounds 0..R-1 again (start the next)
    const int R = n_rounds_all;
    const int v_begin = KIND == 0 ? R : 0;
    const int v_end = KIND == 3 ? R : (KIND == 2 || KIND == 4 ? R + 1 : 2 * R + 1);
    const uint32_t d_lo = td.s_hdr;
    const uint32_t d_hi = td.s_len;
    const uint32_t win = (uint32_t)A.win_dwords;
    uint32_t win_lo = d_lo;
    auto load_window = [&](uint32_t lo) {
        const uint32_t n4 = (min(lo + win, d_hi) - lo) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(tstream + lo);
        uint4 *dst = reinterpret_cast<uint4 *>(cbuf);
        // (the lane's index is laundered: left alone, the compiler hoists the per-lane source address out of the rounds loop and
        // holds it in two registers for the whole kernel -- in the 80-register kernels that was a spill to scratch; refills are rare)
        uint32_t t0 = (uint32_t)tid;
        asm volatile("" : "+v"(t0));
        for (uint32_t i = t0; i < n4; i += 4 * kTileThreads) {
            uint4 v0 = src[i], v1, v2, v3;
            const bool b1 = i + kTileThreads < n4, b2 = i + 2 * kTileThreads < n4, b3 = i + 3 * kTileThreads < n4;
            if (b1) v1 = src[i + kTileThreads];
            if (b2) v2 = src[i + 2 * kTileThreads];
            if (b3) v3 = src[i + 3 * kTileThreads];
            dst[i] = v0;
            if (b1) dst[i + kTileThreads] = v1;
            if (b2) dst[i + 2 * kTileThreads] = v2;
            if (b3) dst[i + 3 * kTileThreads] = v3;
        }
    };
    // Issue everything the tile needs from HBM back to back and BRANCH-FREE (a lane without work reads a valid
    // dummy address): a divergent `if` around a load makes the compiler wait for it at the end of the block,
    // which would serialise the tile's loads into several HBM round trips.
    f32x4 X[PPT];
    float pvx[PPT], pvy[PPT], pvz[PPT];
    uint32_t wi[PPT];
    const int mypal = WPAL ? __float_as_int(A.wpal[tid & (kMaxMassPalette - 1)]) : 0;   // lane l holds palette entry l
#pragma unroll
    for (int m = 0; m < PPT; ++m) {
        const int gc = max(g[m], 0);
        // (GHOSTS: a select of the address, not a branch around the load -- see above)
        const bool ghost = GHOSTS && gc >= A.n_owned;
        const float *px = ghost ? A.ghost_src + 6 * (size_t)(gc - A.n_owned) : A.pos.xyz + 3 * (size_t)gc;
        X[m].x = px[0]; X[m].y = px[1]; X[m].z = px[2];
        if (WPAL) { wi[m] = A.w8[A.w_uniform ? 0 : gc]; X[m].w = 0.0f; } else { wi[m] = 0; X[m].w = A.pos.w[gc]; }
        pvx[m] = pvy[m] = pvz[m] = 0.0f;
        if (KIND != 0 && KIND != 3 && KIND != 4) {
            const float *pp = ghost ? px + 3 : A.prev + 3 * (size_t)gc;
            pvx[m] = pp[0]; pvy[m] = pp[1]; pvz[m] = pp[2];
        }
    }
    const bool rounds_in_lds = n_rounds_all <= A.rounds_dwords;
    const uint32_t rw = tstream[max(min(tid, n_rounds_all - 1), 0)];        // (an empty program still has a 16-byte header)
    // programs of at most 64 rounds: every wave also keeps round word `lane` in a register and reads it back with
    // v_readlane (no LDS round trip at the head of each round): +3 % at 64^3, +1 % at 256^3
    const bool rounds_in_lanes = n_rounds_all <= 64;
    const uint32_t rwl = tstream[max(min(tid & 63, n_rounds_all - 1), 0)];
    // wave items (see kItem*): lane l of a wave holds the wave's step l (tiles without items re-read round word 0)
    uint32_t itreg = 0;
    if (QUADS && kTileThreads == 64 * A.item_waves) {
        const int ns = td.n_steps;
        itreg = tstream[ns > 0 ? td.s_items + (uint32_t)((tid >> 6) * ns + min(tid & 63, ns - 1)) : 0u];
    }
    const int n_pal = td.n_pal;
    // palette follows the round words; lanes without an entry re-read round word 0 (always inside the tile's stream)
    const uint32_t palw = tstream[tid < n_pal ? ((n_rounds_all + 3) & ~3) + tid : 0];
    // uint4 per lane in the first sweep of the window (issued with the particle loads; idle lanes re-read the tile's
    // header, so a sweep nobody needs is a wasted load per lane): 2 cover the 4-byte slots of a 512-particle tile at either
    // width; longer windows finish in the loop below
#ifndef SB_KW
#define SB_KW (QUADS ? 4 : 2)
#endif
    constexpr int kW = SB_KW;
    const bool lane_packed = !QUADS && kTileThreads == kLanePackLanes && td.packed_lanes == (uint32_t)kLanePackLanes;     // (uniform)
    const bool lane_packed_full = !WPAL && lane_packed && td.n_pal == 0;       // per-spring rest lengths behind the index words
    // 256-lane workgroups: one 8-byte word per lane (kWidePack*), loaded by the lane itself below
    const bool wide_packed = !QUADS && kTileThreads == kWidePackLanes && td.packed_lanes == (uint32_t)kWidePackLanes;       // (uniform)
    const bool any_packed = lane_packed || wide_packed;
    // (a lane-packed tile never stages its data in LDS, so the tiling's window need not hold it: its loads are not cut at `win`)
    const uint32_t n4_first = wide_packed ? 0u : lane_packed ? (lane_packed_full ? 2u * (uint32_t)kLanePackLanes : (uint32_t)kLanePackLanes)
                                          : (min(win_lo + win, d_hi) - win_lo) >> 2;
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(tstream + win_lo);
    u32x4 wv[kW];
#pragma unroll
    for (int q = 0; q < kW; ++q) {
        const uint32_t i = tid + q * kTileThreads;
        // idle lanes read the first 16 bytes of the tile's stream (its round words: always present and aligned)
        wv[q] = i < n4_first ? wsrc[i] : *reinterpret_cast<const u32x4 *>(tstream);
    }
    // lane-packed full slots: the rest lengths of the lane's two round-2 slots (8 bytes per lane behind the two 16-byte sweeps); every
    // other tile re-reads its header here (one address for all lanes)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 wc = {0u, 0u};       // (loaded behind the staging barrier, see there)
    // wide-packed slots: the lane's own 8-byte word, issued with the first batch (every other tile re-reads its header: one address for all lanes)
    u32x2 wp = {0u, 0u};
    if (!QUADS && kTileThreads == kWidePackLanes)
        wp = *reinterpret_cast<const u32x2 *>(wide_packed ? tstream + win_lo + 2u * (uint32_t)tid : tstream);
    // One common use of every loaded value: the scheduler cannot sink a load below it, so all loads are issued
    // first and a single wait follows (left alone it emits load, wait, LDS write, load, wait, ... to save registers).
#pragma unroll
    for (int m = 0; m < PPT; ++m) asm volatile("" ::"v"(X[m].x), "v"(pvx[m]), "v"(wi[m]), "v"(X[m].w));
    asm volatile("" ::"v"(mypal));
    if (WPAL) {
#pragma unroll
        for (int m = 0; m < PPT; ++m) X[m].w = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(wi[m] << 2), mypal));
    }
#pragma unroll
    for (int q = 0; q < kW; ++q) asm volatile("" ::"v"(wv[q].x));
    asm volatile("" ::"v"(rw), "v"(palw), "v"(rwl), "v"(itreg));
    if (!QUADS && kTileThreads == kWidePackLanes) asm volatile("" ::"v"(wp.x));
#pragma unroll
    for (int m = 0; m < PPT; ++m)
        if (g[m] >= 0) *reinterpret_cast<f32x4 *>(lds_pos + tid + m * kTileThreads) = X[m];
    if (rounds_in_lds && tid < n_rounds_all) s_rounds[tid] = rw;
    if (tid < n_pal) s_pal[tid] = __uint_as_float(palw);
    if (kTileThreads < kMaxRoundsLds && rounds_in_lds)
        for (int q = tid + kTileThreads; q < n_rounds_all; q += kTileThreads) s_rounds[q] = tstream[q];
    if (kTileThreads < kMaxPalette)
        for (int q = tid + kTileThreads; q < n_pal; q += kTileThreads) s_pal[q] = __uint_as_float(tstream[((n_rounds_all + 3) & ~3) + q]);
    {
        u32x4 *dst = reinterpret_cast<u32x4 *>(cbuf);
#pragma unroll
        for (int q = 0; q < kW; ++q) {
            const uint32_t i = tid + q * kTileThreads;
            if (i < n4_first && !any_packed) dst[i] = wv[q];      // (lane-packed slots stay in wv[0], wide-packed ones in wp)
        }
        if (!any_packed)
            for (uint32_t i = tid + kW * kTileThreads; i < n4_first; i += kTileThreads) dst[i] = wsrc[i];
    }
    __syncthreads();   // also covers the staging loads
    // Issued HERE, not with the first batch: at that point every register of the budget (80 at six waves per SIMD) is in flight, and the
    // two rest lengths are first needed by the tile's THIRD round -- the load's latency hides behind the first two.
    if (!WPAL && !QUADS && kTileThreads == kLanePackLanes && lane_packed_full)
        wc = *reinterpret_cast<const u32x2 *>(tstream + win_lo + 8u * (uint32_t)kLanePackLanes + 2u * (uint32_t)tid);
#if defined(SB_ABLATE) && SB_ABLATE == 5   // timing experiment only: dispatch + descriptor + every load of the tile, nothing else
    if (n_local >= 0) { if (lds_pos[tid].x == 1.2345e-30f && cbuf[tid] == 0x12345678u) A.vel[0] = tp.h; return; }
#endif

    // MARK step (SPEC.md §2): collide + velocity update of the substep that just finished, integrate of the next one
    auto mark_step = [&]() {
#pragma unroll
        for (int m = 0; m < PPT; ++m)
            if (g[m] >= 0) {
                const int l = tid + m * kTileThreads;
                float4 P = lds_pos[l];
                if (KIND != 0 && tp.plane_on && P.w > 0.0f) {   // collide: end of the substep that just finished
                    float a = tp.pnx * P.x, b = tp.pny * P.y, c = tp.pnz * P.z;
                    float pen = ((a + b) + c) - tp.pd;
                    if (pen < 0.0f) {
                        float dx = pen * tp.pnx, dy = pen * tp.pny, dz = pen * tp.pnz;
                        P.x = P.x - dx; P.y = P.y - dy; P.z = P.z - dz;
                        if (KIND == 2 || KIND == 4) lds_pos[l] = P;
                    }
                }
                if (KIND == 4) continue;     // (a peek ends with the collided positions: no velocity, no state write)
                float vx, vy, vz;
                const size_t o = 3 * (size_t)g[m];
                if (KIND == 0) {
                    vx = A.vel[o + 0]; vy = A.vel[o + 1]; vz = A.vel[o + 2];
                } else {
                    float dx = P.x - pvx[m], dy = P.y - pvy[m], dz = P.z - pvz[m];
                    float qx = dx * tp.inv_h, qy = dy * tp.inv_h, qz = dz * tp.inv_h;
                    vx = qx * tp.kd; vy = qy * tp.kd; vz = qz * tp.kd;
                }
                bool moved = false;
                if (KIN && P.w == 0.0f) {     // kinematic particle: SPEC.md 2 -- after the velocity of the tick that ended, before the integrate
                    const int ks = A.kin_map[g[m]];
                    if (ks >= 0) {
                        float *t = A.kin_target + 3 * (size_t)ks;
                        const float tx = t[0], ty = t[1], tz = t[2];
                        if (tx == tx) { P.x = tx; P.y = ty; P.z = tz; moved = true; t[0] = __int_as_float(0x7fc00000); }
                    }
                }
                if (KIND == 2) {
                    A.vel[o + 0] = vx; A.vel[o + 1] = vy; A.vel[o + 2] = vz;
                } else {
                    if (A.store_through & 1) store3_through(A.prev + o, P.x, P.y, P.z);
                    else { A.prev[o + 0] = P.x; A.prev[o + 1] = P.y; A.prev[o + 2] = P.z; }
                    if (P.w > 0.0f) {
                        vx = vx + tp.hgx; vy = vy + tp.hgy; vz = vz + tp.hgz;
                        float hx = tp.h * vx, hy = tp.h * vy, hz = tp.h * vz;
                        P.x = P.x + hx; P.y = P.y + hy; P.z = P.z + hz;
                        lds_pos[l] = P;
                    } else if (KIN && moved) lds_pos[l] = P;
                }
            }
    };

    // Short programs of dictionary-coded distance groups (every tile of a regular mesh: 3 groups) keep their constraint
    // slots and rest lengths in registers: one batch of LDS reads ahead of the first round instead of two dependent LDS
    // round trips (slot, then palette entry) at the head of every round, in both passes. Same constraints, same order.
    constexpr int kRegRounds = THREADS >= 256 ? kRegRoundsWide : kRegRoundsNarrow;       // (kernel_types.hpp: the host packs lanes by the same constants)
    // (round 3: also for tiles whose slots are NOT dictionary-coded -- per-spring rest lengths, {i | j<<16, rest} pairs: the same 2
    // registers per constraint, only the decode differs -- so that a mesh with varied rest lengths keeps the short path)
    if (kRegRounds > 0 && !QUADS && n_rounds_all <= kRegRounds && n_rounds_all > 0 && (kRegFullSlots || n_pal > 0) && (any_packed || d_hi - d_lo <= win)) {
        const bool tile_compact = n_pal > 0;      // (uniform per tile: build_device codes all of a tile's groups one way)
        uint32_t rs[kRegRounds > 0 ? kRegRounds : 1][kCPL];
        float rl[kRegRounds > 0 ? kRegRounds : 1][kCPL];
        int rcnt[kRegRounds > 0 ? kRegRounds : 1];
        {
            uint32_t o = 0;
#pragma unroll
            for (int r = 0; r < kRegRounds; ++r) {
                rcnt[r] = 0;
                if (r < n_rounds_all) {
                    rcnt[r] = (int)((uint32_t)__builtin_amdgcn_readlane((int)rwl, r) & 1023u);
                    if (kCPL == 2 && lane_packed) {
                        // the lane's own 16-byte word, loaded with the window's first sweep: fields 2 r and 2 r + 1
#pragma unroll
                        for (int u = 0; u < kCPL; ++u) {
                            const int bit = kLanePackFieldBits * (2 * r + u), w0 = bit >> 5, sh = bit & 31;
                            const uint32_t lo = wv[0][w0] >> sh;
                            const uint32_t hi = (sh + kLanePackFieldBits > 32) ? (wv[0][w0 + 1 < 4 ? w0 + 1 : 3] << ((32 - sh) & 31)) : 0u;
                            const uint32_t f = (lo | hi) & ((1u << kLanePackFieldBits) - 1u);
                            rs[r][u] = (f & 511u) | (((f >> 9) & 511u) << 12) | ((f >> 18) << 24);
                            if (!WPAL) {     // (full slots: the rest length travels beside the index word; compact tiles overwrite rl from the palette below)
                                const int fld = 2 * r + u;
                                rl[r][u] = __uint_as_float(fld < 4 ? wv[1][fld < 4 ? fld : 0] : wc[fld >= 4 ? fld - 4 : 0]);
                            }
                        }
                    } else if (kCPL == 1 && wide_packed) {
                        // the lane's own 8-byte word: field r (three 21-bit fields)
                        const uint64_t two = (uint64_t)wp.x | ((uint64_t)wp.y << 32);
                        const uint32_t f = (uint32_t)(two >> (kLanePackFieldBits * (r < kLanePackRounds ? r : 0))) & ((1u << kLanePackFieldBits) - 1u);
                        rs[r][0] = (f & 511u) | (((f >> 9) & 511u) << 12) | ((f >> 18) << 24);
                    } else if (tile_compact) {
#pragma unroll
                        for (int u = 0; u < kCPL; ++u) {
                            const int c = tid + u * kTileThreads;
                            rs[r][u] = cbuf[o + (c < rcnt[r] ? c : 0)];
                        }
                        o += ((uint32_t)rcnt[r] + 3u) & ~3u;
                    } else {
#pragma unroll
                        for (int u = 0; u < kCPL; ++u) {
                            const int c = tid + u * kTileThreads;
                            const uint2 e = *reinterpret_cast<const uint2 *>(cbuf + o + 2 * (c < rcnt[r] ? c : 0));
                            // re-code the 16 | 16 bit index pair as the 12 | 12 bit form the rounds decode (tiles of this path hold at
                            // most 512 particles)
                            rs[r][u] = (e.x & 0xfffu) | ((e.x >> 16) << 12);
                            rl[r][u] = __uint_as_float(e.y);
                        }
                        o += (2u * (uint32_t)rcnt[r] + 3u) & ~3u;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kCPL; ++u) { rs[r][u] = 0; rl[r][u] = 0.0f; }
                }
            }
            if (tile_compact) {
#pragma unroll
                for (int r = 0; r < kRegRounds; ++r)
#pragma unroll
                    for (int u = 0; u < kCPL; ++u) rl[r][u] = s_pal[rs[r][u] >> 24];
            }
        }
        auto reg_round = [&](const uint32_t (&e)[kCPL], const float (&L0)[kCPL], int cnt) {
            if (kCPL == 1) {
                if (tid < cnt) {
                    const int i = e[0] & 0xfffu, k = (e[0] >> 12) & 0xfffu;
                    float4 a = lds_pos[i], b = lds_pos[k];
                    if (project_distance(a, b, L0[0], tp.at_d)) { lds_pos[i] = a; lds_pos[k] = b; }
                }
            } else {
                int ci[kCPL], ck[kCPL];
                bool con[kCPL];
                f32x4 ca[kCPL], cb[kCPL];
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    con[u] = tid + u * kTileThreads < cnt;
                    ci[u] = e[u] & 0xfffu; ck[u] = (e[u] >> 12) & 0xfffu;
                    ca[u] = *reinterpret_cast<const f32x4 *>(lds_pos + ci[u]);
                    cb[u] = *reinterpret_cast<const f32x4 *>(lds_pos + ck[u]);
                }
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    float4 a = make_float4(ca[u].x, ca[u].y, ca[u].z, ca[u].w), b = make_float4(cb[u].x, cb[u].y, cb[u].z, cb[u].w);
                    con[u] = project_distance_nobranch(a, b, L0[u], tp.at_d) && con[u];
                    ca[u].x = a.x; ca[u].y = a.y; ca[u].z = a.z; cb[u].x = b.x; cb[u].y = b.y; cb[u].z = b.z;
                }
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    f32x3 *da = con[u] ? reinterpret_cast<f32x3 *>(lds_pos + ci[u]) : lds_spare;
                    f32x3 *db = con[u] ? reinterpret_cast<f32x3 *>(lds_pos + ck[u]) : lds_spare;
                    *da = (f32x3){ca[u].x, ca[u].y, ca[u].z};
                    *db = (f32x3){cb[u].x, cb[u].y, cb[u].z};
                }
            }
            lds_barrier();
        };
#if defined(SB_ABLATE) && SB_ABLATE == 1   // timing experiment only: the launch without its rounds
        constexpr bool kRunRounds = false;
#else
        constexpr bool kRunRounds = true;
#endif
        if (KIND != 0 && kRunRounds) {
#pragma unroll
            for (int r = 0; r < kRegRounds; ++r) if (r < n_rounds_all) reg_round(rs[r], rl[r], rcnt[r]);
        }
        if (KIND != 3) { mark_step(); lds_barrier(); }
        if ((KIND == 0 || KIND == 1) && kRunRounds) {
#pragma unroll
            for (int r = 0; r < kRegRounds; ++r) if (r < n_rounds_all) reg_round(rs[r], rl[r], rcnt[r]);
        }
    } else if (QUADS && kTileThreads == 64 * A.item_waves && td.n_steps > 0 && d_hi - d_lo <= win) {
        // ---- wave items: the whole data of the tile is in the window, every wave walks its own list of steps ----------------
        const int n_steps = td.n_steps;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const uint32_t *items = tstream + td.s_items + (uint32_t)(wave * n_steps);
        float *lds_f = reinterpret_cast<float *>(lds_pos);
        const int q = tid & 3;
        auto run_pass = [&]() {
            if (n_steps > 64) itreg = items[min(lane, n_steps - 1)];      // (the prologue loaded the first 64 steps)
#if defined(SB_ABLATE) && SB_ABLATE == 1   // timing experiment only: the launch without its steps
            if (n_steps >= 0) return;
#endif
#pragma unroll 1
            for (int st = 0; st < n_steps; ++st) {
                if ((st & 63) == 0 && st > 0) itreg = items[min(st + lane, n_steps - 1)];
                const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)itreg, st & 63);
                const uint32_t type = it & 7u;
                const int cnt = (int)((it >> kItemCountShift) & 127u);
                const uint32_t *slots = cbuf + (it >> kItemOffsetShift);
                if (type == kItemBending) {
                    // sixteen lanes per hinge (see project_bending_row): every quad of the row reads the four particles, quad k
                    // writes particle k back
                    const int c = lane >> 4;
                    if (c < cnt) {
                        const uint4 e = *reinterpret_cast<const uint4 *>(slots + 4 * c);
                        const int k = (lane >> 2) & 3;
                        const int o0 = 4 * (int)(e.x & 0xffffu) + q, o1 = 4 * (int)(e.x >> 16) + q;
                        const int o2 = 4 * (int)(e.y & 0xffffu) + q, o3 = 4 * (int)(e.y >> 16) + q;
                        const float P[4] = {lds_f[o0], lds_f[o1], lds_f[o2], lds_f[o3]};
                        float Xk = k == 0 ? P[0] : (k == 1 ? P[1] : (k == 2 ? P[2] : P[3]));
                        const int ok_off = k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3));
#if defined(SB_ABLATE) && SB_ABLATE == 2   // timing experiment only (WRONG results): LDS traffic and barriers without the arithmetic
                        const bool ok = true; Xk += __uint_as_float(e.z) * P[1];
#else
                        const bool ok = project_bending_row(P, __uint_as_float(e.z), __uint_as_float(e.w), tp.at_b, k, (lane & 48) << 2, Xk);
#endif
                        if (ok && q < 3) lds_f[ok_off] = Xk;
                    }
                } else if (type == kItemVolume) {
                    // four lanes per constraint (see project_volume_quad): lane q of a quad reads component q of the four
                    // particles (lane 3: their inverse masses), writes component q back
                    const int c = lane >> 2;
                    if (c < cnt) {
                        const uint4 e = *reinterpret_cast<const uint4 *>(slots + 4 * c);
                        const int o0 = 4 * (int)(e.x & 0xffffu) + q, o1 = 4 * (int)(e.x >> 16) + q;
                        const int o2 = 4 * (int)(e.y & 0xffffu) + q, o3 = 4 * (int)(e.y >> 16) + q;
                        float P[4] = {lds_f[o0], lds_f[o1], lds_f[o2], lds_f[o3]};
#if defined(SB_ABLATE) && SB_ABLATE == 2
                        const bool ok = true; P[0] += __uint_as_float(e.z); P[1] -= P[2]; P[3] += P[0];
#else
                        const bool ok = project_volume_quad(P, __uint_as_float(e.z), tp.at_v);
#endif
                        if (ok && q < 3) { lds_f[o0] = P[0]; lds_f[o1] = P[1]; lds_f[o2] = P[2]; lds_f[o3] = P[3]; }
                    }
                } else if (type != kItemIdle) {
                    if (lane < cnt) {
                        int i, k;
                        float L0;
                        if (type == kItemDistCompact) {
                            const uint32_t e = slots[lane];
                            i = e & 0xfffu; k = (e >> 12) & 0xfffu; L0 = s_pal[e >> 24];
                        } else {
                            const uint2 e = *reinterpret_cast<const uint2 *>(slots + 2 * lane);
                            i = e.x & 0xffffu; k = e.x >> 16; L0 = __uint_as_float(e.y);
                        }
                        float4 a = lds_pos[i], b = lds_pos[k];
#if defined(SB_ABLATE) && SB_ABLATE == 2
                        a.x += L0; b.x -= tp.at_d; lds_pos[i] = a; lds_pos[k] = b;
#else
                        if (project_distance(a, b, L0, tp.at_d)) { lds_pos[i] = a; lds_pos[k] = b; }
#endif
                    }
                }
                if (it & (1u << kItemBarrierBit)) lds_barrier();
            }
        };
        if (KIND != 0) run_pass();
        if (KIND != 3) { mark_step(); lds_barrier(); }
        if (KIND == 0 || KIND == 1) run_pass();
    } else {
    uint32_t off = d_lo;    // dword offset (from the tile's stream start) of the current group's data
#if defined(SB_ABLATE) && SB_ABLATE == 1   // timing experiment only: memory traffic without the rounds
    for (int v = v_begin; v < v_end; v += 100000) {
#else
    for (int v = v_begin; v < v_end; ++v) {
#endif
        if (v == R) { mark_step(); lds_barrier(); continue; }     // (virtual) MARK step between the two passes
        const int r = v < R ? v : v - R - 1;
        if (v == R + 1) off = d_lo;          // the second pass walks the same list again
        uint32_t w;
        if (rounds_in_lanes) w = (uint32_t)__builtin_amdgcn_readlane((int)rwl, __builtin_amdgcn_readfirstlane(r));
        else if (rounds_in_lds) w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rounds[r]);
        else w = (uint32_t)__builtin_amdgcn_readfirstlane((int)tstream[r]);
        // group word: distance | volume << 10 | bending << 20 constraint counts, bit 30 = dictionary-coded distance slots
        const int cnt = w & 1023u, n_vol = (w >> 10) & 1023u, n_bend = (w >> 20) & 1023u;
        const bool compact = (w >> 30) & 1u;
        const uint32_t dsize = compact ? ((cnt + 3u) & ~3u) : ((2u * cnt + 3u) & ~3u);
        const uint32_t size = dsize + 4u * (uint32_t)(n_vol + n_bend);
        if (off + size > win_lo + win || off < win_lo) {   // refill the window (uniform; rare for small tiles)
            lds_barrier();
            win_lo = off;
            load_window(win_lo);
            __syncthreads();
        }
        const uint32_t *base = cbuf + (off - win_lo);
        if (QUADS) {
            // A group may hold constraints of all three types (they share no particle): its hinges, tets and springs go
            // to different WAVES -- wave slot sw covers 16 four-lane constraints or 64 springs, hinges first (the longest
            // projection starts first) -- so a group lasts as long as its slowest type, not the sum of the three.
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int n_wb = (n_bend + 15) >> 4, n_wv = (n_vol + 15) >> 4, n_wd = (cnt + 63) >> 6;
            const uint32_t *qbase = base + dsize;                 // the group's tets, then its hinges
            float *lds_f = reinterpret_cast<float *>(lds_pos);
            const int q = tid & 3;
            // slots are dealt to the waves boustrophedon (row 0: wave 0..NW-1, row 1: NW-1..0, ...), so the wave that got a
            // hinge slot in one row gets the cheapest slot of the next
            constexpr int kWavesPerTile = kTileThreads / 64;
            const int n_slots = n_wb + n_wv + n_wd;
#pragma unroll 1
            for (int row = 0; row * kWavesPerTile < n_slots; ++row) {
                const int sw = row * kWavesPerTile + ((row & 1) ? kWavesPerTile - 1 - wave : wave);
                if (sw >= n_slots) continue;
                if (sw < n_wb + n_wv) {
                    // four lanes per constraint (see project_volume_quad): lane q of a quad reads component q of the four
                    // particles (lane 3: their inverse masses), writes component q back
                    const bool bend = sw < n_wb;
                    const int c = (bend ? sw : sw - n_wb) * 16 + (lane >> 2);
                    if (c < (bend ? n_bend : n_vol)) {
                        const uint4 e = *reinterpret_cast<const uint4 *>(qbase + 4 * (bend ? n_vol + c : c));
                        const int o0 = 4 * (int)(e.x & 0xffffu) + q, o1 = 4 * (int)(e.x >> 16) + q;
                        const int o2 = 4 * (int)(e.y & 0xffffu) + q, o3 = 4 * (int)(e.y >> 16) + q;
                        float P[4] = {lds_f[o0], lds_f[o1], lds_f[o2], lds_f[o3]};
                        const bool ok = bend ? project_bending_quad(P, __uint_as_float(e.z), __uint_as_float(e.w), tp.at_b, q)
                                             : project_volume_quad(P, __uint_as_float(e.z), tp.at_v);
                        if (ok && q < 3) { lds_f[o0] = P[0]; lds_f[o1] = P[1]; lds_f[o2] = P[2]; lds_f[o3] = P[3]; }
                    }
                } else {
                    const int c = (sw - n_wb - n_wv) * 64 + lane;
                    if (c < cnt) {
                        int i, k;
                        float L0;
                        if (compact) {
                            const uint32_t e = base[c];
                            i = e & 0xfffu; k = (e >> 12) & 0xfffu; L0 = s_pal[e >> 24];
                        } else {
                            const uint2 e = *reinterpret_cast<const uint2 *>(base + 2 * c);
                            i = e.x & 0xffffu; k = e.x >> 16; L0 = __uint_as_float(e.y);
                        }
                        float4 a = lds_pos[i], b = lds_pos[k];
                        if (project_distance(a, b, L0, tp.at_d)) { lds_pos[i] = a; lds_pos[k] = b; }
                    }
                }
            }
        } else if (kCPL == 1) {
            if (tid < cnt) {
                int i, k;
                float L0;
                if (compact) {
                    const uint32_t e = base[tid];
                    i = e & 0xfffu; k = (e >> 12) & 0xfffu; L0 = s_pal[e >> 24];
                } else {
                    const uint2 e = *reinterpret_cast<const uint2 *>(base + 2 * tid);
                    i = e.x & 0xffffu; k = e.x >> 16; L0 = __uint_as_float(e.y);
                }
                float4 a = lds_pos[i], b = lds_pos[k];
#if defined(SB_ABLATE) && SB_ABLATE == 2   // timing experiment only: LDS traffic + barriers without the arithmetic
                a.x += L0; b.x -= tp.at_d;
                lds_pos[i] = a; lds_pos[k] = b;
#else
                if (project_distance(a, b, L0, tp.at_d)) { lds_pos[i] = a; lds_pos[k] = b; }
#endif
            }
        } else {
            // kCPL independent constraints per lane: fetch all slots, gather all operands, project, scatter. Lanes
            // past the end of the round re-read its first slot (always present) and drop the result.
            int ci[kCPL], ck[kCPL];
            float cL0[kCPL];
            bool con[kCPL];
            f32x4 ca[kCPL], cb[kCPL];
            if (!compact) {
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = tid + u * kTileThreads;
                    con[u] = c < cnt;
                    const uint2 e = *reinterpret_cast<const uint2 *>(base + 2 * (con[u] ? c : 0));
                    ci[u] = e.x & 0xffffu; ck[u] = e.x >> 16; cL0[u] = __uint_as_float(e.y);
                }
            } else {
                uint32_t ce[kCPL];
#pragma unroll
                for (int u = 0; u < kCPL; ++u) {
                    const int c = tid + u * kTileThreads;
                    con[u] = c < cnt;
                    ce[u] = base[con[u] ? c : 0];
                }
#pragma unroll
                for (int u = 0; u < kCPL; ++u) { ci[u] = ce[u] & 0xfffu; ck[u] = (ce[u] >> 12) & 0xfffu; cL0[u] = s_pal[ce[u] >> 24]; }
            }
#pragma unroll
            for (int u = 0; u < kCPL; ++u) {
                ca[u] = *reinterpret_cast<const f32x4 *>(lds_pos + ci[u]);
                cb[u] = *reinterpret_cast<const f32x4 *>(lds_pos + ck[u]);
            }
#pragma unroll
            for (int u = 0; u < kCPL; ++u) {
                float4 a = make_float4(ca[u].x, ca[u].y, ca[u].z, ca[u].w), b = make_float4(cb[u].x, cb[u].y, cb[u].z, cb[u].w);
                con[u] = project_distance_nobranch(a, b, cL0[u], tp.at_d) && con[u];
                ca[u].x = a.x; ca[u].y = a.y; ca[u].z = a.z; cb[u].x = b.x; cb[u].y = b.y; cb[u].z = b.z;
            }
            // Unconditional stores: an idle lane or a skipped constraint writes to a spare LDS slot instead. With the
            // stores under `if (ok)` the compiler sinks each projection into its own branch and runs them one after
            // the other; this way the independent chains are scheduled together.
#pragma unroll
            for (int u = 0; u < kCPL; ++u) {
                f32x3 *da = con[u] ? reinterpret_cast<f32x3 *>(lds_pos + ci[u]) : lds_spare;
                f32x3 *db = con[u] ? reinterpret_cast<f32x3 *>(lds_pos + ck[u]) : lds_spare;
                *da = (f32x3){ca[u].x, ca[u].y, ca[u].z};
                *db = (f32x3){cb[u].x, cb[u].y, cb[u].z};
            }
        }
        off += size;
#if defined(SB_ABLATE) && SB_ABLATE == 3   // timing experiment only (WRONG results): rounds without the workgroup barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        lds_barrier();
#endif
    }
    }
#pragma unroll
    for (int m = 0; m < PPT; ++m)
        if (g[m] >= 0) {
            const float4 P = lds_pos[tid + m * kTileThreads];
            if (KIND == 4) { float *o = A.peek_out + 3 * (size_t)g[m]; o[0] = P.x; o[1] = P.y; o[2] = P.z; }
            else if (A.store_through & 2) store3_through(A.pos.xyz + 3 * (size_t)g[m], P.x, P.y, P.z);
            else pv_store(A.pos, g[m], P);
        }
}

// (Round 3, VERDICT r2 item 8: ONE persistent launch per tick for small launches -- tiles handing over through per-tile counters in
// memory instead of kernel boundaries -- was built and measured: 64^3 0.125 -> 0.177 ms per tick. A hand-off through memory costs more
// than the dispatch it replaces. Recorded in profiles/r03m_persistent_tick_experiment_negative.txt; the code is in the git history.)

}  // namespace sbk
