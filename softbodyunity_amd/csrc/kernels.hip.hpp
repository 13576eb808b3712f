// kernels.hip.hpp — gfx950 device code of the soft-body hot path (SPEC.md §2, §4-§6).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
// Every arithmetic statement mirrors oracle/oracle.c one operation at a time; the file is compiled
// with -ffp-contract=off and correctly-rounded fp32 divide/sqrt so results are bit-identical to the
// oracle. MFMA is not used: the path is a bandwidth-bound gather/scatter (BASELINE.json:5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbk {

struct TickParams {           // SPEC.md §2 host-side scalars, uploaded once per (dt, S)
    float h, inv_h, hgx, hgy, hgz, kd, at_d, at_v, at_b;
    float pnx, pny, pnz, pd;  // ground plane n.x >= d (SPEC.md §2 step 2b)
    int32_t plane_on;
    float pad[2];
};

// One LDS tile (or pack of tiles) = one workgroup. 128 B, read with scalar loads.
// The tile's constraint stream lives at stream[s_begin ...], 16-byte aligned, in dwords:
//   [round words, padded to 4] [rest-length palette, padded to 4] [rounds' data]
// group word: bits 0-9 distance, 10-19 volume, 20-29 bending constraint count (each <= 256), bit 30 = the distance slots
// are dictionary-coded. A group's constraints share no particle (one barrier per group). Its data: the distance slots,
// count x {i | j<<16, rest length}, or -- when a tile's distance constraints use at most 256 distinct rest lengths
// (regular meshes) -- count x {i | j<<12 | palette index<<24} (one dword each) with the values in the tile's palette,
// padded to 4 dwords; then the volume slots, then the bending slots, each {i0|i1<<16, i2|i3<<16, rest.x, rest.y}.
struct TileDesc {
    int32_t n_local, run_count, n_rounds;
    int32_t gather_begin;      // KIND 3 (T2 tiles): the tile's particles are gather[gather_begin .. +n_local) instead of runs
    uint32_t s_begin;          // dword offset of the tile's stream
    uint32_t s_hdr;            // dwords of round words + palette (each padded to 4): round data starts at s_begin + s_hdr
    uint32_t s_len;            // total dwords (multiple of 4)
    uint32_t packed_lanes;     // 0, or 128: LANE-PACKED slots (see kLanePack*) for workgroups of that many lanes -- the launch must use that width
    int32_t run_overflow;      // runs beyond kInlineRuns live at runs_overflow[run_overflow ...]
    int32_t n_pal;             // palette entries (0 = no dictionary coding), stored after the round words
    int32_t n_steps;           // wave items per wave (meshes with tets / hinges; 0 = none), see kItem* below
    uint32_t s_items;          // dword offset (from s_begin) of the items: kItemWaves runs of n_steps dwords, inside the header
    int2 runs[10];             // {first particle (device numbering), first tile-local index}; unused entries: {0, INT_MAX}
};
constexpr int kInlineRuns = 10;
// Lane-packed slots (round 3, second session). A register-resident spring tile run by 128-lane workgroups gives every lane two slots in
// each of its (at most three) rounds and keeps them in registers for both passes; the slots need 9 + 9 bits of tile-local indices and a
// palette index. Stored as ONE 16-byte word per lane -- six 21-bit fields {i:9 | j:9 | palette:3}, field 2 r + u = the lane's slot u of
// round r (constraint lane + 128 u of that round; beyond the round's count the field is 0) -- the tile's data is 2 KiB instead of 3 KiB
// (4 bytes per slot), it arrives in the lane's first window load and never touches LDS. build_device decides per tile (solver.hip).
// A tile whose slots are NOT dictionary-coded (per-spring rest lengths) packs the same way with its rest values behind the index words:
// [128 x 16 B index word][128 x 16 B: rest of fields 0..3][128 x 8 B: rest of fields 4, 5] = 40 bytes per lane instead of 48
// (n_pal == 0 marks this form; only in the kernels that read inverse masses as floats, WPAL = false -- the host packs accordingly).
constexpr int kLanePackLanes = 128, kLanePackRounds = 3, kLanePackFieldBits = 21, kLanePackMaxPalette = 8;
constexpr uint32_t kLanePackDwordsCompact = 4 * kLanePackLanes, kLanePackDwordsFull = 10 * kLanePackLanes;
// Wave items (meshes with tets / hinges, 4-wave tiles): the host deals every group's work to the four waves ahead of time --
// hinges first, then tets (16 four-lane constraints per wave), then springs (64 per wave), boustrophedon over the rows of
// a group -- and stores for every wave one dword per STEP (= one row of one group): what to project, how many, where the
// slots lie in the tile's data, and whether the group ends here (workgroup barrier). The kernel's loop over a list is then a
// v_readlane, three bit-field extracts and one uniform branch per step instead of decoding the group word, the window test
// and the slot arithmetic (90 scalar instructions and a dozen branches per group on the critical path of every group).
constexpr int kItemWaves = 4;        // dealt for 4-wave tiles (8 with SB_QUAD_LANES=512)
constexpr uint32_t kItemIdle = 0, kItemDistCompact = 1, kItemDistFull = 2, kItemVolume = 3, kItemBending = 4;
constexpr int kItemCountShift = 3, kItemBarrierBit = 10, kItemOffsetShift = 11;     // type:3 | count:7 | barrier:1 | dword offset:21
constexpr int kMaxRoundsLds = 128;   // round words cached in LDS; longer programs read them from memory
constexpr int kMaxPalette = 256;     // rest-length dictionary entries per tile

// Positions in HBM: packed xyz (12 B) + the static inverse mass in a side array (never rewritten).
typedef float f32x3_t __attribute__((ext_vector_type(3)));
struct PosView {
    float *xyz;        // 3 floats per local particle
    const float *w;    // inverse mass per local particle
};
__device__ __forceinline__ float4 pv_load(const PosView &P, int g) {
    const size_t o = 3 * (size_t)g;
    return make_float4(P.xyz[o], P.xyz[o + 1], P.xyz[o + 2], P.w[g]);
}
__device__ __forceinline__ void pv_store(const PosView &P, int g, const float4 &v) {
    const size_t o = 3 * (size_t)g;
    P.xyz[o] = v.x; P.xyz[o + 1] = v.y; P.xyz[o + 2] = v.z;
}

// 12-byte store that writes through the L2 (sc0 sc1). The 8 XCDs' L2s are not coherent with each other, so a kernel ends with a
// write-back of every dirty line; in a launch of a few thousand tiles (every tile resident at once, the kernel a chain of
// latencies) that write-back is a third of the kernel -- 64^3: a launch with its rounds removed takes 5.9 us, of which 3.2 us are
// the MARK step's and the final stores plus the end of the kernel. Written through, the state leaves the chip while the kernel
// still runs (64^3: 7.9 -> 6.1 us per launch). Large launches keep ordinary stores (256^3: write-through is 4 % slower).
__device__ __forceinline__ void store3_through(float *p, float x, float y, float z) {
    f32x3_t v = {x, y, z};
    asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

constexpr int kMaxMassPalette = 64;   // distinct inverse masses that fit the one-byte-per-particle coding (one per lane)

struct TileArgs {
    PosView pos;              // packed xyz + inverse mass per local particle
    const uint8_t *w8;        // WPAL kernels: index of the particle's inverse mass in wpal (1 B instead of 4 B per read)
    const float *wpal;        // kMaxMassPalette entries
    float *prev;              // packed xyz
    float *vel;               // packed xyz (read by KIND 0, written by KIND 2)
    const TileDesc *tiles;
    const int2 *runs_overflow;
    const uint32_t *stream;
    const TickParams *tp;
    const int32_t *gather;    // KIND 3: particle lists of the sparse T2 tiles (device numbering)
    int32_t max_local;        // LDS carve: [max_local float4][rounds_dwords][pal_dwords][win_dwords][16 spare bytes]
    int32_t rounds_dwords;    // round words cached in LDS (multiple of 4, <= kMaxRoundsLds); longer programs read memory
    int32_t pal_dwords;       // largest rest-length dictionary of the tiling, padded to 4 (0 = none)
    int32_t win_dwords;       // constraint window held in LDS (multiple of 4, >= the largest round)
    int32_t tile_base;        // this launch covers tiles tile_base + blockIdx.x (boundary / interior split)
    int32_t w_uniform;        // WPAL kernels: every particle has the same inverse mass -> all lanes read index 0 (one cache line, no per-particle byte)
    int32_t item_waves;       // waves per tile the wave items of this tiling were dealt for (0 = none)
    int32_t store_through;    // bit 0: previous positions, bit 1: positions are stored through the L2 (small launches, see store3_through)
    // GHOSTS kernels (world > 1, T1 launches of a lattice-type plan): particle g >= n_owned is ghost g - n_owned and its position
    // and previous position are read straight from the receive buffer of the exchange that just ended (6 floats per ghost, in
    // ghost order) instead of from the arrays -- the unpack kernel between the exchange and this launch is gone
    const float *ghost_src;
    int32_t n_owned;
    // KIND 4 (peek): where the tick-end positions of the launch's tiles go (packed xyz, device numbering); the state arrays stay as
    // they are
    float *peek_out;
    // KIND 5 (kinematic targets inside the fused tick boundary): kin_map[g] = slot of pinned particle g (-1: free particle),
    // kin_target[3 slot ..] = its pending target or NaN; the lane that applies a target writes NaN back
    const int32_t *kin_map;
    float *kin_target;
};
// (A PACK variant -- T0 tiles writing the send buffer themselves, entries {tile-local index, send slot} per tile -- was built and
// measured in round 3: bit-exact, but 0.786 -> 0.861 ms per tick in the serialised W = 8 loopback schedule and no change in the
// overlapped one (profiles/r03g_loopback_w8_fused_pack_ab_not_kept.txt); removed.)
constexpr int kHaloNone = 0, kHaloGhosts = 1;

// Lanes per tile (template parameter THREADS of tile_kernel). A round holds up to kRoundSlots independent constraints,
// so a lane projects kRoundSlots / THREADS of them per round. Fewer waves per tile = more tiles resident per CU (the
// wave slots, not LDS, cap a 256-lane workgroup at 8 tiles per CU; 128 lanes reach the 12-14 that LDS allows): +4.4 %
// at 256^3. A launch whose tiles all fit on the chip at once is latency-bound instead and wants the wide workgroup
// (64^3: 256 lanes are 22 % faster), so the host picks per launch (solver.hip launch_tile).
constexpr int kNarrowTileThreads = 128;                     // small tiles, launches that oversubscribe the chip
constexpr int kQuadTileThreads = 512;                       // tiles with four-lane constraints (tets, hinges): 8 wave slots per group row
constexpr int kWideTileThreads = 256;                       // small tiles in latency-bound launches, and all large tiles
constexpr int kRoundSlots = 256;                            // plan.hpp kRoundThreads
constexpr int kSmallTile = 512, kLargeTile = 1024;          // the two particle capacities the kernels are built for
typedef float f32x3 __attribute__((ext_vector_type(3)));

typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vectors: one 16-byte load/store, no struct copies
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    float t0 = a.y * b.z, t1 = a.z * b.y, t2 = a.z * b.x, t3 = a.x * b.z, t4 = a.x * b.y, t5 = a.y * b.x;
    return {t0 - t1, t2 - t3, t4 - t5};
}
__device__ __forceinline__ float dot3(V3 a, V3 b) {
    float xx = a.x * b.x, yy = a.y * b.y, zz = a.z * b.z;
    return (xx + yy) + zz;
}
__device__ __forceinline__ V3 addscaled3(V3 x, float s, V3 g) {
    float a = s * g.x, b = s * g.y, c = s * g.z;
    return {x.x + a, x.y + b, x.z + c};
}
__device__ __forceinline__ V3 xyz(float4 p) { return {p.x, p.y, p.z}; }

// Correctly rounded sqrt for x >= 2^-96 (also right for +inf; NaN stays NaN): v_sqrt_f32 is good to 1 ulp, the two
// residuals pick the neighbour when it is closer. This is the compiler's own sqrtf expansion without its rescaling
// of tiny arguments and its special-case select (7 VALU instructions fewer per constraint); SPEC.md §4 skips the
// constraints whose argument would need them.
__device__ __forceinline__ float sqrt_rn_normal(float x) {
#ifdef SB_LIBM_SQRT   // A/B timing builds only
    return sqrtf(x);
#endif
    float s = __builtin_amdgcn_sqrtf(x);
    float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    return ru > 0.0f ? su : s;
}

// SPEC.md §4. Returns false when the constraint is skipped.
__device__ __forceinline__ bool project_distance(float4 &a, float4 &b, float L0, float at) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float L2 = (xx + yy) + zz;
    float ws = (a.w + b.w) + at;
    if (!(L2 >= 0x1p-96f) || !(ws > 0.0f)) return false;
    float L = sqrt_rn_normal(L2);
    float C = L - L0;
    float wl = ws * L;
    float s = (-C) / wl;
    float si = a.w * s, sj = b.w * s;
    float ax = si * dx, ay = si * dy, az = si * dz;
    float bx = sj * dx, by = sj * dy, bz = sj * dz;
    a.x = a.x + ax; a.y = a.y + ay; a.z = a.z + az;
    b.x = b.x - bx; b.y = b.y - by; b.z = b.z - bz;
    return true;
}

// Same arithmetic without the early return (results of skipped constraints are simply not stored): lets the
// compiler interleave the independent projections a lane performs in one round.
__device__ __forceinline__ bool project_distance_nobranch(float4 &a, float4 &b, float L0, float at) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float L2 = (xx + yy) + zz;
    float ws = (a.w + b.w) + at;
    const bool ok = (L2 >= 0x1p-96f) && (ws > 0.0f);
    float L = sqrt_rn_normal(L2);
    float C = L - L0;
    float wl = ws * L;
    float s = (-C) / wl;
    float si = a.w * s, sj = b.w * s;
    float ax = si * dx, ay = si * dy, az = si * dz;
    float bx = sj * dx, by = sj * dy, bz = sj * dz;
    a.x = a.x + ax; a.y = a.y + ay; a.z = a.z + az;
    b.x = b.x - bx; b.y = b.y - by; b.z = b.z - bz;
    return ok;
}

// SPEC.md §5.
__device__ __forceinline__ bool project_volume(float4 &p0, float4 &p1, float4 &p2, float4 &p3, float R6, float at_v) {
    V3 x0 = xyz(p0), x1 = xyz(p1), x2 = xyz(p2), x3 = xyz(p3);
    V3 e1 = sub3(x1, x0), e2 = sub3(x2, x0), e3 = sub3(x3, x0);
    V3 g1 = cross3(e2, e3), g2 = cross3(e3, e1), g3 = cross3(e1, e2);
    V3 g0;
    { float t = g1.x + g2.x; t = t + g3.x; g0.x = -t; }
    { float t = g1.y + g2.y; t = t + g3.y; g0.y = -t; }
    { float t = g1.z + g2.z; t = t + g3.z; g0.z = -t; }
    float C6 = dot3(e1, g1) - R6;
    float a0 = p0.w * dot3(g0, g0), a1 = p1.w * dot3(g1, g1), a2 = p2.w * dot3(g2, g2), a3 = p3.w * dot3(g3, g3);
    float den = (((a0 + a1) + a2) + a3) + at_v;
    if (!(den > 0.0f)) return false;
    float s = (-C6) / den;
    x0 = addscaled3(x0, p0.w * s, g0); x1 = addscaled3(x1, p1.w * s, g1);
    x2 = addscaled3(x2, p2.w * s, g2); x3 = addscaled3(x3, p3.w * s, g3);
    p0.x = x0.x; p0.y = x0.y; p0.z = x0.z; p1.x = x1.x; p1.y = x1.y; p1.z = x1.z;
    p2.x = x2.x; p2.y = x2.y; p2.z = x2.z; p3.x = x3.x; p3.y = x3.y; p3.z = x3.z;
    return true;
}

// SPEC.md §6. rest = (cos phi0, sin phi0).
__device__ __forceinline__ bool project_bending(float4 &pa, float4 &pb, float4 &pc, float4 &pd, float2 rest, float at_b) {
    V3 xa = xyz(pa), xb = xyz(pb), xc = xyz(pc), xd = xyz(pd);
    V3 e = sub3(xb, xa);
    float el2 = dot3(e, e);
    float el = sqrtf(el2);
    V3 ac = sub3(xa, xc), bc = sub3(xb, xc), bd = sub3(xb, xd), ad = sub3(xa, xd);
    V3 n1 = cross3(ac, bc), n2 = cross3(bd, ad);
    float q1 = dot3(n1, n1), q2 = dot3(n2, n2);
    if (!(el > 0.0f) || !(q1 > 0.0f) || !(q2 > 0.0f)) return false;
    V3 m1 = {n1.x / q1, n1.y / q1, n1.z / q1}, m2 = {n2.x / q2, n2.y / q2, n2.z / q2};
    V3 gc = {el * m1.x, el * m1.y, el * m1.z}, gd = {el * m2.x, el * m2.y, el * m2.z};
    V3 cb = sub3(xc, xb), db = sub3(xd, xb);
    float ta1 = dot3(cb, e) / el, ta2 = dot3(db, e) / el;
    float tb1 = dot3(ac, e) / el, tb2 = dot3(ad, e) / el;
    V3 ga, gb;
    { float p = ta1 * m1.x, q = ta2 * m2.x; ga.x = p + q; float r = tb1 * m1.x, t = tb2 * m2.x; gb.x = r + t; }
    { float p = ta1 * m1.y, q = ta2 * m2.y; ga.y = p + q; float r = tb1 * m1.y, t = tb2 * m2.y; gb.y = r + t; }
    { float p = ta1 * m1.z, q = ta2 * m2.z; ga.z = p + q; float r = tb1 * m1.z, t = tb2 * m2.z; gb.z = r + t; }
    float s1 = sqrtf(q1), s2 = sqrtf(q2);
    V3 u1 = {n1.x / s1, n1.y / s1, n1.z / s1}, u2 = {n2.x / s2, n2.y / s2, n2.z / s2};
    float cs = dot3(u1, u2);
    V3 cr = cross3(u1, u2);
    float sn = -(dot3(cr, e) / el);
    float t0 = sn * rest.x, t1 = cs * rest.y;
    float C = t0 - t1;
    float a0 = pa.w * dot3(ga, ga), a1 = pb.w * dot3(gb, gb), a2 = pc.w * dot3(gc, gc), a3 = pd.w * dot3(gd, gd);
    float den = (((a0 + a1) + a2) + a3) + at_b;
    if (!(den > 0.0f)) return false;
    float s = (-C) / den;
    xa = addscaled3(xa, pa.w * s, ga); xb = addscaled3(xb, pb.w * s, gb);
    xc = addscaled3(xc, pc.w * s, gc); xd = addscaled3(xd, pd.w * s, gd);
    pa.x = xa.x; pa.y = xa.y; pa.z = xa.z; pb.x = xb.x; pb.y = xb.y; pb.z = xb.z;
    pc.x = xc.x; pc.y = xc.y; pc.z = xc.z; pd.x = xd.x; pd.y = xd.y; pd.z = xd.z;
    return true;
}

// ---- 4-vertex constraints on FOUR lanes each (tile kernels) -------------------------------------------------------
// A tet or hinge projection is a long serial chain (130 / 450 instructions on one lane) and a round of an irregular
// tile holds only a few dozen of them, so the tile kernels spread one constraint over a quad of lanes: lane q = 0,1,2
// of the quad carries component q of every 3-vector, lane 3 carries the inverse masses; dot and cross products combine
// the lanes with DPP quad permutes (no LDS, no extra instructions once folded into the consumer), and independent scalar
// divisions / square roots are dealt one to a lane. Every operation is the one SPEC.md §5/§6 prescribes, in the same
// order with the same operands, so the bits equal the one-lane functions above (and the oracle).
constexpr int qp_ctrl(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
template <int CTRL>
__device__ __forceinline__ float qperm(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float qb0(float v) { return qperm<qp_ctrl(0, 0, 0, 0)>(v); }
__device__ __forceinline__ float qb1(float v) { return qperm<qp_ctrl(1, 1, 1, 1)>(v); }
__device__ __forceinline__ float qb2(float v) { return qperm<qp_ctrl(2, 2, 2, 2)>(v); }
__device__ __forceinline__ float qb3(float v) { return qperm<qp_ctrl(3, 3, 3, 3)>(v); }
__device__ __forceinline__ float qrot1(float v) { return qperm<qp_ctrl(1, 2, 0, 3)>(v); }   // lane c <- component (c+1) mod 3
__device__ __forceinline__ float qrot2(float v) { return qperm<qp_ctrl(2, 0, 1, 3)>(v); }   // lane c <- component (c+2) mod 3
// dot3: (xx + yy) + zz, the value in every lane of the quad
__device__ __forceinline__ float qdot(float a, float b) {
    const float t = a * b;
    const float xx = qb0(t), yy = qb1(t), zz = qb2(t);
    return (xx + yy) + zz;
}
// cross3: lane c gets a[c+1]*b[c+2] - a[c+2]*b[c+1]
__device__ __forceinline__ float qcross(float a, float b) {
    const float t0 = qrot1(a) * qrot2(b), t1 = qrot2(a) * qrot1(b);
    return t0 - t1;
}

// SPEC.md §5 on a quad. P[k]: lane q < 3 holds component q of particle k, lane 3 its inverse mass.
__device__ __forceinline__ bool project_volume_quad(float (&P)[4], float R6, float at_v) {
    const float W0 = qb3(P[0]), W1 = qb3(P[1]), W2 = qb3(P[2]), W3 = qb3(P[3]);
    const float e1 = P[1] - P[0], e2 = P[2] - P[0], e3 = P[3] - P[0];
    const float g1 = qcross(e2, e3), g2 = qcross(e3, e1), g3 = qcross(e1, e2);
    float t = g1 + g2; t = t + g3;
    const float g0 = -t;
    const float C6 = qdot(e1, g1) - R6;
    const float a0 = W0 * qdot(g0, g0), a1 = W1 * qdot(g1, g1), a2 = W2 * qdot(g2, g2), a3 = W3 * qdot(g3, g3);
    const float den = (((a0 + a1) + a2) + a3) + at_v;
    if (!(den > 0.0f)) return false;
    const float s = (-C6) / den;
    const float s0 = W0 * s, s1 = W1 * s, s2 = W2 * s, s3 = W3 * s;
    const float d0 = s0 * g0, d1 = s1 * g1, d2 = s2 * g2, d3 = s3 * g3;
    P[0] = P[0] + d0; P[1] = P[1] + d1; P[2] = P[2] + d2; P[3] = P[3] + d3;
    return true;
}

// SPEC.md §6 on a quad. rest = (cos phi0, sin phi0). q = lane & 3.
__device__ __forceinline__ bool project_bending_quad(float (&P)[4], float rest_c, float rest_s, float at_b, int q) {
    const float xa = P[0], xb = P[1], xc = P[2], xd = P[3];
    const float Wa = qb3(xa), Wb = qb3(xb), Wc = qb3(xc), Wd = qb3(xd);
    const float e = xb - xa;
    const float el2 = qdot(e, e);
    const float el = sqrtf(el2);
    const float ac = xa - xc, bc = xb - xc, bd = xb - xd, ad = xa - xd;
    const float n1 = qcross(ac, bc), n2 = qcross(bd, ad);
    const float q1 = qdot(n1, n1), q2 = qdot(n2, n2);
    if (!(el > 0.0f) || !(q1 > 0.0f) || !(q2 > 0.0f)) return false;
    const float m1 = n1 / q1, m2 = n2 / q2;
    const float gc = el * m1, gd = el * m2;
    const float cb = xc - xb, db = xd - xb;
    // four independent scalar divisions by el: one to a lane, then broadcast
    const float na1 = qdot(cb, e), na2 = qdot(db, e), nb1 = qdot(ac, e), nb2 = qdot(ad, e);
    const float num = q == 0 ? na1 : (q == 1 ? na2 : (q == 2 ? nb1 : nb2));
    const float quo = num / el;
    const float ta1 = qb0(quo), ta2 = qb1(quo), tb1 = qb2(quo), tb2 = qb3(quo);
    float ga, gb;
    { const float p = ta1 * m1, r = ta2 * m2; ga = p + r; }
    { const float p = tb1 * m1, r = tb2 * m2; gb = p + r; }
    // two independent square roots: lane 0 takes q1, the others q2
    const float sq = sqrtf(q == 0 ? q1 : q2);
    const float s1 = qb0(sq), s2 = qb1(sq);
    const float u1 = n1 / s1, u2 = n2 / s2;
    const float cs = qdot(u1, u2);
    const float cr = qcross(u1, u2);
    const float sn = -(qdot(cr, e) / el);
    const float t0 = sn * rest_c, t1 = cs * rest_s;
    const float C = t0 - t1;
    const float a0 = Wa * qdot(ga, ga), a1 = Wb * qdot(gb, gb), a2 = Wc * qdot(gc, gc), a3 = Wd * qdot(gd, gd);
    const float den = (((a0 + a1) + a2) + a3) + at_b;
    if (!(den > 0.0f)) return false;
    const float s = (-C) / den;
    const float sa = Wa * s, sb = Wb * s, sc = Wc * s, sd = Wd * s;
    const float da = sa * ga, db2 = sb * gb, dc = sc * gc, dd = sd * gd;
    P[0] = xa + da; P[1] = xb + db2; P[2] = xc + dc; P[3] = xd + dd;
    return true;
}

// SPEC.md §6 on a ROW of 16 lanes (wave items path). A hinge is the longest projection of a step -- on four lanes it is 280
// instructions against 90 for a tet, and a step lasts as long as its slowest wave: with every hinge priced like a tet the 100 k
// surrogate's tick is 1.62 instead of 1.93 ms -- and the groups of a tile hold one to three hinges, so lanes are not what is
// scarce. Its two triangles and its independent divisions are therefore spread over the four quads of a DPP row:
//   quad 0: n1, m1 = n1/q1, ta1 -> ga          quad 1: n2, m2 = n2/q2, ta2 -> gb
//   quad 2: n1, u1 = n1/sqrt(q1), tb1, cos / sin of the angle, C -> gc      quad 3: n2, u2 = n2/sqrt(q2), tb2 -> gd
// (one cross product, one square root, two divisions per lane instead of two, two and six), the quads exchange values with row
// rotations (row_ror) and ds_bpermute broadcasts, and quad k finishes with the update of particle k. Every value is computed by
// the operation SPEC.md prescribes on the operands it prescribes, so the bits equal project_bending (and the oracle).
// k = quad of the row (0..3), q = lane & 3, row_base4 = 4 * (first lane of the row). Xk: in/out, component q of particle k.
template <int CTRL>
__device__ __forceinline__ float rperm(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_next(float v) { return rperm<0x12c>(v); }    // lane i <- lane i + 4  (row_ror:12, measured: lane i reads lane i - n)
__device__ __forceinline__ float from_plus2(float v) { return rperm<0x128>(v); }   // lane i <- lane i + 8
__device__ __forceinline__ float from_prev(float v) { return rperm<0x124>(v); }    // lane i <- lane i - 4
__device__ __forceinline__ float row_bcast_quad(float v, int row_base4, int quad) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(row_base4 + 16 * quad, __float_as_int(v)));
}
__device__ __forceinline__ bool project_bending_row(const float (&P)[4], float rest_c, float rest_s, float at_b, int k, int row_base4,
                                                    float &Xk) {
    const float xa = P[0], xb = P[1], xc = P[2], xd = P[3];
    const float e = xb - xa;
    const float el2 = qdot(e, e);
    const float el = sqrtf(el2);
    const float ac = xa - xc, bc = xb - xc, bd = xb - xd, ad = xa - xd;
    const bool side2 = (k & 1) != 0;
    const float Pv = side2 ? bd : ac, Qv = side2 ? ad : bc;
    const float n = qcross(Pv, Qv);                  // n1 in quads 0, 2; n2 in quads 1, 3
    const float qq = qdot(n, n);                     // q1 / q2
    const float qo = from_next(qq);                  // the other triangle's (quad k + 1 holds the other side)
    if (!(el > 0.0f) || !(qq > 0.0f) || !(qo > 0.0f)) return false;
    const float sq = sqrtf(qq);
    const float r = n / (k >= 2 ? sq : qq);          // m1, m2, u1, u2
    const float cb = xc - xb, db = xd - xb;
    const float V = k == 0 ? cb : (k == 1 ? db : (k == 2 ? ac : ad));
    const float t = qdot(V, e) / el;                 // ta1, ta2, tb1, tb2
    const float t2 = from_plus2(t);                  // quad 0: tb1, quad 1: tb2
    const float p_own = t * r, p_x = t2 * r;         // quad 0: ta1*m1, tb1*m1; quad 1: ta2*m2, tb2*m2
    const float ga = p_own + from_next(p_own);       // (quad 0)  ta1*m1 + ta2*m2
    const float gb = from_prev(p_x) + p_x;           // (quad 1)  tb1*m1 + tb2*m2
    const float g23 = from_plus2(el * r);            // quad 2: gc = el*m1, quad 3: gd = el*m2
    const float g = k == 0 ? ga : (k == 1 ? gb : g23);
    // quad 2: u1 = r, u2 = quad 3's r
    const float u2 = from_next(r);
    const float cs = qdot(r, u2);
    const float cr = qcross(r, u2);
    const float sn = -(qdot(cr, e) / el);
    const float t0 = sn * rest_c, t1 = cs * rest_s;
    const float C = row_bcast_quad(t0 - t1, row_base4, 2);
    const float xk = Xk;
    const float Wk = qb3(xk);
    const float ak = Wk * qdot(g, g);
    const float a0 = row_bcast_quad(ak, row_base4, 0), a1 = row_bcast_quad(ak, row_base4, 1);
    const float a2 = row_bcast_quad(ak, row_base4, 2), a3 = row_bcast_quad(ak, row_base4, 3);
    const float den = (((a0 + a1) + a2) + a3) + at_b;
    if (!(den > 0.0f)) return false;
    const float s = (-C) / den;
    const float sk = Wk * s;
    const float d = sk * g;
    Xk = xk + d;
    return true;
}

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
    // (a lane-packed tile never stages its data in LDS, so the tiling's window need not hold it: its loads are not cut at `win`)
    const uint32_t n4_first = lane_packed ? (lane_packed_full ? 2u * (uint32_t)kLanePackLanes : (uint32_t)kLanePackLanes)
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
            if (i < n4_first && !lane_packed) dst[i] = wv[q];      // (lane-packed slots stay in wv[0])
        }
        if (!lane_packed)
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

#ifndef SB_REG_ROUNDS
#define SB_REG_ROUNDS 4          // 256-lane workgroups (one constraint per lane and round)
#endif
#ifndef SB_REG_ROUNDS_NARROW
#define SB_REG_ROUNDS_NARROW 3   // 128-lane workgroups (two per lane): the register budget of 6 waves per SIMD allows 3
#endif
    // Short programs of dictionary-coded distance groups (every tile of a regular mesh: 3 groups) keep their constraint
    // slots and rest lengths in registers: one batch of LDS reads ahead of the first round instead of two dependent LDS
    // round trips (slot, then palette entry) at the head of every round, in both passes. Same constraints, same order.
    constexpr int kRegRounds = THREADS >= 256 ? SB_REG_ROUNDS : SB_REG_ROUNDS_NARROW;
    // (round 3: also for tiles whose slots are NOT dictionary-coded -- per-spring rest lengths, {i | j<<16, rest} pairs: the same 2
    // registers per constraint, only the decode differs -- so that a mesh with varied rest lengths keeps the short path)
#ifdef SB_REG_COMPACT_ONLY      // A/B timing builds only: the round-2 condition (dictionary-coded tiles only)
    constexpr bool kRegFullSlots = false;
#else
    constexpr bool kRegFullSlots = true;
#endif
    if (kRegRounds > 0 && !QUADS && n_rounds_all <= kRegRounds && n_rounds_all > 0 && (kRegFullSlots || n_pal > 0) && (lane_packed || d_hi - d_lo <= win)) {
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

// ---- peer-store halo transport (opt-in, SB_HALO_TRANSPORT=peer; solver.hip) ---------------------------------------
// Instead of pack -> ncclSend/ncclRecv -> unpack, the push kernel stores every peer's ghosts STRAIGHT into that peer's
// mailbox (one fine-grained device allocation per rank, mapped into the senders by IPC handle or, inside one process, by
// plain pointer) and raises a flag there; its last workgroup then waits until the flags of this rank's own senders have
// arrived, so the unpack kernel behind it in the stream can copy the ghosts into the arrays and acknowledge. A segment
// has two buffers used alternately (epoch & 1): exchange e writes the buffer exchange e-2 used, and a rank finishes
// exchange e only after its receivers acknowledged e-1, so no sender ever waits before writing. Epochs count the
// exchanges of a halo slot, live in device memory and are advanced by the kernels themselves: the launches sit in a
// captured hipGraph unchanged. Every wait is bounded: a flag that never arrives sets an error word instead of hanging.
constexpr int kMaxPeers = 8;
struct PeerSlot {
    int32_t n_send, n_recv;                 // peers this rank sends to / receives from on this halo slot
    int32_t send_off[kMaxPeers + 1];        // first ghost of every send peer in send_idx, order of the send peers
    int32_t send_cap[kMaxPeers];            // ghosts pushed to that peer (a loopback self-exchange may push fewer than it packs)
    float *remote_data[kMaxPeers];          // per send peer: where this rank's segment starts inside the peer's mailbox
    int32_t remote_stride[kMaxPeers];       // floats between the two buffers of that segment (exchanges alternate: epoch & 1)
    int32_t my_stride[kMaxPeers];           // per recv peer: the same for the segments in this rank's mailbox
    int32_t recv_off[kMaxPeers + 1];        // first ghost of every recv peer in recv_idx, order of the recv peers
    int32_t recv_cnt[kMaxPeers];            // ghosts per recv peer
    const float *my_data[kMaxPeers];        // per recv peer: its segment in this rank's mailbox (16-byte aligned)
    uint32_t *remote_data_flag[kMaxPeers];  // per send peer: the peer's "data from this rank arrived" word
    uint32_t *my_ack_flag[kMaxPeers];       // per send peer: local word the peer writes when it has consumed the segment
    uint32_t *my_data_flag[kMaxPeers];      // per recv peer: local word the peer writes when its data is in the mailbox
    uint32_t *remote_ack_flag[kMaxPeers];   // per recv peer: the peer's "consumed" word for this rank
    uint32_t *local;                        // this rank's own (ordinary, cached) words for the slot: [0] exchanges completed so far,
                                            // [1] push / [2] unpack workgroups finished
    uint32_t *error;                        // set to 1 when a wait gave up
};
constexpr int kPeerSpinLimit = 1 << 24;

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

// ---- table validator (debug entry sb_debug_validate; SURVEY.md 5 "race detection") ---------------------------------------------------
// The only race this design can have is two constraints of one group (or two tiles of one launch) touching the same particle. The
// planner's output is checked on the host (tests/test_plan.py); THIS kernel checks what the tile kernels actually read -- the uploaded
// descriptors, run tables / particle lists, group words, dictionary-coded or full slots, four-vertex slots and wave items, after
// packing, lane dealing and cost ordering -- with the tile kernels' own decoding rules. One workgroup per device tile.
struct ValidateCounters {
    unsigned long long tiles, groups, constraints;
    // 0 index out of range (particle or tile-local), 1 a particle twice in one group, 2 a particle staged by two tiles of one launch,
    // 3 the group walk leaves the tile's stream, 4 malformed run table / particle list, 5 wave items disagree with the group words
    unsigned int errors[6];
    int first[4];      // tile, group, kind of the first error seen (-1 = none), spare
};
constexpr int kValidateThreads = 256;
__global__ __launch_bounds__(kValidateThreads) void validate_tiles_kernel(const TileDesc *tiles, int tile_begin, const int2 *runs_overflow, const uint32_t *stream,
                                                                         const int32_t *gather, int use_gather, int n_particles, int item_waves,
                                                                         int32_t *owner, ValidateCounters *out) {
    __shared__ uint32_t bitmap[kLargeTile / 32];
    __shared__ unsigned int s_err[6];
    __shared__ int s_first_group, s_first_kind;
    const int tid = threadIdx.x;
    const int tile = tile_begin + (int)blockIdx.x;
    const TileDesc td = tiles[tile];
    const uint32_t *ts = stream + td.s_begin;
    if (tid < 6) s_err[tid] = 0;
    if (tid == 0) { s_first_group = -1; s_first_kind = -1; }
    __syncthreads();
    auto flag = [&](int kind, int group) {
        atomicAdd(&s_err[kind], 1u);
        if (atomicCAS(&s_first_kind, -1, kind) == -1) s_first_group = group;
    };
    const bool size_ok = td.n_local >= 0 && td.n_local <= kLargeTile && td.n_rounds >= 0 && td.s_hdr <= td.s_len;
    if (!size_ok) { if (tid == 0) flag(4, -1); }
    // ---- particles: the tile kernels' own lookup (runs sorted by first local index, or the explicit list of a sparse tile) ----
    if (size_ok) {
        if (!use_gather && tid == 0) {
            bool ok = td.run_count >= (td.n_local > 0 ? 1 : 0);
            int prev_y = -1;
            for (int r = 0; r < td.run_count && ok; ++r) {
                const int2 rn = r < kInlineRuns ? td.runs[r] : runs_overflow[td.run_overflow + r - kInlineRuns];
                ok = rn.y > prev_y && rn.y < td.n_local && rn.x >= 0 && (r > 0 || rn.y == 0);
                prev_y = rn.y;
            }
            if (!ok) flag(4, -1);
        }
        for (int l = tid; l < td.n_local; l += kValidateThreads) {
            int g;
            if (use_gather) g = gather[td.gather_begin + l];
            else {
                g = -1;
                for (int r = 0; r < td.run_count; ++r) {
                    const int2 rn = r < kInlineRuns ? td.runs[r] : runs_overflow[td.run_overflow + r - kInlineRuns];
                    if (rn.y <= l) g = rn.x + (l - rn.y);
                }
            }
            if (g < 0 || g >= n_particles) flag(0, -1);
            else if (atomicCAS(&owner[g], -1, tile) != -1) flag(2, -1);
        }
    }
    // ---- groups: walk the data exactly as the kernels do; inside a group every particle at most once ----
    unsigned int tot[3] = {0, 0, 0};
    uint32_t off = td.s_hdr;
    bool walk_ok = size_ok;
    if (size_ok && td.packed_lanes) {
        // lane-packed slots (kLanePack*): one 16-byte word per lane, field 2 r + u = slot lane + 128 u of round r
        const uint32_t P = td.packed_lanes;
        // (n_pal == 0: the full form, rest lengths behind the index words)
        if (P != (uint32_t)kLanePackLanes || td.n_rounds > kLanePackRounds || td.n_pal < 0 || td.n_pal > kLanePackMaxPalette || td.n_local > kSmallTile ||
            td.s_hdr + (td.n_pal > 0 ? kLanePackDwordsCompact : kLanePackDwordsFull) > td.s_len) { if (tid == 0) flag(3, -1); walk_ok = false; }
        for (int r = 0; walk_ok && r < td.n_rounds; ++r) {
            const uint32_t w = ts[r];
            const uint32_t cnt = w & 1023u;
            if (cnt > 2u * P || ((w >> 10) & 0xfffffu) != 0u) { if (tid == 0) flag(3, r); walk_ok = false; break; }
            for (int q = tid; q < kLargeTile / 32; q += kValidateThreads) bitmap[q] = 0;
            __syncthreads();
            for (uint32_t c = tid; c < cnt; c += kValidateThreads) {
                const uint32_t lane = c % P, u = c / P, bit = (uint32_t)kLanePackFieldBits * (2u * (uint32_t)r + u), w0 = bit >> 5, sh = bit & 31u;
                const uint32_t *wd = ts + td.s_hdr + 4u * lane;
                uint64_t two = (uint64_t)wd[w0] | ((uint64_t)(w0 + 1 < 4 ? wd[w0 + 1] : 0u) << 32);
                const uint32_t f = (uint32_t)(two >> sh) & ((1u << kLanePackFieldBits) - 1u);
                const uint32_t p0 = f & 511u, p1 = (f >> 9) & 511u;
                if ((f >> 18) >= (uint32_t)max(td.n_pal, 1)) flag(0, r);
                for (int e = 0; e < 2; ++e) {
                    const uint32_t p = e ? p1 : p0;
                    if (p >= (uint32_t)td.n_local) flag(0, r);
                    else if (atomicOr(&bitmap[p >> 5], 1u << (p & 31)) & (1u << (p & 31))) flag(1, r);
                }
            }
            __syncthreads();
            tot[0] += cnt;
        }
    } else
    for (int r = 0; walk_ok && r < td.n_rounds; ++r) {
        const uint32_t w = ts[r];
        const uint32_t cnt = w & 1023u, n_vol = (w >> 10) & 1023u, n_bend = (w >> 20) & 1023u;
        const bool compact = (w >> 30) & 1u;
        const uint32_t dsize = compact ? ((cnt + 3u) & ~3u) : ((2u * cnt + 3u) & ~3u);
        const uint32_t size = dsize + 4u * (n_vol + n_bend);
        if (off + size > td.s_len || cnt > (uint32_t)kRoundSlots || n_vol > (uint32_t)kRoundSlots || n_bend > (uint32_t)kRoundSlots ||
            (compact && td.n_pal <= 0)) { if (tid == 0) flag(3, r); walk_ok = false; break; }
        for (int q = tid; q < kLargeTile / 32; q += kValidateThreads) bitmap[q] = 0;
        __syncthreads();
        auto mark = [&](uint32_t p) {
            if (p >= (uint32_t)td.n_local) { flag(0, r); return; }
            if (atomicOr(&bitmap[p >> 5], 1u << (p & 31)) & (1u << (p & 31))) flag(1, r);
        };
        for (uint32_t c = tid; c < cnt; c += kValidateThreads) {
            uint32_t i, k;
            if (compact) { const uint32_t e = ts[off + c]; i = e & 0xfffu; k = (e >> 12) & 0xfffu; if ((e >> 24) >= (uint32_t)td.n_pal) flag(0, r); }
            else { const uint32_t e = ts[off + 2 * c]; i = e & 0xffffu; k = e >> 16; }
            mark(i); mark(k);
        }
        for (uint32_t c = tid; c < n_vol + n_bend; c += kValidateThreads) {
            const uint32_t e0 = ts[off + dsize + 4 * c], e1 = ts[off + dsize + 4 * c + 1];
            mark(e0 & 0xffffu); mark(e0 >> 16); mark(e1 & 0xffffu); mark(e1 >> 16);
        }
        __syncthreads();
        tot[0] += cnt; tot[1] += n_vol; tot[2] += n_bend;
        off += size;
    }
    // ---- wave items: the same work dealt to the waves ahead of time; must add up to the group words and end every group on every wave ----
    if (walk_ok && td.n_steps > 0 && item_waves > 0 && tid == 0) {
        unsigned int it_tot[3] = {0, 0, 0};
        bool ok = td.s_items + (uint32_t)(item_waves * td.n_steps) <= td.s_hdr;
        for (int wv = 0; ok && wv < item_waves; ++wv) {
            int barriers = 0;
            for (int st = 0; st < td.n_steps; ++st) {
                const uint32_t it = ts[td.s_items + (uint32_t)(wv * td.n_steps + st)];
                const uint32_t type = it & 7u, c = (it >> kItemCountShift) & 127u, o = it >> kItemOffsetShift;
                if (it & (1u << kItemBarrierBit)) ++barriers;
                if (type == kItemIdle) continue;
                const uint32_t bytes4 = type == kItemDistCompact ? c : (type == kItemDistFull ? 2u * c : 4u * c);
                if (type > kItemBending || td.s_hdr + o + bytes4 > td.s_len) { ok = false; break; }
                it_tot[type == kItemVolume ? 1 : (type == kItemBending ? 2 : 0)] += c;
            }
            ok = ok && barriers == td.n_rounds;
        }
        if (!ok || it_tot[0] != tot[0] || it_tot[1] != tot[1] || it_tot[2] != tot[2]) flag(5, -1);
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&out->tiles, 1ull);
        atomicAdd(&out->groups, (unsigned long long)(walk_ok ? td.n_rounds : 0));
        atomicAdd(&out->constraints, (unsigned long long)tot[0] + tot[1] + tot[2]);
        bool any = false;
        for (int k = 0; k < 6; ++k) if (s_err[k]) { atomicAdd(&out->errors[k], s_err[k]); any = true; }
        if (any && atomicCAS(&out->first[0], -1, tile) == -1) { out->first[1] = s_first_group; out->first[2] = s_first_kind; }
    }
}
// One global colour: no particle twice (owner[] cleared to -1 before every colour).
__global__ __launch_bounds__(256) void validate_gcolour_kernel(const int32_t *idx, int per_constraint, int count, int n_particles, int colour, int32_t *owner,
                                                               ValidateCounters *out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    for (int c = 0; c < per_constraint; ++c) {
        const int g = idx[(size_t)per_constraint * k + c];
        int kind = -1;
        if (g < 0 || g >= n_particles) kind = 0;
        else if (atomicCAS(&owner[g], -1, k) != -1) kind = 1;
        if (kind >= 0) {
            atomicAdd(&out->errors[kind], 1u);
            if (atomicCAS(&out->first[0], -1, -2 - colour) == -1) { out->first[1] = k; out->first[2] = kind; }
        }
    }
    if (threadIdx.x == 0) atomicAdd(&out->constraints, (unsigned long long)min(256, count - (int)blockIdx.x * 256));
}

}  // namespace sbk
