// device_math.hip.hpp — gfx950 device helpers of the soft-body hot path: loads / stores of the packed state, the constraint projections (SPEC.md §4-§6).
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree).
// Every arithmetic statement mirrors oracle/oracle.c one operation at a time; the file is compiled
// with -ffp-contract=off and correctly-rounded fp32 divide/sqrt so results are bit-identical to the
// oracle. MFMA is not used: the path is a bandwidth-bound gather/scatter (BASELINE.json:5).
#pragma once
#include "kernel_types.hpp"

namespace sbk {

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

}  // namespace sbk
