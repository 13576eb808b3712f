// SoftbodyCpuSolver.cs — the C# CPU FixedUpdate path: a sequential restatement of SPEC.md, identical
// operation for operation to oracle/oracle.c (which is what the test-suite actually runs, because no C#
// toolchain exists in the build image). The reference repository has no CPU solver to copy
// (/root/reference/README.md:1 is its only line).
//
// C# evaluates float expressions in at least float precision but MAY use higher precision for
// intermediates; every intermediate below is therefore stored to a float local (an explicit cast forces
// rounding to binary32, ECMA-335 §I.12.1.3), and no Math.FusedMultiplyAdd is used.
using System;
using UnityEngine;

namespace SoftbodyMI355X
{
    public sealed class SoftbodyCpuSolver
    {
        readonly Softbody sb;
        readonly byte[][] orderTypeByParity;   // SPEC.md §3: substep k of a tick walks the order of parity k & 1
        readonly int[][] orderIdByParity;
        readonly Vector3[] prev;

        public SoftbodyCpuSolver(Softbody owner, byte[][] type, int[][] id)
        {
            sb = owner; orderTypeByParity = type; orderIdByParity = id;
            prev = new Vector3[owner.positions.Length];
        }

        public void Step(float dt, int substeps)
        {
            // SPEC.md §2 host-side scalars
            float S = (float)substeps;
            float h = (float)(dt / S);
            float invH = (float)(1.0f / h);
            Vector3 g = sb.Gravity;
            float hgx = (float)(h * g.x), hgy = (float)(h * g.y), hgz = (float)(h * g.z);
            float td = (float)(sb.Damping * h);
            float kd = (float)(1.0f - td); if (kd < 0f) kd = 0f;
            float h2 = (float)(h * h);
            float atD = (float)(sb.ComplianceD / h2);
            float av = (float)(sb.ComplianceV / h2);
            float atV = (float)(36.0f * av);
            float atB = (float)(sb.ComplianceB / h2);
            Vector3[] x = sb.positions; Vector3[] v = sb.velocities; float[] w = sb.inverseMass;
            int n = x.Length;
            for (int it = 0; it < substeps; ++it)
            {
                byte[] orderType = orderTypeByParity[it & 1];
                int[] orderId = orderIdByParity[it & 1];
                for (int p = 0; p < n; ++p)
                {
                    prev[p] = x[p];
                    if (w[p] > 0f)
                    {
                        float vx = (float)(v[p].x + hgx), vy = (float)(v[p].y + hgy), vz = (float)(v[p].z + hgz);
                        v[p] = new Vector3(vx, vy, vz);
                        float ax = (float)(h * vx), ay = (float)(h * vy), az = (float)(h * vz);
                        x[p] = new Vector3((float)(x[p].x + ax), (float)(x[p].y + ay), (float)(x[p].z + az));
                    }
                }
                for (long k = 0; k < orderId.LongLength; ++k)
                {
                    int id = orderId[k];
                    switch (orderType[k])
                    {
                        case 0: ProjectDistance(x, w, sb.distanceIJ[2 * id], sb.distanceIJ[2 * id + 1], sb.distanceRest[id], atD); break;
                        case 1: ProjectVolume(x, w, sb.volumeIJKL, 4 * id, (float)(6.0f * sb.volumeRest[id]), atV); break;
                        default: ProjectBending(x, w, sb.bendingIJKL, 4 * id, sb.bendingRestCosSin[2 * id], sb.bendingRestCosSin[2 * id + 1], atB); break;
                    }
                }
                if (sb.GroundPlane)   // SPEC.md §2 step 2b
                {
                    Vector3 pn = sb.GroundNormal; float pd = sb.GroundOffset;
                    for (int p = 0; p < n; ++p)
                    {
                        if (!(w[p] > 0f)) continue;
                        float a = (float)(pn.x * x[p].x), b = (float)(pn.y * x[p].y), c = (float)(pn.z * x[p].z);
                        float pen = (float)((float)((float)(a + b) + c) - pd);
                        if (pen < 0f)
                            x[p] = new Vector3((float)(x[p].x - (float)(pen * pn.x)), (float)(x[p].y - (float)(pen * pn.y)), (float)(x[p].z - (float)(pen * pn.z)));
                    }
                }
                for (int p = 0; p < n; ++p)
                {
                    float dx = (float)(x[p].x - prev[p].x), dy = (float)(x[p].y - prev[p].y), dz = (float)(x[p].z - prev[p].z);
                    float qx = (float)(dx * invH), qy = (float)(dy * invH), qz = (float)(dz * invH);
                    v[p] = new Vector3((float)(qx * kd), (float)(qy * kd), (float)(qz * kd));
                }
            }
        }

        // SPEC.md §4
        static void ProjectDistance(Vector3[] x, float[] w, int i, int j, float L0, float at)
        {
            float wi = w[i], wj = w[j];
            float dx = (float)(x[i].x - x[j].x), dy = (float)(x[i].y - x[j].y), dz = (float)(x[i].z - x[j].z);
            float xx = (float)(dx * dx), yy = (float)(dy * dy), zz = (float)(dz * dz);
            float L2 = (float)((float)(xx + yy) + zz);
            float ws = (float)((float)(wi + wj) + at);
            if (!(L2 >= 1.262177448e-29f) || !(ws > 0f)) return;   // 2^-96 (SPEC.md §4): coincident endpoints give no direction
            float L = (float)Math.Sqrt(L2);        // sqrt of a float in double, rounded to float == correctly rounded sqrtf
            float C = (float)(L - L0);
            float wl = (float)(ws * L);
            float s = (float)((-C) / wl);
            float si = (float)(wi * s), sj = (float)(wj * s);
            float ax = (float)(si * dx), ay = (float)(si * dy), az = (float)(si * dz);
            float bx = (float)(sj * dx), by = (float)(sj * dy), bz = (float)(sj * dz);
            x[i] = new Vector3((float)(x[i].x + ax), (float)(x[i].y + ay), (float)(x[i].z + az));
            x[j] = new Vector3((float)(x[j].x - bx), (float)(x[j].y - by), (float)(x[j].z - bz));
        }

        struct F3 { public float x, y, z; public F3(float a, float b, float c) { x = a; y = b; z = c; } }
        static F3 Sub(Vector3 a, Vector3 b) => new F3((float)(a.x - b.x), (float)(a.y - b.y), (float)(a.z - b.z));
        static F3 Cross(F3 a, F3 b)
        {
            float t0 = (float)(a.y * b.z), t1 = (float)(a.z * b.y), t2 = (float)(a.z * b.x);
            float t3 = (float)(a.x * b.z), t4 = (float)(a.x * b.y), t5 = (float)(a.y * b.x);
            return new F3((float)(t0 - t1), (float)(t2 - t3), (float)(t4 - t5));
        }
        static float Dot(F3 a, F3 b)
        {
            float xx = (float)(a.x * b.x), yy = (float)(a.y * b.y), zz = (float)(a.z * b.z);
            return (float)((float)(xx + yy) + zz);
        }
        static Vector3 AddScaled(Vector3 p, float s, F3 g)
        {
            float a = (float)(s * g.x), b = (float)(s * g.y), c = (float)(s * g.z);
            return new Vector3((float)(p.x + a), (float)(p.y + b), (float)(p.z + c));
        }

        // SPEC.md §5
        static void ProjectVolume(Vector3[] x, float[] w, int[] q, int o, float R6, float atV)
        {
            int i0 = q[o], i1 = q[o + 1], i2 = q[o + 2], i3 = q[o + 3];
            F3 e1 = Sub(x[i1], x[i0]), e2 = Sub(x[i2], x[i0]), e3 = Sub(x[i3], x[i0]);
            F3 g1 = Cross(e2, e3), g2 = Cross(e3, e1), g3 = Cross(e1, e2);
            F3 g0 = new F3(-(float)((float)(g1.x + g2.x) + g3.x), -(float)((float)(g1.y + g2.y) + g3.y), -(float)((float)(g1.z + g2.z) + g3.z));
            float C6 = (float)(Dot(e1, g1) - R6);
            float a0 = (float)(w[i0] * Dot(g0, g0)), a1 = (float)(w[i1] * Dot(g1, g1)), a2 = (float)(w[i2] * Dot(g2, g2)), a3 = (float)(w[i3] * Dot(g3, g3));
            float den = (float)((float)((float)((float)(a0 + a1) + a2) + a3) + atV);
            if (!(den > 0f)) return;
            float s = (float)((-C6) / den);
            x[i0] = AddScaled(x[i0], (float)(w[i0] * s), g0); x[i1] = AddScaled(x[i1], (float)(w[i1] * s), g1);
            x[i2] = AddScaled(x[i2], (float)(w[i2] * s), g2); x[i3] = AddScaled(x[i3], (float)(w[i3] * s), g3);
        }

        // SPEC.md §6
        static void ProjectBending(Vector3[] x, float[] w, int[] q, int o, float c0, float s0, float atB)
        {
            int ia = q[o], ib = q[o + 1], ic = q[o + 2], id = q[o + 3];
            F3 e = Sub(x[ib], x[ia]);
            float el = (float)Math.Sqrt(Dot(e, e));
            F3 ac = Sub(x[ia], x[ic]), bc = Sub(x[ib], x[ic]), bd = Sub(x[ib], x[id]), ad = Sub(x[ia], x[id]);
            F3 n1 = Cross(ac, bc), n2 = Cross(bd, ad);
            float q1 = Dot(n1, n1), q2 = Dot(n2, n2);
            if (!(el > 0f) || !(q1 > 0f) || !(q2 > 0f)) return;
            F3 m1 = new F3((float)(n1.x / q1), (float)(n1.y / q1), (float)(n1.z / q1));
            F3 m2 = new F3((float)(n2.x / q2), (float)(n2.y / q2), (float)(n2.z / q2));
            F3 gc = new F3((float)(el * m1.x), (float)(el * m1.y), (float)(el * m1.z));
            F3 gd = new F3((float)(el * m2.x), (float)(el * m2.y), (float)(el * m2.z));
            F3 cb = Sub(x[ic], x[ib]), db = Sub(x[id], x[ib]);
            float ta1 = (float)(Dot(cb, e) / el), ta2 = (float)(Dot(db, e) / el);
            float tb1 = (float)(Dot(ac, e) / el), tb2 = (float)(Dot(ad, e) / el);
            F3 ga = new F3((float)((float)(ta1 * m1.x) + (float)(ta2 * m2.x)), (float)((float)(ta1 * m1.y) + (float)(ta2 * m2.y)), (float)((float)(ta1 * m1.z) + (float)(ta2 * m2.z)));
            F3 gb = new F3((float)((float)(tb1 * m1.x) + (float)(tb2 * m2.x)), (float)((float)(tb1 * m1.y) + (float)(tb2 * m2.y)), (float)((float)(tb1 * m1.z) + (float)(tb2 * m2.z)));
            float s1 = (float)Math.Sqrt(q1), s2 = (float)Math.Sqrt(q2);
            F3 u1 = new F3((float)(n1.x / s1), (float)(n1.y / s1), (float)(n1.z / s1));
            F3 u2 = new F3((float)(n2.x / s2), (float)(n2.y / s2), (float)(n2.z / s2));
            float cs = Dot(u1, u2);
            float sn = -(float)(Dot(Cross(u1, u2), e) / el);
            float C = (float)((float)(sn * c0) - (float)(cs * s0));
            float a0 = (float)(w[ia] * Dot(ga, ga)), a1 = (float)(w[ib] * Dot(gb, gb)), a2 = (float)(w[ic] * Dot(gc, gc)), a3 = (float)(w[id] * Dot(gd, gd));
            float den = (float)((float)((float)((float)(a0 + a1) + a2) + a3) + atB);
            if (!(den > 0f)) return;
            float s = (float)((-C) / den);
            x[ia] = AddScaled(x[ia], (float)(w[ia] * s), ga); x[ib] = AddScaled(x[ib], (float)(w[ib] * s), gb);
            x[ic] = AddScaled(x[ic], (float)(w[ic] * s), gc); x[id] = AddScaled(x[id], (float)(w[id] * s), gd);
        }
    }
}
