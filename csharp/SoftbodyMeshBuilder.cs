// SoftbodyMeshBuilder.cs — Unity Mesh -> particles + constraint graph for the Softbody component
// (SURVEY.md §8f item 2). C# twin of softbodyunity_amd/mesh.py::from_triangle_mesh / from_tet_mesh, which is
// what the test-suite runs (tests/test_authoring.py): no C# toolchain exists in the build image. The reference
// repository has no authoring code to mirror (/root/reference/README.md:1 is its only line).
using System;
using System.Collections.Generic;
using UnityEngine;

namespace SoftbodyMI355X
{
    public static class SoftbodyMeshBuilder
    {
        /// Weld duplicated render vertices (UV / normal seams), create one distance constraint per triangle edge
        /// and one bending hinge per edge shared by exactly two triangles (rest = (cos, sin) of the rest dihedral).
        public static int[] FromTriangleMesh(Softbody sb, Vector3[] vertices, int[] triangles, float weldEps = 1e-6f)
        {
            var key2p = new Dictionary<(long, long, long), int>();
            var particleOfVertex = new int[vertices.Length];
            var pos = new List<Vector3>();
            for (int v = 0; v < vertices.Length; ++v)
            {
                var k = ((long)Math.Round(vertices[v].x / weldEps), (long)Math.Round(vertices[v].y / weldEps), (long)Math.Round(vertices[v].z / weldEps));
                if (!key2p.TryGetValue(k, out int p)) { p = pos.Count; key2p[k] = p; pos.Add(vertices[v]); }
                particleOfVertex[v] = p;
            }
            var edges = new SortedDictionary<(int, int), List<int>>();   // edge -> opposite vertices
            for (int t = 0; t + 2 < triangles.Length; t += 3)
            {
                int a = particleOfVertex[triangles[t]], b = particleOfVertex[triangles[t + 1]], c = particleOfVertex[triangles[t + 2]];
                if (a == b || b == c || a == c) continue;
                AddEdge(edges, a, b, c); AddEdge(edges, b, c, a); AddEdge(edges, c, a, b);
            }
            var ij = new List<int>(); var rest = new List<float>();
            var hinge = new List<int>(); var hingeRest = new List<float>();
            foreach (var e in edges)
            {
                int i = e.Key.Item1, j = e.Key.Item2;
                ij.Add(i); ij.Add(j); rest.Add((pos[i] - pos[j]).magnitude);
                if (e.Value.Count != 2) continue;
                int c = e.Value[0], d = e.Value[1];
                Vector3 A = pos[i], B = pos[j], C = pos[c], D = pos[d];
                Vector3 n1 = Vector3.Cross(A - C, B - C), n2 = Vector3.Cross(B - D, A - D), eh = B - A;
                if (n1.sqrMagnitude < 1e-24f || n2.sqrMagnitude < 1e-24f || eh.sqrMagnitude < 1e-24f) continue;
                n1.Normalize(); n2.Normalize(); eh.Normalize();
                hinge.Add(i); hinge.Add(j); hinge.Add(c); hinge.Add(d);
                hingeRest.Add(Vector3.Dot(n1, n2));                       // cos phi0
                hingeRest.Add(-Vector3.Dot(Vector3.Cross(n1, n2), eh));   // sin phi0 (SPEC.md §6)
            }
            sb.restPositions = pos.ToArray();
            sb.positions = pos.ToArray();
            sb.distanceIJ = ij.ToArray(); sb.distanceRest = rest.ToArray();
            sb.bendingIJKL = hinge.ToArray(); sb.bendingRestCosSin = hingeRest.ToArray();
            var rt = new List<int>();          // render triangles in particle indices (GPU vertex normals, SPEC.md 6a)
            for (int t = 0; t + 2 < triangles.Length; t += 3)
            {
                int a = particleOfVertex[triangles[t]], b = particleOfVertex[triangles[t + 1]], c = particleOfVertex[triangles[t + 2]];
                if (a != b && b != c && a != c) { rt.Add(a); rt.Add(b); rt.Add(c); }
            }
            sb.renderTriangles = rt.ToArray();
            return particleOfVertex;   // mesh.vertices[v] = positions[particleOfVertex[v]] after each FixedUpdate
        }

        /// Tetrahedral mesh: edge springs + one volume constraint per (positively oriented) tet.
        public static void FromTetMesh(Softbody sb, Vector3[] nodes, int[] tets)
        {
            var edges = new SortedSet<(int, int)>();
            var vol = new List<float>(); var idx = new List<int>();
            for (int t = 0; t + 3 < tets.Length; t += 4)
            {
                int a = tets[t], b = tets[t + 1], c = tets[t + 2], d = tets[t + 3];
                float v6 = Vector3.Dot(nodes[b] - nodes[a], Vector3.Cross(nodes[c] - nodes[a], nodes[d] - nodes[a]));
                if (v6 < 0f) { int tmp = b; b = c; c = tmp; v6 = -v6; }
                idx.Add(a); idx.Add(b); idx.Add(c); idx.Add(d); vol.Add(v6 / 6f);
                int[] q = { a, b, c, d };
                for (int x = 0; x < 4; ++x) for (int y = x + 1; y < 4; ++y) edges.Add((Math.Min(q[x], q[y]), Math.Max(q[x], q[y])));
            }
            var ij = new List<int>(); var rest = new List<float>();
            foreach (var e in edges) { ij.Add(e.Item1); ij.Add(e.Item2); rest.Add((nodes[e.Item1] - nodes[e.Item2]).magnitude); }
            sb.restPositions = (Vector3[])nodes.Clone(); sb.positions = (Vector3[])nodes.Clone();
            sb.distanceIJ = ij.ToArray(); sb.distanceRest = rest.ToArray();
            sb.volumeIJKL = idx.ToArray(); sb.volumeRest = vol.ToArray();
        }

        static void AddEdge(SortedDictionary<(int, int), List<int>> edges, int a, int b, int opp)
        {
            var k = (Math.Min(a, b), Math.Max(a, b));
            if (!edges.TryGetValue(k, out var l)) { l = new List<int>(); edges[k] = l; }
            l.Add(opp);
        }
    }
}
