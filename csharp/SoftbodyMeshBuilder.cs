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

        /// Gmsh .msh, ASCII 2.2 or 4.1: the 4-node tetrahedra (element type 4) of the file -> FromTetMesh. Node tags need not be
        /// consecutive; nodes no tet uses (geometry points) are dropped. Mirrors softbodyunity_amd/mesh.py read_gmsh.
        public static void ReadGmsh(Softbody sb, string path)
        {
            var sect = new Dictionary<string, List<string>>();
            string name = null; List<string> cur = null;
            foreach (var raw in System.IO.File.ReadAllLines(path))
            {
                var ln = raw.Trim();
                if (ln.Length == 0) continue;
                if (ln.StartsWith("$End")) { name = null; cur = null; }
                else if (ln.StartsWith("$")) { name = ln.Substring(1); cur = new List<string>(); sect[name] = cur; }
                else if (cur != null) cur.Add(ln);
            }
            var inv = System.Globalization.CultureInfo.InvariantCulture;
            var fmt = sect["MeshFormat"][0].Split((char[])null, StringSplitOptions.RemoveEmptyEntries);
            if (int.Parse(fmt[1]) != 0) throw new FormatException("binary .msh files are not read: export ASCII");
            bool v2 = double.Parse(fmt[0], inv) < 3.0;
            var tag = new List<long>(); var xyz = new List<Vector3>(); var tet = new List<long>();
            var N = sect["Nodes"]; var E = sect["Elements"];
            Func<string, string[]> split = l => l.Split((char[])null, StringSplitOptions.RemoveEmptyEntries);
            if (v2)
            {
                int n = int.Parse(N[0]);
                for (int i = 1; i <= n; ++i) { var t = split(N[i]); tag.Add(long.Parse(t[0])); xyz.Add(new Vector3(float.Parse(t[1], inv), float.Parse(t[2], inv), float.Parse(t[3], inv))); }
                int m = int.Parse(E[0]);
                for (int i = 1; i <= m; ++i)
                {
                    var t = split(E[i]);
                    if (int.Parse(t[1]) != 4) continue;
                    int first = 3 + int.Parse(t[2]);
                    for (int q = 0; q < 4; ++q) tet.Add(long.Parse(t[first + q]));
                }
            }
            else
            {
                int k = 1, blocks = int.Parse(split(N[0])[0]);
                for (int b = 0; b < blocks; ++b)
                {
                    int cnt = int.Parse(split(N[k++])[3]);
                    for (int i = 0; i < cnt; ++i) tag.Add(long.Parse(N[k + i]));
                    k += cnt;
                    for (int i = 0; i < cnt; ++i) { var t = split(N[k + i]); xyz.Add(new Vector3(float.Parse(t[0], inv), float.Parse(t[1], inv), float.Parse(t[2], inv))); }
                    k += cnt;
                }
                k = 1; blocks = int.Parse(split(E[0])[0]);
                for (int b = 0; b < blocks; ++b)
                {
                    var h = split(E[k++]); int etype = int.Parse(h[2]), cnt = int.Parse(h[3]);
                    if (etype == 4) for (int i = 0; i < cnt; ++i) { var t = split(E[k + i]); for (int q = 1; q <= 4; ++q) tet.Add(long.Parse(t[q])); }
                    k += cnt;
                }
            }
            if (tet.Count == 0) throw new FormatException("no 4-node tetrahedra (element type 4) in the file");
            var indexOf = new Dictionary<long, int>();
            for (int i = 0; i < tag.Count; ++i) indexOf[tag[i]] = i;
            var used = new SortedSet<int>();
            foreach (var t in tet) used.Add(indexOf[t]);
            var remap = new Dictionary<int, int>(); var nodes = new List<Vector3>();
            foreach (int u in used) { remap[u] = nodes.Count; nodes.Add(xyz[u]); }
            var tets = new int[tet.Count];
            for (int i = 0; i < tet.Count; ++i) tets[i] = remap[indexOf[tet[i]]];
            FromTetMesh(sb, nodes.ToArray(), tets);
        }

        static void AddEdge(SortedDictionary<(int, int), List<int>> edges, int a, int b, int opp)
        {
            var k = (Math.Min(a, b), Math.Max(a, b));
            if (!edges.TryGetValue(k, out var l)) { l = new List<int>(); edges[k] = l; }
            l.Add(opp);
        }
    }
}
