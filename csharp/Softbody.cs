// Softbody.cs — the Unity component whose FixedUpdate path this repository accelerates.
//
// Start()       -> sb_group_create + sb_group_set_particles/sb_group_set_*_constraints + sb_group_finalize   (plan, partition, upload)
// FixedUpdate() -> sb_group_step(Time.fixedDeltaTime, substeps) + sb_group_get_positions                     (one tick, SPEC.md §2)
//                  asyncReadback: sb_group_readback_begin/end + GPU vertex normals instead (one tick of latency, no stall)
// OnDestroy()   -> sb_group_destroy
//
// A Unity player is ONE process: the component talks to the plugin through include/softbody_group.h, where `deviceCount` GPUs sit
// behind one handle (the mesh authored once, one call per tick, positions gathered in the component's numbering); deviceCount = 1 is
// a plain single-GPU solver behind the same calls. (The per-rank entry points of softbody.h -- sb_create with rank / world,
// sb_comm_init -- are what a one-process-per-GPU job uses: bench.py.)
//
// `useGpu = false` never creates a solver handle and needs no GPU: the schedule comes from the host-only
// planner (sb_plan_build / sb_plan_get_order / sb_plan_destroy) and the tick runs in SoftbodyCpuSolver (the C#
// restatement of SPEC.md) — the "reference C# CPU FixedUpdate path" that
// BASELINE.json:5 compares against. The reference repository ships no such component
// (/root/reference/README.md:1 is its only line); names follow Unity conventions.
//
// NOT COMPILED IN THIS ENVIRONMENT (no C# toolchain, no UnityEngine.dll). softbodyunity_amd/softbody.py
// is the line-for-line Python mirror the tests drive.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

namespace SoftbodyMI355X
{
    [RequireComponent(typeof(MeshFilter))]
    public class Softbody : MonoBehaviour
    {
        [Header("Solver")]
        [SerializeField] int substeps = 20;
        [SerializeField] Vector3 gravity = new Vector3(0f, -9.81f, 0f);
        [SerializeField] float damping = 0f;
        [SerializeField] float distanceCompliance = 0f;
        [SerializeField] float volumeCompliance = 0f;
        [SerializeField] float bendingCompliance = 0f;
        [SerializeField] bool groundPlane = false;
        [SerializeField] Vector3 groundNormal = new Vector3(0f, 1f, 0f);
        [SerializeField] float groundOffset = 0f;
        [Header("Device")]
        [SerializeField] bool useGpu = true;
        [Tooltip("First HIP device ordinal; with deviceCount > 1 the body is split spatially over devices device .. device + deviceCount - 1.")]
        [SerializeField] int device = 0;
        [Tooltip("GPUs of this node the body is partitioned over (1, 2, 4, 8): ghost particles travel over xGMI every substep (RCCL).")]
        [SerializeField] int deviceCount = 1;
        [Tooltip("Ghost exchange between the GPUs: 0 = RCCL send/recv (default), 1 = peer-store mailboxes (SoftbodyNative.Transport*).")]
        [SerializeField] int haloTransport = SoftbodyNative.TransportRccl;
        [Tooltip("Hand every GPU only its block of a lattice-like mesh (block partition) instead of letting every GPU plan the whole mesh.")]
        [SerializeField] bool blockPartition = false;
        [Tooltip("Never cut windows: every GPU is handed and plans the whole mesh, whatever the partition (the plugin decides by itself otherwise and verifies its choice).")]
        [SerializeField] bool wholeMeshOnEveryGpu = false;
        [Tooltip("Target particles per LDS tile; 0 = automatic (512, or 256 when the mesh has volume or bending constraints).")]
        [SerializeField] int tileParticles = 0;
        [Tooltip("Render from the previous tick's snapshot: the D2H copy and the normals (computed on the GPU) overlap the next tick.")]
        [SerializeField] bool asyncReadback = false;
        [Tooltip("With asyncReadback: copy only the particles the render triangles use (a volumetric body renders its surface only); the MeshFilter's mesh is rebuilt over that set.")]
        [SerializeField] bool renderSetOnly = false;

        // constraint graph (filled by an authoring script or SoftbodyMeshBuilder before Start)
        public Vector3[] restPositions;
        public Vector3[] positions;
        public Vector3[] velocities;
        public float[] inverseMass;
        public int[] distanceIJ; public float[] distanceRest;
        public int[] volumeIJKL; public float[] volumeRest;
        public int[] bendingIJKL; public float[] bendingRestCosSin;
        public int[] renderTriangles;          // particle indices, 3 per triangle (SoftbodyMeshBuilder: mesh.triangles through particleOfVertex)

        IntPtr handle = IntPtr.Zero;
        SoftbodyCpuSolver cpu;
        Mesh mesh;
        GCHandle posPin;
        Vector3[] normals;
        bool snapshotPending;

        void Start()
        {
            mesh = GetComponent<MeshFilter>().mesh;
            if (positions == null) { positions = mesh.vertices; restPositions = mesh.vertices; }
            int n = positions.Length;
            if (velocities == null) velocities = new Vector3[n];
            if (inverseMass == null) { inverseMass = new float[n]; for (int i = 0; i < n; ++i) inverseMass[i] = 1f; }

            if (!useGpu)
            {
                // CPU path (BASELINE.json:7, "CPU C# FixedUpdate only ... no GPU"): no solver handle, no device. The schedule
                // comes from the host-only planner entry points (sb_plan_build works on a machine without a GPU); the tick
                // itself is SoftbodyCpuSolver walking the order that planner publishes (SPEC.md §3).
                var opts = new SbPlanOpts { rank = 0, world = 1, tileParticles = tileParticles };
                IntPtr plan = IntPtr.Zero;
                Vector3[] rest = restPositions ?? positions;
                int md = distanceRest != null ? distanceRest.Length : 0;
                int mv = volumeRest != null ? volumeRest.Length : 0;
                int mb = bendingRestCosSin != null ? bendingRestCosSin.Length / 2 : 0;
                Pin(rest, r => Pin(distanceIJ ?? new int[0], dI => Pin(volumeIJKL ?? new int[0], vI => Pin(bendingIJKL ?? new int[0], bI =>
                    SoftbodyNative.Check(SoftbodyNative.sb_plan_build(r, n, dI, md, vI, mv, bI, mb, ref opts, out plan), "sb_plan_build")))));
                try
                {
                    long m = SoftbodyNative.sb_plan_order_count(plan);
                    var type = new byte[2][]; var id = new int[2][];
                    for (int parity = 0; parity < 2; ++parity)
                    {
                        type[parity] = new byte[m]; id[parity] = new int[m];
                        int par = parity;
                        Pin(type[par], t => Pin(id[par], i => SoftbodyNative.Check(SoftbodyNative.sb_plan_get_order(plan, par, t, i), "sb_plan_get_order")));
                    }
                    cpu = new SoftbodyCpuSolver(this, type, id);
                }
                finally { SoftbodyNative.sb_plan_destroy(plan); }
                posPin = GCHandle.Alloc(positions, GCHandleType.Pinned);
                return;
            }

            var d = new SbDesc();
            SoftbodyNative.sb_desc_default(ref d);
            // (device / rank / world of the descriptor are filled in per GPU by the group)
            d.gravityX = gravity.x; d.gravityY = gravity.y; d.gravityZ = gravity.z;
            d.damping = damping; d.tileParticles = tileParticles;
            d.haloTransport = haloTransport;
            d.partition = blockPartition ? SoftbodyNative.PartitionBlocks : SoftbodyNative.PartitionAuto;
            var devices = new int[Math.Max(deviceCount, 1)];
            for (int r = 0; r < devices.Length; ++r) devices[r] = device + r;
            SoftbodyNative.Check(SoftbodyNative.sb_group_create(ref d, devices, devices.Length, wholeMeshOnEveryGpu ? SoftbodyNative.GroupWholeMesh : 0u, out handle), "sb_group_create");

            // Vector3 is a blittable sequential struct of 3 floats: Vector3[] pins directly to float xyz
            Pin(positions, p => Pin(velocities, v => Pin(inverseMass, w =>
                SoftbodyNative.Check(SoftbodyNative.sb_group_set_particles(handle, p, v, w, n), "sb_group_set_particles"))));
            if (restPositions != null)
                Pin(restPositions, r => SoftbodyNative.Check(SoftbodyNative.sb_group_set_rest_positions(handle, r, n), "sb_group_set_rest_positions"));
            if (distanceRest != null && distanceRest.Length > 0)
                Pin(distanceIJ, i => Pin(distanceRest, r => SoftbodyNative.Check(
                    SoftbodyNative.sb_group_set_distance_constraints(handle, i, r, distanceRest.Length, distanceCompliance), "sb_group_set_distance_constraints")));
            if (volumeRest != null && volumeRest.Length > 0)
                Pin(volumeIJKL, i => Pin(volumeRest, r => SoftbodyNative.Check(
                    SoftbodyNative.sb_group_set_volume_constraints(handle, i, r, volumeRest.Length, volumeCompliance), "sb_group_set_volume_constraints")));
            if (bendingRestCosSin != null && bendingRestCosSin.Length > 0)
                Pin(bendingIJKL, i => Pin(bendingRestCosSin, r => SoftbodyNative.Check(
                    SoftbodyNative.sb_group_set_bending_constraints(handle, i, r, bendingRestCosSin.Length / 2, bendingCompliance), "sb_group_set_bending_constraints")));
            if (groundPlane)
                SoftbodyNative.Check(SoftbodyNative.sb_group_set_ground_plane(handle, groundNormal.x, groundNormal.y, groundNormal.z, groundOffset, 1), "sb_group_set_ground_plane");
            SoftbodyNative.Check(SoftbodyNative.sb_group_finalize(handle), "sb_group_finalize");
            posPin = GCHandle.Alloc(positions, GCHandleType.Pinned);
            if (asyncReadback && renderTriangles != null && renderTriangles.Length >= 3)
            {
                SoftbodyNative.Check(SoftbodyNative.sb_group_set_render_triangles(handle, renderTriangles, renderTriangles.Length / 3), "sb_group_set_render_triangles");
                normals = new Vector3[positions.Length];
                if (renderSetOnly)
                {
                    // the plugin's render set = the particles the triangles use, ascending: rebuild the mesh over exactly that set
                    SoftbodyNative.Check(SoftbodyNative.sb_group_set_readback_render_set_only(handle, 1), "sb_group_set_readback_render_set_only");
                    var used = new System.Collections.Generic.SortedSet<int>(renderTriangles);
                    var compactOf = new System.Collections.Generic.Dictionary<int, int>();
                    var compactPos = new Vector3[used.Count];
                    foreach (int p in used) { compactPos[compactOf.Count] = positions[p]; compactOf[p] = compactOf.Count; }
                    var tri = new int[renderTriangles.Length];
                    for (int k = 0; k < tri.Length; ++k) tri[k] = compactOf[renderTriangles[k]];
                    posPin.Free();
                    positions = compactPos; normals = new Vector3[compactPos.Length];
                    posPin = GCHandle.Alloc(positions, GCHandleType.Pinned);
                    mesh.Clear(); mesh.vertices = positions; mesh.triangles = tri;
                }
            }
        }

        void FixedUpdate()
        {
            if (useGpu && asyncReadback)
            {
                // show the snapshot taken after the PREVIOUS tick (its copy and its normals ran beside this tick's kernels),
                // then queue this tick's snapshot: the main thread never waits for the GPU to finish a tick
                SoftbodyNative.Check(SoftbodyNative.sb_group_step(handle, Time.fixedDeltaTime, substeps), "sb_group_step");
                SoftbodyNative.Check(SoftbodyNative.sb_group_readback_begin(handle), "sb_group_readback_begin");
                if (snapshotPending)
                {
                    SoftbodyNative.Check(SoftbodyNative.sb_group_readback_end(handle, out IntPtr pos), "sb_group_readback_end");
                    CopyVectors(pos, positions);
                    if (normals != null)
                    {
                        SoftbodyNative.Check(SoftbodyNative.sb_group_readback_get_normals(handle, out IntPtr nrm), "sb_group_readback_get_normals");
                        CopyVectors(nrm, normals);
                    }
                }
                snapshotPending = true;
                mesh.vertices = positions;
                if (normals != null) mesh.normals = normals; else mesh.RecalculateNormals();
                return;
            }
            if (useGpu)
            {
                SoftbodyNative.Check(SoftbodyNative.sb_group_step(handle, Time.fixedDeltaTime, substeps), "sb_group_step");
                SoftbodyNative.Check(SoftbodyNative.sb_group_get_positions(handle, posPin.AddrOfPinnedObject(), positions.Length), "sb_group_get_positions");
            }
            else
            {
                cpu.Step(Time.fixedDeltaTime, substeps);
            }
            mesh.vertices = positions;
            mesh.RecalculateNormals();
        }

        void OnDestroy()
        {
            if (posPin.IsAllocated) posPin.Free();
            if (handle != IntPtr.Zero) { SoftbodyNative.sb_group_destroy(handle); handle = IntPtr.Zero; }
        }

        /// <summary>Attachments: move pinned particles (inverse mass 0) to new positions before the next FixedUpdate; their constrained
        /// neighbours are pulled along (SPEC.md 2, sb_group_set_kinematic_positions: every GPU takes the pins it owns). ids index the particle arrays.</summary>
        public void MoveKinematic(int[] ids, Vector3[] targets)
        {
            if (ids.Length != targets.Length) throw new ArgumentException("ids and targets differ in length");
            if (handle == IntPtr.Zero)
            {
                // CPU branch: the same assignment on the arrays the C# solver steps
                for (int k = 0; k < ids.Length; ++k)
                {
                    if (inverseMass[ids[k]] != 0f) throw new ArgumentException("only pinned particles (inverse mass 0) are kinematic");
                    positions[ids[k]] = targets[k];
                }
                return;
            }
            Pin(ids, pi => Pin(targets, pt =>
            {
                int rc = SoftbodyNative.sb_group_set_kinematic_positions(handle, pi, pt, ids.Length);
                if (rc != 0) throw new InvalidOperationException(SoftbodyNative.LastError());
            }));
        }

        /// <summary>Debug aid (GPU backend): a GPU kernel re-reads every table the solver's kernels read and counts the one kind of
        /// fault that could make a tick racy -- a particle twice in a group of concurrently projected constraints, or in two tiles of
        /// one launch (softbody.h, sb_debug_validate). True = clean.</summary>
        public bool ValidateTables(out SbValidateReport report)
        {
            report = default(SbValidateReport);
            if (handle == IntPtr.Zero) return true;          // CPU branch: nothing uploaded
            bool clean = true;
            for (int r = 0; r < SoftbodyNative.sb_group_rank_count(handle); ++r)       // every GPU's tables
            {
                SoftbodyNative.Check(SoftbodyNative.sb_group_get_rank(handle, r, out IntPtr solver), "sb_group_get_rank");
                int rc = SoftbodyNative.sb_debug_validate(solver, 0, out report);
                if (rc != 0) throw new InvalidOperationException(SoftbodyNative.LastError());
                clean &= report.errors0 + report.errors1 + report.errors2 + report.errors3 + report.errors4 + report.errors5 == 0;
                if (!clean) break;
            }
            return clean;
        }

        // accessors for SoftbodyCpuSolver
        internal Vector3 Gravity => gravity;
        internal float Damping => damping;
        internal float ComplianceD => distanceCompliance;
        internal float ComplianceV => volumeCompliance;
        internal float ComplianceB => bendingCompliance;
        internal bool GroundPlane => groundPlane;
        internal Vector3 GroundNormal => groundNormal;
        internal float GroundOffset => groundOffset;

        static unsafe void CopyVectors(IntPtr src, Vector3[] dst)
        {
            fixed (Vector3* d = dst) Buffer.MemoryCopy((void*)src, d, (long)dst.Length * 12, (long)dst.Length * 12);
        }

        static void Pin<T>(T[] a, Action<IntPtr> f)
        {
            var h = GCHandle.Alloc(a, GCHandleType.Pinned);
            try { f(h.AddrOfPinnedObject()); } finally { h.Free(); }
        }
    }
}
