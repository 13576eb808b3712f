// SoftbodyNative.cs — P/Invoke layer over libsoftbody_mi355x.so (include/softbody.h + softbody_group.h / softbody_plan.h / softbody_debug.h, ABI version 8).
//
// One [DllImport] per exported function, same name and argument order as the header; the Python twin
// used by the test-suite is softbodyunity_amd/native.py (tests/test_abi.py keeps the three in sync).
// The reference repository holds no C# to mirror (/root/reference/README.md:1 is its only line), so
// this file is the builder-defined binding SURVEY.md §8b calls for.
//
// NOT COMPILED IN THIS ENVIRONMENT: no C# toolchain (dotnet/mono/mcs/csc) and no UnityEngine.dll exist
// in the build image; see INTEGRATION.md.
using System;
using System.Runtime.InteropServices;

namespace SoftbodyMI355X
{
    [StructLayout(LayoutKind.Sequential)]
    public struct SbDesc
    {
        public int device;
        public int rank;
        public int world;
        public int partDimX, partDimY, partDimZ;
        public float gravityX, gravityY, gravityZ;
        public float damping;
        public int tileParticles;
        public int useGraph;
        public int partition;        // SoftbodyNative.Partition*
        public uint planFlags;       // SoftbodyNative.PlanNo*: every rank the same (checked across the ranks at sb_finalize)
        public int haloTransport;    // SoftbodyNative.Transport*
        public int haloSchedule;     // SoftbodyNative.Schedule*
        public uint debugFlags;      // test-only, leave 0
        public int reserved0, reserved1, reserved2;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SbStats
    {
        public long nParticlesOwned, nParticlesLocal;
        public long nDistanceLocal, nVolumeLocal, nBendingLocal;
        public int nTilings, nGlobalColours;
        public long nTilesT0, nTilesT1, tileConstraintsT0, tileConstraintsT1;
        public long constraintsInTiles, constraintsInGlobal;
        public long haloParticlesT1, haloParticlesGlobal, deviceBytes;
        public long nT2Layers, nT2Tiles, t2Constraints;
        // compulsory HBM bytes of one launch: mid-tick on T0 / T1, first, last kernel of a tick, all T2 layers of a substep
        public long launchBytesMidT0, launchBytesMidT1, launchBytesFirst, launchBytesLast, launchBytesT2;
        // partition (world > 1)
        public int partition, haloPeers;
        public long partitionCost, partitionCostMax, partitionCostTotal;
        public long haloParticlesRecv;
        public ulong planHash;
        public int haloSchedule, haloUnpackFused;
        public long readbackPeeks, readbackPeekTiles, ticksFused, ticksFusedKinematic;
        public long lanePackedTilesT0, lanePackedTilesT1;
        public int haloAutoState, haloAutoTicks;          // SB_SCHEDULE_AUTO's calibration: 0 none, 1 measuring, 2 decided; ticks timed
        public double haloAutoMsSerial, haloAutoMsOverlap; // per tick, slowest rank
          // workgroups whose spring slots are lane-packed (16 B per lane)   // position reads served by a peek; T0 workgroups of one render-set peek (-1: none set up)
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SbValidateReport
    {
        public long tilesChecked, groupsChecked, constraintsChecked;
        // 0 index out of range, 1 a particle twice in one group / colour, 2 a particle staged by two tiles of one launch,
        // 3 a group's data leaves the tile's stream, 4 malformed run table / particle list, 5 wave items disagree with the group words
        public long errors0, errors1, errors2, errors3, errors4, errors5;
        public int firstStage, firstTile, firstGroup, firstKind;   // -1 = no error
    }

    [StructLayout(LayoutKind.Sequential, CharSet = CharSet.Ansi)]
    public struct SbRuntimeInfo
    {
        public int hipRuntimeVersion, hipDriverVersion, rcclVersion, rcclHeaderVersion;
        public int rcclWasResident, captureSerialOk, captureOverlapOk, reserved;
        [MarshalAs(UnmanagedType.ByValTStr, SizeConst = 256)] public string hipLibrary;
        [MarshalAs(UnmanagedType.ByValTStr, SizeConst = 256)] public string rcclLibrary;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SbPlanOpts
    {
        public int rank, world;
        public int partDimX, partDimY, partDimZ;
        public int tileParticles;
        public int partition;
        public uint planFlags;
        public IntPtr domain;        // SbDomain* (sharded authoring: the input is this rank's window of a larger mesh) or IntPtr.Zero
        public IntPtr globalId;      // int* ids of the window's particles in the whole mesh, ascending, or IntPtr.Zero
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SbDomain
    {
        public long nGlobal;
        public double loX, loY, loZ, hiX, hiY, hiZ;
        public double spacing;
        public double fill;          // fraction of the bounding box the mesh occupies (1 for a lattice)
        public int fourVertexConstraints, reserved;
    }

    // include/softbody_debug.h: A/B measurement switches (a product host never sets them) and exchange timing
    [StructLayout(LayoutKind.Sequential)]
    public struct SbTuning
    {
        public uint flags;           // SoftbodyNative.Tune*
        public int tileLanes, quadLanes, narrowMinTiles, storeThroughMaxTiles, storeThroughLarge, peekMinTiles, ldsPadBytes, winDwords, prevOffsetBytes;
        public int reserved0, reserved1, reserved2, reserved3, reserved4, reserved5;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SbExchangeTiming
    {
        public long exchanges;
        public double packMs, transportMs, totalMs, exposedWaitMs;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SbPhaseInfo
    {
        public int kind, type, tiling, haloSlot;
        public long orderBegin, orderEnd, taskBegin, taskEnd;
    }

    public static class SoftbodyNative
    {
        const string Lib = "softbody_mi355x";
        const CallingConvention CC = CallingConvention.Cdecl;
        public const int UniqueIdBytes = 128;
        public const int PartitionAuto = 0, PartitionBlocks = 1, PartitionRcb = 2;
        public const uint PlanNoT2 = 1, PlanNoThirdList = 2, PlanNoClusterLayers = 4, PlanNoMixedGroups = 8, PlanNoBankOrder = 16, PlanNoTileMerge = 32;
        public static uint PlanBalancedLists(int n) => (uint)n << 8;   // irregular meshes: 1..3 balanced extra lists; 0 = default (2)
        public const int TransportRccl = 0, TransportPeer = 1;
        public const int ScheduleAuto = 0, ScheduleSerialEager = 1, ScheduleSerialGraph = 2, ScheduleOverlapEager = 3, ScheduleOverlapGraph = 4;
        public const uint GroupWalk = 1;       // sb_group_create flags: no plugin threads, the calling thread walks the tick across the ranks
        public const uint GroupWholeMesh = 2;  // never cut windows: every rank plans the whole mesh (sb_group_finalize)

        [DllImport(Lib, CallingConvention = CC)] public static extern void sb_desc_default(ref SbDesc d);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_create(ref SbDesc desc, out IntPtr solver);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_destroy(IntPtr s);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_particles(IntPtr s, IntPtr posXyz, IntPtr velXyz, IntPtr invMass, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_rest_positions(IntPtr s, IntPtr restXyz, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_distance_constraints(IntPtr s, IntPtr ij, IntPtr restLen, int m, float compliance);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_volume_constraints(IntPtr s, IntPtr ijkl, IntPtr restVol, int m, float compliance);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_bending_constraints(IntPtr s, IntPtr ijkl, IntPtr restCosSin, int m, float compliance);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_ground_plane(IntPtr s, float nx, float ny, float nz, float d, int enabled);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_domain(IntPtr s, ref SbDomain domain, IntPtr globalId, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_domain_from_mesh(IntPtr restXyz, int n, IntPtr distIj, int mD, IntPtr volIjkl, int mV, IntPtr bendIjkl, int mB, out SbDomain domain);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_domain_window(ref SbDomain domain, ref SbPlanOpts opts, double[] lo3, double[] hi3);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_finalize(IntPtr s);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_comm_unique_id(IntPtr outId128);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_comm_init(IntPtr s, IntPtr id128);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_step(IntPtr s, float dt, int substeps);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_get_positions(IntPtr s, IntPtr posXyzOut, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_get_velocities(IntPtr s, IntPtr velXyzOut, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_state(IntPtr s, IntPtr posXyz, IntPtr velXyz, int n);
        // kinematic particles (attachments): move particles with inverse mass 0 between two ticks (SPEC.md 2); a rank applies the ones it owns
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_kinematic_positions(IntPtr s, IntPtr ids, IntPtr posXyz, int count);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_readback_begin(IntPtr s);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_readback_end(IntPtr s, out IntPtr posXyz);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_render_triangles(IntPtr s, int[] triAbc, int m);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_readback_get_normals(IntPtr s, out IntPtr normalXyz);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_readback_render_set_only(IntPtr s, int renderSetOnly);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_readback_get_render_set(IntPtr s, out IntPtr ids, out int count);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_get_owner(IntPtr s, IntPtr ownerRankOut, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_profile_begin(IntPtr s);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_profile_end(IntPtr s, out float elapsedMs);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_synchronize(IntPtr s);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_step_profiled(IntPtr s, float dt, int substeps, IntPtr slotMsOut, IntPtr slotLaunchesOut, int nSlots);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_launch(IntPtr s, float dt, int substeps, int it, int gcolour);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_halo_pack(IntPtr s, int slot, IntPtr hostOut, long capacityFloats, out long countFloats);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_halo_unpack(IntPtr s, int slot, IntPtr hostIn, long countFloats);
        // table validator ("race detector"): a GPU kernel re-reads every table the tile kernels read and counts violations (softbody.h)
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_validate(IntPtr s, int injectFault, out SbValidateReport report);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_get_stats(IntPtr s, out SbStats stats);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_runtime_info(out SbRuntimeInfo info);
        // opt-in peer-store halo transport (SbDesc.haloTransport = TransportPeer): see softbody.h
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_peer_mailbox_handle(IntPtr s, byte[] handle64);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_peer_connect(IntPtr s, int rank, byte[] handle64, IntPtr sameProcessPeer);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_build(IntPtr restXyz, int n, IntPtr distIj, int mD, IntPtr volIjkl, int mV, IntPtr bendIjkl, int mB, ref SbPlanOpts opts, out IntPtr plan);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_destroy(IntPtr plan);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_get_plan(IntPtr s, out IntPtr plan);
        [DllImport(Lib, CallingConvention = CC)] public static extern long sb_plan_order_count(IntPtr plan);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_order(IntPtr plan, int parity, IntPtr typeOut, IntPtr idOut);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_phase_count(IntPtr plan, int parity);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_phases(IntPtr plan, int parity, [Out] SbPhaseInfo[] phases);
        [DllImport(Lib, CallingConvention = CC)] public static extern long sb_plan_task_count(IntPtr plan, int parity);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_tasks(IntPtr plan, int parity, IntPtr taskOffOut);
        [DllImport(Lib, CallingConvention = CC)] public static extern long sb_plan_group_count(IntPtr plan, int parity);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_groups(IntPtr plan, int parity, IntPtr groupOffOut);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_owner(IntPtr plan, IntPtr ownerRankOut);
        [DllImport(Lib, CallingConvention = CC)] public static extern long sb_plan_local_count(IntPtr plan, out long owned);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_local_particles(IntPtr plan, IntPtr globalIdOut);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_halo_slot_count(IntPtr plan);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_halo_counts(IntPtr plan, int slot, IntPtr sendCountPerRank, IntPtr recvCountPerRank);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_halo(IntPtr plan, int slot, int peer, IntPtr sendIds, IntPtr recvIds);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_pair_hashes(IntPtr plan, IntPtr outPerRank);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_plan_get_local_order_mask(IntPtr plan, int parity, IntPtr maskOut);
        // ---- one process, several GPUs (include/softbody_group.h): what Softbody.cs calls when deviceCount > 1 ----
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_create(ref SbDesc desc, int[] devices, int nDevices, uint flags, out IntPtr group);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_destroy(IntPtr g);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_particles(IntPtr g, IntPtr posXyz, IntPtr velXyz, IntPtr invMass, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_rest_positions(IntPtr g, IntPtr restXyz, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_distance_constraints(IntPtr g, IntPtr ij, IntPtr restLen, int m, float compliance);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_volume_constraints(IntPtr g, IntPtr ijkl, IntPtr restVol, int m, float compliance);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_bending_constraints(IntPtr g, IntPtr ijkl, IntPtr restCosSin, int m, float compliance);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_ground_plane(IntPtr g, float nx, float ny, float nz, float d, int enabled);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_finalize(IntPtr g);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_step(IntPtr g, float dt, int substeps);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_get_positions(IntPtr g, IntPtr posXyzOut, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_get_velocities(IntPtr g, IntPtr velXyzOut, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_state(IntPtr g, IntPtr posXyz, IntPtr velXyz, int n);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_kinematic_positions(IntPtr g, IntPtr ids, IntPtr posXyz, int count);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_render_triangles(IntPtr g, int[] triAbc, int m);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_set_readback_render_set_only(IntPtr g, int renderSetOnly);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_readback_begin(IntPtr g);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_readback_end(IntPtr g, out IntPtr posXyz);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_readback_get_normals(IntPtr g, out IntPtr normalXyz);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_readback_get_render_set(IntPtr g, out IntPtr ids, out int count);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_synchronize(IntPtr g);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_rank_count(IntPtr g);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_group_get_rank(IntPtr g, int rank, out IntPtr solver);
        // ---- include/softbody_debug.h: measurement harnesses only ----
        [DllImport(Lib, CallingConvention = CC)] public static extern void sb_tuning_default(ref SbTuning t);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_set_tuning(IntPtr s, ref SbTuning t);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_exchange_timing(IntPtr s, int enabled);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_exchange_timing_read(IntPtr s, out SbExchangeTiming t);
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_debug_last_words(int fd, byte[] text, long len, int exitCode);
        [DllImport(Lib, CallingConvention = CC)] public static extern IntPtr sb_last_error();
        [DllImport(Lib, CallingConvention = CC)] public static extern int sb_abi_version();

        public static string LastError() => Marshal.PtrToStringAnsi(sb_last_error());

        public static void Check(int rc, string what)
        {
            if (rc != 0) throw new InvalidOperationException($"{what} failed ({rc}): {LastError()}");
        }
    }
}
