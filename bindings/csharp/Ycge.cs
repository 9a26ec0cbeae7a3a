// bindings/csharp/Ycge.cs - P/Invoke declarations of include/ycge.h (libycge_hip.so, ABI 8).
//
// Drop into ConsoleGame/RayTracing/Native/.  Style follows the reference's own P/Invokes ([DllImport] + [StructLayout(LayoutKind.Sequential)]
// blittable structs, Renderer/Win32TerminalRenderer.cs:122-151).  Every structure is blittable (fixed buffers, no marshalled arrays), so `ref`
// passes the address of the managed value itself.
//
// MACHINE-CHECKED: tests/test_csharp_binding.py parses this file and holds every struct (field order, types, offsets, size) and every
// DllImport (name, parameter kinds, return kind) to yetanotherconsolegameengine_amd/abi.py, which tests/test_host_cpu.py holds to the
// header through the built library.  One declaration per line, one struct field list per `public` statement: keep it parseable.
using System;
using System.Runtime.InteropServices;

namespace ConsoleGame.RayTracing.Native
{
    [StructLayout(LayoutKind.Sequential)]
    public struct YVec3               // ycge_vec3
    {
        public float X, Y, Z;
        public YVec3(float x, float y, float z) { X = x; Y = y; Z = z; }
        public YVec3(ConsoleGame.RayTracing.Vec3 v) { X = v.X; Y = v.Y; Z = v.Z; }
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YMaterial           // ycge_material
    {
        public int Kind;              // YMaterialKind
        public YVec3 Albedo;
        public YVec3 AlbedoB;
        public float CheckerScale;
        public float Specular;
        public float Reflectivity;
        public YVec3 Emission;
        public float Transparency;
        public float IndexOfRefraction;
        public YVec3 TransmissionColor;
        public int Texture;           // index into YScene.Textures (Kind == Textured)
        public int Reserved;
        public double TextureWeight;  // Material.TextureWeight / UVScale as the doubles they are (Material.cs:17-18)
        public double UvScale;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YTexture            // ycge_texture
    {
        public int Width, Height;
        public IntPtr Pixels;         // static: Texture.pixels (int[] of RGBA32.ToInt()), pinned for the call
        public int FrameBytesPerPixel; // live (Texture(IFrameReader, useRGBA, flipU, flipV)): 3 = BGR, 4 = BGRA; 0 = static
        public int FlipU, FlipV;
        public IntPtr Frame;          // live: reader.GetCurrentFramePtr() at upload (may be null)
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct YPrim        // ycge_prim
    {
        public int Type;              // YPrimType
        public int Material;
        public int Ref;
        public int Reserved;
        public fixed float P[12];
        public float Specular;
        public float Reflectivity;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YMesh               // ycge_mesh
    {
        public IntPtr Triangles;      // 9 floats per triangle: A, B, C
        public int NTriangles;
        public int Material;
        public IntPtr TriMaterial;    // optional int per triangle
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YVoxelLookup        // ycge_voxel_lookup
    {
        public int MatId, MetaId, Material;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YGrid               // ycge_grid
    {
        public int Nx, Ny, Nz;
        public YVec3 MinCorner;
        public YVec3 VoxelSize;
        public IntPtr Cells;          // 2 ints per cell, (ix * ny + iy) * nz + iz
        public IntPtr Lookup;         // YVoxelLookup[]
        public int NLookup;
        public int DefaultMaterial;
        public int Wireframe;
        public float WireWidthFraction;
        public float WireMaxDistance;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YLight              // ycge_light
    {
        public YVec3 Position;
        public YVec3 Color;
        public float Intensity;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YScene              // ycge_scene
    {
        public IntPtr Materials;
        public int NMaterials;
        public IntPtr Prims;
        public int NPrims;
        public IntPtr Meshes;
        public int NMeshes;
        public IntPtr Grids;
        public int NGrids;
        public IntPtr Lights;
        public int NLights;
        public YVec3 AmbientColor;
        public float AmbientIntensity;
        public YVec3 BackgroundTop;
        public YVec3 BackgroundBottom;
        public int IsVolumeScene;
        public int NTextures;
        public IntPtr Textures;
        public int HasDynamicTextures;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct YConfig      // ycge_config - fill with ycge_config_default, then set the size
    {
        public int AbiVersion, FbWidth, FbHeight, SuperSample;
        public float FovDeg;
        public int Device, Rank, WorldSize;
        public int DiffuseBounces, MaxMirrorBounces, MaxRefractions;
        public float MirrorThreshold, Eps;
        public ulong SeedSalt;
        public float TaaAlpha, MotionTransReset, MotionRotReset, DiffuseSigmaDeg;
        public int TaaClampRadius;
        public float TaaLuminancePad;
        public int AtrousIterations;
        public float AtrousCPhi, AtrousNPhi, AtrousZPhi, AtrousAPhi;
        public int CaptureDebug, CountWork;
        public int SlabAlbedo;
        public int NDevices;
        public fixed int Devices[8];
        public int AtrousInplaceExact;
        public int TileRing;
        public int MultiDeviceExchange;   // YExchange: 0 = the peers push their tiles (default), 1 = RCCL all-gather inside ycge_render_frame
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct YFrameStats  // ycge_frame_stats
    {
        public long Frame;
        public int HistoryReset, FanBlocks;
        public double TraceMs, TaaMs, PostMs, TotalMs;
        public ulong NRays, NBox, NTri, NPrim, NVox;
        public float Exposure, ExposureSerialChunks;
        public ulong NRaysDark;
        public int NDevicesTraced;
        public fixed int DeviceTiles[8];
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct YFlightInfo         // ycge_flight_info
    {
        public int TwoTraceStreams, PlacedGate, PostGate, PostPair, FramesOutstanding, StagePipeline;
        public ulong PlacedWaits;
    }

    public enum YStatus { Ok = 0, InvalidArg = -1, NoScene = -2, Device = -3, Unsupported = -4, OutOfMemory = -5, StackDepth = -6, NoDeviceCode = -7, Internal = -8 }
    public enum YMaterialKind { Constant = 0, Checker = 1, Textured = 2 }
    public enum YExchange { PeerPush = 0, Rccl = 1 }
    public enum YPrimType { Sphere = 0, Plane = 1, Disk = 2, XYRect = 3, XZRect = 4, YZRect = 5, Box = 6, CylinderY = 7, Triangle = 8, Mesh = 9, VolumeGrid = 10 }
    public enum YBuffer { Rays = 0, PrimId = 1, SubId = 2, HitT = 3, CurrentHdr = 4, GAlbedo = 5, GNormal = 6, GDepth = 7, SkyMask = 8, TaaHistory = 9, PrevNormal = 10, PrevDepth = 11, PrevSky = 12, Denoised = 13, RngState = 14 }
    public enum YAccel { SceneNodes = 0, SceneLeafIndex = 1, MeshNodes = 2, MeshLeafIndex = 3 }

    internal static unsafe class Ycge
    {
        public const int AbiVersion = 9;
        public const int MaxDevices = 8;
        private const string Lib = "ycge_hip";       // libycge_hip.so

        [DllImport(Lib)] public static extern int ycge_config_default(ref YConfig cfg);
        [DllImport(Lib)] public static extern int ycge_create(ref YConfig cfg, out IntPtr ctx);
        [DllImport(Lib)] public static extern void ycge_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr ycge_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int ycge_scene_upload(IntPtr ctx, ref YScene scene);
        [DllImport(Lib)] public static extern int ycge_validate_scene(ref YScene scene, byte* msg, UIntPtr msgBytes);
        [DllImport(Lib)] public static extern int ycge_scene_update_lights(IntPtr ctx, YLight* lights, int nLights, YVec3* ambientColor, float ambientIntensity, YVec3* top, YVec3* bottom);
        [DllImport(Lib)] public static extern int ycge_scene_update_objects(IntPtr ctx, YPrim* prims, int nPrims);
        [DllImport(Lib)] public static extern int ycge_scene_update_texture(IntPtr ctx, int textureIndex, IntPtr frame, UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_resize(IntPtr ctx, int fbWidth, int fbHeight, int superSample);
        [DllImport(Lib)] public static extern int ycge_set_camera(IntPtr ctx, float* pos, float yaw, float pitch, float fovDeg);
        [DllImport(Lib)] public static extern int ycge_render_frame(IntPtr ctx, float* outTopBottomSdr, YFrameStats* stats);
        [DllImport(Lib)] public static extern int ycge_render_frame_async(IntPtr ctx);
        [DllImport(Lib)] public static extern int ycge_render_frame_async_sdr(IntPtr ctx, float* outTopBottomSdr);
        [DllImport(Lib)] public static extern int ycge_wait(IntPtr ctx);
        [DllImport(Lib)] public static extern int ycge_async_trace_times(IntPtr ctx, float* msOut, int capacity, out int nOut);
        [DllImport(Lib)] public static extern int ycge_flight_query(IntPtr ctx, out YFlightInfo info);
        [DllImport(Lib)] public static extern int ycge_exchange_query(IntPtr ctx, out int mode, out int world);
        // one process per GPU (INTEGRATION.md section 5): device pointers and HIP streams travel as IntPtr
        [DllImport(Lib)] public static extern int ycge_tile_slab_bytes(IntPtr ctx, out UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_trace_tiles(IntPtr ctx, IntPtr dSlab, IntPtr stream, YFrameStats* stats);
        [DllImport(Lib)] public static extern int ycge_resolve_gathered(IntPtr ctx, IntPtr dAllSlabs, IntPtr stream, float* outTopBottomSdr, YFrameStats* stats);
        [DllImport(Lib)] public static extern int ycge_halo_counts(IntPtr ctx, long* sendCounts, long* recvCounts);
        [DllImport(Lib)] public static extern int ycge_history_slab_bytes(IntPtr ctx, out UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_trace_tiles_resident(IntPtr ctx, IntPtr dHaloSend, IntPtr stream, YFrameStats* stats);
        [DllImport(Lib)] public static extern int ycge_trace_tiles_resident_batch(IntPtr ctx, int n, float* poses, IntPtr* dHaloSends, IntPtr stream);
        [DllImport(Lib)] public static extern int ycge_resolve_tiles_resident(IntPtr ctx, IntPtr dHaloRecv, IntPtr dHistorySlab, IntPtr stream, YFrameStats* stats);
        [DllImport(Lib)] public static extern int ycge_unpack_history(IntPtr ctx, IntPtr dAllHistorySlabs, IntPtr stream);
        // tests / tools
        [DllImport(Lib)] public static extern int ycge_read_buffer(IntPtr ctx, int which, IntPtr dst, UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_set_frame_counter(IntPtr ctx, long frameCounter);
        [DllImport(Lib)] public static extern int ycge_read_timed_steps(IntPtr ctx, out ulong laneSteps);
        [DllImport(Lib)] public static extern int ycge_device_count();
        // page-locked memory for the SDR frame (ABI 8): the library's own, or whole pages the caller owns
        [DllImport(Lib)] public static extern UIntPtr ycge_host_page_size();
        [DllImport(Lib)] public static extern int ycge_alloc_host_buffer(UIntPtr bytes, out IntPtr buffer);
        [DllImport(Lib)] public static extern int ycge_free_host_buffer(IntPtr buffer);
        [DllImport(Lib)] public static extern int ycge_pin_host_buffer(IntPtr buffer, UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_unpin_host_buffer(IntPtr buffer);
        [DllImport(Lib)] public static extern int ycge_accel_size(IntPtr ctx, int which, int index, out UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_read_accel(IntPtr ctx, int which, int index, IntPtr dst, UIntPtr bytes);
        [DllImport(Lib)] public static extern int ycge_device_info(IntPtr ctx, byte* name, UIntPtr nameBytes, out int computeUnits);

        /// <summary>Status codes back into the exceptions the reference throws on misuse (e.g. Scenes/Scene.cs:73).</summary>
        public static void Check(IntPtr ctx, int rc)
        {
            if (rc == 0) return;
            string msg = Marshal.PtrToStringAnsi(ycge_last_error(ctx)) ?? "";
            switch ((YStatus)rc)
            {
                case YStatus.NoScene: throw new InvalidOperationException(msg);
                case YStatus.InvalidArg: throw new ArgumentException(msg);
                case YStatus.Unsupported: throw new NotSupportedException(msg);
                case YStatus.OutOfMemory: throw new OutOfMemoryException(msg);
                case YStatus.Internal: throw new InvalidOperationException("ycge internal error (a C++ exception was stopped at the C-ABI): " + msg);
                default: throw new Exception("ycge " + ((YStatus)rc) + ": " + msg);
            }
        }
    }
}
