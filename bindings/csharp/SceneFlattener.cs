// bindings/csharp/SceneFlattener.cs - a reference Scene as the POD records of include/ycge.h (the `Upload` of HipRaytraceWrapper).
//
// What the library needs and where the reference keeps it:
//   Scene.Objects IN ORDER (leaf order and tie-breaking depend on it)          Scenes/Scene.cs:12
//   one tagged material record per distinct Material / material function       Material.cs:7-18, Surfaces.cs:11, Scenes/Scenes.cs:408-428
//   a Mesh's triangles A, B, C and their materials                             Mesh.cs:16 (ctor argument; see Accessors.md: Mesh.Triangles)
//   a Box's material function and its two overrides                            BoundedObjects.cs:76 (private faces[] of public rects)
//   a VolumeGrid's cells, de-bricked through the arithmetic of IndexOf        VolumeGrid.cs:16-37, 235-252
//   a Texture's pixels, or its frame reader                                    Renderer/Texture.cs:15-20
//   lights, ambient, sky, `scene is VolumeScene`, HasDynamicTextures           Scene.cs:13-16, 30; RaytraceRenderer.cs:171, 761
//
// Private members are read through an additive accessor when the maintainer has added one (Accessors.md) and through reflection on the
// unmodified reference otherwise - except Mesh.Triangles, which the reference does not keep at all.
//
// Everything a FlatScene points to lives in unmanaged memory or in pinned managed arrays owned by the FlatScene; the library copies what
// it needs during ycge_scene_upload, so `using (var flat = SceneFlattener.Flatten(scene)) upload(flat)` is the whole life cycle.
using System;
using System.Collections.Generic;
using System.Reflection;
using System.Runtime.CompilerServices;
using System.Runtime.InteropServices;
using ConsoleGame.RayTracing.Objects;
using ConsoleGame.RayTracing.Scenes;
using ConsoleGame.Renderer;

namespace ConsoleGame.RayTracing.Native
{
    /// <summary>Construction sites that know the shape of a material function say so here (Accessors.md); the flattener asks this table
    /// first and probes the delegate only when it finds nothing.</summary>
    public static class MaterialFuncRegistry
    {
        private static readonly ConditionalWeakTable<Delegate, object> table = new ConditionalWeakTable<Delegate, object>();
        public static T Register<T>(T func, YMaterial shape) where T : Delegate { table.AddOrUpdate(func, shape); return func; }
        public static bool TryGet(Delegate func, out YMaterial shape)
        {
            if (func != null && table.TryGetValue(func, out object o)) { shape = (YMaterial)o; return true; }
            shape = default; return false;
        }
    }

    internal sealed unsafe class FlatScene : IDisposable
    {
        public YScene Scene;
        public YPrim[] Prims = Array.Empty<YPrim>();
        public YLight[] Lights = Array.Empty<YLight>();
        public readonly List<Texture> Textures = new List<Texture>();        // index in YScene.Textures -> the Texture it came from
        public readonly List<Mesh> MeshOwners = new List<Mesh>();            // the Mesh objects behind YScene.Meshes, in order
        public readonly List<VolumeGrid> GridOwners = new List<VolumeGrid>();  // the VolumeGrid objects behind YScene.Grids, in order
        public YMaterial[] Materials = Array.Empty<YMaterial>();
        private readonly List<GCHandle> pins = new List<GCHandle>();
        private readonly List<IntPtr> blocks = new List<IntPtr>();

        public IntPtr Pin(Array a) { if (a == null || a.Length == 0) return IntPtr.Zero; var h = GCHandle.Alloc(a, GCHandleType.Pinned); pins.Add(h); return h.AddrOfPinnedObject(); }
        public void* Alloc(long bytes) { void* p = NativeMemory.Alloc((nuint)Math.Max(1, bytes)); blocks.Add((IntPtr)p); return p; }
        public void Dispose()
        {
            foreach (GCHandle h in pins) if (h.IsAllocated) h.Free();
            foreach (IntPtr p in blocks) NativeMemory.Free((void*)p);
            pins.Clear(); blocks.Clear();
        }
    }

    internal static unsafe class SceneFlattener
    {
        private const BindingFlags Any = BindingFlags.Instance | BindingFlags.Public | BindingFlags.NonPublic;

        /// <summary>A member by name: a public accessor if one was added, the private field of the unmodified reference otherwise.</summary>
        private static T Member<T>(object o, params string[] names)
        {
            Type t = o.GetType();
            foreach (string n in names)
            {
                PropertyInfo p = t.GetProperty(n, Any);
                if (p != null && p.GetValue(o) is T pv) return pv;
                FieldInfo f = t.GetField(n, Any);
                if (f != null && f.GetValue(o) is T fv) return fv;
            }
            throw new MissingMemberException(t.Name, string.Join("/", names));
        }
        private static bool HasMember(object o, string name) => o.GetType().GetProperty(name, Any) != null || o.GetType().GetField(name, Any) != null;

        // ------------------------------------------------------------------------------------------------ materials
        private sealed class MaterialTable
        {
            public readonly List<YMaterial> Records = new List<YMaterial>();
            public readonly List<Texture> Textures = new List<Texture>();
            private readonly Dictionary<Delegate, int> byFunc = new Dictionary<Delegate, int>();

            public int Add(YMaterial m)
            {
                for (int i = 0; i < Records.Count; i++) if (Same(Records[i], m)) return i;
                Records.Add(m); return Records.Count - 1;
            }
            private static bool Same(YMaterial a, YMaterial b)
            {
                return new ReadOnlySpan<byte>(&a, sizeof(YMaterial)).SequenceEqual(new ReadOnlySpan<byte>(&b, sizeof(YMaterial)));
            }
            public int TextureIndex(Texture t)
            {
                int i = Textures.IndexOf(t);
                if (i < 0) { Textures.Add(t); i = Textures.Count - 1; }
                return i;
            }
            /// <summary>a plain Material struct: constant, or textured when it carries a DiffuseTexture (RaytraceRenderer.cs:724-735)</summary>
            public YMaterial Record(Material m)
            {
                var r = new YMaterial();
                r.Kind = (int)YMaterialKind.Constant;
                r.Albedo = new YVec3(m.Albedo); r.AlbedoB = new YVec3(m.Albedo); r.CheckerScale = 1.0f;
                r.Specular = (float)m.Specular; r.Reflectivity = (float)m.Reflectivity; r.Emission = new YVec3(m.Emission);
                r.Transparency = (float)m.Transparency; r.IndexOfRefraction = (float)m.IndexOfRefraction; r.TransmissionColor = new YVec3(m.TransmissionColor);
                r.Texture = -1; r.TextureWeight = m.TextureWeight; r.UvScale = m.UVScale;
                if (m.DiffuseTexture != null) { r.Kind = (int)YMaterialKind.Textured; r.Texture = TextureIndex(m.DiffuseTexture); }
                return r;
            }
            public int Add(Material m) => Add(Record(m));
            public int Add(Func<Vec3, Vec3, float, Material> f)
            {
                if (f == null) throw new ArgumentNullException(nameof(f), "an object without a material function");
                if (byFunc.TryGetValue(f, out int known)) return known;
                YMaterial r = MaterialFuncRegistry.TryGet(f, out YMaterial tagged) ? tagged : MaterialFuncProbe.Classify(f, this);
                int i = Add(r);
                byFunc[f] = i;
                return i;
            }
        }

        /// <summary>The reference passes opaque delegates Func&lt;Vec3, Vec3, float, Material&gt; (Surfaces.cs:11).  Three shapes occur in its
        /// builders (Scenes/Scenes.cs:408-428): a constant material (Solid, Emissive, or a captured Material), and Checker(a, b, scale) =
        /// parity of floor(x / scale) + floor(z / scale).  A delegate of another shape is refused - never approximated.</summary>
        private static class MaterialFuncProbe
        {
            private static readonly Vec3 Up = new Vec3(0.0f, 1.0f, 0.0f);

            private static bool SameButAlbedo(Material a, Material b)
            {
                return a.Specular == b.Specular && a.Reflectivity == b.Reflectivity && Eq(a.Emission, b.Emission) && a.Transparency == b.Transparency &&
                       a.IndexOfRefraction == b.IndexOfRefraction && Eq(a.TransmissionColor, b.TransmissionColor) && ReferenceEquals(a.DiffuseTexture, b.DiffuseTexture) &&
                       a.TextureWeight.Equals(b.TextureWeight) && a.UVScale.Equals(b.UVScale);
            }
            private static bool Eq(Vec3 a, Vec3 b) => a.X.Equals(b.X) && a.Y.Equals(b.Y) && a.Z.Equals(b.Z);
            private static bool Same(Material a, Material b) => SameButAlbedo(a, b) && Eq(a.Albedo, b.Albedo);

            private static IEnumerable<Vec3> ProbePoints(float scaleHint)
            {
                // a fixed pseudo-random cloud over several magnitudes (xorshift; no dependence on System.Random's implementation) ...
                uint s = 0x9E3779B9u;
                float Next() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (s >> 8) * (1.0f / 16777216.0f); }
                for (int k = 0; k < 4096; k++)
                {
                    float mag = k < 1024 ? 1.0f : k < 2048 ? 16.0f : k < 3072 ? 256.0f : 0.0625f;
                    yield return new Vec3((Next() - 0.5f) * 2.0f * mag, (Next() - 0.5f) * 2.0f * mag, (Next() - 0.5f) * 2.0f * mag);
                }
                // ... and the cell boundaries of the hinted scale with their neighbouring floats, both axes, both signs
                if (scaleHint > 0.0f)
                    for (int k = -8; k <= 8; k++)
                    {
                        float b = k * scaleHint;
                        foreach (float x in new[] { MathF.BitDecrement(b), b, MathF.BitIncrement(b) })
                        {
                            yield return new Vec3(x, 0.0f, 0.37f * scaleHint);
                            yield return new Vec3(0.37f * scaleHint, 0.0f, x);
                            yield return new Vec3(x, 1.0f, x);
                        }
                    }
            }

            public static YMaterial Classify(Func<Vec3, Vec3, float, Material> f, MaterialTable table)
            {
                Material first = f(new Vec3(1e-6f, 0.0f, 1e-6f), Up, 0.0f);        // cell (0, 0) of any checker with scale > 1e-6: colour A
                Material other = first; bool two = false;
                foreach (Vec3 p in ProbePoints(0.0f))
                {
                    Material m = f(p, Up, 0.0f);
                    if (Same(m, first)) continue;
                    if (!two) { other = m; two = true; continue; }
                    if (!Same(m, other)) throw new NotSupportedException("a material function with more than two values: register its shape with MaterialFuncRegistry (Accessors.md)");
                }
                if (!two) return table.Record(first);
                if (!SameButAlbedo(first, other) || first.DiffuseTexture != null)
                    throw new NotSupportedException("a two-valued material function that is not a checker of two albedos: register its shape with MaterialFuncRegistry");
                // the first colour change along +x from the origin is at x = scale (floor(x / scale) becomes 1): bracket it, bisect to adjacent floats
                float lo = 1e-6f, hi = 0.0f;
                for (float x = 1.0f / 1048576.0f; x <= 1048576.0f; x *= 2.0f)
                {
                    if (!Same(f(new Vec3(x, 0.0f, 1e-6f), Up, 0.0f), first)) { hi = x; break; }
                    lo = x;
                }
                if (hi == 0.0f) throw new NotSupportedException("no checker boundary found along +x below 2^20");
                while (MathF.BitIncrement(lo) < hi)
                {
                    float mid = lo + (hi - lo) * 0.5f;
                    if (mid <= lo || mid >= hi) break;
                    if (Same(f(new Vec3(mid, 0.0f, 1e-6f), Up, 0.0f), first)) lo = mid; else hi = mid;
                }
                foreach (float scale in new[] { hi, MathF.BitIncrement(hi), MathF.BitDecrement(hi), lo })
                {
                    if (!(scale > 0.0f)) continue;
                    bool ok = true;
                    foreach (Vec3 p in ProbePoints(scale))
                    {
                        // Scenes/Scenes.cs:420-424, operation for operation
                        int cx = (int)MathF.Floor(p.X / scale);
                        int cz = (int)MathF.Floor(p.Z / scale);
                        bool check = (cx + cz & 1) == 0;
                        if (!Same(f(p, Up, 0.0f), check ? first : other)) { ok = false; break; }
                    }
                    if (!ok) continue;
                    YMaterial r = table.Record(first);
                    r.Kind = (int)YMaterialKind.Checker;
                    r.AlbedoB = new YVec3(other.Albedo);
                    r.CheckerScale = scale;
                    return r;
                }
                throw new NotSupportedException("a two-coloured material function that no checker scale reproduces: register its shape with MaterialFuncRegistry");
            }
        }

        // ------------------------------------------------------------------------------------------------ objects
        private static YPrim Prim(YPrimType type, int material, float specular, float reflectivity, int reference, params float[] p)
        {
            var q = new YPrim();
            q.Type = (int)type; q.Material = material; q.Ref = reference; q.Reserved = 0;
            for (int i = 0; i < 12; i++) q.P[i] = i < p.Length ? p[i] : 0.0f;
            q.Specular = specular; q.Reflectivity = reflectivity;
            return q;
        }

        /// <summary>Scene.Objects, in order, as ycge_prim records.  A full upload passes onMesh / onGrid and numbers meshes and grids as it meets
        /// them; an object update passes the owners of the last upload and only references them by their position there.</summary>
        private static YPrim[] Objects(Scene scene, MaterialTable mats, List<Mesh> meshOwners, List<VolumeGrid> gridOwners, Action<Mesh> onMesh, Action<VolumeGrid> onGrid)
        {
            var prims = new List<YPrim>(scene.Objects.Count);
            int nMeshes = 0, nGrids = 0;
            foreach (Hittable o in scene.Objects)
            {
                switch (o)
                {
                    case Sphere s: prims.Add(Prim(YPrimType.Sphere, mats.Add(s.Mat), 0, 0, -1, s.Center.X, s.Center.Y, s.Center.Z, s.Radius)); break;
                    case Plane pl: prims.Add(Prim(YPrimType.Plane, mats.Add(pl.MaterialFunc), pl.Specular, pl.Reflectivity, -1, pl.Point.X, pl.Point.Y, pl.Point.Z, pl.Normal.X, pl.Normal.Y, pl.Normal.Z)); break;
                    case Disk d: prims.Add(Prim(YPrimType.Disk, mats.Add(d.MaterialFunc), d.Specular, d.Reflectivity, -1, d.Center.X, d.Center.Y, d.Center.Z, d.Normal.X, d.Normal.Y, d.Normal.Z, d.Radius)); break;
                    case XYRect r: prims.Add(Prim(YPrimType.XYRect, mats.Add(r.MaterialFunc), r.Specular, r.Reflectivity, -1, r.X0, r.X1, r.Y0, r.Y1, r.Z)); break;
                    case XZRect r: prims.Add(Prim(YPrimType.XZRect, mats.Add(r.MaterialFunc), r.Specular, r.Reflectivity, -1, r.X0, r.X1, r.Z0, r.Z1, r.Y)); break;
                    case YZRect r: prims.Add(Prim(YPrimType.YZRect, mats.Add(r.MaterialFunc), r.Specular, r.Reflectivity, -1, r.Y0, r.Y1, r.Z0, r.Z1, r.X)); break;
                    case Box b:
                    {
                        // the six faces share the ctor's function and overrides (BoundedObjects.cs:78-90); face 0 is an XYRect with public members
                        Func<Vec3, Vec3, float, Material> f; float spec, refl;
                        if (HasMember(b, "MaterialFunc")) { f = Member<Func<Vec3, Vec3, float, Material>>(b, "MaterialFunc"); spec = Member<float>(b, "Specular"); refl = Member<float>(b, "Reflectivity"); }
                        else { var face = (XYRect)Member<Hittable[]>(b, "faces")[0]; f = face.MaterialFunc; spec = face.Specular; refl = face.Reflectivity; }
                        prims.Add(Prim(YPrimType.Box, mats.Add(f), spec, refl, -1, b.Min.X, b.Min.Y, b.Min.Z, b.Max.X, b.Max.Y, b.Max.Z));
                        break;
                    }
                    case CylinderY c: prims.Add(Prim(YPrimType.CylinderY, mats.Add(c.Mat), 0, 0, -1, c.Center.X, c.Center.Y, c.Center.Z, c.Radius, c.YMin, c.YMax, c.Capped ? 1.0f : 0.0f)); break;
                    case Triangle t: prims.Add(Prim(YPrimType.Triangle, mats.Add(t.Mat), 0, 0, -1, t.A.X, t.A.Y, t.A.Z, t.B.X, t.B.Y, t.B.Z, t.C.X, t.C.Y, t.C.Z)); break;
                    case Mesh m:
                    {
                        int idx = meshOwners != null ? meshOwners.IndexOf(m) : -1;
                        if (onMesh != null) { idx = nMeshes; onMesh(m); }
                        if (idx < 0) throw new InvalidOperationException("a Mesh the last upload did not hold: upload the scene again");
                        prims.Add(Prim(YPrimType.Mesh, -1, 0, 0, idx)); nMeshes++;
                        break;
                    }
                    case VolumeGrid g:
                    {
                        int idx = gridOwners != null ? gridOwners.IndexOf(g) : -1;
                        if (onGrid != null) { idx = nGrids; onGrid(g); }
                        if (idx < 0) throw new InvalidOperationException("a VolumeGrid the last upload did not hold: upload the scene again");
                        prims.Add(Prim(YPrimType.VolumeGrid, -1, 0, 0, idx)); nGrids++;
                        break;
                    }
                    default:
                        throw new NotSupportedException("not a Hittable the path knows: " + o.GetType().Name);
                }
            }
            return prims.ToArray();
        }

        private static YMesh MeshRecord(Mesh m, MaterialTable mats, FlatScene flat)
        {
            // the triangles as MeshLoader.FromObj left them.  MeshBVH keeps A, B - A, C - A (MeshBVH.cs:32-36): B and C cannot be recovered from
            // those bit for bit, so the one accessor the reference needs is Mesh.Triangles (Accessors.md).
            if (!HasMember(m, "Triangles"))
                throw new NotSupportedException("Mesh.Triangles is missing: keep the ctor's List<Triangle> in Mesh (bindings/csharp/Accessors.md) - MeshBVH stores edges, not vertices");
            IReadOnlyList<Triangle> tris = Member<IReadOnlyList<Triangle>>(m, "Triangles");
            int n = tris.Count;
            float* v = (float*)flat.Alloc((long)n * 9 * sizeof(float));
            int* tm = null;
            int first = n > 0 ? mats.Add(tris[0].Mat) : mats.Add(new Material(Vec3.Zero, 0.0, 0.0, Vec3.Zero));
            int lastMat = first; bool uniform = true;
            var perTri = new int[n];
            for (int i = 0; i < n; i++)
            {
                Triangle t = tris[i];
                float* q = v + (long)i * 9;
                q[0] = t.A.X; q[1] = t.A.Y; q[2] = t.A.Z; q[3] = t.B.X; q[4] = t.B.Y; q[5] = t.B.Z; q[6] = t.C.X; q[7] = t.C.Y; q[8] = t.C.Z;
                perTri[i] = mats.Add(t.Mat);
                if (perTri[i] != first) uniform = false;
            }
            if (!uniform) { tm = (int*)flat.Alloc((long)n * sizeof(int)); for (int i = 0; i < n; i++) tm[i] = perTri[i]; }
            return new YMesh { Triangles = (IntPtr)v, NTriangles = n, Material = first, TriMaterial = (IntPtr)tm };
        }

        private static YGrid GridRecord(VolumeGrid g, MaterialTable mats, FlatScene flat)
        {
            int nx = Member<int>(g, "nx"), ny = Member<int>(g, "ny"), nz = Member<int>(g, "nz");
            int[] mat = Member<int[]>(g, "mat"), meta = Member<int[]>(g, "meta");
            var lookup = Member<Func<int, int, Material>>(g, "materialLookup");
            int nbx = (nx + 7) >> 3, nby = (ny + 7) >> 3;
            long nCells = (long)nx * ny * nz;
            int* cells = (int*)flat.Alloc(nCells * 2 * sizeof(int));
            var pairs = new Dictionary<long, int>();
            var table = new List<YVoxelLookup>();
            for (int ix = 0; ix < nx; ix++)
                for (int iy = 0; iy < ny; iy++)
                    for (int iz = 0; iz < nz; iz++)
                    {
                        // the address arithmetic of VolumeGrid.IndexOf / Morton3_3bits (VolumeGrid.cs:235-252): 8x8x8 bricks, x fastest across
                        // bricks, the low three bits of x, y, z interleaved inside one
                        int brick = (((iz >> 3) * nby) + (iy >> 3)) * nbx + (ix >> 3);
                        int lx = ix & 7, ly = iy & 7, lz = iz & 7;
                        int morton = ((lx & 1) << 0) | ((ly & 1) << 1) | ((lz & 1) << 2) | ((lx & 2) << 2) | ((ly & 2) << 3) | ((lz & 2) << 4) | ((lx & 4) << 4) | ((ly & 4) << 5) | ((lz & 4) << 6);
                        int src = brick * 512 + morton;
                        long dst = (((long)ix * ny + iy) * nz + iz) * 2;
                        int mid = mat[src], mt = meta[src];
                        cells[dst] = mid; cells[dst + 1] = mt;
                        if (mid <= 0) continue;                                   // air is never looked up (VolumeGrid.cs:161-189)
                        long key = ((long)mid << 32) | (uint)mt;
                        if (pairs.ContainsKey(key)) continue;
                        pairs[key] = table.Count;
                        table.Add(new YVoxelLookup { MatId = mid, MetaId = mt, Material = mats.Add(lookup(mid, mt)) });
                    }
            YVoxelLookup[] lut = table.ToArray();
            Vec3 mn = Member<Vec3>(g, "minCorner"), vs = Member<Vec3>(g, "voxelSize");
            return new YGrid
            {
                Nx = nx, Ny = ny, Nz = nz, MinCorner = new YVec3(mn), VoxelSize = new YVec3(vs),
                Cells = (IntPtr)cells, Lookup = flat.Pin(lut), NLookup = lut.Length, DefaultMaterial = -1,
                Wireframe = Member<bool>(g, "wireframe") ? 1 : 0,
                WireWidthFraction = Member<float>(g, "wireWidthFrac"), WireMaxDistance = Member<float>(g, "wireMaxDistance"),
            };
        }

        private static YTexture TextureRecord(Texture t, FlatScene flat)
        {
            var r = new YTexture { Width = t.width, Height = t.height };
            if (Member<bool>(t, "IsDynamic", "isDynamic"))
            {
                var reader = Member<NullEngine.Video.IFrameReader>(t, "DynamicReader", "dynamicReader");
                r.FrameBytesPerPixel = Member<int>(t, "DynamicBytesPerPixel", "dynamicBytesPerPixel");
                r.FlipU = Member<bool>(t, "FlipU", "flipU") ? 1 : 0; r.FlipV = Member<bool>(t, "FlipV", "flipV") ? 1 : 0;
                r.Frame = reader.GetCurrentFramePtr();
            }
            else r.Pixels = flat.Pin(Member<int[]>(t, "Pixels", "pixels"));
            return r;
        }

        public static YLight[] LightRecords(Scene scene)
        {
            var l = new YLight[scene.Lights.Count];
            for (int i = 0; i < l.Length; i++) l[i] = new YLight { Position = new YVec3(scene.Lights[i].Position), Color = new YVec3(scene.Lights[i].Color), Intensity = scene.Lights[i].Intensity };
            return l;
        }

        /// <summary>The whole scene, for ycge_scene_upload (what `scene.RebuildBVH()` at the end of the reference ctor stands for, RaytraceRenderer.cs:107).</summary>
        public static FlatScene Flatten(Scene scene)
        {
            var flat = new FlatScene();
            try
            {
                var mats = new MaterialTable();
                var meshes = new List<YMesh>(); var grids = new List<YGrid>();
                flat.Prims = Objects(scene, mats, null, null,
                    m => { meshes.Add(MeshRecord(m, mats, flat)); flat.MeshOwners.Add(m); },
                    g => { grids.Add(GridRecord(g, mats, flat)); flat.GridOwners.Add(g); });
                flat.Materials = mats.Records.ToArray();
                flat.Textures.AddRange(mats.Textures);
                var tex = new YTexture[mats.Textures.Count];
                for (int i = 0; i < tex.Length; i++) tex[i] = TextureRecord(mats.Textures[i], flat);
                flat.Lights = LightRecords(scene);
                YMesh[] meshArr = meshes.ToArray(); YGrid[] gridArr = grids.ToArray();
                flat.Scene = new YScene
                {
                    Materials = flat.Pin(flat.Materials), NMaterials = flat.Materials.Length,
                    Prims = flat.Pin(flat.Prims), NPrims = flat.Prims.Length,
                    Meshes = flat.Pin(meshArr), NMeshes = meshArr.Length,
                    Grids = flat.Pin(gridArr), NGrids = gridArr.Length,
                    Lights = flat.Pin(flat.Lights), NLights = flat.Lights.Length,
                    AmbientColor = new YVec3(scene.Ambient.Color), AmbientIntensity = scene.Ambient.Intensity,
                    BackgroundTop = new YVec3(scene.BackgroundTop), BackgroundBottom = new YVec3(scene.BackgroundBottom),
                    IsVolumeScene = scene is VolumeScene ? 1 : 0,
                    NTextures = tex.Length, Textures = flat.Pin(tex),
                    HasDynamicTextures = scene.HasDynamicTextures ? 1 : 0,
                };
                return flat;
            }
            catch { flat.Dispose(); throw; }
        }

        /// <summary>Only Scene.Objects again, against the materials / meshes / grids of the last upload - for ycge_scene_update_objects after an
        /// entity moved (Scene.cs:122-127).  Returns null when the objects need a record the last upload did not hold (a new material, mesh or
        /// grid): the caller uploads the scene again.</summary>
        public static YPrim[] ObjectsAgainst(Scene scene, FlatScene uploaded)
        {
            var mats = new MaterialTable();
            foreach (YMaterial m in uploaded.Materials) mats.Records.Add(m);
            mats.Textures.AddRange(uploaded.Textures);
            YPrim[] prims;
            try { prims = Objects(scene, mats, uploaded.MeshOwners, uploaded.GridOwners, null, null); }
            catch (InvalidOperationException) { return null; }
            return mats.Records.Count == uploaded.Materials.Length && mats.Textures.Count == uploaded.Textures.Count ? prims : null;
        }
    }
}
