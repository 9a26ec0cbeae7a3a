// tools/ReferenceDump/SceneFile.cs - a YSC1 scene file (tools/scene_file.py) back into REFERENCE objects, through the reference's own
// public constructors: what the file holds is exactly what include/ycge.h takes, so "the reference on this file" and "the library on this
// file" are the same scene by construction.
using System;
using System.Collections.Generic;
using System.IO;
using System.Reflection;
using System.Runtime.CompilerServices;
using System.Runtime.InteropServices;
using ConsoleGame.RayTracing;
using ConsoleGame.RayTracing.Native;
using ConsoleGame.RayTracing.Objects;
using ConsoleGame.RayTracing.Scenes;
using ConsoleGame.Renderer;

namespace ReferenceDump
{
    internal sealed class LoadedScene
    {
        public int FbWidth, FbHeight, SuperSample;
        public float FovDeg, Yaw, Pitch;
        public Vec3 Pos;
        public Scene Scene;
        public readonly List<Mesh> Meshes = new List<Mesh>();
    }

    internal static unsafe class SceneFile
    {
        private sealed class Reader
        {
            private readonly byte[] b; private int o;
            public Reader(byte[] bytes) { b = bytes; }
            public T Take<T>() where T : unmanaged { T v = MemoryMarshal.Read<T>(new ReadOnlySpan<byte>(b, o, sizeof(T))); o += sizeof(T); return v; }
            public T[] Take<T>(long n) where T : unmanaged
            {
                var a = new T[n];
                MemoryMarshal.Cast<byte, T>(new ReadOnlySpan<byte>(b, o, checked((int)(n * sizeof(T))))).CopyTo(a);
                o += (int)(n * sizeof(T));
                return a;
            }
            public bool AtEnd => o == b.Length;
        }

        private static Vec3 V(YVec3 v) => new Vec3(v.X, v.Y, v.Z);

        public static LoadedScene Load(string path)
        {
            byte[] raw = File.ReadAllBytes(path);
            if (raw.Length > 2 && raw[0] == 0x1f && raw[1] == 0x8b)       // gzip (scene.ysc.gz: voxel worlds are mostly air)
            {
                using var src = new System.IO.Compression.GZipStream(new MemoryStream(raw), System.IO.Compression.CompressionMode.Decompress);
                using var dst = new MemoryStream();
                src.CopyTo(dst);
                raw = dst.ToArray();
            }
            var r = new Reader(raw);
            if (r.Take<uint>() != 0x31435359u /* "YSC1" */ || r.Take<uint>() != 1u) throw new InvalidDataException(path + ": not a YSC1 version 1 file");
            var ls = new LoadedScene();
            ls.FbWidth = r.Take<int>(); ls.FbHeight = r.Take<int>(); ls.SuperSample = r.Take<int>(); ls.FovDeg = r.Take<float>();
            float px = r.Take<float>(), py = r.Take<float>(), pz = r.Take<float>();
            ls.Pos = new Vec3(px, py, pz); ls.Yaw = r.Take<float>(); ls.Pitch = r.Take<float>();
            YScene ys = r.Take<YScene>();
            YMaterial[] mats = r.Take<YMaterial>(ys.NMaterials);
            YPrim[] prims = r.Take<YPrim>(ys.NPrims);
            YLight[] lights = r.Take<YLight>(ys.NLights);
            var meshRecords = new List<(YMesh rec, float[] tris, int[] triMat)>();
            for (int i = 0; i < ys.NMeshes; i++)
            {
                YMesh m = r.Take<YMesh>();
                float[] tris = r.Take<float>(9L * m.NTriangles);
                int[] tm = r.Take<uint>() != 0 ? r.Take<int>(m.NTriangles) : null;
                meshRecords.Add((m, tris, tm));
            }
            var gridRecords = new List<(YGrid rec, int[] cells, YVoxelLookup[] lut)>();
            for (int i = 0; i < ys.NGrids; i++)
            {
                YGrid g = r.Take<YGrid>();
                int[] cells = r.Take<int>(2L * g.Nx * g.Ny * g.Nz);
                gridRecords.Add((g, cells, r.Take<YVoxelLookup>(g.NLookup)));
            }
            var textures = new Texture[ys.NTextures];
            for (int i = 0; i < ys.NTextures; i++)
            {
                YTexture t = r.Take<YTexture>();
                if (t.FrameBytesPerPixel != 0) throw new NotSupportedException("live textures need a frame reader: not dumped");
                textures[i] = StaticTexture(t.Width, t.Height, r.Take<int>((long)t.Width * t.Height));
            }
            if (!r.AtEnd) throw new InvalidDataException(path + ": bytes left over");

            // ---- materials: the struct, or the delegate shapes of Scenes/Scenes.cs:408-428
            Material Mat(int i)
            {
                YMaterial m = mats[i];
                var mm = new Material(V(m.Albedo), m.Specular, m.Reflectivity, V(m.Emission), m.Transparency, m.IndexOfRefraction, V(m.TransmissionColor));
                if (m.Kind == (int)YMaterialKind.Textured) { mm.DiffuseTexture = textures[m.Texture]; mm.TextureWeight = m.TextureWeight; mm.UVScale = m.UvScale; }
                return mm;
            }
            var funcs = new Dictionary<int, Func<Vec3, Vec3, float, Material>>();
            Func<Vec3, Vec3, float, Material> MatFunc(int i)
            {
                if (funcs.TryGetValue(i, out var known)) return known;
                YMaterial m = mats[i];
                Func<Vec3, Vec3, float, Material> f;
                if (m.Kind == (int)YMaterialKind.Checker)
                {
                    Material a = Mat(i), b = Mat(i); b.Albedo = V(m.AlbedoB); float scale = m.CheckerScale;
                    f = (pos, n, u) =>
                    {
                        int cx = (int)MathF.Floor(pos.X / scale);
                        int cz = (int)MathF.Floor(pos.Z / scale);
                        return ((cx + cz) & 1) == 0 ? a : b;
                    };
                }
                else { Material c = Mat(i); f = (pos, n, u) => c; }
                funcs[i] = f;
                return f;
            }

            // ---- objects, in file order = Scene.Objects order
            Scene scene = ys.IsVolumeScene != 0 ? new VolumeScene() : new Scene();
            scene.Objects.Clear(); scene.Lights.Clear();
            foreach (YPrim q in prims)
            {
                YPrim qq = q;                         // (fixed buffers of a foreach variable cannot be indexed directly)
                float[] p = new float[12]; for (int k = 0; k < 12; k++) p[k] = qq.P[k];
                switch ((YPrimType)q.Type)
                {
                    case YPrimType.Sphere: scene.Objects.Add(new Sphere(new Vec3(p[0], p[1], p[2]), p[3], Mat(q.Material))); break;
                    case YPrimType.Plane: scene.Objects.Add(new Plane(new Vec3(p[0], p[1], p[2]), new Vec3(p[3], p[4], p[5]), MatFunc(q.Material), q.Specular, q.Reflectivity)); break;
                    case YPrimType.Disk: scene.Objects.Add(new Disk(new Vec3(p[0], p[1], p[2]), new Vec3(p[3], p[4], p[5]), p[6], MatFunc(q.Material), q.Specular, q.Reflectivity)); break;
                    case YPrimType.XYRect: scene.Objects.Add(new XYRect(p[0], p[1], p[2], p[3], p[4], MatFunc(q.Material), q.Specular, q.Reflectivity)); break;
                    case YPrimType.XZRect: scene.Objects.Add(new XZRect(p[0], p[1], p[2], p[3], p[4], MatFunc(q.Material), q.Specular, q.Reflectivity)); break;
                    case YPrimType.YZRect: scene.Objects.Add(new YZRect(p[0], p[1], p[2], p[3], p[4], MatFunc(q.Material), q.Specular, q.Reflectivity)); break;
                    case YPrimType.Box: scene.Objects.Add(new Box(new Vec3(p[0], p[1], p[2]), new Vec3(p[3], p[4], p[5]), MatFunc(q.Material), q.Specular, q.Reflectivity)); break;
                    case YPrimType.CylinderY: scene.Objects.Add(new CylinderY(new Vec3(p[0], p[1], p[2]), p[3], p[4], p[5], p[6] != 0.0f, Mat(q.Material))); break;
                    case YPrimType.Triangle: scene.Objects.Add(new Triangle(new Vec3(p[0], p[1], p[2]), new Vec3(p[3], p[4], p[5]), new Vec3(p[6], p[7], p[8]), Mat(q.Material))); break;
                    case YPrimType.Mesh:
                    {
                        var (rec, tris, tm) = meshRecords[q.Ref];
                        var list = new List<Triangle>(rec.NTriangles);
                        Vec3 mn = new Vec3(float.PositiveInfinity, float.PositiveInfinity, float.PositiveInfinity), mx = new Vec3(float.NegativeInfinity, float.NegativeInfinity, float.NegativeInfinity);
                        for (int t = 0; t < rec.NTriangles; t++)
                        {
                            int k = t * 9;
                            var A = new Vec3(tris[k], tris[k + 1], tris[k + 2]); var B = new Vec3(tris[k + 3], tris[k + 4], tris[k + 5]); var Cc = new Vec3(tris[k + 6], tris[k + 7], tris[k + 8]);
                            list.Add(new Triangle(A, B, Cc, Mat(tm != null ? tm[t] : rec.Material)));
                            foreach (Vec3 v in new[] { A, B, Cc })
                            {
                                mn = new Vec3(MathF.Min(mn.X, v.X), MathF.Min(mn.Y, v.Y), MathF.Min(mn.Z, v.Z));
                                mx = new Vec3(MathF.Max(mx.X, v.X), MathF.Max(mx.Y, v.Y), MathF.Max(mx.Z, v.Z));
                            }
                        }
                        var mesh = new Mesh(list, mn, mx);          // (BoundsMin / BoundsMax are stored, never read by the tracer: Mesh.cs:11-21, 35-38)
                        ls.Meshes.Add(mesh);
                        scene.Objects.Add(mesh);
                        break;
                    }
                    case YPrimType.VolumeGrid:
                    {
                        var (g, cells, lut) = gridRecords[q.Ref];
                        var arr = new (int, int)[g.Nx, g.Ny, g.Nz];
                        for (int ix = 0; ix < g.Nx; ix++) for (int iy = 0; iy < g.Ny; iy++) for (int iz = 0; iz < g.Nz; iz++)
                        {
                            long c = (((long)ix * g.Ny + iy) * g.Nz + iz) * 2;
                            arr[ix, iy, iz] = (cells[c], cells[c + 1]);
                        }
                        var table = new Dictionary<long, Material>();
                        foreach (YVoxelLookup e in lut) table[((long)e.MatId << 32) | (uint)e.MetaId] = Mat(e.Material);
                        int dflt = g.DefaultMaterial;
                        Func<int, int, Material> lookup = (id, meta) => table.TryGetValue(((long)id << 32) | (uint)meta, out Material mm) ? mm : dflt >= 0 ? Mat(dflt) : throw new KeyNotFoundException($"voxel ({id}, {meta}) has no material in the file");
                        scene.Objects.Add(new VolumeGrid(arr, V(g.MinCorner), V(g.VoxelSize), lookup, g.Wireframe != 0, g.WireWidthFraction, g.WireMaxDistance));
                        break;
                    }
                    default: throw new InvalidDataException("unknown primitive type " + q.Type);
                }
            }
            foreach (YLight l in lights) scene.Lights.Add(new PointLight(V(l.Position), V(l.Color), l.Intensity));
            scene.Ambient = new AmbientLight(V(ys.AmbientColor), ys.AmbientIntensity);
            scene.BackgroundTop = V(ys.BackgroundTop); scene.BackgroundBottom = V(ys.BackgroundBottom);
            scene.HasDynamicTextures = ys.HasDynamicTextures != 0;
            ls.Scene = scene;
            return ls;
        }

        /// <summary>A static Texture from its pixel ints.  The reference builds one only from an image file (Renderer/Texture.cs:25-49); the
        /// sampler reads `pixels`, `width`, `height` and `isDynamic == false` (:142-163), so those are set on an uninitialised object.</summary>
        private static Texture StaticTexture(int w, int h, int[] pixels)
        {
            var t = (Texture)RuntimeHelpers.GetUninitializedObject(typeof(Texture));
            t.width = w; t.height = h;
            typeof(Texture).GetField("pixels", BindingFlags.Instance | BindingFlags.NonPublic).SetValue(t, pixels);
            return t;
        }
    }
}
