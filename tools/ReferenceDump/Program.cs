// tools/ReferenceDump/Program.cs - the REFERENCE's RaytraceRenderer, run on a scene file of this repository, its buffers written out raw.
//
//     dotnet run -c Release -p:ReferenceRoot=<checkout of NullandKale/YetAnotherConsoleGameEngine> -- <scene.ysc> <out_dir> [frames = 3]
//
// Per frame k = 1..frames (frameCounter starts at 0 and is incremented before use, RaytraceRenderer.cs:24, 175), little-endian, row-major
// x + y * hiW as Fast2D stores them (Fast2D.cs:21-24):
//     f<k>_rays.f32          6 per pixel: origin, direction          RaytraceRenderer.cs:45   (private Fast2D<Ray> rays)
//     f<k>_current_hdr.f32   3 per pixel                             :155  currentHdr
//     f<k>_g_albedo.f32, f<k>_g_normal.f32   3 per pixel, f<k>_g_depth.f32   1 per pixel      :46-48
//     f<k>_sky.u8            1 per pixel                             :54   skyMask
//     f<k>_taa_history.f32   3 per pixel                             :68   taaHistory (after TemporalBlendWithClamp)
//     f<k>_sdr.f32           6 per chexel: top rgb, bottom rgb       what fb.SetChexel received (:260-261; ChexelColor keeps the Vec3)
// Once: the scene BVH and every mesh BVH as the reference built them (private SoA arrays of Objects/BVH.cs:11-25, MeshBVH.cs:18-39), in the
// record layout of ycge_read_accel (10 x 4 bytes per node: min xyz, max xyz, left, right, start, count; then the leaf index array):
//     accel_scene_nodes.bin, accel_scene_leaf.i32, accel_mesh<i>_nodes.bin, accel_mesh<i>_leaf.i32
// and meta.json (sizes, pose, runtime description).  tests/test_reference_goldens.py of the repository reads exactly these names.
using System;
using System.Collections.Generic;
using System.IO;
using System.Reflection;
using System.Runtime.InteropServices;
using ConsoleGame.RayTracing;
using ConsoleGame.RayTracing.Objects;
using ConsoleGame.RayTracing.Scenes;
using ConsoleGame.Renderer;

namespace ReferenceDump
{
    internal static class Program
    {
        private const BindingFlags Priv = BindingFlags.Instance | BindingFlags.NonPublic | BindingFlags.Public;

        private static T Field<T>(object o, string name)
        {
            for (Type t = o.GetType(); t != null; t = t.BaseType)
            {
                FieldInfo f = t.GetField(name, Priv | BindingFlags.DeclaredOnly);
                if (f != null) return (T)f.GetValue(o);
            }
            throw new MissingFieldException(o.GetType().Name, name);
        }

        private static void WriteFloats(string path, ReadOnlySpan<float> v) => File.WriteAllBytes(path, MemoryMarshal.AsBytes(v).ToArray());
        private static void WriteVec3(string path, Vec3[] v)
        {
            var f = new float[v.Length * 3];
            for (int i = 0; i < v.Length; i++) { f[3 * i] = v[i].X; f[3 * i + 1] = v[i].Y; f[3 * i + 2] = v[i].Z; }
            WriteFloats(path, f);
        }

        private static void DumpAccel(string dir, string name, object bvh, string leafField)
        {
            float[] a = Field<float[]>(bvh, "nodeMinX"), b = Field<float[]>(bvh, "nodeMinY"), c = Field<float[]>(bvh, "nodeMinZ");
            float[] d = Field<float[]>(bvh, "nodeMaxX"), e = Field<float[]>(bvh, "nodeMaxY"), f = Field<float[]>(bvh, "nodeMaxZ");
            int[] l = Field<int[]>(bvh, "nodeLeft"), r = Field<int[]>(bvh, "nodeRight"), s = Field<int[]>(bvh, "nodeStart"), n = Field<int[]>(bvh, "nodeCount");
            int used = Field<int>(bvh, "nodeCountUsed");
            using (var w = new BinaryWriter(File.Create(Path.Combine(dir, name + "_nodes.bin"))))
                for (int i = 0; i < used; i++) { w.Write(a[i]); w.Write(b[i]); w.Write(c[i]); w.Write(d[i]); w.Write(e[i]); w.Write(f[i]); w.Write(l[i]); w.Write(r[i]); w.Write(s[i]); w.Write(n[i]); }
            int[] leaf = Field<int[]>(bvh, leafField);
            File.WriteAllBytes(Path.Combine(dir, name + "_leaf.i32"), MemoryMarshal.AsBytes<int>(leaf).ToArray());
        }

        public static int Main(string[] args)
        {
            if (args.Length < 2) { Console.Error.WriteLine("usage: ReferenceDump <scene.ysc> <out_dir> [frames = 3]"); return 2; }
            string outDir = args[1];
            int frames = args.Length > 2 ? int.Parse(args[2]) : 3;
            Directory.CreateDirectory(outDir);
            LoadedScene ls = SceneFile.Load(args[0]);
            var fb = new Framebuffer(ls.FbWidth, ls.FbHeight, 0, 0);
            // (pxW, pxH are stored and never read: RaytraceRenderer.cs:74-87 derives the trace grid from the framebuffer and ss)
            var rt = new RaytraceRenderer(fb, ls.Scene, ls.FovDeg, ls.FbWidth * ls.SuperSample, ls.FbHeight * ls.SuperSample, ls.SuperSample);
            rt.SetFov(ls.FovDeg);
            rt.SetCamera(ls.Pos, ls.Yaw, ls.Pitch);

            DumpAccel(outDir, "accel_scene", Field<object>(ls.Scene, "bvh"), "leafObjIndex");
            for (int i = 0; i < ls.Meshes.Count; i++) DumpAccel(outDir, "accel_mesh" + i, Field<object>(ls.Meshes[i], "bvh"), "leafTriIndex");

            int hiW = ls.FbWidth * ls.SuperSample, hiH = ls.FbHeight * 2 * ls.SuperSample;
            for (int k = 1; k <= frames; k++)
            {
                rt.TryFlipAndBlit(fb);
                string p = Path.Combine(outDir, "f" + k + "_");
                Ray[] rays = Field<Fast2D<Ray>>(rt, "rays").Buffer;
                var rf = new float[rays.Length * 6];
                for (int i = 0; i < rays.Length; i++) { rf[6 * i] = rays[i].Origin.X; rf[6 * i + 1] = rays[i].Origin.Y; rf[6 * i + 2] = rays[i].Origin.Z; rf[6 * i + 3] = rays[i].Dir.X; rf[6 * i + 4] = rays[i].Dir.Y; rf[6 * i + 5] = rays[i].Dir.Z; }
                WriteFloats(p + "rays.f32", rf);
                WriteVec3(p + "current_hdr.f32", Field<Fast2D<Vec3>>(rt, "currentHdr").Buffer);
                WriteVec3(p + "g_albedo.f32", Field<Fast2D<Vec3>>(rt, "gAlbedo").Buffer);
                WriteVec3(p + "g_normal.f32", Field<Fast2D<Vec3>>(rt, "gNormal").Buffer);
                WriteFloats(p + "g_depth.f32", Field<Fast2D<float>>(rt, "gDepth").Buffer);
                bool[] sky = Field<Fast2D<bool>>(rt, "skyMask").Buffer;
                var sb = new byte[sky.Length]; for (int i = 0; i < sky.Length; i++) sb[i] = sky[i] ? (byte)1 : (byte)0;
                File.WriteAllBytes(p + "sky.u8", sb);
                WriteVec3(p + "taa_history.f32", Field<Fast2D<Vec3>>(rt, "taaHistory").Buffer);
                var sdr = new float[ls.FbWidth * ls.FbHeight * 6];
                for (int cy = 0; cy < ls.FbHeight; cy++)
                    for (int cx = 0; cx < ls.FbWidth; cx++)
                    {
                        Chexel c = fb.GetChexel(cx, cy);
                        int i = (cx + cy * ls.FbWidth) * 6;
                        sdr[i] = c.ForegroundColor.color_f32.X; sdr[i + 1] = c.ForegroundColor.color_f32.Y; sdr[i + 2] = c.ForegroundColor.color_f32.Z;
                        sdr[i + 3] = c.BackgroundColor.color_f32.X; sdr[i + 4] = c.BackgroundColor.color_f32.Y; sdr[i + 5] = c.BackgroundColor.color_f32.Z;
                    }
                WriteFloats(p + "sdr.f32", sdr);
                Console.WriteLine("frame " + k + " dumped");
            }
            File.WriteAllText(Path.Combine(outDir, "meta.json"),
                "{\"fb_width\": " + ls.FbWidth + ", \"fb_height\": " + ls.FbHeight + ", \"super_sample\": " + ls.SuperSample + ", \"hi_w\": " + hiW + ", \"hi_h\": " + hiH +
                ", \"frames\": " + frames + ", \"n_meshes\": " + ls.Meshes.Count + ", \"scene_file\": \"" + Path.GetFileName(args[0]) + "\"" +
                ", \"runtime\": \"" + RuntimeInformation.FrameworkDescription + " on " + RuntimeInformation.OSDescription.Replace("\"", "'") + " " + RuntimeInformation.ProcessArchitecture + "\"}\n");
            return 0;
        }
    }
}
