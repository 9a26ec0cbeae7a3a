// bindings/csharp/HipRaytraceWrapper.cs - the third IConsoleRenderer (next to RaytraceWrapper / VideoWrapper, RaytraceEntity.cs:20-50).
//
// RaytraceEntity is a `partial class` and its IConsoleRenderer is private: this file adds the wrapper as one more part of the class, so
// no existing file changes.  A construction site then reads
//     this.renderer = new HipRaytraceWrapper(fb, activeScene, activeScene.DefaultFovDeg, rtSuperSample);
// where it reads `new RaytraceWrapper(new RaytraceRenderer(fb, activeScene, fov, rtWidth, rtHeight, rtSuperSample))` today
// (RaytraceEntity.cs:97-98, 240-241, 262-263).
//
// One TryFlipAndBlit = one ycge_render_frame: ray generation, trace, TAA, denoise, exposure, tonemap and downsample on the GPU
// (RaytraceRenderer.cs:157-267), then the SetChexel loop of :260-261 on the host, unchanged.
using System;
using System.Runtime.CompilerServices;
using ConsoleGame.RayTracing;
using ConsoleGame.RayTracing.Native;
using ConsoleGame.RayTracing.Objects;
using ConsoleGame.RayTracing.Scenes;
using ConsoleGame.Renderer;

public partial class RaytraceEntity
{
    private sealed unsafe class HipRaytraceWrapper : IConsoleRenderer, IDisposable
    {
        private IntPtr ctx;
        private readonly Scene scene;
        private int fbW, fbH, ss;
        private float fov;
        private Vec3 pos; private float yaw, pitch;
        private float* sdr;                         // fbW * fbH * {top rgb, bottom rgb}: page-locked memory of the library (ycge_alloc_host_buffer)
        private FlatScene uploaded;                 // what the device holds (records only; its pins are released after the upload)
        private ulong objectsSignature;
        private bool forceUpload;
        private YLight[] lightsSent = Array.Empty<YLight>();
        private YVec3 ambientSent, topSent, bottomSent; private float ambientIntensitySent;

        public HipRaytraceWrapper(Framebuffer fb, Scene scene, float fovDeg, int superSample)
        {
            this.scene = scene ?? throw new ArgumentNullException(nameof(scene));
            var cfg = new YConfig();
            Ycge.Check(IntPtr.Zero, Ycge.ycge_config_default(ref cfg));
            cfg.FbWidth = fbW = fb.Width; cfg.FbHeight = fbH = fb.Height; cfg.SuperSample = ss = Math.Max(1, superSample); cfg.FovDeg = fov = fovDeg;
            Ycge.Check(IntPtr.Zero, Ycge.ycge_create(ref cfg, out ctx));
            AllocSdr();
            Upload();                               // the reference ctor ends with scene.RebuildBVH() (RaytraceRenderer.cs:107)
        }

        private void AllocSdr()
        {
            if (sdr != null) { Ycge.ycge_wait(ctx); Ycge.ycge_free_host_buffer((IntPtr)sdr); sdr = null; }
            Ycge.Check(ctx, Ycge.ycge_alloc_host_buffer((UIntPtr)((ulong)fbW * (ulong)fbH * 6 * sizeof(float)), out IntPtr p));
            sdr = (float*)p;
        }

        // ---- scene: full upload, and what changes between frames (Scene.Update: entities move objects, DayNightCycle.cs:80-89 moves lights and sky)
        private void Upload()
        {
            uploaded?.Dispose();
            uploaded = SceneFlattener.Flatten(scene);
            YScene s = uploaded.Scene;
            Ycge.Check(ctx, Ycge.ycge_scene_upload(ctx, ref s));
            uploaded.Dispose();                     // (frees pins and unmanaged copies; the record arrays and owner lists stay readable)
            objectsSignature = ObjectsSignature();
            RememberLights(uploaded.Lights);
        }

        private ulong ObjectsSignature()
        {
            // identity and bounds of every object, in order: an entity that moves, adds or removes geometry changes it (Scene.cs:122-127)
            ulong h = 1469598103934665603UL;
            void Mix(uint v) { h = (h ^ v) * 1099511628211UL; }
            foreach (Hittable o in scene.Objects)
            {
                Mix((uint)RuntimeHelpers.GetHashCode(o));
                if (o.TryGetBounds(out float a, out float b, out float c, out float d, out float e, out float f, out _, out _, out _))
                { Mix(BitConverter.SingleToUInt32Bits(a)); Mix(BitConverter.SingleToUInt32Bits(b)); Mix(BitConverter.SingleToUInt32Bits(c)); Mix(BitConverter.SingleToUInt32Bits(d)); Mix(BitConverter.SingleToUInt32Bits(e)); Mix(BitConverter.SingleToUInt32Bits(f)); }
            }
            return h;
        }

        private void RememberLights(YLight[] l)
        {
            lightsSent = l; ambientSent = new YVec3(scene.Ambient.Color); ambientIntensitySent = scene.Ambient.Intensity;
            topSent = new YVec3(scene.BackgroundTop); bottomSent = new YVec3(scene.BackgroundBottom);
        }
        private static bool Same(YVec3 a, YVec3 b) => a.X.Equals(b.X) && a.Y.Equals(b.Y) && a.Z.Equals(b.Z);
        private static bool Same(YLight[] a, YLight[] b)
        {
            if (a.Length != b.Length) return false;
            for (int i = 0; i < a.Length; i++) if (!Same(a[i].Position, b[i].Position) || !Same(a[i].Color, b[i].Color) || !a[i].Intensity.Equals(b[i].Intensity)) return false;
            return true;
        }

        private void SyncScene()
        {
            if (forceUpload || ObjectsSignature() != objectsSignature)
            {
                YPrim[] prims = forceUpload ? null : SceneFlattener.ObjectsAgainst(scene, uploaded);
                forceUpload = false;
                if (prims == null) Upload();        // a new mesh, grid or material (chunk streaming): the whole scene again
                else
                {
                    fixed (YPrim* p = prims) Ycge.Check(ctx, Ycge.ycge_scene_update_objects(ctx, p, prims.Length));     // only the scene-level BVH is rebuilt, as in the reference
                    uploaded.Prims = prims;
                    objectsSignature = ObjectsSignature();
                }
            }
            YLight[] lights = SceneFlattener.LightRecords(scene);
            YVec3 amb = new YVec3(scene.Ambient.Color), top = new YVec3(scene.BackgroundTop), bot = new YVec3(scene.BackgroundBottom);
            if (!Same(lights, lightsSent) || !Same(amb, ambientSent) || !Same(top, topSent) || !Same(bot, bottomSent) || !scene.Ambient.Intensity.Equals(ambientIntensitySent))
            {
                fixed (YLight* l = lights) Ycge.Check(ctx, Ycge.ycge_scene_update_lights(ctx, l, lights.Length, &amb, scene.Ambient.Intensity, &top, &bot));
                RememberLights(lights);
            }
            // live textures: the frame the reader shows NOW is the frame this TryFlipAndBlit samples (Renderer/Texture.cs:113-116)
            for (int i = 0; i < uploaded.Textures.Count; i++)
            {
                Texture t = uploaded.Textures[i];
                if (!SceneFlattenerAccess.IsDynamic(t)) continue;
                IntPtr frame = SceneFlattenerAccess.CurrentFrame(t, out int bpp);
                if (frame != IntPtr.Zero) Ycge.Check(ctx, Ycge.ycge_scene_update_texture(ctx, i, frame, (UIntPtr)((ulong)t.width * (ulong)t.height * (ulong)bpp)));
            }
        }

        /// <summary>Forces the next frame to upload the scene again (for a host that edits a scene in ways the signature cannot see).</summary>
        public void Invalidate() { forceUpload = true; }

        // ---- IConsoleRenderer
        public void SetCamera(Vec3 p, float y, float pt) { pos = p; yaw = y; pitch = pt; Push(); }
        public void SetFov(float f) { fov = f; Push(); }
        private void Push()
        {
            float* p = stackalloc float[3] { pos.X, pos.Y, pos.Z };
            Ycge.Check(ctx, Ycge.ycge_set_camera(ctx, p, yaw, pitch, fov));
        }

        public void Resize(Framebuffer fb, int superSample)
        {
            fbW = fb.Width; fbH = fb.Height; ss = Math.Max(1, superSample);
            Ycge.Check(ctx, Ycge.ycge_resize(ctx, fbW, fbH, ss));          // drops the TAA history (RaytraceRenderer.cs:137)
            AllocSdr();
        }

        public void TryFlipAndBlit(Framebuffer fb)
        {
            if (fb.Width != fbW || fb.Height != fbH) Resize(fb, ss);       // RaytraceRenderer.cs:119-120 does the same check
            SyncScene();
            Ycge.Check(ctx, Ycge.ycge_render_frame(ctx, sdr, null));
            for (int cy = 0; cy < fbH; cy++)
                for (int cx = 0; cx < fbW; cx++)
                {
                    float* c = sdr + ((long)cx + (long)cy * fbW) * 6;      // {top rgb, bottom rgb}
                    fb.SetChexel(cx, cy, new Chexel('▀', new Vec3(c[0], c[1], c[2]), new Vec3(c[3], c[4], c[5])));      // RaytraceRenderer.cs:260-261
                }
        }

        public void Dispose()
        {
            if (ctx != IntPtr.Zero) { Ycge.ycge_destroy(ctx); ctx = IntPtr.Zero; }      // (waits for everything in flight)
            if (sdr != null) { Ycge.ycge_free_host_buffer((IntPtr)sdr); sdr = null; }
            uploaded?.Dispose();
        }
    }
}

namespace ConsoleGame.RayTracing.Native
{
    /// <summary>The two questions the wrapper asks a Texture per frame (accessor first, private field of the unmodified reference otherwise).</summary>
    internal static class SceneFlattenerAccess
    {
        private const System.Reflection.BindingFlags Any = System.Reflection.BindingFlags.Instance | System.Reflection.BindingFlags.Public | System.Reflection.BindingFlags.NonPublic;
        private static object Get(object o, params string[] names)
        {
            foreach (string n in names)
            {
                var p = o.GetType().GetProperty(n, Any); if (p != null) return p.GetValue(o);
                var f = o.GetType().GetField(n, Any); if (f != null) return f.GetValue(o);
            }
            return null;
        }
        public static bool IsDynamic(ConsoleGame.Renderer.Texture t) => Get(t, "IsDynamic", "isDynamic") is bool b && b;
        public static IntPtr CurrentFrame(ConsoleGame.Renderer.Texture t, out int bytesPerPixel)
        {
            bytesPerPixel = Get(t, "DynamicBytesPerPixel", "dynamicBytesPerPixel") is int n ? n : 0;
            return Get(t, "DynamicReader", "dynamicReader") is NullEngine.Video.IFrameReader r ? r.GetCurrentFramePtr() : IntPtr.Zero;
        }
    }
}
