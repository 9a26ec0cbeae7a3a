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
//
// Options (HipRaytraceOptions, all off by default = the reference's behaviour call for call):
//   FrameLate  - TryFlipAndBlit queues frame N with ycge_render_frame_async_sdr and blits frame N - 1, which it waited for first: the GPU's
//                3.6 ms (trace + TAA + the exact post stage at 1080p) run beside the host's Update / input / presenter instead of in front
//                of them.  One frame of latency; the frames themselves are the same, in the same order (tests/test_gpu_timed_variants.py).
//   Devices    - one process, several GPUs: the ONE call drives them all (config.n_devices / devices[]).
//   Exchange   - how their tiles come together on Devices[0]: YExchange.PeerPush (xGMI peer writes) or YExchange.Rccl (ONE ncclAllGather of
//                the tile slabs inside ycge_render_frame - SURVEY 8(e); the library dlopens librccl.so and falls back to PeerPush without it;
//                ExchangeInUse says which).
using System;
using System.Runtime.CompilerServices;
using ConsoleGame.RayTracing;
using ConsoleGame.RayTracing.Native;
using ConsoleGame.RayTracing.Objects;
using ConsoleGame.RayTracing.Scenes;
using ConsoleGame.Renderer;

public sealed class HipRaytraceOptions
{
    public bool FrameLate;
    public int[] Devices;
    public YExchange Exchange = YExchange.PeerPush;
}

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
        private float* sdrLate;                     // FrameLate: the second array - the frame in flight fills one while the host blits the other
        private readonly bool frameLate;
        private bool inFlight;                      // FrameLate: a frame queued by the last TryFlipAndBlit has not been waited for yet
        private FlatScene uploaded;                 // what the device holds (records only; its pins are released after the upload)
        private ulong objectsSignature;
        private bool forceUpload;
        private YLight[] lightsSent = Array.Empty<YLight>();
        private YVec3 ambientSent, topSent, bottomSent; private float ambientIntensitySent;

        public HipRaytraceWrapper(Framebuffer fb, Scene scene, float fovDeg, int superSample, HipRaytraceOptions options = null)
        {
            this.scene = scene ?? throw new ArgumentNullException(nameof(scene));
            var cfg = new YConfig();
            Ycge.Check(IntPtr.Zero, Ycge.ycge_config_default(ref cfg));
            cfg.FbWidth = fbW = fb.Width; cfg.FbHeight = fbH = fb.Height; cfg.SuperSample = ss = Math.Max(1, superSample); cfg.FovDeg = fov = fovDeg;
            if (options?.Devices != null && options.Devices.Length > 0)
            {
                if (options.Devices.Length > 8) throw new ArgumentException("at most 8 devices (YCGE_MAX_DEVICES)");
                cfg.NDevices = options.Devices.Length;
                for (int i = 0; i < options.Devices.Length; i++) cfg.Devices[i] = options.Devices[i];
                cfg.MultiDeviceExchange = (int)options.Exchange;
            }
            frameLate = options != null && options.FrameLate && cfg.NDevices <= 1;      // (frames in flight are the single-device form)
            Ycge.Check(IntPtr.Zero, Ycge.ycge_create(ref cfg, out ctx));
            AllocSdr();
            Upload();                               // the reference ctor ends with scene.RebuildBVH() (RaytraceRenderer.cs:107)
        }

        /// <summary>What the devices' tiles really travel by (the library falls back to the peer push where librccl.so is missing).</summary>
        public YExchange ExchangeInUse { get { Ycge.Check(ctx, Ycge.ycge_exchange_query(ctx, out int mode, out _)); return (YExchange)mode; } }

        private void AllocSdr()
        {
            if (sdr != null || sdrLate != null) { Ycge.ycge_wait(ctx); inFlight = false; }
            if (sdr != null) { Ycge.ycge_free_host_buffer((IntPtr)sdr); sdr = null; }
            if (sdrLate != null) { Ycge.ycge_free_host_buffer((IntPtr)sdrLate); sdrLate = null; }
            UIntPtr bytes = (UIntPtr)((ulong)fbW * (ulong)fbH * 6 * sizeof(float));
            Ycge.Check(ctx, Ycge.ycge_alloc_host_buffer(bytes, out IntPtr p));
            sdr = (float*)p;
            if (frameLate) { Ycge.Check(ctx, Ycge.ycge_alloc_host_buffer(bytes, out IntPtr q)); sdrLate = (float*)q; }
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
            if (frameLate)
            {
                // the frame queued by the LAST call is finished first (it ran beside the host's Update in between), then this call's frame is queued
                // into the other array and the finished one is blitted: one frame late, never torn (the device writes `sdrLate`, the host reads `sdr`)
                bool have = inFlight;
                if (inFlight) { Ycge.Check(ctx, Ycge.ycge_wait(ctx)); inFlight = false; float* t = sdr; sdr = sdrLate; sdrLate = t; }
                SyncScene();
                Ycge.Check(ctx, Ycge.ycge_render_frame_async_sdr(ctx, sdrLate));
                inFlight = true;
                if (!have) return;                                             // (the very first call has nothing to show yet: the framebuffer keeps what it had)
            }
            else
            {
                SyncScene();
                Ycge.Check(ctx, Ycge.ycge_render_frame(ctx, sdr, null));
            }
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
            if (sdrLate != null) { Ycge.ycge_free_host_buffer((IntPtr)sdrLate); sdrLate = null; }
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
