"""Host-side mirror of the reference's render driver for the hot path.

`PathTracer` reproduces the caller contract of Assets/Scripts/PathTracer.cs (the C# MonoBehaviour cannot run
here): the inspector fields (PathTracer.cs:24-50), one `OnRenderImage` = one pass with the uniform block of
PathTracer.cs:230-249, the ping-pong / `_currentSample` bookkeeping of :268-272 and `Reset()` (:318-322).
`BVHScene` mirrors what BVHScene.cs does with the native plugin: BuildBVH -> sizes -> borrowed pointers ->
upload (BVHScene.cs:629-667), then binds the buffers (BVHScene.cs:140-167) through PTSetScene.

All compute goes through the C-ABI of libunity-webgpu-pathtracer-plugin.so (include/ptmi_plugin.h); nothing
here renders, traverses or shades on the host.
"""
import ctypes as C

import numpy as np

from . import abi, plugin
from .scenes import Scene, frame_params


class BVHScene:
    """Scene buffers as BVHScene.cs holds them, uploaded into HBM by PTSetScene."""

    def __init__(self, scene: Scene, build_device: int = None):
        """build_device = None: BuildBVH (the reference's CPU builder, byte-identical); k: PTBuildBVHDevice on HIP device k."""
        self.scene = scene
        self.build_device = build_device
        self.build_ms = {}
        self.tlas_data = None
        self.tlas_index_offset = 0
        self.gpu_instances = None
        if scene.use_tlas:
            self._build_two_level(scene)
        else:
            # BVHScene.cs:629-659: BuildBVH over the world-space triangle soup, copy node/triangle bytes out
            self.bvh_nodes, self.bvh_tris = plugin.build_cwbvh(scene.vertices, device=build_device, timing=self.build_ms)
        self.tri_attrs = np.ascontiguousarray(scene.tri_attrs)
        self.materials = np.ascontiguousarray(scene.materials, dtype=np.float32)
        self.lights = np.ascontiguousarray(scene.lights, dtype=np.float32)
        self.texture_data = np.ascontiguousarray(scene.texture_data, dtype=np.uint32)

    def _build_two_level(self, scene: Scene):
        """BVHScene.cs:600-758 with _useTLAS: one BLAS per unique mesh (local space), node / triangle buffers back to
        back, one GPUInstance + BLASInstance per renderer, BuildTLAS over the renderers' world bounds, TLASData =
        TLAS nodes followed by the instance indices at TLASIndexOffset."""
        from .scenes import instance_world_bounds
        nodes, tris, node_off, tri_off = [], [], [], []
        n_off = t_off = 0
        for t0, n in scene.mesh_ranges:
            nb, tb = plugin.build_cwbvh(scene.vertices[t0 * 3:(t0 + n) * 3], device=self.build_device)
            nodes.append(nb)
            tris.append(tb)
            node_off.append(n_off)
            tri_off.append(t_off)
            n_off += nb.nbytes
            t_off += tb.nbytes
        self.bvh_nodes = np.concatenate(nodes)
        self.bvh_tris = np.concatenate(tris)
        gi = np.zeros(len(scene.instances), dtype=abi.GPU_INSTANCE)
        bi = np.zeros(len(scene.instances), dtype=abi.BLAS_INSTANCE)
        for k, (mesh, l2w, material) in enumerate(scene.instances):
            w2l = np.linalg.inv(l2w)
            gi[k]["localToWorld"] = l2w.T.reshape(16).astype(np.float32)       # Matrix4x4 memory order: (r, c) at c*4 + r
            gi[k]["worldToLocal"] = w2l.T.reshape(16).astype(np.float32)
            gi[k]["bvhOffset"] = node_off[mesh] // 80                            # kBVHNodeSize
            gi[k]["triOffset"] = tri_off[mesh] // 16                             # kBVHTriSize: float4 units
            gi[k]["triAttributeOffset"] = scene.mesh_ranges[mesh][0]             # in triangles
            gi[k]["materialIndex"] = material
            t0, n = scene.mesh_ranges[mesh]
            lo, hi = instance_world_bounds(scene.vertices[t0 * 3:(t0 + n) * 3], l2w)
            bi[k]["localToWorld"], bi[k]["worldToLocal"] = gi[k]["localToWorld"], gi[k]["worldToLocal"]
            bi[k]["aabbMin"], bi[k]["aabbMax"], bi[k]["blasIndex"] = lo, hi, k
        tlas_nodes, tlas_idx = plugin.build_tlas(bi)
        self.tlas_index_offset = tlas_nodes.nbytes // 4
        self.tlas_data = np.concatenate([tlas_nodes.view(np.float32), tlas_idx.view(np.float32)])
        self.gpu_instances = gi
        self.blas_instances = bi

    def desc(self) -> abi.PTSceneDesc:
        d = abi.PTSceneDesc()
        d.bvhNodes = self.bvh_nodes.ctypes.data
        d.bvhNodesBytes = self.bvh_nodes.nbytes
        d.bvhTris = self.bvh_tris.ctypes.data
        d.bvhTrisBytes = self.bvh_tris.nbytes
        d.triAttrs = self.tri_attrs.ctypes.data
        d.triAttrsBytes = self.tri_attrs.nbytes
        d.materials = self.materials.ctypes.data
        d.materialCount = self.materials.shape[0]
        d.lights = self.lights.ctypes.data if self.lights.size else None
        d.lightCount = self.lights.shape[0]
        d.textureData = self.texture_data.ctypes.data if self.texture_data.size else None
        d.textureDataUints = self.texture_data.size
        d.features = self.scene.features
        if self.tlas_data is not None:
            d.tlasData = self.tlas_data.ctypes.data
            d.tlasDataFloats = self.tlas_data.size
            d.tlasIndexOffset = self.tlas_index_offset
            d.instanceCount = self.gpu_instances.shape[0]
            d.gpuInstances = self.gpu_instances.ctypes.data
        if self.scene.environment_texture is not None:
            self.env_texture = np.ascontiguousarray(self.scene.environment_texture, dtype=np.float32)
            d.envTexture = self.env_texture.ctypes.data
            d.envHeight, d.envWidth = self.env_texture.shape[:2]
        return d

    def PrepareShader(self, ctx):
        """BVHScene.PrepareShader (BVHScene.cs:140-167): bind the buffers to the kernel."""
        plugin.check(plugin.load_library().PTSetScene(ctx, C.byref(self.desc())))


class PathTracer:
    """One render context on one GPU, driven exactly as PathTracer.cs drives the compute shader."""

    def __init__(self, scene: Scene, device: int = 0, width: int = 256, height: int = 256,
                 samplesPerPass: int = 1, maxSamples: int = 100000, maxRayBounces: int = 4,
                 useRussianRoulette: bool = True, fireflyFilter: bool = False, maxFireflyLuminance: float = 10.0,
                 rank: int = 0, world_size: int = 1, reference_dispatch: bool = False, schedule: int = None,
                 build_device: int = None):
        self.lib = plugin.load_library()
        self.scene = scene
        self.width, self.height = width, height
        # inspector fields (PathTracer.cs:24-50)
        self.samplesPerPass = samplesPerPass
        self.maxSamples = maxSamples
        self.maxRayBounces = maxRayBounces
        self.useRussianRoulette = useRussianRoulette
        self.fireflyFilter = fireflyFilter
        self.maxFireflyLuminance = maxFireflyLuminance
        self.reference_dispatch = reference_dispatch
        # tonemapping settings (PathTracer.cs:41-48)
        self.tonemapMode = abi.TONEMAP_LOTTES
        self.sRGB = False
        self.exposure, self.brightness, self.contrast, self.saturation, self.vignette = 1.0, 1.0, 1.0, 1.0, 0.0
        self._currentSample = 0
        ctx = C.c_void_p()
        plugin.check(self.lib.PTCreate(device, C.byref(ctx)))
        self.ctx = ctx
        self._bvhScene = BVHScene(scene, build_device=build_device)
        self._bvhScene.PrepareShader(self.ctx)
        if world_size > 1:
            plugin.check(self.lib.PTSetTileOwnership(self.ctx, rank, world_size))
        self.rank, self.world_size = rank, world_size
        if schedule is not None:
            self.set_schedule(schedule)

    # ---- PathTracer.cs:318-322
    def Reset(self):
        self._currentSample = 0
        plugin.check(self.lib.PTResetFrames(self.ctx))

    def params(self, seed: int) -> abi.PTFrameParams:
        groups = (0, 0)
        if self.reference_dispatch:
            # Mathf.CeilToInt(_outputWidth / dx) with integer operands (PathTracer.cs:207-208): floor division
            groups = (self.width // 8, self.height // 8)
        return frame_params(self.scene, self.width, self.height, spp=self.samplesPerPass,
                            current_sample=self._currentSample, seed=seed, max_bounces=self.maxRayBounces,
                            russian_roulette=self.useRussianRoulette, firefly=self.fireflyFilter,
                            max_firefly_luminance=self.maxFireflyLuminance, dispatch_groups=groups)

    # ---- PathTracer.cs:188-280 (the render half; the tonemap blit is out of scope)
    def OnRenderImage(self, seed: int):
        """One progressive pass.  `seed` is RngSeedRoot (the C# host draws a fresh random value per frame, :233)."""
        if self._currentSample < self.maxSamples:
            p = self.params(seed)
            plugin.check(self.lib.PTRenderPass(self.ctx, C.byref(p)))
        if self._currentSample < self.maxSamples:
            self._currentSample += max(1, self.samplesPerPass)
        if self._currentSample < self.maxSamples:
            plugin.check(self.lib.PTFlipFrames(self.ctx))
            self._flipped = True
        else:
            self._flipped = False

    def render_pass(self, p: abi.PTFrameParams):
        """Raw access: one pass with explicit params into the internal ping-pong frames (no bookkeeping)."""
        plugin.check(self.lib.PTRenderPass(self.ctx, C.byref(p)))

    def render_pass_to(self, p: abi.PTFrameParams, d_output: int, d_accumulated: int = 0):
        plugin.check(self.lib.PTRenderPassTo(self.ctx, C.byref(p), C.c_void_p(d_output), C.c_void_p(d_accumulated or None)))

    def render_batch_to(self, params, d_output: int, d_accumulated: int = 0):
        """PTRenderPassBatchTo: `params` = list of PTFrameParams (1..8 passes differing in RngSeedRoot / CurrentSample only)."""
        arr = (abi.PTFrameParams * len(params))(*params)
        plugin.check(self.lib.PTRenderPassBatchTo(self.ctx, arr, len(params), C.c_void_p(d_output), C.c_void_p(d_accumulated or None)))

    def flip(self):
        plugin.check(self.lib.PTFlipFrames(self.ctx))

    def synchronize(self):
        plugin.check(self.lib.PTSynchronize(self.ctx))

    def readback(self, last_output: bool = True) -> np.ndarray:
        """Current Output frame as (H, W, 4) float32.  After OnRenderImage flipped the targets the most recent
        output is frame[1-cur]; `last_output` accounts for that."""
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        flipped = last_output and getattr(self, "_flipped", False)
        if flipped:
            plugin.check(self.lib.PTFlipFrames(self.ctx))
        try:
            plugin.check(self.lib.PTReadback(self.ctx, out.ctypes.data_as(C.c_void_p), out.size))
        finally:
            if flipped:
                plugin.check(self.lib.PTFlipFrames(self.ctx))
        return out

    def frame_pointer(self, which: int = -1) -> int:
        return self.lib.PTGetFramePointer(self.ctx, which)

    def stream(self) -> int:
        return self.lib.PTGetStream(self.ctx)

    def set_stats_level(self, level: int):
        plugin.check(self.lib.PTSetStatsLevel(self.ctx, level))

    def stats(self) -> abi.PTStats:
        st = abi.PTStats()
        plugin.check(self.lib.PTGetStats(self.ctx, C.byref(st)))
        return st

    def reset_stats(self):
        plugin.check(self.lib.PTResetStats(self.ctx))

    def set_profiling(self, on: bool):
        plugin.check(self.lib.PTSetProfiling(self.ctx, 1 if on else 0))

    def timings(self) -> abi.PTTimings:
        t = abi.PTTimings()
        plugin.check(self.lib.PTGetTimings(self.ctx, C.byref(t)))
        return t

    def reset_timings(self):
        plugin.check(self.lib.PTResetTimings(self.ctx))

    # ---- PathTracer.cs:255-266: the presentation blit of _outputRT[_currentRT]
    def present_params(self) -> abi.PTPresentParams:
        q = abi.PTPresentParams()
        q.OutputWidth, q.OutputHeight = self.width, self.height
        q.Mode, q.sRGB = int(self.tonemapMode), 1 if self.sRGB else 0
        q.Exposure, q.Brightness, q.Contrast, q.Saturation, q.Vignette = (self.exposure, self.brightness, self.contrast,
                                                                          self.saturation, self.vignette)
        return q

    def present(self, q: abi.PTPresentParams = None) -> np.ndarray:
        """The displayable image of the current Output frame, (H, W, 4) float32."""
        q = q or self.present_params()
        out = np.empty((q.OutputHeight, q.OutputWidth, 4), dtype=np.float32)
        plugin.check(self.lib.PTPresentToHost(self.ctx, C.byref(q), out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def present_to(self, q: abi.PTPresentParams, d_src: int, d_dst: int):
        plugin.check(self.lib.PTPresent(self.ctx, C.byref(q), C.c_void_p(d_src) if d_src else None, C.c_void_p(d_dst)))

    def set_schedule(self, schedule: int):
        plugin.check(self.lib.PTSetSchedule(self.ctx, schedule))

    def schedule(self) -> int:
        """The schedule the next pass runs with (auto resolved against the current scene)."""
        rc = self.lib.PTGetSchedule(self.ctx)
        if rc < 0:
            plugin.check(rc)
        return rc

    def set_wavefront_iterations(self, n: int):
        plugin.check(self.lib.PTSetWavefrontIterations(self.ctx, n))

    def set_passes_in_flight(self, n: int):
        """0 = default for the hardware queues the process asked for (GPU_MAX_HW_QUEUES); 1 = passes back to back."""
        plugin.check(self.lib.PTSetPassesInFlight(self.ctx, n))

    def set_sub_frames(self, n: int):
        """PTSetSubFrames: cut every pass into n interleaved sub-frames with their own launch sequences (1 = off)."""
        plugin.check(self.lib.PTSetSubFrames(self.ctx, n))

    def passes_in_flight(self) -> int:
        return self.lib.PTGetPassesInFlight(self.ctx)

    def close(self):
        if self.ctx:
            self.lib.PTDestroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
