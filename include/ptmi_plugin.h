/*
 * ptmi_plugin.h — C-ABI of libunity-webgpu-pathtracer-plugin.so, the MI355X drop-in.
 *
 * Part 1 re-exports, symbol for symbol, the native plugin the Unity C# host binds with
 *   [DllImport("unity-webgpu-pathtracer-plugin")]  (Assets/Scripts/util/TinyBVH.cs:15-50)
 * and that the reference builds from Assets/Plugins/Web/plugin.cpp (Plugin/CMakeLists.txt:7-11).
 *
 * Part 2 adds the render entry points.  The reference has no render FFI: it renders
 * through Unity's ComputeShader API (Assets/Scripts/PathTracer.cs:226-252).  The PT*
 * functions below take exactly what those calls bind: the buffers of
 * BVHScene.PrepareShader (Assets/Scripts/util/BVHScene.cs:140-167) and the uniforms of
 * PathTracer.OnRenderImage (PathTracer.cs:230-249), and replace DispatchCompute (:251).
 *
 * Conventions: cdecl, blittable arguments only (pointers, ints, floats, POD structs);
 * inputs are borrowed for the duration of the call; outputs returned by Get*Data are
 * owned by the library until the matching Destroy*.  Functions that returned C++ `bool`
 * in the reference return a full-width int 0/1 here (ABI-compatible with both the 1-byte
 * and the 4-byte marshalling of C# `bool`).  Nothing in this library ever calls exit():
 * degenerate input yields a negative handle / error code and PTGetLastError() text.
 * Like the reference (unsynchronised globals, plugin.cpp:5-6) the handle tables are
 * single-threaded by contract.
 */
#ifndef PTMI_PLUGIN_H
#define PTMI_PLUGIN_H

#include <stdint.h>
#include "ptmi_layouts.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define PT_API __declspec(dllexport)
#else
#define PT_API __attribute__((visibility("default")))
#endif

/* ------------------------------------------------------------------------------------
 * Part 1 — acceleration-structure exports (replaces Assets/Plugins/Web/plugin.h:14-29)
 * ------------------------------------------------------------------------------------ */

/* plugin.cpp:35-40.  Synchronous CWBVH build over 3*triangleCount 16-byte vertices
 * (w ignored).  Returns the handle (first free slot), or -1 on triangleCount <= 0 / NULL. */
PT_API int   BuildBVH(const PTFloat4* vertices, int triangleCount);
/* plugin.cpp:42-52.  Out-of-range or already-destroyed handle is a no-op. */
PT_API void  DestroyBVH(int index);
/* plugin.cpp:54-58 */
PT_API int   IsBVHReady(int index);
/* plugin.cpp:23-33.  Opaque object pointer (never dereferenced by the C# host). */
PT_API void* GetBVHPtr(int index);
PT_API void* GetBVH(int index);
/* plugin.cpp:60-64.  BYTES of CWBVH node data (nodes * 80); 0 for an invalid handle. */
PT_API int   GetCWBVHNodesSize(int index);
/* plugin.cpp:66-70.  BYTES of CWBVH triangle data (triangles * 48). */
PT_API int   GetCWBVHTrisSize(int index);
/* plugin.cpp:72-86.  Borrowed pointers, valid until DestroyBVH(index). */
PT_API int   GetCWBVHData(int index, PTFloat4** bvhNodes, PTFloat4** bvhTris);

/* No reference counterpart: the same CWBVH format built ON the MI355X -- LBVH (63-bit Morton keys, radix
 * sort, Karras radix tree), greedy surface-area collapse to 8-wide, CWBVH encode, all in HIP kernels.  The tree differs from
 * BuildBVH's binned-SAH tree (so the bytes differ), every ray finds the same closest hit.  The handle lives in BuildBVH's
 * table: GetCWBVHNodesSize / GetCWBVHTrisSize / GetCWBVHData / IsBVHReady / DestroyBVH apply.  Returns -1 on degenerate input
 * or when deviceIndex is not a HIP device (PTGetBVHBuildError() has the text); there is no CPU fallback behind this entry. */
PT_API int   PTBuildBVHDevice(int deviceIndex, const PTFloat4* vertices, int triangleCount);
PT_API const char* PTGetBVHBuildError(void);
/* Build time of a handle in milliseconds: host wall time of BuildBVH, device time (kernels only) of PTBuildBVHDevice. */
PT_API double PTGetBVHBuildMs(int index);

/* plugin.cpp:111-118.  2-wide SAH BVH over the instances' world AABBs, Aila-Laine layout. */
PT_API int   BuildTLAS(const PTBlasInstance* instances, int instanceCount);
PT_API void  DestroyTLAS(int index);                                     /* plugin.cpp:120-130 */
PT_API int   IsTLASReady(int index);                                     /* plugin.cpp:132-136 */
PT_API int   GetTLASNodesSize(int index);                                /* plugin.cpp:138-142, bytes = nodes*64 */
PT_API int   GetTLASData(int index, PTFloat4** tlasNodes, uint32_t** tlasIndices); /* plugin.cpp:144-158 */

/* ------------------------------------------------------------------------------------
 * Part 2 — render entry points (replace the ComputeShader dispatch of PathTracer.cs:226-252)
 * ------------------------------------------------------------------------------------ */

typedef struct PTContext PTContext;   /* opaque; one per GPU (one process per GPU) */

/* Feature bits = the reference's shader keywords (PathTracer.compute:6-9). */
#define PT_FEATURE_HAS_LIGHTS    0x1u   /* HAS_LIGHTS   (PathTracer.cs:372,469) */
#define PT_FEATURE_HAS_TEXTURES  0x2u   /* HAS_TEXTURES (PathTracer.cs:185)     */
#define PT_FEATURE_HAS_TLAS      0x4u   /* HAS_TLAS     (BVHScene.cs:145-149): two-level traversal, util/tlas.hlsl       */
#define PT_FEATURE_HAS_ENVIRONMENT_TEXTURE 0x8u   /* HAS_ENVIRONMENT_TEXTURE (PathTracer.cs:119-143): util/sky.hlsl:7-88 */

/* Versioning of the two input structs: the FIRST member of PTSceneDesc and PTFrameParams is structSize = sizeof(the struct
 * the host was compiled against).  New members are only ever appended; the library reads min(structSize, its own sizeof)
 * bytes and treats the rest as zero, and rejects a structSize smaller than the first published layout (PT_ERR_INVALID_ARG:
 * almost always a host that forgot to set it).  So a host built against an older header keeps working unchanged. */

/* The buffers BVHScene.PrepareShader binds (BVHScene.cs:151-166) + Lights (PathTracer.cs:471).
 * All pointers are HOST pointers; PTSetScene copies them into HBM and VALIDATES every index the kernels will follow
 * (material indices, texture descriptors, CWBVH child / triangle ranges, primitive indices, TLAS nodes and instance offsets):
 * a scene that would make a kernel read out of bounds is refused with PT_ERR_INVALID_ARG instead of faulting the GPU. */
typedef struct PTSceneDesc {
    uint32_t        structSize;      uint32_t _pad3;             /* = sizeof(PTSceneDesc) of the host's header */
    const void*     bvhNodes;        uint64_t bvhNodesBytes;     /* PTCwbvhNode[]          "BVHNodes"; below 4 GiB (53 M nodes): the kernels address nodes with 32-bit byte offsets */
    const void*     bvhTris;         uint64_t bvhTrisBytes;      /* PTCwbvhTri[]           "BVHTris"  */
    const void*     triAttrs;        uint64_t triAttrsBytes;     /* PTTriangleAttributes[] "TriangleAttributesBuffer" */
    const void*     materials;       uint32_t materialCount;  uint32_t _pad0;  /* PTMaterialData[] "Materials" */
    const void*     lights;          uint32_t lightCount;     uint32_t _pad1;  /* PTLight[] "Lights" + LightCount */
    const uint32_t* textureData;     uint64_t textureDataUints;  /* "TextureData": K x {w,h,offset,0} then texels */
    uint32_t        features;        uint32_t _pad2;             /* PT_FEATURE_* */
    /* HAS_TLAS only (BVHScene.cs:155-160, 700-758).  bvhNodes / bvhTris / triAttrs then hold the per-mesh BLASes back to back
     * and each PTGpuInstance carries its offsets into them. */
    const float*    tlasData;        uint64_t tlasDataFloats;    /* "TLASData": PTTlasNode[] (16 floats each) then the instance indices */
    uint32_t        tlasIndexOffset; uint32_t instanceCount;     /* "TLASIndexOffset" (in floats); number of PTGpuInstance           */
    const void*     gpuInstances;                                /* PTGpuInstance[]        "BLASInstances"                            */
    /* HAS_ENVIRONMENT_TEXTURE only: the RGBAFloat copy of PathTracer.environmentTexture (PathTracer.cs:125-137), envWidth x
     * envHeight float4, in the order of the readback array `data[i]` the reference builds its CDF from (PathTracer.cs:297-306).
     * PTSetScene builds "EnvironmentCDF" / EnvironmentCdfSum from it exactly as OnEnvTexReadback does (sequential fp32 sum of
     * Color.grayscale).  SampleLevel(uv) is restated as bilinear with clamp addressing; memory row r is the texel row at
     * v = 1 - (r + 0.5) / envHeight, which makes SampleEnvMap's `uv.y = 1 - uv.y` (sky.hlsl:71) address the row whose CDF entry
     * it picked.  (Sampler state and row order belong to Unity's platform layer: parity unpinned there, see DESIGN.md.) */
    const float*    envTexture;      uint32_t envWidth;  uint32_t envHeight;
} PTSceneDesc;
#define PT_SCENE_DESC_MIN_SIZE  ((uint32_t)(uintptr_t)&((PTSceneDesc*)0)->envTexture)      /* layout before the environment texture */

/* The uniform block PathTracer.OnRenderImage sets every frame (PathTracer.cs:230-249;
 * declarations util/globals.hlsl:7-17, util/camera.hlsl:7-10, PathTracer.compute:40-41).
 * Matrices are 16 floats in Unity Matrix4x4 memory order: element (row r, col c) at [c*4+r],
 * applied to column vectors (mul(M, v)). */
typedef struct PTFrameParams {
    uint32_t structSize;              /* = sizeof(PTFrameParams) of the host's header (see "Versioning" above) */
    uint32_t _pad0;
    float    CamInvProj[16];          /* _camera.projectionMatrix.inverse  (PathTracer.cs:230) */
    float    CamToWorld[16];          /* _camera.cameraToWorldMatrix       (PathTracer.cs:231) */
    uint32_t RngSeedRoot;             /* fresh random value per frame      (PathTracer.cs:233) */
    uint32_t MaxRayBounces;           /* host passes max(.,1)              (PathTracer.cs:234) */
    int32_t  SamplesPerPass;          /* host passes max(1,.)              (PathTracer.cs:235) */
    uint32_t OutputWidth;
    uint32_t OutputHeight;
    uint32_t CurrentSample;           /* samples already accumulated       (PathTracer.cs:238) */
    int32_t  EnvironmentMode;         /* 0 environment colour, 1 basic sky (PathTracer.cs:239) */
    float    EnvironmentIntensity;
    float    EnvironmentColor[4];
    float    EnvironmentMapRotation;  /* only read by HAS_ENVIRONMENT_TEXTURE (PathTracer.cs:243)  */
    float    FocalLength;
    float    Aperture;
    int32_t  UseFireflyFilter;
    float    MaxFireflyLuminance;
    int32_t  UseRussianRoulette;
    /* DispatchCompute(kernel, dispatchX, dispatchY, 1) group counts of 8x8 threads
     * (PathTracer.cs:203-208,251).  0,0 = cover every pixel (ceil).  The C# host uses integer
     * division there, so for sizes that are not multiples of 8 it leaves edge pixels untouched;
     * pass its values to reproduce that exactly. */
    uint32_t DispatchGroupsX;
    uint32_t DispatchGroupsY;
} PTFrameParams;
#define PT_FRAME_PARAMS_MIN_SIZE ((uint32_t)(uintptr_t)&((PTFrameParams*)0)->DispatchGroupsX) /* DispatchGroups default to 0, 0 */

/* Work counters, accumulated over passes since the last PTResetStats.  One "ray" is one
 * call of RayIntersectBvh (util/bvh.hlsl:126): SURVEY.md §8(d). */
typedef struct PTStats {
    uint64_t paths;                /* samples started (pixels * SamplesPerPass)            */
    uint64_t closestHitRays;       /* RayIntersect calls      (util/bvh.hlsl:217)          */
    uint64_t shadowRays;           /* ShadowRayIntersect calls (util/bvh.hlsl:228)         */
    uint64_t nodeVisits;           /* CWBVH nodes fetched + tested (80 B each)             */
    uint64_t triTests;             /* IntersectTriangle calls (48 B each)                  */
    uint64_t attrFetches;          /* TriangleAttributes fetched (128 B each)              */
    uint64_t materialFetches;      /* MaterialData fetched (128 B each)                    */
    uint64_t lightFetches;         /* Light records read (64 B each)                       */
    uint64_t texelFetches;         /* texels read (4 B each)                               */
    uint64_t texDescriptorFetches; /* texture descriptors read (16 B each)                 */
    uint64_t pixelsWritten;        /* Output texels written (16 B each)                    */
    uint64_t pixelsRead;           /* AccumulatedOutput texels read (16 B each)            */
    uint64_t maxStackDepth;        /* deepest traversal stack seen (reference limit is 32) */
    uint64_t stackOverflows;       /* rays that would have exceeded BVH_STACK_SIZE         */
    uint64_t tlasNodeVisits;       /* HAS_TLAS: TLAS nodes read (64 B each)                */
    uint64_t instanceVisits;       /* HAS_TLAS: BLAS instances entered (144 B each)        */
} PTStats;

/* Device-side timing of the render kernels, measured with HIP events recorded on the
 * context's own stream around each launch (enabled by PTSetProfiling). */
typedef struct PTTimings {
    uint64_t passes;               /* passes timed since the last reset          */
    double   kernelMsTotal;        /* sum of event-to-event kernel time, ms      */
    double   kernelMsLast;         /* the most recent pass                        */
    uint64_t kernelLaunches;       /* device kernels launched in those passes     */
} PTTimings;

/* Error codes (negative int returns). */
#define PT_OK                 0
#define PT_ERR_INVALID_ARG   -1
#define PT_ERR_NO_DEVICE     -2   /* HIP runtime / MI355X not available: the product path never falls back to a CPU */
#define PT_ERR_HIP           -3
#define PT_ERR_NO_SCENE      -4
#define PT_ERR_UNSUPPORTED   -5

/* Create a render context on HIP device `deviceIndex`.  Fails with PT_ERR_NO_DEVICE when
 * there is no GPU: there is no CPU fallback. */
PT_API int PTCreate(int deviceIndex, PTContext** outCtx);
PT_API int PTDestroy(PTContext* ctx);

/* Copy the scene buffers into HBM (replaces ComputeBuffer.SetData, BVHScene.cs:640-667). */
PT_API int PTSetScene(PTContext* ctx, const PTSceneDesc* scene);

/* Screen-tile sharding for one-process-per-GPU rendering (no reference counterpart;
 * SURVEY.md §8e).  The frame is cut into 16x16-pixel blocks; block (bx,by) belongs to rank
 * (bx + by) % worldSize.  A context renders only its own blocks and writes exact zeros
 * (rgba = 0) elsewhere, so that a sum over ranks reproduces the single-GPU frame bit for bit.
 * rank 0 / worldSize 1 (the default) owns everything. */
PT_API int PTSetTileOwnership(PTContext* ctx, int rank, int worldSize);

/* Frame assembly for tile sharding: a rank's OWNED tiles packed densely (16 bytes per owned pixel, in the order the kernels
 * enumerate pixels) instead of a whole zero-padded frame -- 1/worldSize of the bytes of a sum-reduce.  One process per GPU:
 * pack on every rank, gather the packed buffers on the root (RCCL gather: bench.py), unpack each rank's buffer there.
 * PTGetOwnedTileSlots: float4 slots of the packed buffer for THIS context's ownership and these params (includes padding of
 * partially covered blocks; differs by at most one block row between ranks -- size a gather by the maximum).
 * All buffers are DEVICE pointers on the context's device; the calls are ordered on the context's stream. */
PT_API int PTGetOwnedTileSlots(PTContext* ctx, const PTFrameParams* params, uint64_t* outFloat4Slots);
PT_API int PTPackOwnedTiles(PTContext* ctx, const PTFrameParams* params, const void* dFrame, void* dPacked);
/* Scatters the packed tiles of rank `rank` of `worldSize` into dFrame (any context may unpack any rank's buffer). */
PT_API int PTUnpackTiles(PTContext* ctx, const PTFrameParams* params, int rank, int worldSize, const void* dPacked, void* dFrame);

/* One process, N devices (a C or C# host without an MPI-style launcher): a group owns one context per device, replicates the
 * scene, gives context i the tiles of rank i of N, and assembles every pass on the first device: each device packs its tiles,
 * hipMemcpyPeerAsync moves them over xGMI, the root unpacks.  The per-device passes are enqueued from one host thread per
 * device.  Progressive accumulation works as on one GPU: every device keeps the history of its own tiles
 * (PTGroupFlipFrames / PTGroupResetFrames mirror PTFlipFrames / PTResetFrames on all of them).  The same device may be
 * listed more than once (rehearsal on a one-GPU box).  Frames are bit-identical to the single-GPU frame. */
typedef struct PTGroup PTGroup;
PT_API int PTCreateMulti(const int* deviceIndices, int deviceCount, PTGroup** outGroup);
PT_API int PTGroupDestroy(PTGroup* group);
PT_API int PTGroupSize(PTGroup* group);
PT_API PTContext* PTGroupGetContext(PTGroup* group, int index);      /* for PTSetSchedule / PTSetStatsLevel / ... per device */
PT_API int PTGroupSetScene(PTGroup* group, const PTSceneDesc* scene);
PT_API int PTGroupRenderPass(PTGroup* group, const PTFrameParams* params);
PT_API int PTGroupRenderPassBatch(PTGroup* group, const PTFrameParams* params, int count);   /* `count` passes per device as one launch sequence (PTRenderPassBatch), one assembly */
PT_API int PTGroupFlipFrames(PTGroup* group);
PT_API int PTGroupResetFrames(PTGroup* group);
PT_API int PTGroupSynchronize(PTGroup* group);
PT_API int PTGroupReadback(PTGroup* group, float* dstRGBA, uint64_t dstFloats);   /* the assembled frame of the last pass */
PT_API void* PTGroupGetAssembledFrame(PTGroup* group);                            /* its device pointer (first device)    */
PT_API int PTGroupGetStats(PTGroup* group, PTStats* out);                         /* summed over devices (max for maxStackDepth) */
PT_API int PTGroupResetStats(PTGroup* group);

/* One progressive pass = one DispatchCompute of the PathTracer kernel (PathTracer.cs:251),
 * using the context's internal ping-pong frames exactly as PathTracer.cs:246-247,268-272:
 * Output = frame[cur], AccumulatedOutput = frame[1-cur]; the caller advances CurrentSample
 * and calls PTFlipFrames()/PTResetFrames() as the C# host flips _currentRT / calls Reset(). */
PT_API int PTRenderPass(PTContext* ctx, const PTFrameParams* params);
PT_API int PTFlipFrames(PTContext* ctx);       /* _currentRT = 1 - _currentRT  (PathTracer.cs:271-272) */
PT_API int PTResetFrames(PTContext* ctx);      /* Reset(): _currentRT = 0      (PathTracer.cs:318-322) */

/* Same pass, but into caller-owned DEVICE buffers (float4 per pixel, row-major):
 * dOutput is written, dAccumulated (may be NULL when CurrentSample == 0) is read. */
PT_API int PTRenderPassTo(PTContext* ctx, const PTFrameParams* params, void* dOutput, const void* dAccumulated);

/* `count` consecutive progressive passes (1..8) enqueued as ONE launch sequence: params[j] are the uniforms of pass j and may
 * differ in RngSeedRoot and CurrentSample only.  The result in dOutput is bit-identical to calling PTRenderPassTo `count` times
 * with the caller ping-ponging two frames (pass j reads what pass j-1 wrote; pass 0 reads dAccumulated when its CurrentSample > 0):
 * the paths of different passes are independent and the running mean of PathTracer.compute:89-98 is applied pass by pass in the
 * resolve kernel -- only the intermediate frames are never stored.  For offline / converging renders (the reference keeps
 * dispatching passes until _maxSamples, PathTracer.cs:194-272) and for tile-sharded rendering, where one rank's share of one
 * pass is too small a launch to fill an MI355X: eight passes of a 1/8 share are one whole frame's worth of paths.  PTStats counts
 * pixelsWritten / pixelsRead as the `count` separate passes would. */
PT_API int PTRenderPassBatchTo(PTContext* ctx, const PTFrameParams* params, int count, void* dOutput, const void* dAccumulated);
/* The same over the context's internal frame pair: Output = the current frame, AccumulatedOutput = the other one, as PTRenderPass;
 * the caller advances _currentSample by count x SamplesPerPass and flips ONCE (the intermediate frames do not exist). */
PT_API int PTRenderPassBatch(PTContext* ctx, const PTFrameParams* params, int count);

/* Block until every pass launched so far has finished. */
PT_API int PTSynchronize(PTContext* ctx);
/* Copy the current Output frame (frame[cur]) to host memory: width*height float4. */
PT_API int PTReadback(PTContext* ctx, float* dstRGBA, uint64_t dstFloats);
/* Device pointer of internal frame `which` (0/1) or of the current Output frame (-1). */
PT_API void* PTGetFramePointer(PTContext* ctx, int which);
/* The presentation blit (Assets/Resources/Presentation.shader:36-73 with util/tonemap.hlsl), i.e. the uniforms
 * PathTracer.cs:255-264 sets on _presentationMaterial.  Mode = TonemapMode (PathTracer.cs:8-14). */
typedef struct PTPresentParams {
    uint32_t OutputWidth, OutputHeight;
    int32_t  Mode;            /* 0 none, 1 ACES, 2 Filmic, 3 Reinhard, 4 Lottes */
    int32_t  sRGB;            /* LinearToSrgb after the operator */
    float    Exposure, Brightness, Contrast, Saturation, Vignette;
} PTPresentParams;
/* dst = present(src): both DEVICE pointers to OutputWidth*OutputHeight float4; src == NULL reads the context's current
 * Output frame.  Ordered on the context stream after every pass launched so far (cmd.Blit after DispatchCompute). */
PT_API int PTPresent(PTContext* ctx, const PTPresentParams* params, const void* dSrc, void* dDst);
/* Presents the context's current Output frame into host memory (width*height float4). */
PT_API int PTPresentToHost(PTContext* ctx, const PTPresentParams* params, float* dstRGBA, uint64_t dstFloats);
/* ---- scene ingestion: the two compute shaders that turn Unity meshes / textures into the path
 * tracer's buffers, run on the MI355X.  Inputs and outputs are HOST memory; the work is staged through HBM. ---- */

/* One Dispatch of Assets/Resources/MeshProcessing.compute as BVHScene.ProcessMeshes sets it up (BVHScene.cs:489-553). */
#define PT_MESH_HAS_32_BIT_INDICES 0x1u   /* mesh.indexFormat == UInt32 (only read when indexBuffer != NULL) */
#define PT_MESH_HAS_NORMALS        0x2u
#define PT_MESH_HAS_TANGENTS       0x4u
#define PT_MESH_HAS_UVS            0x8u
typedef struct PTMeshDesc {
    const void* vertexBuffer;  uint64_t vertexBufferBytes;   /* mesh.GetVertexBuffer(0): interleaved, fp32 attributes      */
    const void* indexBuffer;   uint64_t indexBufferBytes;    /* mesh.GetIndexBuffer(); NULL = HAS_INDEX_BUFFER off          */
    uint32_t VertexStride, PositionOffset, NormalOffset, TangentOffset, UVOffset;   /* bytes (Utilities.FindVertexAttribute) */
    uint32_t MaterialIndex;
    uint32_t TriangleCount, OutputTriangleStart;
    float    LocalToWorld[16], WorldToLocal[16];              /* Unity Matrix4x4 memory order, as in PTFrameParams          */
    uint32_t flags;            uint32_t _pad;                 /* PT_MESH_HAS_*                                                */
} PTMeshDesc;
/* ProcessMeshes + the readback (BVHScene.cs:429-560): runs the kernel once per mesh and returns
 * VertexPositionBuffer (3 float4 per triangle, w = 0: BuildBVH's input) and TriangleAttributesBuffer (PTTriangleAttributes
 * per triangle).  totalTriangles = the size of both outputs; every mesh writes [OutputTriangleStart, +TriangleCount). */
PT_API int PTProcessMeshes(PTContext* ctx, const PTMeshDesc* meshes, uint32_t meshCount, uint32_t totalTriangles,
                           float* outVertexPositions, void* outTriangleAttributes);

/* The texture loop of BVHScene.UpdateMaterialData (BVHScene.cs:386-417) over Assets/Resources/CopyTextureData.compute:
 * every texture (RGBA fp32 texels, row y = Texture.Load(int3(x, y, 0))) becomes a {w, h, offset, 0} descriptor and w*h RGBA8
 * words.  outTextureData must hold 4*count + sum(w*h) uints. */
typedef struct PTTextureDesc {
    const float* texels;  uint32_t width, height;  int32_t hasAlpha;  uint32_t _pad;
} PTTextureDesc;
PT_API int PTCopyTextureData(PTContext* ctx, const PTTextureDesc* textures, uint32_t count, uint32_t* outTextureData, uint64_t outUints);

/* The hipStream_t (as void*) all passes of this context are launched on. */
PT_API void* PTGetStream(PTContext* ctx);

/* Counters: level 0 = rays/paths only (always on, free), 1 = full PTStats (slower kernel variant). */
PT_API int PTSetStatsLevel(PTContext* ctx, int level);
PT_API int PTGetStats(PTContext* ctx, PTStats* out);
PT_API int PTResetStats(PTContext* ctx);

PT_API int PTSetProfiling(PTContext* ctx, int enabled);
PT_API int PTGetTimings(PTContext* ctx, PTTimings* out);
PT_API int PTResetTimings(PTContext* ctx);

/* Select the kernel schedule.  All schedules produce bit-identical frames and counters:
 *  -1  auto (default): 1, except for scenes whose BVH has <= 16 nodes (no traversal to speak of), which use 0
 *   0  megakernel: one lane per pixel, one launch per pass
 *   1  wavefront: slot-indexed path state, refill trace kernel + shade kernel, host-sync-free, passes overlap
 *   2  wavefront with the plain one-ray-per-lane trace kernel
 *   3  wavefront with the persistent dynamic-chunk trace kernel
 *   4  fused persistent wavefront: ONE launch per pass; a persistent wave owns 128 path contexts, alternates between tracing
 *      their rays and shading them, and hands a context whose pixel has finished the next pixel of the frame.  One path-state
 *      set of ~0.5 M contexts instead of one slot per pixel and pass in flight; the fastest schedule when passes are NOT
 *      pipelined (a host that presents every pass).  HAS_TLAS scenes run schedule 1's kernels. */
PT_API int PTSetSchedule(PTContext* ctx, int schedule);
/* The schedule (0..4) the next pass will run with the current scene: resolves -1 (auto).  Negative = error code. */
PT_API int PTGetSchedule(PTContext* ctx);
/* Wavefront schedules: number of trace+shade iterations launched per pass before the cleanup kernel finishes whatever is
 * still alive.  0 (default) = SamplesPerPass * (MaxRayBounces + 2) + 4.  Any value gives the same frame; it only moves
 * work between the wavefront kernels and the cleanup kernel (tuning / test knob). */
PT_API int PTSetWavefrontIterations(PTContext* ctx, int iterations);

/* Wavefront schedules: how many consecutive passes may be in flight at once, each on its own path-state set and HIP stream
 * (the pixel write of a pass is decoupled from its path loop, so pass k+1 traces while pass k drains; only the resolves are
 * ordered).  1 = passes run back to back (what a host that reads every frame back gets anyway).  0 (default) = sized to the
 * hardware queues the host asked the HIP runtime for: GPU_MAX_HW_QUEUES >= 16 -> 12, >= 8 -> 6, otherwise 3 (streams that
 * share a hardware queue serialise; the library never changes the environment itself).  Each set costs ~290 B per owned pixel
 * and is allocated on the first pass that can use it; lowering the number frees the sets no longer used.  Same frame for
 * every value. */
PT_API int PTSetPassesInFlight(PTContext* ctx, int passes);
PT_API int PTGetPassesInFlight(PTContext* ctx);
/* Wavefront schedules: cut every pass into `subFrames` (1 .. passes in flight) interleaved subsets of the context's 16x16 pixel
 * blocks, each rendered by its own launch sequence on its own state set and stream into the same output frame.  For a host that
 * consumes every frame before it asks for the next (the reference presents every pass, PathTracer.cs:251-272) this gives ONE
 * pass the overlap that otherwise only several passes in flight have.  Same frame for every value; 1 (default) = off. */
PT_API int PTSetSubFrames(PTContext* ctx, int subFrames);

/* Text of the last error on the calling thread ("" if none). */
PT_API const char* PTGetLastError(void);
/* Library/ABI version: (major << 16) | minor. */
PT_API int PTGetVersion(void);

#ifdef __cplusplus
}
#endif

#endif /* PTMI_PLUGIN_H */
