/*
 * ptmi_layouts.h — byte layouts of every buffer that crosses the drop-in boundary.
 *
 * These are the buffer contracts of the reference's compute-shader path; the MI355X
 * kernels read exactly these bytes, so a host that packs scenes for the reference
 * (BVHScene.cs / PathTracer.cs) can hand the same arrays to this library.
 * Citations are relative to /root/reference (see SURVEY.md for the abbreviations).
 *
 * Plain C (C99) and C++ both include this header. No torch / HIP types appear here.
 */
#ifndef PTMI_LAYOUTS_H
#define PTMI_LAYOUTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 16-byte vector, the unit of every geometry buffer (tinybvh::bvhvec4; HLSL float4). */
typedef struct PTFloat4 { float x, y, z, w; } PTFloat4;

/* CWBVH node, 80 bytes = 5 x float4.  Assets/Resources/util/common.hlsl:13-20,
 * writer Assets/Plugins/Web/tiny_bvh.h:5925-5975.
 *   n0: lo.xyz | u8 ex, ey, ez, imask
 *   n1: u32 childBaseIndex | u32 triBaseIndex (float4 units) | u8 meta[8]
 *   n2..n4: u8 qlox[8] qloy[8] | qloz[8] qhix[8] | qhiy[8] qhiz[8]               */
typedef struct PTCwbvhNode {
    float    lo[3];
    uint8_t  ex, ey, ez, imask;
    uint32_t childBaseIndex;
    uint32_t triBaseIndex;
    uint8_t  meta[8];
    uint8_t  qlox[8], qloy[8], qloz[8];
    uint8_t  qhix[8], qhiy[8], qhiz[8];
} PTCwbvhNode;

/* One CWBVH triangle = 3 x float4: (e2 = v2-v0), (e1 = v1-v0), (v0, primIdx bits).
 * tiny_bvh.h:5964-5968, reader util/bvh.hlsl:23-27,49.                            */
typedef struct PTCwbvhTri {
    PTFloat4 e2;
    PTFloat4 e1;
    float    v0[3];
    uint32_t primIdx;
} PTCwbvhTri;

/* Per-triangle shading attributes, 128 bytes.  util/triangle_attributes.hlsl:5-31,
 * writer MeshProcessing.compute:108-138.                                           */
typedef struct PTTriangleAttributes {
    float normal0[3];  float pad0;
    float normal1[3];  float pad1;
    float normal2[3];  float pad2;
    float tangent0[3]; float pad3;
    float tangent1[3]; float pad4;
    float tangent2[3]; float pad5;
    float uv0[2];      float uv1[2];
    float uv2[2];      uint32_t materialIndex; float pad6;
} PTTriangleAttributes;

/* MaterialData, 32 floats = 128 bytes.  util/common.hlsl:72-83, packing order
 * Assets/Scripts/util/BVHScene.cs:236-282.                                         */
typedef struct PTMaterialData {
    float baseColor[3];   float opacity;           /* data1: linear rgb, a*(1-transmission) */
    float emission[3];    float alphaCutoff;       /* data2 */
    float metallic, roughness, normalScale, ior;   /* data3 */
    float alphaMode, anisotropic, specular, specularTint; /* data4 (alphaMode: 0 opaque 1 blend 2 mask) */
    float sheen, sheenTint, subsurface, clearCoat; /* data5 */
    float clearCoatGloss, oneMinusOpacity;         /* data6 */
    float texBaseColor, texMetallicRoughness;      /* textures1 (index as float, <0 = none) */
    float texNormal, texEmission, texOcclusion, texPad; /* textures2 */
    float uvScale[2], uvOffset[2];                 /* texture1Transform */
} PTMaterialData;

/* Analytic light, 16 floats = 64 bytes.  util/common.hlsl:147-160, packing
 * Assets/Scripts/PathTracer.cs:442-460 (type = Unity LightType as int bits).       */
typedef struct PTLight {
    float    position[3]; uint32_t type;   /* 0 spot, 1 directional, 2 point, 3 rectangle */
    float    emission[3]; float range;
    float    u[3];        float area;
    float    v[3];        float pad;
} PTLight;

#define PT_LIGHT_TYPE_SPOT        0u
#define PT_LIGHT_TYPE_DIRECTIONAL 1u
#define PT_LIGHT_TYPE_POINT       2u
#define PT_LIGHT_TYPE_RECTANGLE   3u

/* TLAS node in Aila-Laine layout, 64 bytes.  tiny_bvh.h:1094-1104.                 */
typedef struct PTTlasNode {
    float lmin[3]; uint32_t left;
    float lmax[3]; uint32_t right;
    float rmin[3]; uint32_t triCount;
    float rmax[3]; uint32_t firstTri;
} PTTlasNode;

/* Input to BuildTLAS, 192 bytes.  BVHScene.cs:26-37 <-> tiny_bvh.h:1442-1456.
 * Only aabbMin/aabbMax are read by the builder (plugin.cpp:111-118 passes no BLAS list). */
typedef struct PTBlasInstance {
    float    localToWorld[16];
    float    worldToLocal[16];
    float    aabbMin[3]; uint32_t blasIndex;
    float    aabbMax[3]; uint32_t mask;
    uint32_t pad[8];
} PTBlasInstance;

/* Per-instance record read by the HAS_TLAS shader variant, 144 bytes.  BVHScene.cs:14-22. */
typedef struct PTGpuInstance {
    float   localToWorld[16];   /* Unity Matrix4x4 memory order: element (r,c) at c*4+r */
    float   worldToLocal[16];
    int32_t bvhOffset;          /* in nodes   */
    int32_t triOffset;          /* in float4s */
    int32_t triAttributeOffset; /* in triangles */
    int32_t materialIndex;
} PTGpuInstance;

/* Constants shared by every implementation.  util/common.hlsl:4-10, util/bvh.hlsl:11,32,47 */
#define PT_EPSILON        0.0001f
#define PT_FAR_PLANE      100000.0f
#define PT_BVH_STACK_SIZE 32
#define PT_AA_SIGMA       0.4246609f   /* PathTracer.compute:31 */

#define PT_SKY_MODE_ENVIRONMENT 0
#define PT_SKY_MODE_BASIC       1
#define PT_ALPHA_MODE_OPAQUE    0
#define PT_ALPHA_MODE_BLEND     1
#define PT_ALPHA_MODE_MASK      2

#ifdef __cplusplus
}
static_assert(sizeof(PTFloat4) == 16, "float4");
static_assert(sizeof(PTCwbvhNode) == 80, "CWBVH node is 80 bytes");
static_assert(sizeof(PTCwbvhTri) == 48, "CWBVH triangle is 48 bytes");
static_assert(sizeof(PTTriangleAttributes) == 128, "TriangleAttributes is 128 bytes");
static_assert(sizeof(PTMaterialData) == 128, "MaterialData is 128 bytes");
static_assert(sizeof(PTLight) == 64, "Light is 64 bytes");
static_assert(sizeof(PTTlasNode) == 64, "TLAS node is 64 bytes");
static_assert(sizeof(PTBlasInstance) == 192, "BLASInstance is 192 bytes");
static_assert(sizeof(PTGpuInstance) == 144, "GPUInstance is 144 bytes");
#else
_Static_assert(sizeof(PTCwbvhNode) == 80, "CWBVH node is 80 bytes");
_Static_assert(sizeof(PTCwbvhTri) == 48, "CWBVH triangle is 48 bytes");
_Static_assert(sizeof(PTTriangleAttributes) == 128, "TriangleAttributes is 128 bytes");
_Static_assert(sizeof(PTMaterialData) == 128, "MaterialData is 128 bytes");
_Static_assert(sizeof(PTLight) == 64, "Light is 64 bytes");
_Static_assert(sizeof(PTTlasNode) == 64, "TLAS node is 64 bytes");
_Static_assert(sizeof(PTBlasInstance) == 192, "BLASInstance is 192 bytes");
_Static_assert(sizeof(PTGpuInstance) == 144, "GPUInstance is 144 bytes");
#endif

#endif /* PTMI_LAYOUTS_H */
