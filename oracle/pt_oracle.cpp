// oracle/pt_oracle.cpp — CPU oracle: scalar restatement of the reference's path-tracing kernel.
//
// TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  One function per shader function, same control flow,
// same RNG draw order, the fp32 expression trees fixed by include/ptmi_math.h.
// Citations are relative to /root/reference/Assets/Resources.
//
// Canonical evaluation rules used throughout (documented in DESIGN.md "Canonical arithmetic"):
//   * every HLSL binary operator is one IEEE fp32 operation, evaluated left to right; no contraction,
//     EXCEPT the CWBVH child-slab computation t = q * scale + pos, which is one fused pt_fma (HLSL
//     backends emit mad there; it is the inner-most operation of the whole path);
//   * dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;  normalize(v) = v * (1 / sqrt(dot(v,v)));
//   * v / s is three true divisions;  rcp(x) = 1/x;  rsqrt(x) = 1/sqrt(x);
//   * lerp(a,b,t) = a + t*(b-a);  reflect(i,n) = i - (2*dot(n,i))*n;
//   * function arguments are evaluated left to right (PathTracer.compute:71 draws two randoms in one call);
//   * int(RandomFloat*LightCount) is clamped to LightCount-1 (RandomFloat can return exactly 1.0).
#include "pt_oracle.h"
#include "ptmi_math.h"

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// vectors
// ------------------------------------------------------------------------------------------
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline f3 V3(float a) { return {a, a, a}; }
inline f3 V3(float a, float b, float c) { return {a, b, c}; }
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
inline f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline f3 operator-(float s, f3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
inline f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
inline f3& operator*=(f3& a, float s) { a = a * s; return a; }
inline f3& operator/=(f3& a, float s) { a = a / s; return a; }
inline f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
inline f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
inline f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline f4 operator*(float s, f4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }

inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(f3 v) { return pt_sqrt(dot(v, v)); }
inline f3 normalize(f3 v) { return v * (1.0f / pt_sqrt(dot(v, v))); }
inline f3 lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }
inline f4 lerp4(f4 a, f4 b, float t) { return a + t * (b - a); }
inline f3 reflect(f3 i, f3 n) { return i - (2.0f * dot(n, i)) * n; }
inline f3 refract(f3 i, f3 n, float eta)
{
    float d = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3(0.0f);
    return eta * i - (eta * d + pt_sqrt(k)) * n;
}
inline float Luminance(f3 c) { return dot(c, V3(0.299f, 0.587f, 0.114f)); }   // util/common.hlsl:195-198

struct Mat3 { f3 r0, r1, r2; };                                              // float3x3 rows
inline f3 ToWorld(const Mat3& b, f3 l) { return b.r0 * l.x + b.r1 * l.y + b.r2 * l.z; }   // util/common.hlsl:386-389
inline f3 ToLocal(const Mat3& b, f3 w) { return {dot(b.r0, w), dot(b.r1, w), dot(b.r2, w)}; }  // :391-394

// util/common.hlsl:346-384 (ONB_METHOD 1)
inline Mat3 GetONB(f3 z)
{
    float lenSq = dot(z, z);
    if (lenSq == 0.0f) return {V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)};
    z = normalize(z);
    float k = 1.0f / pt_max(1.0f + z.z, 0.00001f);
    float a = z.y * k;
    float b = z.y * a;
    float c = -z.x * a;
    f3 x = normalize(V3(z.z + b, c, -z.x));
    f3 y = normalize(V3(c, 1.0f - b, -z.y));
    return {x, y, z};
}

// Unity Matrix4x4 memory order: element (r, c) at [c*4 + r]; mul(M, v) with column vector v.
inline f4 mul44(const float* m, f4 v)
{
    f4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}

// ------------------------------------------------------------------------------------------
// scene view + per-thread counters
// ------------------------------------------------------------------------------------------
struct SceneView {
    const PTCwbvhNode* nodes; uint64_t nodeCount;
    const PTFloat4* tris; uint64_t triFloat4s;
    const PTTriangleAttributes* attrs; uint64_t attrCount;
    const PTMaterialData* materials; uint32_t materialCount;
    const PTLight* lights; int32_t lightCount;
    const uint32_t* tex; uint64_t texUints;
    const float* tlas; uint64_t tlasFloats; uint32_t tlasIndexOffset;
    const PTGpuInstance* instances; uint32_t instanceCount;
    bool hasLights, hasTextures, hasTlas;
    // HAS_ENVIRONMENT_TEXTURE: "EnvironmentTexture" texels (RGBAFloat), "EnvironmentCDF", EnvironmentCdfSum, width, height
    const PTFloat4* envTexture; std::vector<float> EnvironmentCDF; float EnvironmentCdfSum;
    int32_t EnvironmentTextureWidth, EnvironmentTextureHeight;
    bool hasEnvironmentTexture;
};

struct Ctx {
    const SceneView* S;
    const PTFrameParams* P;
    PTStats st;
    bool shadowAnyHit;
    bool nanRayEarlyOut;
    OracleOptions* opt;     // ray dump target (single-threaded use)
};

struct Ray { f3 origin; f3 direction; };

// util/common.hlsl:173-193
struct RayHit {
    f3 position; float distance;
    f2 barycentric; uint32_t triIndex; uint32_t triAddr;
    f3 normal; uint32_t steps;
    int32_t materialIndex;
    f3 ffnormal; uint32_t intersectType;
    f2 uv;
};
enum { INTERSECT_TRIANGLE = 0, INTERSECT_LIGHT = 1 };

// util/common.hlsl:85-135
struct Material {
    f3 baseColor; float opacity;
    f3 emission; float alphaMode;
    float alphaCutoff, anisotropic, metallic, roughness;
    float subsurface, specularTint, sheen, sheenTint;
    float clearcoat, clearcoatRoughness, specTrans, ior;
    float ax, ay, eta, occlusion;
};

inline float RandomFloat(uint32_t& s) { return pt_random_float(&s); }

// ------------------------------------------------------------------------------------------
// util/bvh.hlsl — CWBVH traversal
// ------------------------------------------------------------------------------------------
inline f2 Interp2(f2 b, f2 a0, f2 a1, f2 a2) { return a0 * (1.0f - b.x - b.y) + a1 * b.x + a2 * b.y; }        // :13-16
inline f3 Interp3(f2 b, f3 a0, f3 a1, f3 a2) { return a0 * (1.0f - b.x - b.y) + a1 * b.x + a2 * b.y; }        // :18-21

// util/bvh.hlsl:23-59
inline void IntersectTriangle(Ctx& c, uint32_t triAddr, const Ray& ray, RayHit& hit)
{
    const PTFloat4* t = c.S->tris + triAddr;
    f3 v0 = V3(t[2].x, t[2].y, t[2].z);
    f3 e1 = V3(t[1].x, t[1].y, t[1].z);
    f3 e2 = V3(t[0].x, t[0].y, t[0].z);
    c.st.triTests++;
    f3 r = cross(ray.direction, e2);
    float a = dot(e1, r);
    if (pt_abs(a) > 0.0000001f) {
        float f = 1.0f / a;
        f3 s = ray.origin - v0;
        float u = f * dot(s, r);
        if (u >= 0.0f && u <= 1.0f) {
            f3 q = cross(s, e1);
            float v = f * dot(ray.direction, q);
            if (v >= 0.0f && u + v <= 1.0f) {
                float d = f * dot(e2, q);
                if (d > 0.0001f && d < hit.distance) {
                    hit.barycentric = {u, v};
                    hit.triAddr = triAddr;
                    hit.triIndex = pt_asuint(t[2].w);
                    hit.distance = d;
                }
            }
        }
    }
}

inline uint32_t ExtractByte(uint32_t v, uint32_t i) { return (v >> (i * 8u)) & 0xFFu; }    // util/common.hlsl:214-217

// util/bvh.hlsl:61-75
inline f3 GetNodeInvDir(uint32_t packed, f3 invDir)
{
    uint32_t ex = (ExtractByte(packed, 0) ^ 0x80u) - 0x80u;
    uint32_t ey = (ExtractByte(packed, 1) ^ 0x80u) - 0x80u;
    uint32_t ez = (ExtractByte(packed, 2) ^ 0x80u) - 0x80u;
    return V3(pt_asfloat((ex + 127u) << 23) * invDir.x,
              pt_asfloat((ey + 127u) << 23) * invDir.y,
              pt_asfloat((ez + 127u) << 23) * invDir.z);
}

// util/bvh.hlsl:77-124.  The node is read as 20 dwords n[0..19] (n0 = 0-3, n1 = 4-7, ...).
inline uint32_t IntersectCWBVHNode(f3 origin, f3 invDir, uint32_t octinv4, float tmax, const uint32_t* n)
{
    uint32_t hitmask = 0;
    f3 nodeInvDir = GetNodeInvDir(n[3], invDir);
    f3 lo = V3(pt_asfloat(n[0]), pt_asfloat(n[1]), pt_asfloat(n[2]));
    f3 nodePos = (lo - origin) * invDir;
    for (int i = 0; i < 2; ++i) {
        uint32_t meta = i == 0 ? n[6] : n[7];
        // n2 = 8..11 (qlox lo/hi, qloy lo/hi), n3 = 12..15 (qloz, qhix), n4 = 16..19 (qhiy, qhiz)
        uint32_t qlox = n[8 + i], qloy = n[10 + i], qloz = n[12 + i];
        uint32_t qhix = n[14 + i], qhiy = n[16 + i], qhiz = n[18 + i];
        uint32_t lox = invDir.x < 0.0f ? qhix : qlox, hix = invDir.x < 0.0f ? qlox : qhix;
        uint32_t loy = invDir.y < 0.0f ? qhiy : qloy, hiy = invDir.y < 0.0f ? qloy : qhiy;
        uint32_t loz = invDir.z < 0.0f ? qhiz : qloz, hiz = invDir.z < 0.0f ? qloz : qhiz;
        uint32_t isInner = (meta & (meta << 1)) & 0x10101010u;
        uint32_t innerMask = (isInner >> 4) * 0xffu;
        uint32_t bitIndex = (meta ^ (octinv4 & innerMask)) & 0x1F1F1F1Fu;
        uint32_t childBits = (meta >> 5) & 0x07070707u;
        for (int j = 0; j < 4; ++j) {
            float tminx = pt_fma((float)ExtractByte(lox, j), nodeInvDir.x, nodePos.x);
            float tmaxx = pt_fma((float)ExtractByte(hix, j), nodeInvDir.x, nodePos.x);
            float tminy = pt_fma((float)ExtractByte(loy, j), nodeInvDir.y, nodePos.y);
            float tmaxy = pt_fma((float)ExtractByte(hiy, j), nodeInvDir.y, nodePos.y);
            float tminz = pt_fma((float)ExtractByte(loz, j), nodeInvDir.z, nodePos.z);
            float tmaxz = pt_fma((float)ExtractByte(hiz, j), nodeInvDir.z, nodePos.z);
            float cmin = pt_max(pt_max(pt_max(tminx, tminy), tminz), 0.0f);
            float cmax = pt_min(pt_min(pt_min(tmaxx, tmaxy), tmaxz), tmax);
            if (cmin <= cmax) {
                uint32_t shiftBits = (childBits >> (j * 8)) & 255u;
                uint32_t bitShift = (bitIndex >> (j * 8)) & 31u;
                hitmask |= shiftBits << bitShift;
            }
        }
    }
    return hitmask;
}

inline uint32_t firstbithigh(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// util/bvh.hlsl:126-215
bool RayIntersectBvh(Ctx& c, const Ray& ray, RayHit& hit, bool isShadowRay)
{
    const SceneView& S = *c.S;
    f3 invDir = V3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    uint32_t octinv4 = (7u - ((ray.direction.x < 0 ? 4u : 0u) | (ray.direction.y < 0 ? 2u : 0u) | (ray.direction.z < 0 ? 1u : 0u))) * 0x1010101u;

    if (c.opt && c.opt->rayDump && c.opt->rayDumpCount < c.opt->rayDumpCapacity) {
        OracleRay& d = c.opt->rayDump[c.opt->rayDumpCount++];
        d.origin[0] = ray.origin.x; d.origin[1] = ray.origin.y; d.origin[2] = ray.origin.z;
        d.direction[0] = ray.direction.x; d.direction[1] = ray.direction.y; d.direction[2] = ray.direction.z;
        d.tmax = hit.distance; d.kind = isShadowRay ? 1.0f : 0.0f;
    }

    // Optional (counting variant only): a ray carrying a NaN can never hit a triangle, yet passes every slab
    // test, so the reference walks the whole tree for it; the GPU kernels return the miss at once.
    if (c.nanRayEarlyOut && (pt_isnan(ray.origin.x) || pt_isnan(ray.origin.y) || pt_isnan(ray.origin.z) ||
                             pt_isnan(ray.direction.x) || pt_isnan(ray.direction.y) || pt_isnan(ray.direction.z))) {
        hit.steps = 0;
        return hit.distance < PT_FAR_PLANE;
    }

    uint32_t stackX[PT_BVH_STACK_SIZE], stackY[PT_BVH_STACK_SIZE];
    uint32_t stackPtr = 0;
    uint32_t ngx = 0, ngy = 0x80000000u;      // the shader's 0x80000001 work-around (:133-147) is an encoding detail
    uint32_t tgx = 0, tgy = 0;
    uint32_t steps = 0;
    bool overflowed = false;

    while (true) {
        if (ngy > 0x00FFFFFFu) {
            uint32_t mask = ngy;
            uint32_t childBitIndex = firstbithigh(mask);
            uint32_t childNodeBaseIndex = ngx;
            ngy &= ~(1u << childBitIndex);
            if (ngy > 0x00FFFFFFu) {
                if (stackPtr < PT_BVH_STACK_SIZE) { stackX[stackPtr] = ngx; stackY[stackPtr] = ngy; }
                else overflowed = true;
                stackPtr++;
                if (stackPtr > c.st.maxStackDepth) c.st.maxStackDepth = stackPtr;
            }
            uint32_t slotIndex = (childBitIndex - 24u) ^ (octinv4 & 255u);
            uint32_t relativeIndex = (uint32_t)__builtin_popcount(mask & ~(0xFFFFFFFFu << slotIndex));
            uint32_t childNodeIndex = childNodeBaseIndex + relativeIndex;
            const uint32_t* n = (const uint32_t*)(S.nodes + childNodeIndex);
            uint32_t hitmask = IntersectCWBVHNode(ray.origin, invDir, octinv4, hit.distance, n);
            c.st.nodeVisits++;
            ngx = n[4];
            ngy = (hitmask & 0xFF000000u) | (n[3] >> 24);
            tgx = n[5];
            tgy = hitmask & 0x00FFFFFFu;
            steps++;
        } else {
            tgx = ngx; tgy = ngy;
            ngx = 0; ngy = 0;
        }
        while (tgy != 0) {
            uint32_t triangleIndex = firstbithigh(tgy);
            uint32_t triAddr = tgx + triangleIndex * 3u;
            IntersectTriangle(c, triAddr, ray, hit);
            tgy -= 1u << triangleIndex;
            if (isShadowRay && c.shadowAnyHit && hit.distance < PT_FAR_PLANE) goto done;
        }
        if (ngy <= 0x00FFFFFFu) {
            if (stackPtr > 0) {
                --stackPtr;
                if (stackPtr < PT_BVH_STACK_SIZE) { ngx = stackX[stackPtr]; ngy = stackY[stackPtr]; }
                else { ngx = 0; ngy = 0; }
            } else break;
        }
    }
done:
    if (overflowed) c.st.stackOverflows++;
    hit.steps = steps;

    if (!isShadowRay && hit.distance < PT_FAR_PLANE) {
        const PTTriangleAttributes& ta = S.attrs[hit.triIndex];
        c.st.attrFetches++;
        hit.position = ray.origin + hit.distance * ray.direction;
        // hit.tangent is computed by the shader (:205) but never consumed (BRDF frames come from GetONB); omitted.
        hit.normal = normalize(Interp3(hit.barycentric, V3(ta.normal0[0], ta.normal0[1], ta.normal0[2]),
                                       V3(ta.normal1[0], ta.normal1[1], ta.normal1[2]),
                                       V3(ta.normal2[0], ta.normal2[1], ta.normal2[2])));
        hit.ffnormal = dot(hit.normal, ray.direction) <= 0.0f ? hit.normal : -hit.normal;
        hit.uv = Interp2(hit.barycentric, f2{ta.uv0[0], ta.uv0[1]}, f2{ta.uv1[0], ta.uv1[1]}, f2{ta.uv2[0], ta.uv2[1]});
        hit.materialIndex = (int32_t)ta.materialIndex;
        hit.intersectType = INTERSECT_TRIANGLE;
    }
    return hit.distance < PT_FAR_PLANE;
}

// ------------------------------------------------------------------------------------------
// util/tlas.hlsl — HAS_TLAS variant: 2-wide TLAS over instances, each instance a CWBVH in its local space
// ------------------------------------------------------------------------------------------
// tlas.hlsl:22-59.  Differences from util/bvh.hlsl:23-59: accept distance > 0 (not 1e-4), triIndex gets the
// instance's attribute offset, returns whether THIS call accepted the triangle.
inline bool IntersectTriangleInst(Ctx& c, const PTGpuInstance& inst, uint32_t triAddr, const Ray& ray, RayHit& hit)
{
    const PTFloat4* t = c.S->tris + triAddr;
    f3 v0 = V3(t[2].x, t[2].y, t[2].z);
    f3 e1 = V3(t[1].x, t[1].y, t[1].z);
    f3 e2 = V3(t[0].x, t[0].y, t[0].z);
    c.st.triTests++;
    bool hitFound = false;
    f3 r = cross(ray.direction, e2);
    float a = dot(e1, r);
    if (pt_abs(a) > 0.0000001f) {
        float f = 1.0f / a;
        f3 s = ray.origin - v0;
        float u = f * dot(s, r);
        if (u >= 0.0f && u <= 1.0f) {
            f3 q = cross(s, e1);
            float v = f * dot(ray.direction, q);
            if (v >= 0.0f && u + v <= 1.0f) {
                float d = f * dot(e2, q);
                if (d > 0.0f && d < hit.distance) {
                    hit.barycentric = {u, v};
                    hit.triAddr = triAddr;
                    hit.triIndex = (uint32_t)inst.triAttributeOffset + pt_asuint(t[2].w);
                    hit.distance = d;
                    hitFound = true;
                }
            }
        }
    }
    return hitFound;
}

// mul(float4(v, 0), M): row vector times matrix (Unity memory order: element (r, c) at [c*4 + r])
inline f3 mulRow3(f3 v, const float* m)
{
    return V3(v.x * m[0] + v.y * m[1] + v.z * m[2] + 0.0f * m[3],
              v.x * m[4] + v.y * m[5] + v.z * m[6] + 0.0f * m[7],
              v.x * m[8] + v.y * m[9] + v.z * m[10] + 0.0f * m[11]);
}

// tlas.hlsl:129-234
bool RayIntersectBvhInstance(Ctx& c, const Ray& worldRay, const PTGpuInstance& inst, bool isShadowRay, RayHit& hit, bool& stopNow)
{
    const SceneView& S = *c.S;
    c.st.instanceVisits++;
    f4 lo4 = mul44(inst.worldToLocal, f4{worldRay.origin.x, worldRay.origin.y, worldRay.origin.z, 1.0f});
    f4 ld4 = mul44(inst.worldToLocal, f4{worldRay.direction.x, worldRay.direction.y, worldRay.direction.z, 0.0f});
    Ray ray = {V3(lo4.x, lo4.y, lo4.z), V3(ld4.x, ld4.y, ld4.z)};          // direction is NOT re-normalised (:133-135)
    f3 invDir = V3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    uint32_t octinv4 = (7u - ((ray.direction.x < 0 ? 4u : 0u) | (ray.direction.y < 0 ? 2u : 0u) | (ray.direction.z < 0 ? 1u : 0u))) * 0x1010101u;
    bool hitFound = false;
    uint32_t stackX[PT_BVH_STACK_SIZE], stackY[PT_BVH_STACK_SIZE];
    uint32_t stackPtr = 0;
    uint32_t ngx = 0, ngy = 0x80000000u, tgx = 0, tgy = 0;
    bool overflowed = false;
    const uint32_t nodeOffset = (uint32_t)inst.bvhOffset;
    while (true) {
        if (ngy > 0x00FFFFFFu) {
            uint32_t mask = ngy;
            uint32_t childBitIndex = firstbithigh(mask);
            uint32_t childNodeBaseIndex = ngx;
            ngy &= ~(1u << childBitIndex);
            if (ngy > 0x00FFFFFFu) {
                if (stackPtr < PT_BVH_STACK_SIZE) { stackX[stackPtr] = ngx; stackY[stackPtr] = ngy; }
                else overflowed = true;
                stackPtr++;
                if (stackPtr > c.st.maxStackDepth) c.st.maxStackDepth = stackPtr;
            }
            uint32_t slotIndex = (childBitIndex - 24u) ^ (octinv4 & 255u);
            uint32_t relativeIndex = (uint32_t)__builtin_popcount(mask & ~(0xFFFFFFFFu << slotIndex));
            uint32_t childNodeIndex = childNodeBaseIndex + relativeIndex;
            const uint32_t* n = (const uint32_t*)(S.nodes + nodeOffset + childNodeIndex);
            uint32_t hitmask = IntersectCWBVHNode(ray.origin, invDir, octinv4, hit.distance, n);
            c.st.nodeVisits++;
            ngx = n[4];
            ngy = (hitmask & 0xFF000000u) | (n[3] >> 24);
            tgx = n[5];
            tgy = hitmask & 0x00FFFFFFu;
        } else {
            tgx = ngx; tgy = ngy;
            ngx = 0; ngy = 0;
        }
        while (tgy != 0) {
            uint32_t triangleIndex = firstbithigh(tgy);
            uint32_t triAddr = tgx + triangleIndex * 3u;
            hitFound = IntersectTriangleInst(c, inst, (uint32_t)inst.triOffset + triAddr, ray, hit) | hitFound;
            tgy -= 1u << triangleIndex;
            if (isShadowRay && c.shadowAnyHit && hitFound) { stopNow = true; goto done; }
        }
        if (ngy <= 0x00FFFFFFu) {
            if (stackPtr > 0) {
                --stackPtr;
                if (stackPtr < PT_BVH_STACK_SIZE) { ngx = stackX[stackPtr]; ngy = stackY[stackPtr]; }
                else { ngx = 0; ngy = 0; }
            } else break;
        }
    }
done:
    if (overflowed) c.st.stackOverflows++;
    if (!isShadowRay && hitFound) {
        const PTTriangleAttributes& ta = S.attrs[hit.triIndex];
        c.st.attrFetches++;
        hit.intersectType = INTERSECT_TRIANGLE;
        f3 lp = ray.origin + hit.distance * ray.direction;
        f4 wp = mul44(inst.localToWorld, f4{lp.x, lp.y, lp.z, 1.0f});
        hit.position = V3(wp.x, wp.y, wp.z);
        hit.distance = length(hit.position - worldRay.origin);             // world-space length; later instances compare their LOCAL t against it (:217)
        hit.uv = Interp2(hit.barycentric, f2{ta.uv0[0], ta.uv0[1]}, f2{ta.uv1[0], ta.uv1[1]}, f2{ta.uv2[0], ta.uv2[1]});
        f3 normal = normalize(Interp3(hit.barycentric, V3(ta.normal0[0], ta.normal0[1], ta.normal0[2]),
                                      V3(ta.normal1[0], ta.normal1[1], ta.normal1[2]),
                                      V3(ta.normal2[0], ta.normal2[1], ta.normal2[2])));
        hit.normal = normalize(mulRow3(normal, inst.worldToLocal));        // inverse-transpose (:223)
        hit.ffnormal = dot(hit.normal, worldRay.direction) <= 0.0f ? hit.normal : -hit.normal;
        hit.materialIndex = inst.materialIndex;
    }
    return hit.distance < PT_FAR_PLANE;
}

inline float selectf(float f, float t, bool c) { return c ? t : f; }          // util/common.hlsl:234-237

// tlas.hlsl:236-332
bool RayIntersectTLAS(Ctx& c, const Ray& ray, RayHit& hit, bool isShadowRay)
{
    const SceneView& S = *c.S;
    if (c.opt && c.opt->rayDump && c.opt->rayDumpCount < c.opt->rayDumpCapacity) {
        OracleRay& d = c.opt->rayDump[c.opt->rayDumpCount++];
        d.origin[0] = ray.origin.x; d.origin[1] = ray.origin.y; d.origin[2] = ray.origin.z;
        d.direction[0] = ray.direction.x; d.direction[1] = ray.direction.y; d.direction[2] = ray.direction.z;
        d.tmax = hit.distance; d.kind = isShadowRay ? 1.0f : 0.0f;
    }
    if (c.nanRayEarlyOut && (pt_isnan(ray.origin.x) || pt_isnan(ray.origin.y) || pt_isnan(ray.origin.z) ||
                             pt_isnan(ray.direction.x) || pt_isnan(ray.direction.y) || pt_isnan(ray.direction.z)))
        return false;
    f3 O = ray.origin;
    f3 D = normalize(ray.direction);
    f3 rD = V3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    bool hitFound = false;
    uint32_t stack[PT_BVH_STACK_SIZE];
    uint32_t nodeIndex = 0, stackPtr = 0;
    const float* T = S.tlas;
    while (true) {
        uint32_t no = nodeIndex * 16u;
        c.st.tlasNodeVisits++;
        uint32_t instanceCount = pt_asuint(T[no + 11]);
        if (instanceCount == 0) {
            f3 lmin = V3(T[no + 0], T[no + 1], T[no + 2]), lmax = V3(T[no + 4], T[no + 5], T[no + 6]);
            f3 rmin = V3(T[no + 8], T[no + 9], T[no + 10]), rmax = V3(T[no + 12], T[no + 13], T[no + 14]);
            uint32_t left = pt_asuint(T[no + 3]), right = pt_asuint(T[no + 7]);
            f3 t1a = (lmin - O) * rD, t2a = (lmax - O) * rD;
            f3 minta = V3(pt_min(t1a.x, t2a.x), pt_min(t1a.y, t2a.y), pt_min(t1a.z, t2a.z));
            f3 maxta = V3(pt_max(t1a.x, t2a.x), pt_max(t1a.y, t2a.y), pt_max(t1a.z, t2a.z));
            float tmina = pt_max(pt_max(pt_max(minta.x, minta.y), minta.z), 0.0f);
            float tmaxa = pt_min(pt_min(pt_min(maxta.x, maxta.y), maxta.z), hit.distance);
            float dist1 = selectf(tmina, PT_FAR_PLANE, tmina > tmaxa);
            f3 t1b = (rmin - O) * rD, t2b = (rmax - O) * rD;
            f3 mintb = V3(pt_min(t1b.x, t2b.x), pt_min(t1b.y, t2b.y), pt_min(t1b.z, t2b.z));
            f3 maxtb = V3(pt_max(t1b.x, t2b.x), pt_max(t1b.y, t2b.y), pt_max(t1b.z, t2b.z));
            float tminb = pt_max(pt_max(pt_max(mintb.x, mintb.y), mintb.z), 0.0f);
            float tmaxb = pt_min(pt_min(pt_min(maxtb.x, maxtb.y), maxtb.z), hit.distance);
            float dist2 = selectf(tminb, PT_FAR_PLANE, tminb > tmaxb);
            if (dist1 > dist2) {
                float h = dist1; dist1 = dist2; dist2 = h;
                uint32_t t = left; left = right; right = t;
            }
            if (dist1 == PT_FAR_PLANE) {
                if (stackPtr > 0) nodeIndex = stack[--stackPtr];
                else break;
            } else {
                nodeIndex = left;
                if (dist2 != PT_FAR_PLANE) {
                    if (stackPtr < PT_BVH_STACK_SIZE) stack[stackPtr] = right;
                    stackPtr++;
                }
            }
        }
        if (instanceCount > 0) {
            uint32_t firstInstance = pt_asuint(T[no + 15]);
            for (uint32_t i = 0; i < instanceCount; ++i) {
                uint32_t instanceIndex = pt_asuint(T[S.tlasIndexOffset + firstInstance + i]);
                bool stopNow = false;
                hitFound = RayIntersectBvhInstance(c, ray, S.instances[instanceIndex], isShadowRay, hit, stopNow) | hitFound;
                if (stopNow) return true;
            }
            if (stackPtr > 0) nodeIndex = stack[--stackPtr];
            else break;
        }
    }
    return hitFound;
}

// ------------------------------------------------------------------------------------------
// util/intersect.hlsl
// ------------------------------------------------------------------------------------------
inline float RectIntersect(f3 pos, f3 u, f3 v, f4 plane, const Ray& r)            // :6-27
{
    f3 n = V3(plane.x, plane.y, plane.z);
    float dt = dot(r.direction, n);
    float t = (plane.w - dot(n, r.origin)) / dt;
    float res = PT_FAR_PLANE;
    if (t > PT_EPSILON) {
        f3 p = r.origin + r.direction * t;
        f3 vi = p - pos;
        float a1 = dot(u, vi);
        if (a1 >= 0.0f && a1 <= 1.0f) {
            float a2 = dot(v, vi);
            if (a2 >= 0.0f && a2 <= 1.0f) res = t;
        }
    }
    return res;
}

inline f3 L3(const float* p) { return V3(p[0], p[1], p[2]); }

inline void IntersectLights(Ctx& c, const Ray& ray, RayHit& hit)                   // :29-54
{
    if (!c.S->hasLights) return;
    for (int i = 0; i < c.S->lightCount; ++i) {
        const PTLight& light = c.S->lights[i];
        c.st.lightFetches++;
        if (light.type == PT_LIGHT_TYPE_RECTANGLE) {
            f3 lu = L3(light.u), lv = L3(light.v), lp = L3(light.position);
            f3 normal = normalize(cross(lu, lv));
            f4 plane = {normal.x, normal.y, normal.z, dot(normal, lp)};
            f3 u = lu / dot(lu, lu);
            f3 v = lv / dot(lv, lv);
            float d = RectIntersect(lp, u, v, plane, ray);
            if (d > 0.0f && d < hit.distance && dot(normal, ray.direction) < 0.0f) {
                hit.distance = d;
                hit.position = ray.origin + d * ray.direction;
                hit.normal = normal;
                hit.ffnormal = dot(hit.normal, ray.direction) <= 0.0f ? hit.normal : -hit.normal;
                hit.triIndex = (uint32_t)i;
                hit.intersectType = INTERSECT_LIGHT;
            }
        }
    }
}

inline bool RayIntersect(Ctx& c, const Ray& ray, RayHit& hit)                      // util/bvh.hlsl:217-226
{
    hit.distance = PT_FAR_PLANE;
    c.st.closestHitRays++;
    if (c.S->hasTlas) RayIntersectTLAS(c, ray, hit, false);                     // util/tlas.hlsl:334-344
    else RayIntersectBvh(c, ray, hit, false);
    IntersectLights(c, ray, hit);
    return hit.distance < PT_FAR_PLANE;
}

inline bool ShadowRayIntersect(Ctx& c, const Ray& ray)                             // util/bvh.hlsl:228-233
{
    RayHit hit;
    memset(&hit, 0, sizeof(hit));
    hit.distance = PT_FAR_PLANE;
    c.st.shadowRays++;
    if (c.S->hasTlas) return RayIntersectTLAS(c, ray, hit, true);               // util/tlas.hlsl:346-352
    return RayIntersectBvh(c, ray, hit, true);
}

// ------------------------------------------------------------------------------------------
// util/texture.hlsl
// ------------------------------------------------------------------------------------------
inline f4 GetTexturePixel(Ctx& c, uint32_t off, uint32_t width, uint32_t height, uint32_t x, uint32_t y)   // :6-23
{
    x = x < width - 1u ? x : width - 1u;
    y = y < height - 1u ? y : height - 1u;
    uint32_t p = c.S->tex[off + (y * width + x)];
    c.st.texelFetches++;
    return {(float)(p & 0xFFu) / 255.0f, (float)((p >> 8) & 0xFFu) / 255.0f,
            (float)((p >> 16) & 0xFFu) / 255.0f, (float)((p >> 24) & 0xFFu) / 255.0f};
}

inline f4 SampleTexture(Ctx& c, int32_t textureIndex, f2 uv, bool linearSample)   // :25-76
{
    if (textureIndex < 0) return {0, 0, 0, 0};
    uint32_t d = (uint32_t)textureIndex * 4u;
    uint32_t width = c.S->tex[d + 0], height = c.S->tex[d + 1], offset = c.S->tex[d + 2];
    c.st.texDescriptorFetches++;
    float u = uv.x, v = uv.y;
    // :41-48, the literal loops.  They never terminate for |u| >= 2^24 or +-inf (u -+ 1 == u): such inputs are DEFINED
    // as 0 in this project (oracle and kernel; ptmi_math.h pt_wrap01 is the kernel's closed form of the same loops).
    if (pt_abs(u) >= 16777216.0f) u = 0.0f;
    if (pt_abs(v) >= 16777216.0f) v = 0.0f;
    while (u > 1.0f) u -= 1.0f;
    while (v > 1.0f) v -= 1.0f;
    while (u < 0.0f) u += 1.0f;
    while (v < 0.0f) v += 1.0f;
    float tu = u * ((float)width - 1.0f);
    float tv = v * ((float)height - 1.0f);
    uint32_t tx = pt_f2u(tu), ty = pt_f2u(tv);
    f4 p1 = GetTexturePixel(c, offset, width, height, tx, ty);
    if (!linearSample) return p1;
    float uF = tu - (float)tx, vF = tv - (float)ty;
    f4 p2 = GetTexturePixel(c, offset, width, height, tx + 1, ty);
    f4 p3 = GetTexturePixel(c, offset, width, height, tx, ty + 1);
    f4 p4 = GetTexturePixel(c, offset, width, height, tx + 1, ty + 1);
    return lerp4(lerp4(p1, p2, uF), lerp4(p3, p4, uF), vF);
}

// ------------------------------------------------------------------------------------------
// util/material.hlsl
// ------------------------------------------------------------------------------------------
Material GetMaterial(Ctx& c, const PTMaterialData& md, const Ray& ray, const RayHit& hit)   // :84-138
{
    const bool T = c.S->hasTextures;
    f2 uv = hit.uv;
    // GetBaseColorOpacity :56-69
    f4 bco = {md.baseColor[0], md.baseColor[1], md.baseColor[2], md.opacity};
    if (T && !(md.texBaseColor < 0.0f)) {
        f2 tuv = {uv.x * md.uvScale[0] + md.uvOffset[0], uv.y * md.uvScale[1] + md.uvOffset[1]};
        f4 px = SampleTexture(c, pt_f2i(md.texBaseColor), tuv, true);
        bco = px * bco;
    }
    Material m;
    m.baseColor = V3(bco.x, bco.y, bco.z);
    m.opacity = bco.w;
    m.alphaMode = md.alphaMode;
    m.alphaCutoff = md.alphaCutoff;
    // GetEmission :8-21
    m.emission = V3(md.emission[0], md.emission[1], md.emission[2]);
    if (T && !(md.texEmission < 0.0f)) {
        f4 px = SampleTexture(c, pt_f2i(md.texEmission), uv, true);
        m.emission = V3(px.x, px.y, px.z);
    }
    // GetMetallicRoughness :40-54
    float metallic = md.metallic, rough = md.roughness;
    if (T && !(md.texMetallicRoughness < 0.0f)) {
        f4 px = SampleTexture(c, pt_f2i(md.texMetallicRoughness), uv, true);
        metallic = px.z;
        rough = px.y * px.y;
    }
    m.metallic = metallic;
    m.roughness = pt_max(rough, 0.001f);
    m.subsurface = md.subsurface;
    m.specularTint = md.specularTint;
    m.sheen = md.sheen;
    m.sheenTint = md.sheenTint;
    m.clearcoat = md.clearCoat;
    m.clearcoatRoughness = pt_lerp(0.1f, 0.001f, md.clearCoatGloss);
    m.specTrans = 1.0f - pt_saturate(bco.w);
    m.ior = pt_clamp(md.ior, 1.001f, 2.0f);
    m.anisotropic = pt_clamp(md.anisotropic, -0.9f, 0.9f);
    // GetOcclusion :71-82
    m.occlusion = 1.0f;
    if (T && !(md.texOcclusion < 0.0f)) {
        float px = SampleTexture(c, pt_f2i(md.texOcclusion), uv, true).x;
        m.occlusion = 1.0f + (px - 1.0f);
    }
    float aspect = pt_sqrt(1.0f - m.anisotropic * 0.9f);
    m.ax = pt_max(0.001f, m.roughness / aspect);
    m.ay = pt_max(0.001f, m.roughness * aspect);
    m.eta = (dot(ray.direction, hit.normal) < 0.0f) ? 1.0f / m.ior : m.ior;
    return m;
}

// ------------------------------------------------------------------------------------------
// util/sampling.hlsl
// ------------------------------------------------------------------------------------------
inline float GTR1(float NDotH, float a)                                            // :6-18
{
    if (a >= 1.0f) return PT_INV_PI;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (PT_PI * pt_log(a2) * t);
}
inline f3 SampleGTR1(float rgh, float r1, float r2)                                // :20-33
{
    float a = pt_max(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * PT_TWO_PI;
    float cosTheta = pt_sqrt((1.0f - pt_pow(a2, 1.0f - r2)) / (1.0f - a2));
    float sinTheta = pt_clamp(pt_sqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
    float sinPhi = pt_sin(phi);
    float cosPhi = pt_cos(phi);
    return V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
inline f3 SampleGGXVNDF(f3 V, float ax, float ay, float r1, float r2)              // :56-74
{
    f3 Vh = normalize(V3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    f3 T1 = lensq > 0.0f ? V3(-Vh.y, Vh.x, 0.0f) * pt_rsqrt(lensq) : V3(1, 0, 0);
    f3 T2 = cross(Vh, T1);
    float r = pt_sqrt(r1);
    float phi = 2.0f * PT_PI * r2;
    float t1 = r * pt_cos(phi);
    float t2 = r * pt_sin(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * pt_sqrt(1.0f - t1 * t1) + s * t2;
    f3 Nh = t1 * T1 + t2 * T2 + pt_sqrt(pt_max(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(V3(ax * Nh.x, ay * Nh.y, pt_max(0.0f, Nh.z)));
}
inline float GTR2Aniso(float NDotH, float HDotX, float HDotY, float ax, float ay)  // :76-82
{
    float a = HDotX / ax;
    float b = HDotY / ay;
    float c = a * a + b * b + NDotH * NDotH;
    return 1.0f / (PT_PI * ax * ay * c * c);
}
inline float SmithG(float NDotV, float alphaG)                                     // :95-100
{
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return (2.0f * NDotV) / (NDotV + pt_sqrt(a + b - a * b));
}
inline float SmithGAniso(float NDotV, float VDotX, float VDotY, float ax, float ay) // :102-108
{
    float a = VDotX * ax;
    float b = VDotY * ay;
    float c = NDotV;
    return (2.0f * NDotV) / (NDotV + pt_sqrt(a * a + b * b + c * c));
}
inline float SchlickWeight(float u)                                                // :110-115
{
    float m = pt_clamp(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
inline float DielectricFresnel(float cosThetaI, float eta)                         // :117-135
{
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = pt_sqrt(pt_max(1.0f - sinThetaTSq, 0.0f));
    float rs = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    float rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}
inline f3 CosineSampleHemisphere(float r1, float r2)                               // :137-146
{
    float r = pt_sqrt(r1);
    float phi = PT_TWO_PI * r2;
    f3 d;
    d.x = r * pt_cos(phi);
    d.y = r * pt_sin(phi);
    d.z = pt_sqrt(pt_max(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return d;
}
inline float PowerHeuristic(float a, float b) { float t = a * a; return t / (b * b + t); }   // :163-167

// ------------------------------------------------------------------------------------------
// util/brdf.hlsl
// ------------------------------------------------------------------------------------------
inline void TintColors(const Material& mat, float eta, float& F0, f3& Csheen, f3& Cspec0)   // :9-23
{
    float lum = Luminance(mat.baseColor);
    f3 ctint = lum > 0.0f ? mat.baseColor / lum : V3(1.0f);
    F0 = (1.0f - eta) / (1.0f + eta);
    F0 *= F0;
    Cspec0 = F0 * lerp3(V3(1.0f), ctint, mat.specularTint);
    Csheen = lerp3(V3(1.0f), ctint, mat.sheenTint);
}

inline f3 EvalDiffuse(const Material& mat, f3 Csheen, f3 V, f3 L, f3 H, float& pdf)          // :25-54
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float LDotH = dot(L, H);
    float Rr = 2.0f * mat.roughness * LDotH * LDotH;
    float FL = SchlickWeight(L.z);
    float FV = SchlickWeight(V.z);
    float Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0f));
    float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
    float Fss90 = 0.5f * Rr;
    float Fss = pt_lerp(1.0f, Fss90, FL) * pt_lerp(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (L.z + V.z) - 0.5f) + 0.5f);
    float FH = SchlickWeight(LDotH);
    f3 Fsheen = FH * mat.sheen * Csheen;
    pdf = L.z * PT_INV_PI;
    return PT_INV_PI * mat.baseColor * pt_lerp(Fd + Fretro, ss, mat.subsurface) + Fsheen;
}

inline f3 EvalMicrofacetReflection(const Material& mat, f3 V, f3 L, f3 H, f3 F, float& pdf)  // :56-71
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float D = GTR2Aniso(H.z, H.x, H.y, mat.ax, mat.ay);
    float G1 = SmithGAniso(pt_abs(V.z), V.x, V.y, mat.ax, mat.ay);
    float G2 = G1 * SmithGAniso(pt_abs(L.z), L.x, L.y, mat.ax, mat.ay);
    pdf = G1 * D / (4.0f * V.z);
    return F * D * G2 / (4.0f * L.z * V.z);
}

inline f3 EvalMicrofacetRefraction(const Material& mat, float eta, f3 V, f3 L, f3 H, f3 F, float& pdf)   // :73-93
{
    pdf = 0.0f;
    if (L.z >= 0.0f) return V3(0.0f);
    float LDotH = dot(L, H);
    float VDotH = dot(V, H);
    float D = GTR2Aniso(H.z, H.x, H.y, mat.ax, mat.ay);
    float G1 = SmithGAniso(pt_abs(V.z), V.x, V.y, mat.ax, mat.ay);
    float G2 = G1 * SmithGAniso(pt_abs(L.z), L.x, L.y, mat.ax, mat.ay);
    float denom = LDotH + VDotH * eta;
    denom *= denom;
    float eta2 = eta * eta;
    float jacobian = pt_abs(LDotH) / denom;
    pdf = G1 * pt_max(0.0f, VDotH) * D * jacobian / V.z;
    f3 sq = V3(pt_pow(mat.baseColor.x, 0.5f), pt_pow(mat.baseColor.y, 0.5f), pt_pow(mat.baseColor.z, 0.5f));
    return sq * (1.0f - F) * D * G2 * pt_abs(VDotH) * jacobian * eta2 / pt_abs(L.z * V.z);
}

inline f3 EvalClearcoat(const Material& mat, f3 V, f3 L, f3 H, float& pdf)                    // :95-112
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float VDotH = dot(V, H);
    float F = pt_lerp(0.04f, 1.0f, SchlickWeight(VDotH));
    float D = GTR1(H.z, mat.clearcoatRoughness);
    float G = SmithG(L.z, 0.25f) * SmithG(V.z, 0.25f);
    float jacobian = 1.0f / (4.0f * VDotH);
    pdf = D * H.z * jacobian;
    return V3(F) * D * G;
}

struct LobeWeights { float dielectricWt, metalWt, glassWt, diffPr, dielectricPr, metalPr, glassPr, clearCtPr; };

inline LobeWeights ComputeLobes(const Material& mat, f3 Cspec0, float Vz)          // :140-161 == :260-281
{
    LobeWeights w;
    w.dielectricWt = (1.0f - mat.metallic) * (1.0f - mat.specTrans);
    w.metalWt = mat.metallic;
    w.glassWt = (1.0f - mat.metallic) * mat.specTrans;
    float schlickWt = SchlickWeight(Vz);
    w.diffPr = w.dielectricWt * Luminance(mat.baseColor);
    w.dielectricPr = w.dielectricWt * Luminance(lerp3(Cspec0, V3(1.0f), schlickWt));
    w.metalPr = w.metalWt * Luminance(lerp3(mat.baseColor, V3(1.0f), schlickWt));
    w.glassPr = w.glassWt;
    w.clearCtPr = 0.25f * mat.clearcoat;
    float invTotalWt = 1.0f / (w.diffPr + w.dielectricPr + w.metalPr + w.glassPr + w.clearCtPr);
    w.diffPr *= invTotalWt;
    w.dielectricPr *= invTotalWt;
    w.metalPr *= invTotalWt;
    w.glassPr *= invTotalWt;
    w.clearCtPr *= invTotalWt;
    return w;
}

f3 _EvalBRDF(const Material& mat, f3 V, f3 N, f3 L, const Mat3& onb, float& pdf)              // :114-225
{
    (void)N;
    pdf = 0.0f;
    f3 f = V3(0.0f);
    V = ToLocal(onb, V);
    L = ToLocal(onb, L);
    f3 H;
    if (L.z > 0.0f) H = normalize(L + V);
    else H = normalize(L + V * mat.eta);
    if (H.z < 0.0f) H = -H;

    f3 Csheen, Cspec0;
    float F0;
    TintColors(mat, mat.eta, F0, Csheen, Cspec0);
    LobeWeights w = ComputeLobes(mat, Cspec0, V.z);

    bool refl = L.z * V.z > 0.0f;
    float tmpPdf = 0.0f;
    float VDotH = pt_abs(dot(V, H));

    if (w.diffPr > 0.0f && refl) {
        f += EvalDiffuse(mat, Csheen, V, L, H, tmpPdf) * w.dielectricWt;
        pdf += tmpPdf * w.diffPr;
    }
    if (w.dielectricPr > 0.0f && refl) {
        float F = 0.0f;
        if (F0 != 1.0f && mat.ior != 0.0f) {
            float invEta = pt_rcp(mat.ior);
            float invF0 = 1.0f - F0;
            invF0 = pt_rcp(invF0);
            F = (DielectricFresnel(VDotH, invEta) - F0) * invF0;
        }
        f += EvalMicrofacetReflection(mat, V, L, H, lerp3(Cspec0, V3(1.0f), F), tmpPdf) * w.dielectricWt;
        pdf += tmpPdf * w.dielectricPr;
    }
    if (w.metalPr > 0.0f && refl) {
        f3 F = lerp3(mat.baseColor, V3(1.0f), SchlickWeight(VDotH));
        f += EvalMicrofacetReflection(mat, V, L, H, F, tmpPdf) * w.metalWt;
        pdf += tmpPdf * w.metalPr;
    }
    if (w.glassPr > 0.0f) {
        float F = DielectricFresnel(VDotH, mat.eta);
        if (refl) {
            f += EvalMicrofacetReflection(mat, V, L, H, V3(F), tmpPdf) * w.glassWt;
            pdf += tmpPdf * w.glassPr * F;
        } else {
            f += EvalMicrofacetRefraction(mat, mat.eta, V, L, H, V3(F), tmpPdf) * w.glassWt;
            pdf += tmpPdf * w.glassPr * (1.0f - F);
        }
    }
    if (w.clearCtPr > 0.0f && refl) {
        f += EvalClearcoat(mat, V, L, H, tmpPdf) * 0.25f * mat.clearcoat;
        pdf += tmpPdf * w.clearCtPr;
    }
    f *= mat.occlusion;
    return f * pt_abs(L.z);
}

inline f3 EvalBRDF(const Material& mat, f3 V, f3 N, f3 L, float& pdf)                         // :227-238
{
    Mat3 onb = GetONB(N);
    return _EvalBRDF(mat, V, N, L, onb, pdf);
}

f3 SampleBRDF(const Material& mat, f3 V, f3 N, f3& L, float& pdf, uint32_t& rng)              // :240-340
{
    pdf = 0.0f;
    float r1 = RandomFloat(rng);
    float r2 = RandomFloat(rng);
    Mat3 onb = GetONB(N);
    V = ToLocal(onb, V);

    f3 Csheen, Cspec0;
    float F0;
    TintColors(mat, mat.eta, F0, Csheen, Cspec0);
    LobeWeights w = ComputeLobes(mat, Cspec0, V.z);

    float cdf0 = w.diffPr;
    float cdf1 = cdf0 + w.dielectricPr;
    float cdf2 = cdf1 + w.metalPr;
    float cdf3 = cdf2 + w.glassPr;
    float r3 = RandomFloat(rng);

    if (r3 < cdf0) {
        L = CosineSampleHemisphere(r1, r2);
    } else if (r3 < cdf2) {
        f3 H = SampleGGXVNDF(V, mat.ax, mat.ay, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-V, H));
    } else if (r3 < cdf3) {
        f3 H = SampleGGXVNDF(V, mat.ax, mat.ay, r1, r2);
        float F = DielectricFresnel(pt_abs(dot(V, H)), mat.eta);
        if (H.z < 0.0f) H = -H;
        r3 = (r3 - cdf2) / (cdf3 - cdf2);
        if (r3 < F) L = normalize(reflect(-V, H));
        else L = normalize(refract(-V, H, mat.eta));
    } else {
        f3 H = SampleGTR1(mat.clearcoatRoughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-V, H));
    }
    L = ToWorld(onb, L);
    V = ToWorld(onb, V);
    return _EvalBRDF(mat, V, N, L, onb, pdf);
}

// ------------------------------------------------------------------------------------------
// util/random.hlsl:34-41, util/sky.hlsl:90-129
// ------------------------------------------------------------------------------------------
inline f3 RandomCosineHemisphere(f3 normal, uint32_t& rng)
{
    float theta = pt_acos(pt_sqrt(RandomFloat(rng)));
    float phi = 2.0f * PT_PI * RandomFloat(rng);
    Mat3 onb = GetONB(normal);
    return pt_sin(theta) * (pt_cos(phi) * onb.r0 + pt_sin(phi) * onb.r1 + pt_cos(theta) * onb.r2);
}

// Texture2D.SampleLevel(sampler, uv, 0) for the environment copy (an ARGBFloat RenderTexture: bilinear filter, clamp
// wrap).  Restated in fp32: texel centres at (i + 0.5)/size, weights = fractional parts, a + t (b - a) blends, x first.
// Array row r (the order of the readback `data[i]`) sits at v = 1 - (r + 0.5)/H.  The hardware's fixed-point filter
// weights and Unity's per-platform row order are not restated: parity unpinned here (oracle/pt_oracle.h).
inline f3 EnvSampleLevel(const SceneView& S, f2 uv)
{
    const int32_t W = S.EnvironmentTextureWidth, H = S.EnvironmentTextureHeight;
    float tx = uv.x * (float)W - 0.5f;
    float ty = (1.0f - uv.y) * (float)H - 0.5f;
    float bx = pt_floor(tx), by = pt_floor(ty);
    float wx = tx - bx, wy = ty - by;
    auto clampi = [](int32_t v, int32_t hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    int32_t ix = pt_f2i(bx), iy = pt_f2i(by);
    int32_t ix1 = ix < 2147483647 ? ix + 1 : ix, iy1 = iy < 2147483647 ? iy + 1 : iy;
    ix = clampi(ix, W - 1); ix1 = clampi(ix1, W - 1); iy = clampi(iy, H - 1); iy1 = clampi(iy1, H - 1);
    auto texel = [&](int32_t x, int32_t y) { const PTFloat4& t = S.envTexture[(size_t)y * W + x]; return V3(t.x, t.y, t.z); };
    f3 t00 = texel(ix, iy), t10 = texel(ix1, iy), t01 = texel(ix, iy1), t11 = texel(ix1, iy1);
    f3 rowA = V3(pt_lerp(t00.x, t10.x, wx), pt_lerp(t00.y, t10.y, wx), pt_lerp(t00.z, t10.z, wx));
    f3 rowB = V3(pt_lerp(t01.x, t11.x, wx), pt_lerp(t01.y, t11.y, wx), pt_lerp(t01.z, t11.z, wx));
    return V3(pt_lerp(rowA.x, rowB.x, wy), pt_lerp(rowA.y, rowB.y, wy), pt_lerp(rowA.z, rowB.z, wy));
}

inline f2 BinarySearch(const SceneView& S, float value)                              // util/sky.hlsl:7-42
{
    const int32_t W = S.EnvironmentTextureWidth, H = S.EnvironmentTextureHeight;
    const float* EnvironmentCDF = S.EnvironmentCDF.data();
    int lower = 0;
    int upper = H - 1;
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        int idx = mid * W + W - 1;
        if (value < EnvironmentCDF[idx]) upper = mid;
        else lower = mid + 1;
    }
    int y = lower < 0 ? 0 : (lower > H - 1 ? H - 1 : lower);
    lower = 0;
    upper = W - 1;
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        int idx = y * W + mid;
        if (value < EnvironmentCDF[idx]) upper = mid;
        else lower = mid + 1;
    }
    int x = lower < 0 ? 0 : (lower > W - 1 ? W - 1 : lower);
    return {(float)x / (float)W, (float)y / (float)H};
}

inline f4 EvalEnvMap(const Ctx& c, f3 r, float intensity)                             // util/sky.hlsl:44-64
{
    const SceneView& S = *c.S;
    float theta = pt_acos(pt_clamp(r.y, -1.0f, 1.0f));
    float r_atan = pt_atan2(r.z, r.x);
    f2 uv = f2{(PT_PI + r_atan) * PT_INV_TWO_PI, 1.0f - theta * PT_INV_PI} + f2{c.P->EnvironmentMapRotation, 0.0f};
    uv.x = pt_fmod(uv.x, 1.0f);
    uv.y = pt_fmod(uv.y, 1.0f);
    if (uv.x < 0.0f) uv.x += 1.0f;
    if (uv.y < 0.0f) uv.y += 1.0f;
    f3 color = EnvSampleLevel(S, uv);
    float pdf = Luminance(color) / S.EnvironmentCdfSum;
    pdf = (pdf * (float)S.EnvironmentTextureWidth * (float)S.EnvironmentTextureHeight) / (PT_TWO_PI * PT_PI * pt_sin(theta));
    f3 ci = color * intensity;
    return {ci.x, ci.y, ci.z, pdf};
}

inline f4 SampleEnvMap(const Ctx& c, f3& color, uint32_t& rngState)                   // util/sky.hlsl:66-88
{
    const SceneView& S = *c.S;
    float rnd = RandomFloat(rngState) * S.EnvironmentCdfSum;
    f2 uv = BinarySearch(S, rnd);
    uv.y = 1.0f - uv.y;
    color = EnvSampleLevel(S, uv);
    float pdf = Luminance(color) / S.EnvironmentCdfSum;
    uv.x -= c.P->EnvironmentMapRotation;
    float phi = uv.x * PT_TWO_PI;
    float theta = uv.y * PT_PI;
    float sinTheta = pt_sin(theta);
    if (sinTheta == 0.0f) pdf = 0.0f;
    return {-sinTheta * pt_cos(phi), pt_cos(theta), -sinTheta * pt_sin(phi),
            (pdf * (float)S.EnvironmentTextureWidth * (float)S.EnvironmentTextureHeight) / (PT_TWO_PI * PT_PI * sinTheta)};
}

inline f4 SampleSkyRadiance(const Ctx& c, f3 direction, uint32_t rayDepth)            // util/sky.hlsl:90-129
{
    const PTFrameParams& P = *c.P;
    f4 radiance = {0, 0, 0, 0};
    float pdf = 1.0f / (4.0f * PT_PI);
    float intensity = 1.0f;
    if ((int32_t)rayDepth > 0) intensity = P.EnvironmentIntensity;     // SampleSkyRadiance(float3, int rayDepth)
    if (P.EnvironmentMode == PT_SKY_MODE_ENVIRONMENT) {
        if (c.S->hasEnvironmentTexture) return EvalEnvMap(c, direction, intensity);   // EnvironmentSky :90-98
        f3 col = V3(P.EnvironmentColor[0], P.EnvironmentColor[1], P.EnvironmentColor[2]) * intensity;
        radiance = {col.x, col.y, col.z, pdf};
    } else if (P.EnvironmentMode == PT_SKY_MODE_BASIC) {
        float a = pt_saturate(0.5f * (direction.y + 1.0f));
        f3 blue = V3(pt_pow(0.5f, 2.2f), pt_pow(0.7f, 2.2f), pt_pow(1.0f, 2.2f));
        f3 col = (1.0f - a) * V3(1.0f) + a * blue;
        col = col * intensity;
        radiance = {col.x, col.y, col.z, pdf};
    }
    return radiance;
}

// ------------------------------------------------------------------------------------------
// util/light.hlsl
// ------------------------------------------------------------------------------------------
struct LightSampleRec { f3 normal; float pdf; f3 emission; float distance; f3 direction; };

f3 EvalLight(Ctx& c, const Ray& ray, const RayHit& hit, const Material& mat, const PTLight& light,
             f3 scatterPos, const LightSampleRec& ls)                                          // :60-114
{
    float falloff = 1.0f;
    if (ls.distance > light.range) falloff = 0.0f;
    else {
        float r = ls.distance / light.range;
        float atten = pt_saturate(1.0f / (1.0f + 25.0f * r * r) * pt_saturate((1.0f - r) * 5.0f));
        falloff *= atten;
    }
    if (light.type == PT_LIGHT_TYPE_RECTANGLE) {
        float cosTheta = dot(normalize(-ls.direction), normalize(ls.normal));
        falloff = cosTheta < 0.0f ? 0.0f : falloff;
    }
    if (light.type == PT_LIGHT_TYPE_SPOT) {
        float cosTheta = dot(normalize(-ls.direction), normalize(ls.normal));
        if (cosTheta < light.v[0]) falloff = 0.0f;
        else if (cosTheta > light.v[0] && cosTheta < light.v[1]) falloff *= (cosTheta - light.v[0]) / (light.v[1] - light.v[0]);
    }
    f3 Li = L3(light.emission) * falloff;
    f3 Ld = V3(0.0f);
    Ray shadowRay = {scatterPos, ls.direction};
    bool inShadow = ShadowRayIntersect(c, shadowRay);
    if (!inShadow) {
        float pdf = 0.0f;
        f3 f = EvalBRDF(mat, -ray.direction, hit.normal, ls.direction, pdf);
        float lightPdf = 1.0f;
        if (ls.pdf > 0.0f) lightPdf = ls.pdf;
        f3 L = Li * f / lightPdf;
        Ld += L;
    }
    return Ld;
}

f3 DirectLight(Ctx& c, const Ray& ray, const RayHit& hit, const Material& mat, uint32_t& rng)   // :117-173
{
    const PTFrameParams& P = *c.P;
    f3 Ld = V3(0.0f);
    f3 scatterPos = hit.position + hit.normal * PT_EPSILON;

    if (P.EnvironmentMode == 0) {
        if (c.S->hasEnvironmentTexture) {                                              // :125-141
            f3 Li = V3(0.0f);
            f4 dirPdf = SampleEnvMap(c, Li, rng);
            f3 lightDir = V3(dirPdf.x, dirPdf.y, dirPdf.z);
            float lightPdf = dirPdf.w;
            Ray shadowRay = {scatterPos, lightDir};
            bool inShadow = ShadowRayIntersect(c, shadowRay);
            if (!inShadow) {
                float spdf = 0.0f;
                f3 sf = EvalBRDF(mat, -ray.direction, hit.ffnormal, lightDir, spdf);
                if (spdf > 0.0f) {
                    float misWeight = PowerHeuristic(lightPdf, spdf);
                    if (misWeight > 0.0f) Ld += misWeight * Li * sf * P.EnvironmentIntensity / lightPdf;
                }
            }
        } else {                                                                       // :143-157
            f3 Li = V3(P.EnvironmentColor[0], P.EnvironmentColor[1], P.EnvironmentColor[2]) * P.EnvironmentIntensity;
            float lightPdf = 1.0f / (4.0f * PT_PI);
            f3 lightDir = normalize(RandomCosineHemisphere(hit.normal, rng));
            Ray shadowRay = {scatterPos, lightDir};
            bool inShadow = ShadowRayIntersect(c, shadowRay);
            if (!inShadow) {
                float spdf = 0.0f;
                f3 sf = EvalBRDF(mat, -ray.direction, hit.ffnormal, lightDir, spdf);
                if (spdf > 0.0f) {
                    float misWeight = PowerHeuristic(lightPdf, spdf);
                    if (misWeight > 0.0f) Ld += misWeight * Li * sf / lightPdf;
                }
            }
        }
    }

    if (c.S->hasLights) {
        LightSampleRec ls;
        memset(&ls, 0, sizeof(ls));
        int32_t lightIndex = pt_f2i(RandomFloat(rng) * (float)c.S->lightCount);
        if (lightIndex > c.S->lightCount - 1) lightIndex = c.S->lightCount - 1;       // r == 1.0
        const PTLight& light = c.S->lights[lightIndex];
        c.st.lightFetches++;
        bool ok = false;
        f3 lp = L3(light.position);
        if (light.type == PT_LIGHT_TYPE_SPOT) {                                        // :36-45
            ls.normal = normalize(L3(light.u));
            ls.emission = L3(light.emission);
            ls.direction = -normalize(scatterPos - lp);
            ls.distance = length(lp - scatterPos);
            ls.pdf = 0.0f;
            ok = true;
        } else if (light.type == PT_LIGHT_TYPE_RECTANGLE) {                            // :7-23
            float r1 = RandomFloat(rng);
            float r2 = RandomFloat(rng);
            f3 lu = L3(light.u), lv = L3(light.v);
            f3 lightSurfacePos = lp + lu * r1 + lv * r2;
            ls.direction = lightSurfacePos - scatterPos;
            ls.distance = length(ls.direction);
            float distSq = ls.distance * ls.distance;
            ls.direction /= ls.distance;
            ls.normal = normalize(cross(lu, lv));
            ls.emission = L3(light.emission) * (float)c.S->lightCount;                 // computed, never used (:19 vs :96)
            ls.pdf = distSq / (light.area * pt_abs(dot(ls.normal, ls.direction)));
            ok = true;
        } else if (light.type == PT_LIGHT_TYPE_POINT) {                                // :25-34
            ls.normal = normalize(scatterPos - lp);
            ls.emission = L3(light.emission);
            ls.direction = -ls.normal;
            ls.distance = length(scatterPos - lp);
            ls.pdf = 0.0f;
            ok = true;
        }
        if (ok) Ld += EvalLight(c, ray, hit, mat, light, scatterPos, ls);
    }
    return Ld;
}

// ------------------------------------------------------------------------------------------
// util/pathtrace.hlsl:10-131
// ------------------------------------------------------------------------------------------
f3 PathTrace(Ctx& c, Ray ray, uint32_t& rng)
{
    const PTFrameParams& P = *c.P;
    f3 radiance = V3(0.0f);
    f3 throughput = V3(1.0f);
    f3 scatterL = V3(0.0f), scatterF = V3(0.0f);
    float scatterPdf = 0.0f;
    const uint32_t maxRayBounces = P.MaxRayBounces > 1u ? P.MaxRayBounces : 1u;
    RayHit hit;
    memset(&hit, 0, sizeof(hit));
    float maxRoughness = 0.0f;

    for (uint32_t rayDepth = 0;; ++rayDepth) {
        bool didHit = RayIntersect(c, ray, hit);
        if (!didHit) {
            f4 sky = SampleSkyRadiance(c, ray.direction, rayDepth);
            float misWeight = 1.0f;
            if (rayDepth > 0) misWeight = PowerHeuristic(scatterPdf, sky.w);
            if (misWeight > 0.0f) radiance += misWeight * V3(sky.x, sky.y, sky.z) * throughput;
            break;
        }
        if (c.S->hasLights && hit.intersectType == INTERSECT_LIGHT) {
            const PTLight& light = c.S->lights[hit.triIndex];
            c.st.lightFetches++;
            radiance += L3(light.emission) * throughput;
            break;
        }
        c.st.materialFetches++;
        Material material = GetMaterial(c, c.S->materials[hit.materialIndex], ray, hit);
        maxRoughness = pt_max(maxRoughness, material.roughness);
        material.roughness = maxRoughness;

        radiance += material.emission * throughput;

        if (rayDepth >= maxRayBounces) break;

        // :84-85 is one boolean expression.  HLSL before HLSL 2021 does NOT short-circuit && / || (they are
        // component-wise operators: every operand is evaluated, side effects included).  FXC is in the reference's
        // pipeline (its own workaround comment, util/bvh.hlsl:133-135) and PathTracer.compute has no `#pragma use_dxc`
        // / -HV 2021, so RandomFloat(rngState) is drawn on EVERY bounce that reaches this line -- for OPAQUE and MASK
        // materials too -- and its value is only USED for BLEND.  (Nothing here can execute HLSL: parity unpinned.)
        const float alphaRnd = RandomFloat(rng);
        if ((material.alphaMode == (float)PT_ALPHA_MODE_MASK && material.opacity < material.alphaCutoff) ||
            (material.alphaMode == (float)PT_ALPHA_MODE_BLEND && alphaRnd > material.opacity)) {
            scatterL = ray.direction;
            rayDepth--;                                   // uint wrap at 0, undone by the loop's ++ (:88)
        } else {
            radiance += DirectLight(c, ray, hit, material, rng) * throughput;
            scatterF = SampleBRDF(material, -ray.direction, hit.ffnormal, scatterL, scatterPdf, rng);
            if (pt_isnan(scatterF.x) || pt_isnan(scatterF.y) || pt_isnan(scatterF.z)) {
                radiance = V3(0.0f, 1.0f, 0.0f);
                break;
            }
            if (scatterPdf > 0.0f) throughput *= scatterF / scatterPdf;
            else break;
        }
        ray.direction = scatterL;
        ray.origin = hit.position + ray.direction * PT_EPSILON;

        if (P.UseRussianRoulette) {
            float rrPcont = pt_min(pt_max(throughput.x, pt_max(throughput.y, throughput.z)) + 0.001f, 0.95f);
            if (RandomFloat(rng) >= rrPcont) break;
            throughput /= rrPcont;
        }
    }
    return radiance;
}

// ------------------------------------------------------------------------------------------
// util/common.hlsl:285-341, util/camera.hlsl:13-42
// ------------------------------------------------------------------------------------------
inline void ConcentricSampleDisk(float u1, float u2, float& dx, float& dy)
{
    float sx = 2.0f * u1 - 1.0f;
    float sy = 2.0f * u2 - 1.0f;
    if (sx == 0.0f && sy == 0.0f) { dx = 0.0f; dy = 0.0f; return; }
    float r, theta;
    if (sx >= -sy) {
        if (sx > sy) { r = sx; theta = sy > 0.0f ? sy / r : 8.0f + sy / r; }
        else { r = sy; theta = 2.0f - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
        else { r = -sy; theta = 6.0f + sx / r; }
    }
    theta *= PT_PI / 4.0f;
    dx = r * pt_cos(theta);
    dy = r * pt_sin(theta);
}

inline Ray GetScreenRay(const Ctx& c, f2 pixelCoords, uint32_t& rng)
{
    const PTFrameParams& P = *c.P;
    f4 o4 = mul44(P.CamToWorld, f4{0.0f, 0.0f, 0.0f, 1.0f});
    f3 origin = V3(o4.x, o4.y, o4.z);
    f2 uv = {pixelCoords.x / (float)P.OutputWidth * 2.0f - 1.0f, pixelCoords.y / (float)P.OutputHeight * 2.0f - 1.0f};
    f4 d4 = mul44(P.CamInvProj, f4{uv.x, uv.y, 0.0f, 1.0f});
    f4 w4 = mul44(P.CamToWorld, f4{d4.x, d4.y, d4.z, 0.0f});
    f3 direction = normalize(V3(w4.x, w4.y, w4.z));
    if (P.Aperture > 0.0f && P.FocalLength > 0.0f) {
        float sampleLensU = RandomFloat(rng);
        float sampleLensV = RandomFloat(rng);
        float lensU, lensV;
        ConcentricSampleDisk(sampleLensU, sampleLensV, lensU, lensV);
        float lensRadius = P.Aperture * 0.5f;
        lensU *= lensRadius;
        lensV *= lensRadius;
        float ft = P.FocalLength;
        f3 focalPoint = origin + direction * ft;
        f4 l4 = mul44(P.CamToWorld, f4{lensU, lensV, 0.0f, 1.0f});
        origin = V3(l4.x, l4.y, l4.z);
        direction = normalize(focalPoint - origin);
    }
    return {origin, direction};
}

// PathTracer.compute:33-38
inline f2 SampleGaussian(float u, float v)
{
    float r = pt_sqrt(-2.0f * pt_log(pt_max(1e-38f, u)));
    float theta = 2.0f * PT_PI * v;
    return {r * pt_cos(theta), r * pt_sin(theta)};
}

// PathTracer.compute:45-100 for one pixel
void RenderPixel(Ctx& c, uint32_t pixelX, uint32_t pixelY, const float* accumulated, float* output)
{
    const PTFrameParams& P = *c.P;
    const uint32_t pixelIndex = pixelY * P.OutputWidth + pixelX;
    f2 pixelCoords = {(float)pixelX, (float)pixelY};
    const int numSamples = P.SamplesPerPass > 1 ? P.SamplesPerPass : 1;
    const float fSamples = (float)numSamples;
    uint32_t rng = pixelIndex * (P.CurrentSample + 1u) + P.RngSeedRoot;
    f3 color = V3(0.0f);
    for (int s = 0; s < numSamples; ++s) {
        c.st.paths++;
        float g0 = RandomFloat(rng);
        float g1 = RandomFloat(rng);
        f2 g = SampleGaussian(g0, g1);
        f2 sub = {0.5f + PT_AA_SIGMA * g.x, 0.5f + PT_AA_SIGMA * g.y};
        f2 pcs = pixelCoords + sub;
        Ray ray = GetScreenRay(c, pcs, rng);
        f3 radiance = PathTrace(c, ray, rng);
        if (P.UseFireflyFilter) {
            float lum = Luminance(radiance);
            if (lum > P.MaxFireflyLuminance) radiance *= P.MaxFireflyLuminance / lum;
        }
        color += radiance;
    }
    float* o = output + (size_t)pixelIndex * 4;
    if (P.CurrentSample > 0) {
        const float* a = accumulated + (size_t)pixelIndex * 4;
        c.st.pixelsRead++;
        float cs = (float)P.CurrentSample;
        f3 acc = (color + V3(a[0], a[1], a[2]) * cs) / (cs + fSamples);
        o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = 1.0f;
    } else {
        f3 acc = color / fSamples;
        o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = 1.0f;
    }
    c.st.pixelsWritten++;
}

SceneView MakeView(const PTSceneDesc* s)
{
    SceneView v;
    v.nodes = (const PTCwbvhNode*)s->bvhNodes; v.nodeCount = s->bvhNodesBytes / 80;
    v.tris = (const PTFloat4*)s->bvhTris; v.triFloat4s = s->bvhTrisBytes / 16;
    v.attrs = (const PTTriangleAttributes*)s->triAttrs; v.attrCount = s->triAttrsBytes / 128;
    v.materials = (const PTMaterialData*)s->materials; v.materialCount = s->materialCount;
    v.lights = (const PTLight*)s->lights; v.lightCount = (int32_t)s->lightCount;
    v.tex = s->textureData; v.texUints = s->textureDataUints;
    v.hasLights = (s->features & PT_FEATURE_HAS_LIGHTS) && s->lightCount > 0;
    v.hasTextures = (s->features & PT_FEATURE_HAS_TEXTURES) && s->textureDataUints > 0;
    v.hasTlas = (s->features & PT_FEATURE_HAS_TLAS) != 0;
    v.tlas = s->tlasData; v.tlasFloats = s->tlasDataFloats; v.tlasIndexOffset = s->tlasIndexOffset;
    v.instances = (const PTGpuInstance*)s->gpuInstances; v.instanceCount = s->instanceCount;
    v.hasEnvironmentTexture = (s->features & PT_FEATURE_HAS_ENVIRONMENT_TEXTURE) != 0;
    v.envTexture = (const PTFloat4*)s->envTexture;
    v.EnvironmentTextureWidth = (int32_t)s->envWidth; v.EnvironmentTextureHeight = (int32_t)s->envHeight;
    v.EnvironmentCdfSum = 0.0f;
    if (v.hasEnvironmentTexture) {
        // PathTracer.cs:297-306 (OnEnvTexReadback): cdf[i] = sum of data[0..i].grayscale, float accumulator;
        // UnityEngine.Color.grayscale = 0.299f * r + 0.587f * g + 0.114f * b
        const size_t n = (size_t)s->envWidth * s->envHeight;
        v.EnvironmentCDF.resize(n);
        float sum = 0.0f;
        for (size_t i = 0; i < n; i++) {
            const PTFloat4& d = v.envTexture[i];
            sum += 0.299f * d.x + 0.587f * d.y + 0.114f * d.z;
            v.EnvironmentCDF[i] = sum;
        }
        v.EnvironmentCdfSum = sum;
    }
    return v;
}

void AddStats(PTStats& a, const PTStats& b)
{
    uint64_t* pa = (uint64_t*)&a;
    const uint64_t* pb = (const uint64_t*)&b;
    const size_t n = sizeof(PTStats) / 8;
    for (size_t i = 0; i < n; i++) {
        if (&pa[i] == &a.maxStackDepth) { if (pb[i] > pa[i]) pa[i] = pb[i]; }
        else pa[i] += pb[i];
    }
}

} // namespace

extern "C" {

int oracle_render(const PTSceneDesc* scene, const PTFrameParams* params, const float* accumulated,
                  float* output, PTStats* stats, OracleOptions* opt)
{
    if (!scene || !params || !output) return -1;
    if (params->CurrentSample > 0 && !accumulated) return -1;
    SceneView S = MakeView(scene);
    OracleOptions defaults;
    memset(&defaults, 0, sizeof(defaults));
    if (!opt) opt = &defaults;
    int x0 = opt->x0, y0 = opt->y0, x1 = opt->x1, y1 = opt->y1;
    uint32_t W = params->OutputWidth, H = params->OutputHeight;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = (int)W; y1 = (int)H; }
    // DispatchCompute coverage (PathTracer.cs:203-208): gx*8 x gy*8 threads, 0 = cover everything
    uint32_t covW = params->DispatchGroupsX ? (params->DispatchGroupsX * 8u < W ? params->DispatchGroupsX * 8u : W) : W;
    uint32_t covH = params->DispatchGroupsY ? (params->DispatchGroupsY * 8u < H ? params->DispatchGroupsY * 8u : H) : H;
    int threads = opt->threads > 0 ? opt->threads : 1;
    if (opt->rayDump) threads = 1;
    int world = opt->tileWorld > 0 ? opt->tileWorld : 1;
    int rank = opt->tileRank;
    opt->rayDumpCount = 0;

    std::vector<PTStats> tstats(threads);
    for (auto& s : tstats) memset(&s, 0, sizeof(PTStats));
    std::atomic<int> nextRow(y0);
    auto worker = [&](int tid) {
        Ctx c;
        c.S = &S; c.P = params; memset(&c.st, 0, sizeof(PTStats));
        c.shadowAnyHit = opt->shadowAnyHit != 0;
        c.nanRayEarlyOut = opt->shadowAnyHit != 0;
        c.opt = opt->rayDump ? opt : nullptr;
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= y1) break;
            for (int x = x0; x < x1; ++x) {
                if ((uint32_t)x >= covW || (uint32_t)y >= covH) continue;
                bool owned = world == 1 || (((x >> 4) + (y >> 4)) % world) == rank;
                if (!owned) {
                    float* o = output + ((size_t)y * W + x) * 4;
                    o[0] = o[1] = o[2] = o[3] = 0.0f;
                    continue;
                }
                RenderPixel(c, (uint32_t)x, (uint32_t)y, accumulated, output);
            }
        }
        tstats[tid] = c.st;
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    if (stats) {
        memset(stats, 0, sizeof(PTStats));
        for (auto& s : tstats) AddStats(*stats, s);
    }
    return 0;
}

int oracle_trace(const PTSceneDesc* scene, const OracleRay* rays, uint64_t n, float* outT,
                 uint32_t* outPrim, uint32_t* outSteps)
{
    if (!scene || !rays) return -1;
    SceneView S = MakeView(scene);
    PTFrameParams P;
    memset(&P, 0, sizeof(P));
    Ctx c;
    c.S = &S; c.P = &P; memset(&c.st, 0, sizeof(PTStats)); c.shadowAnyHit = false; c.nanRayEarlyOut = false; c.opt = nullptr;
    for (uint64_t i = 0; i < n; i++) {
        Ray r = {V3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]),
                 V3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2])};
        RayHit hit;
        memset(&hit, 0, sizeof(hit));
        hit.distance = rays[i].tmax;
        hit.triIndex = 0xFFFFFFFFu;
        bool shadow = rays[i].kind != 0.0f;
        // traverse without the attribute fetch: call the loop through the shadow path, then report
        float tmax = rays[i].tmax;
        RayIntersectBvh(c, r, hit, true);
        bool found = hit.distance < tmax;
        if (shadow) {
            if (outT) outT[i] = found ? 0.0f : tmax;
            if (outPrim) outPrim[i] = found ? 1u : 0xFFFFFFFFu;
        } else {
            if (outT) outT[i] = hit.distance;
            if (outPrim) outPrim[i] = found ? hit.triIndex : 0xFFFFFFFFu;
        }
        if (outSteps) outSteps[i] = hit.steps;
    }
    return 0;
}

// The same restated traversal as oracle_trace, reporting the full closest-hit record per ray: out4[i] = {t, u, v, prim bits}
// (prim = 0xFFFFFFFF, u = v = 0 on a miss).  Compared bit for bit with the reference's own CWBVH walker
// (BVH8_CWBVH::Intersect, tiny_bvh.h:7006-7114, through oracle/ref_driver_cwbvh.cpp) by tests/test_oracle.py.
int oracle_trace_uv(const PTSceneDesc* scene, const OracleRay* rays, uint64_t n, float* out4, uint64_t* outNodeVisits, uint64_t* outTriTests)
{
    if (!scene || !rays || !out4) return -1;
    SceneView S = MakeView(scene);
    PTFrameParams P;
    memset(&P, 0, sizeof(P));
    Ctx c;
    c.S = &S; c.P = &P; memset(&c.st, 0, sizeof(PTStats)); c.shadowAnyHit = false; c.nanRayEarlyOut = false; c.opt = nullptr;
    for (uint64_t i = 0; i < n; i++) {
        Ray r = {V3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]),
                 V3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2])};
        RayHit hit;
        memset(&hit, 0, sizeof(hit));
        hit.distance = rays[i].tmax;
        hit.triIndex = 0xFFFFFFFFu;
        RayIntersectBvh(c, r, hit, true);                 // the shadow path = the traversal loop without the attribute fetch
        const bool found = hit.distance < rays[i].tmax;
        const uint32_t prim = found ? hit.triIndex : 0xFFFFFFFFu;
        out4[i * 4 + 0] = hit.distance;
        out4[i * 4 + 1] = found ? hit.barycentric.x : 0.0f;
        out4[i * 4 + 2] = found ? hit.barycentric.y : 0.0f;
        memcpy(&out4[i * 4 + 3], &prim, 4);
    }
    if (outNodeVisits) *outNodeVisits = c.st.nodeVisits;
    if (outTriTests) *outTriTests = c.st.triTests;
    return 0;
}

float oracle_random_float(uint32_t* state) { return pt_random_float(state); }

float oracle_math(int fn, float x, float y)
{
    switch (fn) {
    case 0: return pt_sin(x);
    case 1: return pt_cos(x);
    case 2: return pt_log(x);
    case 3: return pt_log2(x);
    case 4: return pt_exp2(x);
    case 5: return pt_pow(x, y);
    case 6: return pt_acos(x);
    case 7: return pt_asin(x);
    case 8: return pt_sqrt(x);
    case 9: return pt_rcp(x);
    case 10: return pt_atan2(x, y);
    case 11: return pt_fmod(x, y);
    case 12: return pt_wrap01(x);
    case 13: {                                    // the literal loops of util/texture.hlsl:41-48 (with this project's guard)
        float u = x;
        if (pt_abs(u) >= 16777216.0f) u = 0.0f;
        while (u > 1.0f) u -= 1.0f;
        while (u < 0.0f) u += 1.0f;
        return u;
    }
    case 14: return pt_unorm8((uint32_t)x);       // the kernels' form of a texel channel ...
    case 15: return (float)((uint32_t)x & 0xFFu) / 255.0f;    // ... and the division of util/texture.hlsl that the oracle performs
    default: return 0.0f;
    }
}

float oracle_env_probe(const PTSceneDesc* scene, const PTFrameParams* params, int what, const float* in, uint64_t n, float* out)
{
    SceneView S = MakeView(scene);
    if (!S.hasEnvironmentTexture) return 0.0f;
    Ctx c;
    c.S = &S; c.P = params; memset(&c.st, 0, sizeof(PTStats)); c.shadowAnyHit = false; c.nanRayEarlyOut = false; c.opt = nullptr;
    for (uint64_t k = 0; k < n; k++) {
        if (what == 0) out[k] = S.EnvironmentCDF[(size_t)in[k]];
        else if (what == 1) { f2 uv = BinarySearch(S, in[k]); out[2 * k] = uv.x; out[2 * k + 1] = uv.y; }
        else if (what == 2) {
            f4 r = EvalEnvMap(c, V3(in[3 * k], in[3 * k + 1], in[3 * k + 2]), 1.0f);
            out[4 * k] = r.x; out[4 * k + 1] = r.y; out[4 * k + 2] = r.z; out[4 * k + 3] = r.w;
        } else if (what == 3) {
            uint32_t rng = pt_asuint(in[k]);
            f3 col = V3(0.0f);
            f4 r = SampleEnvMap(c, col, rng);
            float* o = out + 8 * k;
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; o[4] = col.x; o[5] = col.y; o[6] = col.z; o[7] = 0.0f;
        }
    }
    return S.EnvironmentCdfSum;
}

// ------------------------------------------------------------------------------------------
// MeshProcessing.compute / CopyTextureData.compute
// ------------------------------------------------------------------------------------------
namespace {
struct ByteAddressBuffer {
    const uint8_t* p;
    uint32_t Load(uint32_t off) const { uint32_t v; memcpy(&v, p + off, 4); return v; }
    f3 LoadFloat3(uint32_t off) const { return V3(pt_asfloat(Load(off)), pt_asfloat(Load(off + 4)), pt_asfloat(Load(off + 8))); }
};
// float4x4 in Unity memory order: M(r, c) = m[c * 4 + r]
inline f4 mulMV(const float* m, f4 v)          // mul(M, v)
{
    f4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}
inline f4 mulVM(f4 v, const float* m)          // mul(v, M)
{
    f4 r;
    r.x = v.x * m[0] + v.y * m[1] + v.z * m[2] + v.w * m[3];
    r.y = v.x * m[4] + v.y * m[5] + v.z * m[6] + v.w * m[7];
    r.z = v.x * m[8] + v.y * m[9] + v.z * m[10] + v.w * m[11];
    r.w = v.x * m[12] + v.y * m[13] + v.z * m[14] + v.w * m[15];
    return r;
}
inline f3 xyz(f4 v) { return V3(v.x, v.y, v.z); }
}

int oracle_process_mesh(const PTMeshDesc* M, float* vertexPositionBuffer, void* triangleAttributesBuffer)   // MeshProcessing.compute:59-139
{
    if (!M || !vertexPositionBuffer || !triangleAttributesBuffer) return -1;
    ByteAddressBuffer VertexBuffer = {(const uint8_t*)M->vertexBuffer};
    std::vector<uint8_t> paddedIndices;
    ByteAddressBuffer IndexBuffer = {nullptr};
    if (M->indexBuffer) {                        // Load2 on the last odd triangle reads one word past the 16-bit data
        paddedIndices.assign((size_t)M->indexBufferBytes + 8, 0);
        memcpy(paddedIndices.data(), M->indexBuffer, (size_t)M->indexBufferBytes);
        IndexBuffer.p = paddedIndices.data();
    }
    PTTriangleAttributes* out = (PTTriangleAttributes*)triangleAttributesBuffer;
    for (uint32_t triIndex = 0; triIndex < M->TriangleCount; triIndex++) {
        uint32_t vertIndices[3];
        if (M->indexBuffer) {
            if (M->flags & PT_MESH_HAS_32_BIT_INDICES) {
                uint32_t byteOffset = triIndex * (3 * 4);
                for (int k = 0; k < 3; k++) vertIndices[k] = IndexBuffer.Load(byteOffset + 4 * k);
            } else {
                uint32_t byteOffset = triIndex * (3 * 2);
                uint32_t alignedOffset = byteOffset & ~3u;
                uint32_t px = IndexBuffer.Load(alignedOffset), py = IndexBuffer.Load(alignedOffset + 4);
                uint32_t offsetInWord = byteOffset % 4;
                vertIndices[0] = (offsetInWord == 0) ? (px & 0xFFFF) : (px >> 16);
                vertIndices[1] = (offsetInWord == 0) ? (px >> 16) : (py & 0xFFFF);
                vertIndices[2] = (offsetInWord == 0) ? (py & 0xFFFF) : (py >> 16);
            }
        } else {
            vertIndices[0] = triIndex * 3; vertIndices[1] = triIndex * 3 + 1; vertIndices[2] = triIndex * 3 + 2;
        }
        int writeTriIndex = (int)(M->OutputTriangleStart + triIndex);
        int writeIndex = writeTriIndex * 3;
        PTTriangleAttributes attr;
        memset(&attr, 0, sizeof(attr));
        float* nrm[3] = {attr.normal0, attr.normal1, attr.normal2};
        float* tan[3] = {attr.tangent0, attr.tangent1, attr.tangent2};
        float* uvs[3] = {attr.uv0, attr.uv1, attr.uv2};
        for (int k = 0; k < 3; k++) {
            uint32_t base = vertIndices[k] * M->VertexStride;
            f3 vert = VertexBuffer.LoadFloat3(base + M->PositionOffset);
            vert = xyz(mulMV(M->LocalToWorld, f4{vert.x, vert.y, vert.z, 1.0f}));
            float* vp = vertexPositionBuffer + 4 * (size_t)(writeIndex + k);
            vp[0] = vert.x; vp[1] = vert.y; vp[2] = vert.z; vp[3] = 0.0f;
            if (M->flags & PT_MESH_HAS_NORMALS) {
                f3 n = VertexBuffer.LoadFloat3(base + M->NormalOffset);
                n = normalize(xyz(mulVM(f4{n.x, n.y, n.z, 0.0f}, M->WorldToLocal)));
                nrm[k][0] = n.x; nrm[k][1] = n.y; nrm[k][2] = n.z;
            }
            if (M->flags & PT_MESH_HAS_TANGENTS) {
                f3 t = VertexBuffer.LoadFloat3(base + M->TangentOffset);
                t = normalize(xyz(mulMV(M->LocalToWorld, f4{t.x, t.y, t.z, 0.0f})));
                tan[k][0] = t.x; tan[k][1] = t.y; tan[k][2] = t.z;
            }
            if (M->flags & PT_MESH_HAS_UVS) {
                uvs[k][0] = pt_asfloat(VertexBuffer.Load(base + M->UVOffset));
                uvs[k][1] = pt_asfloat(VertexBuffer.Load(base + M->UVOffset + 4));
            }
        }
        attr.materialIndex = M->MaterialIndex;
        out[writeTriIndex] = attr;
    }
    return 0;
}

int oracle_copy_texture_data(const PTTextureDesc* textures, uint32_t count, uint32_t* TextureData)
{
    if (!textures || !TextureData) return -1;
    int textureDescriptorOffset = 0;
    int textureDataOffset = (int)count * 4;                                           // BVHScene.cs:388-389
    for (uint32_t t = 0; t < count; t++) {
        const uint32_t TextureWidth = textures[t].width, TextureHeight = textures[t].height;
        for (uint32_t pixelY = 0; pixelY < TextureHeight; pixelY++)
            for (uint32_t pixelX = 0; pixelX < TextureWidth; pixelX++) {              // CopyTextureData.compute:13-36
                uint32_t pixelIndex = pixelY * TextureWidth + pixelX;
                TextureData[textureDescriptorOffset + 0] = TextureWidth;
                TextureData[textureDescriptorOffset + 1] = TextureHeight;
                TextureData[textureDescriptorOffset + 2] = (uint32_t)textureDataOffset;
                TextureData[textureDescriptorOffset + 3] = 0;
                const float* texel = textures[t].texels + 4 * (size_t)pixelIndex;
                uint32_t r = pt_f2u(texel[0] * 255.0f) & 0xff;
                uint32_t g = pt_f2u(texel[1] * 255.0f) & 0xff;
                uint32_t b = pt_f2u(texel[2] * 255.0f) & 0xff;
                uint32_t a = textures[t].hasAlpha ? pt_f2u(texel[3] * 255.0f) & 0xff : 0xff;
                TextureData[pixelIndex + (uint32_t)textureDataOffset] = (a << 24) | (b << 16) | (g << 8) | r;
            }
        textureDataOffset += (int)(TextureWidth * TextureHeight);                      // :413-414
        textureDescriptorOffset += 4;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// util/tonemap.hlsl + Presentation.shader FragBlit
// ------------------------------------------------------------------------------------------
namespace {
inline f3 pow_f3(f3 v, float e) { return V3(pt_pow(v.x, e), pt_pow(v.y, e), pt_pow(v.z, e)); }
inline f3 lerp_f3(f3 x, f3 y, f3 s) { return x + s * (y - x); }
inline f3 greaterThan(f3 a, f3 b) { return V3(a.x > b.x ? 1.0f : 0.0f, a.y > b.y ? 1.0f : 0.0f, a.z > b.z ? 1.0f : 0.0f); }   // util/common.hlsl:269-272
inline f3 saturate3(f3 v) { return V3(pt_saturate(v.x), pt_saturate(v.y), pt_saturate(v.z)); }
inline f3 mul3x3(const float* m, f3 c)          // mul(float3x3 M, float3 v): rows dotted with v
{
    return V3(dot(V3(m[0], m[1], m[2]), c), dot(V3(m[3], m[4], m[5]), c), dot(V3(m[6], m[7], m[8]), c));
}

inline f3 LinearToSrgb(f3 rgb)                                                         // tonemap.hlsl:6-11
{
    f3 low = rgb * 12.92f;
    f3 high = pow_f3(rgb, 1.0f / 2.4f) * 1.055f - V3(0.055f);
    return lerp_f3(low, high, greaterThan(rgb, V3(0.0031308f)));
}

inline f3 ACES(f3 color)                                                               // :21-46
{
    static const float ACESInputMat[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
    static const float ACESOutputMat[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};
    color = mul3x3(ACESInputMat, color);
    f3 a = color * (color + 0.0245786f) - V3(0.000090537f);
    f3 b = color * (0.983729f * color + 0.4329510f) + 0.238081f;
    color = a / b;
    return mul3x3(ACESOutputMat, color);
}

inline f3 Filmic(f3 x)                                                                 // :49-54
{
    f3 X = V3(pt_max(0.0f, x.x - 0.004f), pt_max(0.0f, x.y - 0.004f), pt_max(0.0f, x.z - 0.004f));
    f3 result = (X * (6.2f * X + 0.5f)) / (X * (6.2f * X + 1.7f) + 0.06f);
    return pow_f3(result, 2.2f);
}

inline f3 Lottes(f3 x)                                                                 // :57-74
{
    const float a = 1.6f, d = 0.977f, hdrMax = 8.0f, midIn = 0.18f, midOut = 0.267f;
    float b = (-pt_pow(midIn, a) + pt_pow(hdrMax, a) * midOut) / ((pt_pow(hdrMax, a * d) - pt_pow(midIn, a * d)) * midOut);
    float c = (pt_pow(hdrMax, a * d) * pt_pow(midIn, a) - pt_pow(hdrMax, a) * pt_pow(midIn, a * d) * midOut) /
              ((pt_pow(hdrMax, a * d) - pt_pow(midIn, a * d)) * midOut);
    return pow_f3(x, a) / (pow_f3(x, a * d) * b + c);
}

inline f3 Reinhard(f3 x) { return x / (V3(1.0f) + x); }                                // :76-79
}

int oracle_present(const PTPresentParams* Q, const float* src, float* dst)             // Presentation.shader:36-73
{
    if (!Q || !src || !dst) return -1;
    for (uint32_t y = 0; y < Q->OutputHeight; y++)
        for (uint32_t x = 0; x < Q->OutputWidth; x++) {
            const size_t i = (size_t)y * Q->OutputWidth + x;
            f3 color = V3(src[4 * i], src[4 * i + 1], src[4 * i + 2]);      // tex2D(_MainTex, i.uv) at the pixel centre
            color = color * Q->Exposure;
            switch (Q->Mode) {
            case 1: color = ACES(color); break;
            case 2: color = Filmic(color); break;
            case 3: color = Reinhard(color); break;
            case 4: color = Lottes(color); break;
            }
            if (Q->sRGB) color = LinearToSrgb(color);
            color = saturate3(lerp_f3(V3(0.5f), color, V3(Q->Contrast)));
            color = pow_f3(color, 1.0f / Q->Brightness);
            f3 l = V3(Luminance(color));
            color = lerp_f3(l, color, V3(Q->Saturation));
            f2 uv = {((float)x + 0.5f) / (float)Q->OutputWidth, ((float)y + 0.5f) / (float)Q->OutputHeight};
            f2 centerUv = {(uv.x - 0.5f) * 2.0f, (uv.y - 0.5f) * 2.0f};
            color = color * (1.0f - (centerUv.x * centerUv.x + centerUv.y * centerUv.y) * Q->Vignette);
            dst[4 * i] = color.x; dst[4 * i + 1] = color.y; dst[4 * i + 2] = color.z; dst[4 * i + 3] = 1.0f;
        }
    return 0;
}

static Material MaterialFromData(const float* material, f3 V, f3 N, float eta)
{
    PTSceneDesc none;
    memset(&none, 0, sizeof(none));
    SceneView S = MakeView(&none);
    PTFrameParams P;
    memset(&P, 0, sizeof(P));
    Ctx c;
    c.S = &S; c.P = &P; memset(&c.st, 0, sizeof(PTStats)); c.shadowAnyHit = false; c.nanRayEarlyOut = false; c.opt = nullptr;
    PTMaterialData md;
    memcpy(&md, material, sizeof(md));
    Ray ray = {V3(0.0f), -V};
    RayHit hit;
    memset(&hit, 0, sizeof(hit));
    hit.normal = N;
    Material m = GetMaterial(c, md, ray, hit);
    if (eta > 0.0f) m.eta = eta;
    return m;
}

void oracle_eval_brdf(const float* material, const float* V, const float* N, const float* L, float eta, float* out)
{
    f3 v = L3(V), n = L3(N), l = L3(L);
    Material m = MaterialFromData(material, v, n, eta);
    float pdf = 0.0f;
    f3 f = EvalBRDF(m, v, n, l, pdf);
    out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = pdf;
}

void oracle_sample_brdf(const float* material, const float* V, const float* N, float eta, uint32_t* rngState,
                        float* outL, float* outFPdf)
{
    f3 v = L3(V), n = L3(N);
    Material m = MaterialFromData(material, v, n, eta);
    f3 l;
    float pdf = 0.0f;
    f3 f = SampleBRDF(m, v, n, l, pdf, *rngState);
    outL[0] = l.x; outL[1] = l.y; outL[2] = l.z;
    outFPdf[0] = f.x; outFPdf[1] = f.y; outFPdf[2] = f.z; outFPdf[3] = pdf;
}

// Batch forms of the two probes above for the distribution tests of tests/test_oracle.py (per lobe: does the pdf that
// EvalBRDF reports integrate to the probability of drawing a valid direction, and are the directions SampleBRDF draws
// distributed as that pdf).  L: n x 3, out: n x 4 (f.xyz, pdf).
void oracle_eval_brdf_batch(const float* material, const float* V, const float* N, float eta, const float* L, uint64_t n, float* out)
{
    f3 v = L3(V), nn = L3(N);
    Material m = MaterialFromData(material, v, nn, eta);
    for (uint64_t i = 0; i < n; i++) {
        float pdf = 0.0f;
        f3 f = EvalBRDF(m, v, nn, L3(L + 3 * i), pdf);
        out[4 * i + 0] = f.x; out[4 * i + 1] = f.y; out[4 * i + 2] = f.z; out[4 * i + 3] = pdf;
    }
}

// n consecutive SampleBRDF calls on ONE RNG chain (3 draws each, util/brdf.hlsl:247-249,258).  outL: n x 3, outFPdf: n x 4.
void oracle_sample_brdf_batch(const float* material, const float* V, const float* N, float eta, uint32_t* rngState, uint64_t n,
                              float* outL, float* outFPdf)
{
    f3 v = L3(V), nn = L3(N);
    Material m = MaterialFromData(material, v, nn, eta);
    for (uint64_t i = 0; i < n; i++) {
        f3 l;
        float pdf = 0.0f;
        f3 f = SampleBRDF(m, v, nn, l, pdf, *rngState);
        outL[3 * i + 0] = l.x; outL[3 * i + 1] = l.y; outL[3 * i + 2] = l.z;
        outFPdf[4 * i + 0] = f.x; outFPdf[4 * i + 1] = f.y; outFPdf[4 * i + 2] = f.z; outFPdf[4 * i + 3] = pdf;
    }
}

} // extern "C"
