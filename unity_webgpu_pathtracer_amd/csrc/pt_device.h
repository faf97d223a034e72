// pt_device.h — device-side building blocks of the MI355X path tracer (gfx950, wave64).
//
// Everything here is __device__ code used by the render kernels in pt_kernels.hip.  It implements the
// reference's hot path (Assets/Resources/PathTracer.compute + Assets/Resources/util/ *.hlsl) on the
// buffer layouts of include/ptmi_layouts.h, with the canonical fp32 arithmetic of include/ptmi_math.h.
// Citations (file:line) are relative to /root/reference/Assets/Resources.
//
// Compile with -ffp-contract=off: products and sums are separate IEEE operations unless written pt_fma.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ptmi_layouts.h"
#include "ptmi_math.h"
#include "ptmi_plugin.h"

#define PT_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------
// small vector algebra (explicit operation order; see DESIGN.md "Canonical arithmetic")
// ------------------------------------------------------------------------------------------
struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };

PT_DEV v3 mk3(float a) { return {a, a, a}; }
PT_DEV v3 mk3(float a, float b, float c) { return {a, b, c}; }
PT_DEV v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
PT_DEV v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_DEV v3 operator*(float s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }
PT_DEV v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PT_DEV v3 operator-(float s, v3 a) { return {s - a.x, s - a.y, s - a.z}; }
PT_DEV v4 operator+(v4 a, v4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
PT_DEV v4 operator-(v4 a, v4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
PT_DEV v4 operator*(v4 a, v4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
PT_DEV v4 operator*(float s, v4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }

PT_DEV float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV v3 cross3(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_DEV float length3(v3 v) { return pt_sqrt(dot3(v, v)); }
PT_DEV v3 normalize3(v3 v) { return v * (1.0f / pt_sqrt(dot3(v, v))); }
PT_DEV v3 lerp3(v3 a, v3 b, float t) { return a + t * (b - a); }
PT_DEV v4 lerp4(v4 a, v4 b, float t) { return a + t * (b - a); }
PT_DEV v3 reflect3(v3 i, v3 n) { return i - (2.0f * dot3(n, i)) * n; }
PT_DEV v3 refract3(v3 i, v3 n, float eta)
{
    float d = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f);
    return eta * i - (eta * d + pt_sqrt(k)) * n;
}
PT_DEV float luminance3(v3 c) { return dot3(c, mk3(0.299f, 0.587f, 0.114f)); }     // util/common.hlsl:195-198

struct Onb { v3 x, y, z; };
PT_DEV v3 onb_to_world(const Onb& b, v3 l) { return b.x * l.x + b.y * l.y + b.z * l.z; }         // util/common.hlsl:386-389
PT_DEV v3 onb_to_local(const Onb& b, v3 w) { return {dot3(b.x, w), dot3(b.y, w), dot3(b.z, w)}; } // :391-394

// util/common.hlsl:346-384, ONB_METHOD 1
PT_DEV Onb make_onb(v3 z)
{
    float lenSq = dot3(z, z);
    if (lenSq == 0.0f) return {mk3(1, 0, 0), mk3(0, 1, 0), mk3(0, 0, 1)};
    z = normalize3(z);
    float k = 1.0f / pt_max(1.0f + z.z, 0.00001f);
    float a = z.y * k;
    float b = z.y * a;
    float c = -z.x * a;
    v3 x = normalize3(mk3(z.z + b, c, -z.x));
    v3 y = normalize3(mk3(c, 1.0f - b, -z.y));
    return {x, y, z};
}

// ------------------------------------------------------------------------------------------
// device views of the scene / frame
// ------------------------------------------------------------------------------------------
struct DScene {
    const uint4*  nodes;        // PTCwbvhNode[] as 5 x uint4
    const float4* tris;         // PTCwbvhTri[] as 3 x float4
    const float4* attrs;        // PTTriangleAttributes[] as 8 x float4
    const float4* materials;    // PTMaterialData[] as 8 x float4
    const float4* lights;       // PTLight[] as 4 x float4
    const float4* lightConst;   // 4 x float4 per light: what the shader recomputes from the record at every use (derive_light_rows)
    const uint32_t* tex;        // TextureData
    int32_t lightCount;
    uint32_t materialCount;
    uint32_t hasLights;
    uint32_t hasTextures;
    // HAS_TLAS (util/tlas.hlsl): TLASData floats, instance records as 9 x float4 (PTGpuInstance)
    const float*  tlas;
    const float4* instances;
    const float4* instByLeaf;   // derived by PTSetScene: entry k = worldToLocal (4 rows), offsets row, {instance index} of the instance that TLAS index slot k
                                // names -- 6 float4, in the order TLAS leaves list them: entering an instance is ONE fetch, not index -> record
    const float*  tlasBfs;      // derived by PTSetScene: the TLAS nodes reachable from node 0, renumbered breadth-first (children indices
                                // rewritten, everything else copied): the refill kernel keeps the first nodes of THIS array in LDS
    uint32_t tlasNodeCount;     // nodes in tlasBfs
    uint32_t tlasIndexOffset;
    uint32_t hasTlas;
    // HAS_ENVIRONMENT_TEXTURE (util/sky.hlsl:7-88): RGBAFloat texels, running sum of their grayscale, its last value
    const float4* envTex;
    const float*  envCdf;
    int32_t envW, envH;
    float envCdfSum;
    uint32_t hasEnvTex;
};

struct Counters {
    uint32_t paths, closestRays, shadowRays, nodeVisits, triTests, attrFetches, materialFetches,
             lightFetches, texelFetches, texDescFetches, pixelsWritten, pixelsRead, maxStack, overflows,
             tlasNodeVisits, instanceVisits;
};

PT_DEV float rnd(uint32_t& s) { return pt_random_float(&s); }

// ------------------------------------------------------------------------------------------
// CWBVH traversal.  util/bvh.hlsl:61-215 (Ylitie et al. 2017 as implemented by the reference)
// ------------------------------------------------------------------------------------------
// Stack: the reference keeps a private uint2[32] per thread (util/bvh.hlsl:11,131).  Here the first
// PT_LDS_STACK entries of every lane live in LDS laid out [entry][lane] (one ds_read/write_b64 per
// lane, bank = 2*lane mod 64: conflict-free whatever depth each lane is at); deeper entries, which the
// benchmark scenes never reach, spill to a private array.
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12
#endif
#define PT_SPILL_STACK (PT_BVH_STACK_SIZE - PT_LDS_STACK)

// GSPILL = true: entries beyond the LDS part go to a slab in HBM (gbase + gidx * (32 - LDSN)) instead of a private
// array.  A private array that is indexed dynamically lives in scratch memory, and a kernel that declares ANY scratch
// makes the dispatcher allocate a scratch wave slot per wave: for the refill trace kernel that allocation, not registers
// or LDS, was what limited the waves in flight (SPI_RA_TMP_STALL_CSN 1.7e9 vs SPI_RA_WAVE_SIMD_FULL_CSN 1.9e7 per five
// passes; removing the 192-byte array: +8 % end to end).  The slab is touched only by rays deeper than LDSN entries.
typedef const volatile __attribute__((address_space(3))) uint32_t* pt_lds_word;    // an LDS address (32 bits, no flat pointer)
#define PT_LDS_WORD(var) ((pt_lds_word)(&(var)))

// The LDS part is addressed through an LDS-address-space pointer (32 bits): through a generic `uint2*` the accesses become FLAT
// instructions as soon as the compiler loses sight of where the pointer came from (the megakernel keeps the struct in scratch).
typedef uint32_t pt_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) pt_u2* pt_lds_u2;
#define PT_LDS_U2(ptr) ((pt_lds_u2)(ptr))
template <int LDSN, bool GSPILL = false>
struct TravStackT {
    static constexpr uint32_t kLds = LDSN;
    static constexpr bool kGlobalSpill = GSPILL;
    pt_lds_u2 lds;              // &smem[0][threadIdx.x]; entry stride = blockDim.x
    uint32_t stride;
    uint2 spill[GSPILL ? 1 : PT_BVH_STACK_SIZE - LDSN];
    uint2* gbase;               // GSPILL: slab base (wave-uniform)
    pt_lds_word gwave;          // GSPILL: LDS word holding this wave's index; the lane's slab row is *gwave * 64 + lane
};
typedef TravStackT<PT_LDS_STACK> TravStack;

// The slab row is formed where it is used, from an LDS word and the lane id: nothing about the (almost never taken) slab
// path lives in a register across the traversal loop, where at 64 VGPRs it would be spilled -- to scratch.
PT_DEV size_t spill_row(pt_lds_word gwave)
{
    uint32_t lane;              // volatile asm: the two instructions stay HERE (as builtins they are hoisted out of the loop and spilled)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return (size_t)(*gwave * 64u + lane);
}

// The empty asm statements keep the LDS access in a branch of its own.  Without them the compiler merges it with the access to
// the private overflow array into ONE access through a selected pointer -- a FLAT load / store, which travels the texture-address
// path even when it resolves to LDS and counts against both memory counters (round 3: the megakernel's stack was all flat).
template <class ST>
PT_DEV void stack_push(ST& st, uint32_t sp, uint2 v)
{
    if (sp < ST::kLds) { st.lds[sp * st.stride] = pt_u2{v.x, v.y}; asm volatile("" ::: "memory"); }
    else if (sp < PT_BVH_STACK_SIZE) {
        if (ST::kGlobalSpill) st.gbase[spill_row(st.gwave) * (PT_BVH_STACK_SIZE - ST::kLds) + (sp - ST::kLds)] = v;
        else st.spill[ST::kGlobalSpill ? 0u : sp - ST::kLds] = v;
    }
}
template <class ST>
PT_DEV uint2 stack_pop(ST& st, uint32_t sp)
{
    if (sp < ST::kLds) { const pt_u2 q = st.lds[sp * st.stride]; uint2 v = make_uint2(q.x, q.y); asm volatile("" : "+v"(v.x), "+v"(v.y)); return v; }
    if (sp < PT_BVH_STACK_SIZE) {
        if (ST::kGlobalSpill) return st.gbase[spill_row(st.gwave) * (PT_BVH_STACK_SIZE - ST::kLds) + (sp - ST::kLds)];
        return st.spill[ST::kGlobalSpill ? 0u : sp - ST::kLds];
    }
    return make_uint2(0u, 0u);
}

PT_DEV uint32_t byte_of(uint32_t v, int i) { return (v >> (i * 8)) & 0xFFu; }

// util/bvh.hlsl:77-124.  n0..n4 are the five 16-byte rows of the node.
PT_DEV uint32_t cwbvh_node_hitmask(v3 origin, v3 invDir, uint32_t octinv4, float tmax,
                                   uint4 n0, uint4 n1, uint4 n2, uint4 n3, uint4 n4)
{
    // GetNodeInvDir :61-75
    uint32_t ex = (byte_of(n0.w, 0) ^ 0x80u) - 0x80u;
    uint32_t ey = (byte_of(n0.w, 1) ^ 0x80u) - 0x80u;
    uint32_t ez = (byte_of(n0.w, 2) ^ 0x80u) - 0x80u;
    float sx = pt_asfloat((ex + 127u) << 23) * invDir.x;
    float sy = pt_asfloat((ey + 127u) << 23) * invDir.y;
    float sz = pt_asfloat((ez + 127u) << 23) * invDir.z;
    float px = (pt_asfloat(n0.x) - origin.x) * invDir.x;
    float py = (pt_asfloat(n0.y) - origin.y) * invDir.y;
    float pz = (pt_asfloat(n0.z) - origin.z) * invDir.z;
    const bool nx = invDir.x < 0.0f, ny = invDir.y < 0.0f, nz = invDir.z < 0.0f;
    uint32_t hitmask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint32_t meta = i == 0 ? n1.z : n1.w;
        uint32_t qlox = i == 0 ? n2.x : n2.y, qloy = i == 0 ? n2.z : n2.w, qloz = i == 0 ? n3.x : n3.y;
        uint32_t qhix = i == 0 ? n3.z : n3.w, qhiy = i == 0 ? n4.x : n4.y, qhiz = i == 0 ? n4.z : n4.w;
        uint32_t lox = nx ? qhix : qlox, hix = nx ? qlox : qhix;
        uint32_t loy = ny ? qhiy : qloy, hiy = ny ? qloy : qhiy;
        uint32_t loz = nz ? qhiz : qloz, hiz = nz ? qloz : qhiz;
        uint32_t isInner = (meta & (meta << 1)) & 0x10101010u;
        // == (isInner >> 4) * 0xff without the quarter-rate v_mul_lo_u32: the empty asm keeps the compiler from folding shift and
        // subtraction back into that multiplication (it did: two v_mul_lo_u32 per node visit in the round-3 ISA)
        uint32_t innerHi = (isInner >> 4) << 8;
        asm("" : "+v"(innerHi));
        uint32_t innerMask = innerHi - (isInner >> 4);
        uint32_t bitIndex = (meta ^ (octinv4 & innerMask)) & 0x1F1F1F1Fu;
        uint32_t childBits = (meta >> 5) & 0x07070707u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float tminx = pt_fma((float)byte_of(lox, j), sx, px);
            float tmaxx = pt_fma((float)byte_of(hix, j), sx, px);
            float tminy = pt_fma((float)byte_of(loy, j), sy, py);
            float tmaxy = pt_fma((float)byte_of(hiy, j), sy, py);
            float tminz = pt_fma((float)byte_of(loz, j), sz, pz);
            float tmaxz = pt_fma((float)byte_of(hiz, j), sz, pz);
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

struct TraceHit {
    float t;            // hit.distance (PT_FAR_PLANE on a miss)
    float u, v;         // hit.barycentric
    uint32_t triIndex;  // hit.triIndex (primitive index from the triangle record)
};

// util/bvh.hlsl:23-59 on the three 16-byte rows of a triangle record.  TMIN = 1e-4 (util/bvh.hlsl:47); the HAS_TLAS
// variant accepts distance > 0 (util/tlas.hlsl:47).  Returns true when the hit was accepted.
template <bool TLAS_EPS>
PT_DEV bool intersect_triangle_rows(float4 t0, float4 t1, float4 t2, uint32_t indexOffset, v3 o, v3 d, TraceHit& hit)
{
    v3 e2 = mk3(t0.x, t0.y, t0.z), e1 = mk3(t1.x, t1.y, t1.z), v0 = mk3(t2.x, t2.y, t2.z);
    v3 r = cross3(d, e2);
    float a = dot3(e1, r);
    if (pt_abs(a) > 0.0000001f) {
        float f = 1.0f / a;
        v3 s = o - v0;
        float u = f * dot3(s, r);
        if (u >= 0.0f && u <= 1.0f) {
            v3 q = cross3(s, e1);
            float v = f * dot3(d, q);
            if (v >= 0.0f && u + v <= 1.0f) {
                float dist = f * dot3(e2, q);
                if (dist > (TLAS_EPS ? 0.0f : 0.0001f) && dist < hit.t) {
                    hit.u = u; hit.v = v;
                    hit.triIndex = indexOffset + pt_asuint(t2.w);
                    hit.t = dist;
                    return true;
                }
            }
        }
    }
    return false;
}

PT_DEV void intersect_triangle(const float4* __restrict__ tris, uint32_t triAddr, v3 o, v3 d, TraceHit& hit)
{
    // all 48 bytes are requested before the first use, so the whole test costs ONE memory round trip
    // (left to itself hipcc sinks the v0 load under the determinant test: two dependent round trips)
    const float4* tp = tris + triAddr;                    // one address, rows at immediate offsets 0 / 16 / 32
    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
    asm volatile("" : "+v"(t0.x), "+v"(t1.x), "+v"(t2.x), "+v"(t2.w));
    intersect_triangle_rows<false>(t0, t1, t2, 0u, o, d, hit);
}

// ---- the node visit of util/bvh.hlsl:150-176, shared by EVERY traversal variant in this file --------------------------
// cwbvh_pick_child: take the nearest hit child off the node group, push what is left, return the child's node index.
template <bool STATS, class ST>
PT_DEV uint32_t cwbvh_pick_child(uint2& ng, uint32_t octinv4, uint32_t& sp, bool& overflow, ST& st, Counters& cn)
{
    const uint32_t mask = ng.y;
    const uint32_t childBitIndex = 31u - (uint32_t)__clz((int)mask);
    const uint32_t childNodeBaseIndex = ng.x;
    ng.y &= ~(1u << childBitIndex);
    if (ng.y > 0x00FFFFFFu) {
        stack_push(st, sp, ng);
        if (sp >= PT_BVH_STACK_SIZE) overflow = true;
        sp++;
        if (STATS) cn.maxStack = sp > cn.maxStack ? sp : cn.maxStack;
    }
    const uint32_t slotIndex = (childBitIndex - 24u) ^ (octinv4 & 255u);
    const uint32_t relativeIndex = (uint32_t)__popc(mask & ~(0xFFFFFFFFu << slotIndex));
    return childNodeBaseIndex + relativeIndex;
}
// cwbvh_apply_node: test the node's eight children and form the new node / triangle groups.
template <bool STATS>
PT_DEV void cwbvh_apply_node(v3 o, v3 invDir, uint32_t octinv4, float tmax, uint4 n0, uint4 n1, uint4 n2, uint4 n3, uint4 n4,
                             uint2& ng, uint2& tg, Counters& cn)
{
    const uint32_t hitmask = cwbvh_node_hitmask(o, invDir, octinv4, tmax, n0, n1, n2, n3, n4);
    if (STATS) cn.nodeVisits++;
    ng.x = n1.x;
    ng.y = (hitmask & 0xFF000000u) | (n0.w >> 24);
    tg.x = n1.y;
    tg.y = hitmask & 0x00FFFFFFu;
}
// cwbvh_visit_node = pick + fetch (five 16-byte rows, requested together) + apply.  nodeOffset: HAS_TLAS instance base.
template <bool STATS, class ST>
PT_DEV void cwbvh_visit_node(const uint4* __restrict__ nodes, uint32_t nodeOffset, v3 o, v3 invDir, uint32_t octinv4, float tmax,
                             uint2& ng, uint2& tg, uint32_t& sp, bool& overflow, ST& st, Counters& cn)
{
    const uint32_t childNodeIndex = cwbvh_pick_child<STATS>(ng, octinv4, sp, overflow, st, cn);
    // 32-bit byte offset from the (wave-uniform) base: the loads take base + offset themselves (SGPR base, VGPR offset) instead of a
    // 64-bit address formed with a quarter-rate v_mad_u64_u32.  PTSetScene refuses node arrays of 4 GiB and more.
    // (index * 80 as ((index << 2) + index) << 4: written as a product the compiler emits v_mul_lo_u32, quarter rate again; the empty
    // asm hides that the two operands of the shift-add are the same value)
    const uint32_t nodeIndex = nodeOffset + childNodeIndex;
    uint32_t nodeIndexCopy = nodeIndex;
    asm("" : "+v"(nodeIndexCopy));
    const uint4* np = (const uint4*)((const char*)nodes + (uint32_t)(((nodeIndexCopy << 2) + nodeIndex) << 4));
    uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
    cwbvh_apply_node<STATS>(o, invDir, octinv4, tmax, n0, n1, n2, n3, n4, ng, tg, cn);
}

// util/bvh.hlsl:126-199 for one lane.  ANYHIT = shadow ray: stops at the first accepted triangle
// (ShadowRayIntersect only uses the boolean, util/bvh.hlsl:228-233, so the result is identical).
//
// A ray with a NaN in its origin or direction can never be accepted by IntersectTriangle (every product
// with the NaN poisons `a` or `u`, and all its comparisons are false), but under HLSL/IEEE min-max
// semantics its slab test passes for EVERY child (cmin = 0, cmax = tmax), so the reference walks the whole
// tree for it.  Such rays do occur: RandomFloat returns exactly 1.0 for 128 of 2^32 states, which makes
// RandomCosineHemisphere return the zero vector and normalize() turn it into NaN (util/random.hlsl:34-41,
// util/light.hlsl:141) -- about two NEE rays per 1080p/8spp frame.  One lane visiting 45k nodes + 250k
// triangles stalls its whole kernel for ~80 ms, so the miss is returned immediately (same result).
template <bool STATS, class ST>
PT_DEV void traverse_cwbvh(const DScene& S, v3 o, v3 d, bool anyHit, TraceHit& hit, ST& st, Counters& cn)
{
    if (pt_isnan(o.x) || pt_isnan(o.y) || pt_isnan(o.z) || pt_isnan(d.x) || pt_isnan(d.y) || pt_isnan(d.z)) return;
    v3 invDir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t octinv4 = (7u - ((d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u))) * 0x1010101u;
    uint32_t sp = 0;
    uint2 ng = make_uint2(0u, 0x80000000u);
    uint2 tg = make_uint2(0u, 0u);
    bool overflow = false;
    while (true) {
        if (ng.y > 0x00FFFFFFu) {
            cwbvh_visit_node<STATS>(S.nodes, 0u, o, invDir, octinv4, hit.t, ng, tg, sp, overflow, st, cn);
        } else {
            tg = ng;
            ng = make_uint2(0u, 0u);
        }
        while (tg.y != 0u) {
            uint32_t triangleIndex = 31u - (uint32_t)__clz((int)tg.y);
            uint32_t triAddr = tg.x + triangleIndex * 3u;
            intersect_triangle(S.tris, triAddr, o, d, hit);
            if (STATS) cn.triTests++;
            tg.y -= 1u << triangleIndex;
            if (anyHit && hit.t < PT_FAR_PLANE) { tg.y = 0u; ng.y = 0u; sp = 0u; }
        }
        if (ng.y <= 0x00FFFFFFu) {
            if (sp > 0u) { --sp; ng = stack_pop(st, sp); }
            else break;
        }
    }
    if (STATS && overflow) cn.overflows++;
}

// ------------------------------------------------------------------------------------------
// HAS_TLAS variant.  util/tlas.hlsl:22-59, 129-332: a 2-wide TLAS over instances, each instance a CWBVH walked in the
// instance's LOCAL space (direction not re-normalised).  Reference quirks kept: triangles accept distance > 0 (not
// 1e-4); after an instance produced a hit, hit.distance becomes the WORLD-space length and later instances compare
// their local parameters against it; the attribute record is fetched once per instance that improved the hit.
// ------------------------------------------------------------------------------------------
struct HitRecord {
    TraceHit h;          // t, u, v, triIndex
    v3 pos;              // HAS_TLAS: world-space hit position (tlas.hlsl:216)
    uint32_t inst;       // HAS_TLAS: instance that owns the hit
};

PT_DEV v4 mul44c(float4 c0, float4 c1, float4 c2, float4 c3, v4 v)   // Matrix4x4 given as its four columns
{
    v4 r;
    r.x = c0.x * v.x + c1.x * v.y + c2.x * v.z + c3.x * v.w;
    r.y = c0.y * v.x + c1.y * v.y + c2.y * v.z + c3.y * v.w;
    r.z = c0.z * v.x + c1.z * v.y + c2.z * v.z + c3.z * v.w;
    r.w = c0.w * v.x + c1.w * v.y + c2.w * v.z + c3.w * v.w;
    return r;
}

PT_DEV bool intersect_triangle_inst(const float4* __restrict__ tris, uint32_t triAddr, uint32_t attrOffset, v3 o, v3 d, TraceHit& hit)
{
    const float4* tp = tris + triAddr;                    // one address, rows at immediate offsets 0 / 16 / 32
    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
    asm volatile("" : "+v"(t0.x), "+v"(t1.x), "+v"(t2.x), "+v"(t2.w));
    return intersect_triangle_rows<true>(t0, t1, t2, attrOffset, o, d, hit);
}

// tlas.hlsl:129-234 for one instance.  Returns hit.distance < FAR_PLANE; sets stopNow for an occluded shadow ray.
template <bool STATS, class ST>
PT_DEV bool traverse_instance(const DScene& S, v3 wo, v3 wd, uint32_t instIndex, bool isShadow, HitRecord& rec, ST& st, Counters& cn,
                              bool& stopNow)
{
    const float4* ip = S.instances + (size_t)instIndex * 9;
    const float4 w0 = ip[4], w1 = ip[5], w2 = ip[6], w3 = ip[7], ints = ip[8];
    if (STATS) cn.instanceVisits++;
    const uint32_t nodeOffset = pt_asuint(ints.x), triOffset = pt_asuint(ints.y), attrOffset = pt_asuint(ints.z);
    v4 lo4 = mul44c(w0, w1, w2, w3, v4{wo.x, wo.y, wo.z, 1.0f});
    v4 ld4 = mul44c(w0, w1, w2, w3, v4{wd.x, wd.y, wd.z, 0.0f});
    const v3 o = mk3(lo4.x, lo4.y, lo4.z), d = mk3(ld4.x, ld4.y, ld4.z);
    v3 invDir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t octinv4 = (7u - ((d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u))) * 0x1010101u;
    bool hitFound = false, overflow = false;
    uint32_t sp = 0;
    uint2 ng = make_uint2(0u, 0x80000000u), tg = make_uint2(0u, 0u);
    while (true) {
        if (ng.y > 0x00FFFFFFu) {
            cwbvh_visit_node<STATS>(S.nodes, nodeOffset, o, invDir, octinv4, rec.h.t, ng, tg, sp, overflow, st, cn);
        } else {
            tg = ng;
            ng = make_uint2(0u, 0u);
        }
        while (tg.y != 0u) {
            uint32_t triangleIndex = 31u - (uint32_t)__clz((int)tg.y);
            uint32_t triAddr = tg.x + triangleIndex * 3u;
            hitFound = intersect_triangle_inst(S.tris, triOffset + triAddr, attrOffset, o, d, rec.h) | hitFound;
            if (STATS) cn.triTests++;
            tg.y -= 1u << triangleIndex;
            if (isShadow && hitFound) { stopNow = true; tg.y = 0u; ng.y = 0u; sp = 0u; }
        }
        if (ng.y <= 0x00FFFFFFu) {
            if (sp > 0u) { --sp; ng = stack_pop(st, sp); }
            else break;
        }
    }
    if (STATS && overflow) cn.overflows++;
    if (!isShadow && hitFound) {
        if (STATS) cn.attrFetches++;
        const float4 l0 = ip[0], l1 = ip[1], l2 = ip[2], l3 = ip[3];
        v3 lp = o + rec.h.t * d;
        v4 wp = mul44c(l0, l1, l2, l3, v4{lp.x, lp.y, lp.z, 1.0f});
        rec.pos = mk3(wp.x, wp.y, wp.z);
        rec.h.t = length3(rec.pos - wo);
        rec.inst = instIndex;
    }
    return rec.h.t < PT_FAR_PLANE;
}

// tlas.hlsl:236-332
template <bool STATS, class ST>
PT_DEV bool traverse_tlas(const DScene& S, v3 O, v3 dir, bool isShadow, HitRecord& rec, ST& st, Counters& cn)
{
    if (pt_isnan(O.x) || pt_isnan(O.y) || pt_isnan(O.z) || pt_isnan(dir.x) || pt_isnan(dir.y) || pt_isnan(dir.z)) return false;
    const v3 D = normalize3(dir);
    const v3 rD = mk3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    bool hitFound = false;
    uint32_t stack[PT_BVH_STACK_SIZE];
    uint32_t nodeIndex = 0u, sp = 0u;
    const float* T = S.tlas;
    while (true) {
        const float4* np = (const float4*)(T + (size_t)nodeIndex * 16u);
        const float4 a = np[0], b = np[1], c = np[2], e = np[3];
        if (STATS) cn.tlasNodeVisits++;
        const uint32_t instanceCount = pt_asuint(c.w);
        if (instanceCount == 0u) {
            uint32_t left = pt_asuint(a.w), right = pt_asuint(b.w);
            v3 t1a = (mk3(a.x, a.y, a.z) - O) * rD, t2a = (mk3(b.x, b.y, b.z) - O) * rD;
            float tmina = pt_max(pt_max(pt_max(pt_min(t1a.x, t2a.x), pt_min(t1a.y, t2a.y)), pt_min(t1a.z, t2a.z)), 0.0f);
            float tmaxa = pt_min(pt_min(pt_min(pt_max(t1a.x, t2a.x), pt_max(t1a.y, t2a.y)), pt_max(t1a.z, t2a.z)), rec.h.t);
            float dist1 = tmina > tmaxa ? PT_FAR_PLANE : tmina;
            v3 t1b = (mk3(c.x, c.y, c.z) - O) * rD, t2b = (mk3(e.x, e.y, e.z) - O) * rD;
            float tminb = pt_max(pt_max(pt_max(pt_min(t1b.x, t2b.x), pt_min(t1b.y, t2b.y)), pt_min(t1b.z, t2b.z)), 0.0f);
            float tmaxb = pt_min(pt_min(pt_min(pt_max(t1b.x, t2b.x), pt_max(t1b.y, t2b.y)), pt_max(t1b.z, t2b.z)), rec.h.t);
            float dist2 = tminb > tmaxb ? PT_FAR_PLANE : tminb;
            if (dist1 > dist2) {
                float h = dist1; dist1 = dist2; dist2 = h;
                uint32_t t = left; left = right; right = t;
            }
            if (dist1 == PT_FAR_PLANE) {
                if (sp > 0u) nodeIndex = stack[--sp];
                else break;
            } else {
                nodeIndex = left;
                if (dist2 != PT_FAR_PLANE) {
                    if (sp < PT_BVH_STACK_SIZE) stack[sp] = right;
                    sp++;
                }
            }
        }
        if (instanceCount > 0u) {
            const uint32_t firstInstance = pt_asuint(e.w);
            for (uint32_t i = 0; i < instanceCount; ++i) {
                const uint32_t instanceIndex = pt_asuint(T[S.tlasIndexOffset + firstInstance + i]);
                bool stopNow = false;
                hitFound = traverse_instance<STATS>(S, O, dir, instanceIndex, isShadow, rec, st, cn, stopNow) | hitFound;
                if (stopNow) return true;
            }
            if (sp > 0u) nodeIndex = stack[--sp];
            else break;
        }
    }
    return hitFound;
}

// ------------------------------------------------------------------------------------------
// surface hit record (util/common.hlsl:173-193) and hit attribute fetch (util/bvh.hlsl:201-212)
// ------------------------------------------------------------------------------------------
struct SurfHit {
    v3 position;
    float distance;
    v3 normal;
    v3 ffnormal;
    v2 uv;
    uint32_t triIndex;
    int32_t materialIndex;
    uint32_t isLight;          // intersectType == INTERSECT_LIGHT
};

PT_DEV v3 interp3(float bu, float bv, v3 a0, v3 a1, v3 a2) { return a0 * (1.0f - bu - bv) + a1 * bu + a2 * bv; }

template <bool STATS>
PT_DEV void fetch_hit_attributes(const DScene& S, v3 o, v3 d, const TraceHit& th, SurfHit& hit, Counters& cn)
{
    const float4* a = S.attrs + (size_t)th.triIndex * 8;
    float4 n0 = a[0], n1 = a[1], n2 = a[2], uv01 = a[6], uv2m = a[7];
    if (STATS) cn.attrFetches++;
    hit.distance = th.t;
    hit.triIndex = th.triIndex;
    hit.position = o + th.t * d;
    hit.normal = normalize3(interp3(th.u, th.v, mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), mk3(n2.x, n2.y, n2.z)));
    hit.ffnormal = dot3(hit.normal, d) <= 0.0f ? hit.normal : -hit.normal;
    float w = 1.0f - th.u - th.v;
    hit.uv.x = uv01.x * w + uv01.z * th.u + uv2m.x * th.v;
    hit.uv.y = uv01.y * w + uv01.w * th.u + uv2m.y * th.v;
    hit.materialIndex = (int32_t)pt_asuint(uv2m.z);
    hit.isLight = 0u;
}

// tlas.hlsl:208-229 for the instance that owns the final hit (the per-instance fetches were counted during traversal)
PT_DEV void fetch_hit_attributes_tlas(const DScene& S, v3 d, const HitRecord& rec, SurfHit& hit)
{
    const float4* a = S.attrs + (size_t)rec.h.triIndex * 8;
    float4 n0 = a[0], n1 = a[1], n2 = a[2], uv01 = a[6], uv2m = a[7];
    const float4* ip = S.instances + (size_t)rec.inst * 9;
    const float4 w0 = ip[4], w1 = ip[5], w2 = ip[6], ints = ip[8];
    hit.distance = rec.h.t;
    hit.triIndex = rec.h.triIndex;
    hit.position = rec.pos;
    float w = 1.0f - rec.h.u - rec.h.v;
    hit.uv.x = uv01.x * w + uv01.z * rec.h.u + uv2m.x * rec.h.v;
    hit.uv.y = uv01.y * w + uv01.w * rec.h.u + uv2m.y * rec.h.v;
    v3 n = normalize3(interp3(rec.h.u, rec.h.v, mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), mk3(n2.x, n2.y, n2.z)));
    // mul(float4(normal, 0), worldToLocal): row vector times matrix = dot with each column
    v3 wn = mk3(n.x * w0.x + n.y * w0.y + n.z * w0.z + 0.0f * w0.w,
                n.x * w1.x + n.y * w1.y + n.z * w1.z + 0.0f * w1.w,
                n.x * w2.x + n.y * w2.y + n.z * w2.z + 0.0f * w2.w);
    hit.normal = normalize3(wn);
    hit.ffnormal = dot3(hit.normal, d) <= 0.0f ? hit.normal : -hit.normal;
    hit.materialIndex = (int32_t)pt_asuint(ints.w);
    hit.isLight = 0u;
}

// ------------------------------------------------------------------------------------------
// analytic lights.  util/intersect.hlsl:6-54
// ------------------------------------------------------------------------------------------
struct DLight { v3 position; uint32_t type; v3 emission; float range; v3 u; float area; v3 v; };

PT_DEV DLight load_light(const DScene& S, int i)
{
    const float4* p = S.lights + (size_t)i * 4;
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    DLight l;
    l.position = mk3(a.x, a.y, a.z); l.type = pt_asuint(a.w);
    l.emission = mk3(b.x, b.y, b.z); l.range = b.w;
    l.u = mk3(c.x, c.y, c.z); l.area = c.w;
    l.v = mk3(d.x, d.y, d.z);
    return l;
}

// Per-light constants.  The shader derives these from the light record at EVERY use (util/intersect.hlsl:30-53 per light per
// closest hit, util/light.hlsl per light sample): a cross product, two normalisations and seven divisions whose operands
// are scene constants -- ~115 VALU instructions per light per shading step, identical in all 64 lanes.  PTSetScene evaluates
// the same expressions once per light with the same device code (pt_derive_lights), so every value is the one the per-call
// evaluation would produce, bit for bit.
//   row 0: N = normalize(cross(u, v)) (rectangle) / normalize(u) (spot), w = dot(N, position)
//   row 1: u / dot(u, u)      row 2: v / dot(v, v)      row 3: normalize(N)  (the shader normalises the unit normal again)
PT_DEV void derive_light_rows(const DLight& light, float4 rows[4])
{
    v3 n = mk3(0.0f), u = mk3(0.0f), v = mk3(0.0f), nn = mk3(0.0f);
    float planeW = 0.0f;
    if (light.type == PT_LIGHT_TYPE_RECTANGLE) {
        n = normalize3(cross3(light.u, light.v));
        planeW = dot3(n, light.position);
        u = light.u / dot3(light.u, light.u);
        v = light.v / dot3(light.v, light.v);
        nn = normalize3(n);
    } else if (light.type == PT_LIGHT_TYPE_SPOT) {
        n = normalize3(light.u);
        nn = normalize3(n);
    }
    rows[0] = make_float4(n.x, n.y, n.z, planeW);
    rows[1] = make_float4(u.x, u.y, u.z, 0.0f);
    rows[2] = make_float4(v.x, v.y, v.z, 0.0f);
    rows[3] = make_float4(nn.x, nn.y, nn.z, 0.0f);
}

// A WAVE-UNIFORM read of scene constants (the light records every lane of a wave walks through in the same order): through the
// constant address space the compiler emits scalar loads (s_load_dwordx4 into SGPRs, served by the scalar cache) instead of 64
// identical lane requests on the vector-memory path -- which costs a broadcast as much as a gather (~55 CU-cycles per instruction,
// tools/microbench/gather16.hip) and parks the result in VGPRs.  Round 3: the light loop of intersect_lights issued ~5 such loads per
// light and shading step, ~6 x 10^8 lane requests per Sponza-class pass, a tenth of all vector-memory requests.
#ifndef PT_UNIFORM_LIGHTS
#define PT_UNIFORM_LIGHTS 0          // scenes with at most this many lights read all of them per wave (scalar loads) in nee_prepare_light.
                                     // Measured with 4 on the two-light Sponza-class scene: 5,088 against 5,102 Mrays/s with 0 (the per-lane gather): off
#endif
typedef float pt_vec4 __attribute__((ext_vector_type(4)));
PT_DEV float4 pt_uniform_load(const float4* p, size_t index)          // p + index must be the same address in every lane of the wave
{
    const __attribute__((address_space(4))) pt_vec4* q = (const __attribute__((address_space(4))) pt_vec4*)(uintptr_t)p;
    const pt_vec4 v = q[index];
    return make_float4(v.x, v.y, v.z, v.w);
}

template <bool STATS>
PT_DEV void intersect_lights(const DScene& S, v3 o, v3 d, SurfHit& hit, Counters& cn)
{
    if (!S.hasLights) return;
    for (int i = 0; i < S.lightCount; ++i) {
        const float4 la = pt_uniform_load(S.lights, (size_t)i * 4);                  // i is wave-uniform
        DLight light;
        light.position = mk3(la.x, la.y, la.z); light.type = pt_asuint(la.w);
        if (STATS) cn.lightFetches++;
        if (light.type == PT_LIGHT_TYPE_RECTANGLE) {
            const float4 c0 = pt_uniform_load(S.lightConst, (size_t)i * 4), c1 = pt_uniform_load(S.lightConst, (size_t)i * 4 + 1),
                         c2 = pt_uniform_load(S.lightConst, (size_t)i * 4 + 2);
            v3 normal = mk3(c0.x, c0.y, c0.z);                   // normalize(cross(light.u, light.v))
            float planeW = c0.w;                                 // dot(normal, light.position)
            v3 u = mk3(c1.x, c1.y, c1.z);                        // light.u / dot(light.u, light.u)
            v3 v = mk3(c2.x, c2.y, c2.z);                        // light.v / dot(light.v, light.v)
            // RectIntersect :6-27
            float dt = dot3(d, normal);
            float t = (planeW - dot3(normal, o)) / dt;
            float res = PT_FAR_PLANE;
            if (t > PT_EPSILON) {
                v3 p = o + d * t;
                v3 vi = p - light.position;
                float a1 = dot3(u, vi);
                if (a1 >= 0.0f && a1 <= 1.0f) {
                    float a2 = dot3(v, vi);
                    if (a2 >= 0.0f && a2 <= 1.0f) res = t;
                }
            }
            if (res > 0.0f && res < hit.distance && dot3(normal, d) < 0.0f) {
                hit.distance = res;
                hit.position = o + res * d;
                hit.normal = normal;
                hit.ffnormal = dot3(hit.normal, d) <= 0.0f ? hit.normal : -hit.normal;
                hit.triIndex = (uint32_t)i;
                hit.isLight = 1u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// textures + material.  util/texture.hlsl:6-76, util/material.hlsl:8-138
// ------------------------------------------------------------------------------------------
struct Material {
    v3 baseColor; float opacity;
    v3 emission; float alphaMode;
    float alphaCutoff, anisotropic, metallic, roughness;
    float subsurface, specularTint, sheen, sheenTint;
    float clearcoat, clearcoatRoughness, specTrans, ior;
    float ax, ay, eta, occlusion;
};

template <bool STATS>
PT_DEV v4 texture_pixel(const DScene& S, uint32_t off, uint32_t width, uint32_t height, uint32_t x, uint32_t y, Counters& cn)
{
    x = x < width - 1u ? x : width - 1u;
    y = y < height - 1u ? y : height - 1u;
    uint32_t p = S.tex[off + (y * width + x)];
    if (STATS) cn.texelFetches++;
    return {pt_unorm8(p), pt_unorm8(p >> 8), pt_unorm8(p >> 16), pt_unorm8(p >> 24)};    // == channel / 255.0f (include/ptmi_math.h)
}

template <bool STATS>
PT_DEV v4 sample_texture(const DScene& S, int32_t textureIndex, v2 uv, Counters& cn)      // linearSample == true at every call site
{
    if (textureIndex < 0) return {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t dsc = (uint32_t)textureIndex * 4u;
    uint32_t width = S.tex[dsc + 0], height = S.tex[dsc + 1], offset = S.tex[dsc + 2];
    if (STATS) cn.texDescFetches++;
    // util/texture.hlsl:41-48: the reference's +-1 wrap loops in closed form (bit-identical, and bounded: ptmi_math.h)
    const float u = pt_wrap01(uv.x), v = pt_wrap01(uv.y);
    float tu = u * ((float)width - 1.0f);
    float tv = v * ((float)height - 1.0f);
    uint32_t tx = pt_f2u(tu), ty = pt_f2u(tv);
    v4 p1 = texture_pixel<STATS>(S, offset, width, height, tx, ty, cn);
    float uF = tu - (float)tx, vF = tv - (float)ty;
    v4 p2 = texture_pixel<STATS>(S, offset, width, height, tx + 1u, ty, cn);
    v4 p3 = texture_pixel<STATS>(S, offset, width, height, tx, ty + 1u, cn);
    v4 p4 = texture_pixel<STATS>(S, offset, width, height, tx + 1u, ty + 1u, cn);
    return lerp4(lerp4(p1, p2, uF), lerp4(p3, p4, uF), vF);
}

// The four texture lookups of a material (base colour, emission, metallic-roughness, occlusion) as TWO memory round trips for
// all of them -- every wanted descriptor, then every texel -- instead of two per texture one after the other (PT_TEX_BATCH).
// Same arithmetic per texture as sample_texture (same counters), only the order of the requests differs.
#ifndef PT_TEX_BATCH
#define PT_TEX_BATCH 1
#endif
template <bool STATS>
PT_DEV void sample_textures4(const DScene& S, const bool want[4], const int32_t index[4], const v2 uvs[4], v4 out[4], Counters& cn)
{
    uint32_t width[4], height[4], offset[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        width[k] = 1u; height[k] = 1u; offset[k] = 0u;
        if (want[k]) {
            const uint32_t dsc = (uint32_t)index[k] * 4u;
            width[k] = S.tex[dsc + 0]; height[k] = S.tex[dsc + 1]; offset[k] = S.tex[dsc + 2];
            if (STATS) cn.texDescFetches++;
        }
    }
    asm volatile("" : "+v"(width[0]), "+v"(width[1]), "+v"(width[2]), "+v"(width[3]));           // all descriptors requested before the first is used
    uint32_t px[4][4];
    float uF[4], vF[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uF[k] = 0.0f; vF[k] = 0.0f;
        px[k][0] = px[k][1] = px[k][2] = px[k][3] = 0u;
        if (want[k]) {
            const float u = pt_wrap01(uvs[k].x), v = pt_wrap01(uvs[k].y);
            const float tu = u * ((float)width[k] - 1.0f);
            const float tv = v * ((float)height[k] - 1.0f);
            const uint32_t tx = pt_f2u(tu), ty = pt_f2u(tv);
            uF[k] = tu - (float)tx; vF[k] = tv - (float)ty;
            const uint32_t w1 = width[k] - 1u, h1 = height[k] - 1u;
            const uint32_t x0 = tx < w1 ? tx : w1, x1 = tx + 1u < w1 ? tx + 1u : w1;
            const uint32_t y0 = ty < h1 ? ty : h1, y1 = ty + 1u < h1 ? ty + 1u : h1;
            const uint32_t* base = S.tex + offset[k];
            px[k][0] = base[y0 * width[k] + x0];
            px[k][1] = base[y0 * width[k] + x1];
            px[k][2] = base[y1 * width[k] + x0];
            px[k][3] = base[y1 * width[k] + x1];
            if (STATS) cn.texelFetches += 4u;
        }
    }
    asm volatile("" : "+v"(px[0][0]), "+v"(px[1][0]), "+v"(px[2][0]), "+v"(px[3][0]));          // all texels requested before the first is used
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out[k] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (want[k]) {
            v4 p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = {pt_unorm8(px[k][q]), pt_unorm8(px[k][q] >> 8), pt_unorm8(px[k][q] >> 16), pt_unorm8(px[k][q] >> 24)};
            out[k] = lerp4(lerp4(p[0], p[1], uF[k]), lerp4(p[2], p[3], uF[k]), vF[k]);
        }
    }
}

template <bool STATS>
PT_DEV Material get_material(const DScene& S, int32_t materialIndex, v3 rayDir, const SurfHit& hit, Counters& cn)
{
    const float4* mp = S.materials + (size_t)materialIndex * 8;
    float4 d1 = mp[0], d2 = mp[1], d3 = mp[2], d4 = mp[3], d5 = mp[4], d6t1 = mp[5], t2 = mp[6], tr = mp[7];
    if (STATS) cn.materialFetches++;
    const bool T = S.hasTextures != 0u;
    v2 uv = hit.uv;
    v4 bco = {d1.x, d1.y, d1.z, d1.w};
#if PT_TEX_BATCH
    const bool want[4] = {T && !(d6t1.z < 0.0f), T && !(t2.y < 0.0f), T && !(d6t1.w < 0.0f), T && !(t2.z < 0.0f)};
    const int32_t tindex[4] = {pt_f2i(d6t1.z), pt_f2i(t2.y), pt_f2i(d6t1.w), pt_f2i(t2.z)};
    const v2 uvs[4] = {{uv.x * tr.x + tr.z, uv.y * tr.y + tr.w}, uv, uv, uv};
    v4 texel[4];
    if (T) sample_textures4<STATS>(S, want, tindex, uvs, texel, cn);
#endif
    if (T && !(d6t1.z < 0.0f)) {                                          // GetBaseColorOpacity :56-69
#if PT_TEX_BATCH
        v4 px = texel[0];
#else
        v2 tuv = {uv.x * tr.x + tr.z, uv.y * tr.y + tr.w};
        v4 px = sample_texture<STATS>(S, pt_f2i(d6t1.z), tuv, cn);
#endif
        bco = px * bco;
    }
    Material m;
    m.baseColor = mk3(bco.x, bco.y, bco.z);
    m.opacity = bco.w;
    m.alphaMode = d4.x;
    m.alphaCutoff = d2.w;
    m.emission = mk3(d2.x, d2.y, d2.z);
    if (T && !(t2.y < 0.0f)) {                                            // GetEmission :8-21
#if PT_TEX_BATCH
        v4 px = texel[1];
#else
        v4 px = sample_texture<STATS>(S, pt_f2i(t2.y), uv, cn);
#endif
        m.emission = mk3(px.x, px.y, px.z);
    }
    float metallic = d3.x, rough = d3.y;
    if (T && !(d6t1.w < 0.0f)) {                                          // GetMetallicRoughness :40-54
#if PT_TEX_BATCH
        v4 px = texel[2];
#else
        v4 px = sample_texture<STATS>(S, pt_f2i(d6t1.w), uv, cn);
#endif
        metallic = px.z;
        rough = px.y * px.y;
    }
    m.metallic = metallic;
    m.roughness = pt_max(rough, 0.001f);
    m.subsurface = d5.z;
    m.specularTint = d4.w;
    m.sheen = d5.x;
    m.sheenTint = d5.y;
    m.clearcoat = d5.w;
    m.clearcoatRoughness = pt_lerp(0.1f, 0.001f, d6t1.x);
    m.specTrans = 1.0f - pt_saturate(bco.w);
    m.ior = pt_clamp(d3.w, 1.001f, 2.0f);
    m.anisotropic = pt_clamp(d4.y, -0.9f, 0.9f);
    m.occlusion = 1.0f;
    if (T && !(t2.z < 0.0f)) {                                            // GetOcclusion :71-82
#if PT_TEX_BATCH
        float px = texel[3].x;
#else
        float px = sample_texture<STATS>(S, pt_f2i(t2.z), uv, cn).x;
#endif
        m.occlusion = 1.0f + (px - 1.0f);
    }
    float aspect = pt_sqrt(1.0f - m.anisotropic * 0.9f);
    m.ax = pt_max(0.001f, m.roughness / aspect);
    m.ay = pt_max(0.001f, m.roughness * aspect);
    m.eta = (dot3(rayDir, hit.normal) < 0.0f) ? 1.0f / m.ior : m.ior;
    return m;
}

// ------------------------------------------------------------------------------------------
// util/sampling.hlsl
// ------------------------------------------------------------------------------------------
PT_DEV float gtr1(float NDotH, float a)
{
    if (a >= 1.0f) return PT_INV_PI;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (PT_PI * pt_log(a2) * t);
}
PT_DEV v3 sample_gtr1(float rgh, float r1, float r2)
{
    float a = pt_max(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * PT_TWO_PI;
    float cosTheta = pt_sqrt((1.0f - pt_pow(a2, 1.0f - r2)) / (1.0f - a2));
    float sinTheta = pt_clamp(pt_sqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
    float sinPhi = pt_sin(phi);
    float cosPhi = pt_cos(phi);
    return mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
PT_DEV v3 sample_ggx_vndf(v3 V, float ax, float ay, float r1, float r2)
{
    v3 Vh = normalize3(mk3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * pt_rsqrt(lensq) : mk3(1, 0, 0);
    v3 T2 = cross3(Vh, T1);
    float r = pt_sqrt(r1);
    float phi = 2.0f * PT_PI * r2;
    float t1 = r * pt_cos(phi);
    float t2 = r * pt_sin(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * pt_sqrt(1.0f - t1 * t1) + s * t2;
    v3 Nh = t1 * T1 + t2 * T2 + pt_sqrt(pt_max(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize3(mk3(ax * Nh.x, ay * Nh.y, pt_max(0.0f, Nh.z)));
}
PT_DEV float gtr2_aniso(float NDotH, float HDotX, float HDotY, float ax, float ay)
{
    float a = HDotX / ax;
    float b = HDotY / ay;
    float c = a * a + b * b + NDotH * NDotH;
    return 1.0f / (PT_PI * ax * ay * c * c);
}
PT_DEV float smith_g(float NDotV, float alphaG)
{
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return (2.0f * NDotV) / (NDotV + pt_sqrt(a + b - a * b));
}
PT_DEV float smith_g_aniso(float NDotV, float VDotX, float VDotY, float ax, float ay)
{
    float a = VDotX * ax;
    float b = VDotY * ay;
    float c = NDotV;
    return (2.0f * NDotV) / (NDotV + pt_sqrt(a * a + b * b + c * c));
}
PT_DEV float schlick_weight(float u)
{
    float m = pt_clamp(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
PT_DEV float dielectric_fresnel(float cosThetaI, float eta)
{
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = pt_sqrt(pt_max(1.0f - sinThetaTSq, 0.0f));
    float rs = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    float rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}
PT_DEV v3 cosine_sample_hemisphere(float r1, float r2)
{
    float r = pt_sqrt(r1);
    float phi = PT_TWO_PI * r2;
    v3 d;
    d.x = r * pt_cos(phi);
    d.y = r * pt_sin(phi);
    d.z = pt_sqrt(pt_max(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return d;
}
PT_DEV float power_heuristic(float a, float b) { float t = a * a; return t / (b * b + t); }

// ------------------------------------------------------------------------------------------
// util/brdf.hlsl
// ------------------------------------------------------------------------------------------
struct Lobes { float dielectricWt, metalWt, glassWt, diffPr, dielectricPr, metalPr, glassPr, clearCtPr; };

// What the three BSDF evaluations of one shading step (environment NEE, light NEE, the sampled direction) derive from the
// MATERIAL alone -- util/brdf.hlsl TintColors -- and the frame of the face-forward normal that two of them share.  The
// shader recomputes both inside every call (brdf.hlsl:118-131, :229, :246); evaluated once per step with the same
// expressions the values are the same bits (~60 instructions per tint, ~30 per frame, 3 divisions each).  Affordable
// since the shade kernel no longer holds its load addresses in registers (pt_wavefront.hip shade_slot): 10 more VGPRs.
struct BsdfShared { float F0; v3 Csheen, Cspec0; };

PT_DEV void tint_colors(const Material& mat, float eta, float& F0, v3& Csheen, v3& Cspec0)
{
    float lum = luminance3(mat.baseColor);
    v3 ctint = lum > 0.0f ? mat.baseColor / lum : mk3(1.0f);
    F0 = (1.0f - eta) / (1.0f + eta);
    F0 *= F0;
    Cspec0 = F0 * lerp3(mk3(1.0f), ctint, mat.specularTint);
    Csheen = lerp3(mk3(1.0f), ctint, mat.sheenTint);
}

PT_DEV Lobes lobe_weights(const Material& mat, v3 Cspec0, float Vz)
{
    Lobes w;
    w.dielectricWt = (1.0f - mat.metallic) * (1.0f - mat.specTrans);
    w.metalWt = mat.metallic;
    w.glassWt = (1.0f - mat.metallic) * mat.specTrans;
    float schlickWt = schlick_weight(Vz);
    w.diffPr = w.dielectricWt * luminance3(mat.baseColor);
    w.dielectricPr = w.dielectricWt * luminance3(lerp3(Cspec0, mk3(1.0f), schlickWt));
    w.metalPr = w.metalWt * luminance3(lerp3(mat.baseColor, mk3(1.0f), schlickWt));
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

PT_DEV v3 eval_diffuse(const Material& mat, v3 Csheen, v3 V, v3 L, v3 H, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    float LDotH = dot3(L, H);
    float Rr = 2.0f * mat.roughness * LDotH * LDotH;
    float FL = schlick_weight(L.z);
    float FV = schlick_weight(V.z);
    float Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0f));
    float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
    float Fss90 = 0.5f * Rr;
    float Fss = pt_lerp(1.0f, Fss90, FL) * pt_lerp(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (L.z + V.z) - 0.5f) + 0.5f);
    float FH = schlick_weight(LDotH);
    v3 Fsheen = FH * mat.sheen * Csheen;
    pdf = L.z * PT_INV_PI;
    return PT_INV_PI * mat.baseColor * pt_lerp(Fd + Fretro, ss, mat.subsurface) + Fsheen;
}

PT_DEV v3 eval_microfacet_reflection(const Material& mat, v3 V, v3 L, v3 H, v3 F, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    float D = gtr2_aniso(H.z, H.x, H.y, mat.ax, mat.ay);
    float G1 = smith_g_aniso(pt_abs(V.z), V.x, V.y, mat.ax, mat.ay);
    float G2 = G1 * smith_g_aniso(pt_abs(L.z), L.x, L.y, mat.ax, mat.ay);
    pdf = G1 * D / (4.0f * V.z);
    return F * D * G2 / (4.0f * L.z * V.z);
}

// The three lobes that end in eval_microfacet_reflection (dielectric, metal, glass reflection) differ in F only: D, G1 and G2 are
// evaluated ONCE per BSDF evaluation (MicrofacetShared) instead of once per lobe -- the same expressions, so the same bits; a lane
// with 0 < metallic < 1 runs two of the lobes itself, and a wave with metals and dielectrics in it ran the ~120 instructions twice.
struct MicrofacetShared { float D, G1, G2; };
PT_DEV MicrofacetShared microfacet_shared(const Material& mat, v3 V, v3 L, v3 H)
{
    MicrofacetShared m;
    m.D = gtr2_aniso(H.z, H.x, H.y, mat.ax, mat.ay);
    m.G1 = smith_g_aniso(pt_abs(V.z), V.x, V.y, mat.ax, mat.ay);
    m.G2 = m.G1 * smith_g_aniso(pt_abs(L.z), L.x, L.y, mat.ax, mat.ay);
    return m;
}
PT_DEV v3 eval_microfacet_reflection_shared(const MicrofacetShared& m, v3 V, v3 L, v3 F, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    pdf = m.G1 * m.D / (4.0f * V.z);
    return F * m.D * m.G2 / (4.0f * L.z * V.z);
}

PT_DEV v3 eval_microfacet_refraction(const Material& mat, float eta, v3 V, v3 L, v3 H, v3 F, float& pdf)
{
    pdf = 0.0f;
    if (L.z >= 0.0f) return mk3(0.0f);
    float LDotH = dot3(L, H);
    float VDotH = dot3(V, H);
    float D = gtr2_aniso(H.z, H.x, H.y, mat.ax, mat.ay);
    float G1 = smith_g_aniso(pt_abs(V.z), V.x, V.y, mat.ax, mat.ay);
    float G2 = G1 * smith_g_aniso(pt_abs(L.z), L.x, L.y, mat.ax, mat.ay);
    float denom = LDotH + VDotH * eta;
    denom *= denom;
    float eta2 = eta * eta;
    float jacobian = pt_abs(LDotH) / denom;
    pdf = G1 * pt_max(0.0f, VDotH) * D * jacobian / V.z;
    v3 sq = mk3(pt_pow(mat.baseColor.x, 0.5f), pt_pow(mat.baseColor.y, 0.5f), pt_pow(mat.baseColor.z, 0.5f));
    return sq * (1.0f - F) * D * G2 * pt_abs(VDotH) * jacobian * eta2 / pt_abs(L.z * V.z);
}

PT_DEV v3 eval_clearcoat(const Material& mat, v3 V, v3 L, v3 H, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    float VDotH = dot3(V, H);
    float F = pt_lerp(0.04f, 1.0f, schlick_weight(VDotH));
    float D = gtr1(H.z, mat.clearcoatRoughness);
    float G = smith_g(L.z, 0.25f) * smith_g(V.z, 0.25f);
    float jacobian = 1.0f / (4.0f * VDotH);
    pdf = D * H.z * jacobian;
    return mk3(F) * D * G;
}

// util/brdf.hlsl:114-225.  V and L are world-space; onb = GetONB(N).
PT_DEV v3 eval_brdf_onb(const Material& mat, v3 V, v3 L, const Onb& onb, float& pdf, const BsdfShared& sh)
{
    pdf = 0.0f;
    v3 f = mk3(0.0f);
    V = onb_to_local(onb, V);
    L = onb_to_local(onb, L);
    v3 H;
    if (L.z > 0.0f) H = normalize3(L + V);
    else H = normalize3(L + V * mat.eta);
    if (H.z < 0.0f) H = -H;

    const v3 Csheen = sh.Csheen, Cspec0 = sh.Cspec0;     // tint_colors(mat, mat.eta), evaluated once per shading step
    const float F0 = sh.F0;
    Lobes w = lobe_weights(mat, Cspec0, V.z);

    bool refl = L.z * V.z > 0.0f;
    float tmpPdf = 0.0f;
    float VDotH = pt_abs(dot3(V, H));

    MicrofacetShared mf = {0.0f, 0.0f, 0.0f};
    if (refl && L.z > 0.0f && (w.dielectricPr > 0.0f || w.metalPr > 0.0f || w.glassPr > 0.0f)) mf = microfacet_shared(mat, V, L, H);
    if (w.diffPr > 0.0f && refl) {
        f = f + eval_diffuse(mat, Csheen, V, L, H, tmpPdf) * w.dielectricWt;
        pdf += tmpPdf * w.diffPr;
    }
    if (w.dielectricPr > 0.0f && refl) {
        float F = 0.0f;
        if (F0 != 1.0f && mat.ior != 0.0f) {
            float invEta = pt_rcp(mat.ior);
            float invF0 = 1.0f - F0;
            invF0 = pt_rcp(invF0);
            F = (dielectric_fresnel(VDotH, invEta) - F0) * invF0;
        }
        f = f + eval_microfacet_reflection_shared(mf, V, L, lerp3(Cspec0, mk3(1.0f), F), tmpPdf) * w.dielectricWt;
        pdf += tmpPdf * w.dielectricPr;
    }
    if (w.metalPr > 0.0f && refl) {
        v3 F = lerp3(mat.baseColor, mk3(1.0f), schlick_weight(VDotH));
        f = f + eval_microfacet_reflection_shared(mf, V, L, F, tmpPdf) * w.metalWt;
        pdf += tmpPdf * w.metalPr;
    }
    if (w.glassPr > 0.0f) {
        float F = dielectric_fresnel(VDotH, mat.eta);
        if (refl) {
            f = f + eval_microfacet_reflection_shared(mf, V, L, mk3(F), tmpPdf) * w.glassWt;
            pdf += tmpPdf * w.glassPr * F;
        } else {
            f = f + eval_microfacet_refraction(mat, mat.eta, V, L, H, mk3(F), tmpPdf) * w.glassWt;
            pdf += tmpPdf * w.glassPr * (1.0f - F);
        }
    }
    if (w.clearCtPr > 0.0f && refl) {
        f = f + eval_clearcoat(mat, V, L, H, tmpPdf) * 0.25f * mat.clearcoat;
        pdf += tmpPdf * w.clearCtPr;
    }
    f = f * mat.occlusion;
    return f * pt_abs(L.z);
}

PT_DEV v3 eval_brdf(const Material& mat, v3 V, v3 N, v3 L, float& pdf, const BsdfShared& sh)     // util/brdf.hlsl:227-238
{
    Onb onb = make_onb(N);
    return eval_brdf_onb(mat, V, L, onb, pdf, sh);
}

// onb = make_onb(N) of the caller (shared with the environment NEE evaluation, which uses the same normal)
PT_DEV v3 sample_brdf(const Material& mat, v3 V, const Onb& onb, v3& L, float& pdf, uint32_t& rng, const BsdfShared& sh) // util/brdf.hlsl:240-340
{
    pdf = 0.0f;
    float r1 = rnd(rng);
    float r2 = rnd(rng);
    V = onb_to_local(onb, V);
    Lobes w = lobe_weights(mat, sh.Cspec0, V.z);
    float cdf0 = w.diffPr;
    float cdf1 = cdf0 + w.dielectricPr;
    float cdf2 = cdf1 + w.metalPr;
    float cdf3 = cdf2 + w.glassPr;
    float r3 = rnd(rng);
    if (r3 < cdf0) {
        L = cosine_sample_hemisphere(r1, r2);
    } else if (r3 < cdf2) {
        v3 H = sample_ggx_vndf(V, mat.ax, mat.ay, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize3(reflect3(-V, H));
    } else if (r3 < cdf3) {
        v3 H = sample_ggx_vndf(V, mat.ax, mat.ay, r1, r2);
        float F = dielectric_fresnel(pt_abs(dot3(V, H)), mat.eta);
        if (H.z < 0.0f) H = -H;
        r3 = (r3 - cdf2) / (cdf3 - cdf2);
        if (r3 < F) L = normalize3(reflect3(-V, H));
        else L = normalize3(refract3(-V, H, mat.eta));
    } else {
        v3 H = sample_gtr1(mat.clearcoatRoughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize3(reflect3(-V, H));
    }
    L = onb_to_world(onb, L);
    V = onb_to_world(onb, V);
    return eval_brdf_onb(mat, V, L, onb, pdf, sh);
}

// util/random.hlsl:34-41
// GetONB(normal) is ~115 instructions (three normalisations and a division) and the shading step needs it three times: for the
// face-forward normal (sampling + environment evaluation), and for the NORMAL in RandomCosineHemisphere and in the light's EvalBRDF.
// On a front-facing hit normal and face-forward normal are the same three floats, hence the same basis: onb_of_normal hands out
// the basis already computed and only lanes whose hit is back-facing compute their own (a branch the wave skips otherwise).
PT_DEV Onb onb_of_normal(v3 normal, v3 ffnormal, const Onb& ffOnb)
{
    Onb o = ffOnb;
    if (!(normal.x == ffnormal.x && normal.y == ffnormal.y && normal.z == ffnormal.z)) o = make_onb(normal);
    return o;
}

PT_DEV v3 random_cosine_hemisphere(const Onb& onb, uint32_t& rng)     // onb = GetONB(normal)
{
    float theta = pt_acos(pt_sqrt(rnd(rng)));
    float phi = 2.0f * PT_PI * rnd(rng);
    return pt_sin(theta) * (pt_cos(phi) * onb.x + pt_sin(phi) * onb.y + pt_cos(theta) * onb.z);
}

// ------------------------------------------------------------------------------------------
// environment texture.  util/sky.hlsl:7-88
// ------------------------------------------------------------------------------------------

// EnvironmentTexture.SampleLevel(samplerEnvironmentTexture, uv, 0).rgb restated: bilinear, clamp addressing, texel
// centres at (i + 0.5) / size, memory row r at v = 1 - (r + 0.5) / H (include/ptmi_plugin.h, PTSceneDesc.envTexture)
PT_DEV v3 env_sample_level(const DScene& S, float u, float v)
{
    const float fx = u * (float)S.envW - 0.5f;
    const float fy = (1.0f - v) * (float)S.envH - 0.5f;
    const float x0f = pt_floor(fx), y0f = pt_floor(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    int32_t x0 = pt_f2i(x0f), y0 = pt_f2i(y0f);
    int32_t x1 = x0 < 2147483647 ? x0 + 1 : x0, y1 = y0 < 2147483647 ? y0 + 1 : y0;
    const int32_t mx = S.envW - 1, my = S.envH - 1;
    x0 = x0 < 0 ? 0 : (x0 > mx ? mx : x0); x1 = x1 < 0 ? 0 : (x1 > mx ? mx : x1);
    y0 = y0 < 0 ? 0 : (y0 > my ? my : y0); y1 = y1 < 0 ? 0 : (y1 > my ? my : y1);
    const float4 a = S.envTex[(size_t)y0 * S.envW + x0], b = S.envTex[(size_t)y0 * S.envW + x1];
    const float4 c = S.envTex[(size_t)y1 * S.envW + x0], d = S.envTex[(size_t)y1 * S.envW + x1];
    const v3 top = mk3(pt_lerp(a.x, b.x, ax), pt_lerp(a.y, b.y, ax), pt_lerp(a.z, b.z, ax));
    const v3 bot = mk3(pt_lerp(c.x, d.x, ax), pt_lerp(c.y, d.y, ax), pt_lerp(c.z, d.z, ax));
    return mk3(pt_lerp(top.x, bot.x, ay), pt_lerp(top.y, bot.y, ay), pt_lerp(top.z, bot.z, ay));
}

// BinarySearch :7-42: the row whose last CDF entry exceeds `value`, then the column inside that row
PT_DEV void env_binary_search(const DScene& S, float value, float& u, float& v)
{
    int32_t lower = 0, upper = S.envH - 1;
    while (lower < upper) {
        const int32_t mid = (lower + upper) >> 1;
        const int32_t idx = mid * S.envW + S.envW - 1;
        if (value < S.envCdf[idx]) upper = mid; else lower = mid + 1;
    }
    const int32_t y = lower < 0 ? 0 : (lower > S.envH - 1 ? S.envH - 1 : lower);
    lower = 0; upper = S.envW - 1;
    while (lower < upper) {
        const int32_t mid = (lower + upper) >> 1;
        const int32_t idx = y * S.envW + mid;
        if (value < S.envCdf[idx]) upper = mid; else lower = mid + 1;
    }
    const int32_t x = lower < 0 ? 0 : (lower > S.envW - 1 ? S.envW - 1 : lower);
    u = (float)x / (float)S.envW;
    v = (float)y / (float)S.envH;
}

// EvalEnvMap :44-64
PT_DEV v4 eval_env_map(const DScene& S, const PTFrameParams& P, v3 r, float intensity)
{
    const float theta = pt_acos(pt_clamp(r.y, -1.0f, 1.0f));
    const float r_atan = pt_atan2(r.z, r.x);
    float u = (PT_PI + r_atan) * PT_INV_TWO_PI + P.EnvironmentMapRotation;
    float v = (1.0f - theta * PT_INV_PI) + 0.0f;
    u = pt_fmod(u, 1.0f);
    v = pt_fmod(v, 1.0f);
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    const v3 color = env_sample_level(S, u, v);
    float pdf = luminance3(color) / S.envCdfSum;
    pdf = (pdf * (float)S.envW * (float)S.envH) / (PT_TWO_PI * PT_PI * pt_sin(theta));
    const v3 c = color * intensity;
    return {c.x, c.y, c.z, pdf};
}

// SampleEnvMap :66-88 (one RNG draw).  Returns (direction, pdf); `color` is the texel BEFORE any intensity.
PT_DEV v4 sample_env_map(const DScene& S, const PTFrameParams& P, v3& color, uint32_t& rng)
{
    const float r = rnd(rng) * S.envCdfSum;
    float u, v;
    env_binary_search(S, r, u, v);
    v = 1.0f - v;
    color = env_sample_level(S, u, v);
    float pdf = luminance3(color) / S.envCdfSum;
    u -= P.EnvironmentMapRotation;
    const float phi = u * PT_TWO_PI;
    const float theta = v * PT_PI;
    const float sinTheta = pt_sin(theta);
    if (sinTheta == 0.0f) pdf = 0.0f;
    return {-sinTheta * pt_cos(phi), pt_cos(theta), -sinTheta * pt_sin(phi),
            (pdf * (float)S.envW * (float)S.envH) / (PT_TWO_PI * PT_PI * sinTheta)};
}

// util/sky.hlsl:90-129
PT_DEV v4 sample_sky_radiance(const DScene& S, const PTFrameParams& P, v3 direction, uint32_t rayDepth)
{
    v4 radiance = {0.0f, 0.0f, 0.0f, 0.0f};
    float pdf = 1.0f / (4.0f * PT_PI);
    float intensity = 1.0f;
    if ((int32_t)rayDepth > 0) intensity = P.EnvironmentIntensity;
    if (P.EnvironmentMode == PT_SKY_MODE_ENVIRONMENT) {
        if (S.hasEnvTex) return eval_env_map(S, P, direction, intensity);          // EnvironmentSky :90-98
        v3 col = mk3(P.EnvironmentColor[0], P.EnvironmentColor[1], P.EnvironmentColor[2]) * intensity;
        radiance = {col.x, col.y, col.z, pdf};
    } else if (P.EnvironmentMode == PT_SKY_MODE_BASIC) {
        float a = pt_saturate(0.5f * (direction.y + 1.0f));
        v3 blue = mk3(pt_pow(0.5f, 2.2f), pt_pow(0.7f, 2.2f), pt_pow(1.0f, 2.2f));
        v3 col = (1.0f - a) * mk3(1.0f) + a * blue;
        col = col * intensity;
        radiance = {col.x, col.y, col.z, pdf};
    }
    return radiance;
}

// ------------------------------------------------------------------------------------------
// next-event estimation, split into "prepare" (all RNG draws + the BSDF value that would be added if
// the shadow ray is unoccluded) and the shadow ray itself, so that the shadow rays of a bounce can be
// traced together with the next bounce ray in one traversal pass.  util/light.hlsl:7-173.
// The BSDF is evaluated eagerly; a contribution is only ADDED when its ray is unoccluded, in the
// reference's order (environment first, then the analytic light), so sums are bit-identical.
// ------------------------------------------------------------------------------------------
struct NeeRay { v3 dir; v3 contribution; uint32_t valid; };

PT_DEV v3 nee_scatter_pos(const SurfHit& hit) { return hit.position + hit.normal * PT_EPSILON; }

// environment half of DirectLight (:123-160)
PT_DEV void nee_prepare_environment(const DScene& S, const PTFrameParams& P, v3 rayDir, const SurfHit& hit, const Material& mat,
                                    uint32_t& rng, NeeRay& out, const BsdfShared& sh, const Onb& ffOnb)
{
    out.valid = 0u;
    out.contribution = mk3(0.0f);
    if (P.EnvironmentMode != 0) return;
    v3 Li, lightDir;
    float lightPdf;
    if (S.hasEnvTex) {                                   // HAS_ENVIRONMENT_TEXTURE :125-141
        v4 dirPdf = sample_env_map(S, P, Li, rng);
        lightDir = mk3(dirPdf.x, dirPdf.y, dirPdf.z);
        lightPdf = dirPdf.w;
    } else {                                             // :143-157
        Li = mk3(P.EnvironmentColor[0], P.EnvironmentColor[1], P.EnvironmentColor[2]) * P.EnvironmentIntensity;
        lightPdf = 1.0f / (4.0f * PT_PI);
        lightDir = normalize3(random_cosine_hemisphere(onb_of_normal(hit.normal, hit.ffnormal, ffOnb), rng));
    }
    out.dir = lightDir;
    out.valid = 1u;
    float spdf = 0.0f;
    v3 sf = eval_brdf_onb(mat, -rayDir, lightDir, ffOnb, spdf, sh);          // ffOnb = make_onb(hit.ffnormal)
    if (spdf > 0.0f) {
        float misWeight = power_heuristic(lightPdf, spdf);
        if (misWeight > 0.0f) {
            if (S.hasEnvTex) out.contribution = misWeight * Li * sf * P.EnvironmentIntensity / lightPdf;
            else out.contribution = misWeight * Li * sf / lightPdf;
        } else out.valid = 2u;     // ray is traced (the reference traces before evaluating) but adds nothing
    } else out.valid = 2u;
}

// analytic-light half of DirectLight (:162-170) + SampleOneLight (:47-58) + EvalLight (:60-114)
template <bool STATS>
PT_DEV void nee_prepare_light(const DScene& S, v3 rayDir, const SurfHit& hit, const Material& mat, v3 scatterPos,
                              uint32_t& rng, NeeRay& out, Counters& cn, const BsdfShared& sh, const Onb& ffOnb)
{
    out.valid = 0u;
    out.contribution = mk3(0.0f);
    if (!S.hasLights) return;
    int32_t lightIndex = pt_f2i(rnd(rng) * (float)S.lightCount);
    if (lightIndex > S.lightCount - 1) lightIndex = S.lightCount - 1;
    // The picked light differs from lane to lane, but with a handful of lights it is cheaper to read EVERY light once per wave
    // through the scalar path (pt_uniform_load) and keep the lane's own than to gather six rows per lane: no vector-memory
    // requests, no dependent round trip in front of the BSDF evaluation.  Same records, same values.
    DLight light;
    float4 lcN, lcNN;
    if (S.lightCount <= PT_UNIFORM_LIGHTS) {
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, c = a, d = a;
        lcN = a; lcNN = a;
        for (int i = 0; i < S.lightCount; ++i) {
            const float4 ua = pt_uniform_load(S.lights, (size_t)i * 4), ub = pt_uniform_load(S.lights, (size_t)i * 4 + 1),
                         uc = pt_uniform_load(S.lights, (size_t)i * 4 + 2), ud = pt_uniform_load(S.lights, (size_t)i * 4 + 3),
                         un = pt_uniform_load(S.lightConst, (size_t)i * 4), unn = pt_uniform_load(S.lightConst, (size_t)i * 4 + 3);
            if (lightIndex == i) { a = ua; b = ub; c = uc; d = ud; lcN = un; lcNN = unn; }
        }
        light.position = mk3(a.x, a.y, a.z); light.type = pt_asuint(a.w);
        light.emission = mk3(b.x, b.y, b.z); light.range = b.w;
        light.u = mk3(c.x, c.y, c.z); light.area = c.w;
        light.v = mk3(d.x, d.y, d.z);
    } else {
        light = load_light(S, lightIndex);
        lcN = S.lightConst[(size_t)lightIndex * 4]; lcNN = S.lightConst[(size_t)lightIndex * 4 + 3];
    }
    if (STATS) cn.lightFetches++;
    v3 lsNormal, lsDirection;
    float lsDistance, lsPdf;
    if (light.type == PT_LIGHT_TYPE_SPOT) {
        lsNormal = mk3(lcN.x, lcN.y, lcN.z);                     // normalize(light.u)
        lsDirection = -normalize3(scatterPos - light.position);
        lsDistance = length3(light.position - scatterPos);
        lsPdf = 0.0f;
    } else if (light.type == PT_LIGHT_TYPE_RECTANGLE) {
        float r1 = rnd(rng);
        float r2 = rnd(rng);
        v3 lightSurfacePos = light.position + light.u * r1 + light.v * r2;
        lsDirection = lightSurfacePos - scatterPos;
        lsDistance = length3(lsDirection);
        float distSq = lsDistance * lsDistance;
        lsDirection = lsDirection / lsDistance;
        lsNormal = mk3(lcN.x, lcN.y, lcN.z);                     // normalize(cross(light.u, light.v))
        lsPdf = distSq / (light.area * pt_abs(dot3(lsNormal, lsDirection)));
    } else if (light.type == PT_LIGHT_TYPE_POINT) {
        lsNormal = normalize3(scatterPos - light.position);
        lsDirection = -lsNormal;
        lsDistance = length3(scatterPos - light.position);
        lsPdf = 0.0f;
    } else {
        return;                 // SampleOneLight returns false: no shadow ray, no contribution
    }
    // EvalLight
    float falloff = 1.0f;
    if (lsDistance > light.range) falloff = 0.0f;
    else {
        float r = lsDistance / light.range;
        float atten = pt_saturate(1.0f / (1.0f + 25.0f * r * r) * pt_saturate((1.0f - r) * 5.0f));
        falloff *= atten;
    }
    const v3 lsNormalN = mk3(lcNN.x, lcNN.y, lcNN.z);            // normalize(lsNormal) of a rectangle / spot light
    if (light.type == PT_LIGHT_TYPE_RECTANGLE) {
        float cosTheta = dot3(normalize3(-lsDirection), lsNormalN);
        falloff = cosTheta < 0.0f ? 0.0f : falloff;
    }
    if (light.type == PT_LIGHT_TYPE_SPOT) {
        float cosTheta = dot3(normalize3(-lsDirection), lsNormalN);
        if (cosTheta < light.v.x) falloff = 0.0f;
        else if (cosTheta > light.v.x && cosTheta < light.v.y) falloff *= (cosTheta - light.v.x) / (light.v.y - light.v.x);
    }
    v3 Li = light.emission * falloff;
    out.dir = lsDirection;
    out.valid = 1u;
    float pdf = 0.0f;
    v3 f = eval_brdf_onb(mat, -rayDir, lsDirection, onb_of_normal(hit.normal, hit.ffnormal, ffOnb), pdf, sh);       // EvalBRDF(..., hit.normal, ...): GetONB(normal)
    float lightPdf = 1.0f;
    if (lsPdf > 0.0f) lightPdf = lsPdf;
    out.contribution = Li * f / lightPdf;
}

// ------------------------------------------------------------------------------------------
// camera.  util/camera.hlsl:13-42, util/common.hlsl:285-341, PathTracer.compute:33-38
// ------------------------------------------------------------------------------------------
PT_DEV v4 mul44(const float* m, v4 v)      // Unity Matrix4x4 memory order (element (r,c) at c*4+r), column vector
{
    v4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}

PT_DEV void concentric_sample_disk(float u1, float u2, float& dx, float& dy)
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

// One camera sample: two jitter draws (left to right), optional two lens draws.  PathTracer.compute:68-75
PT_DEV void generate_camera_ray(const PTFrameParams& P, uint32_t pixelX, uint32_t pixelY, uint32_t& rng, v3& origin, v3& direction)
{
    float g0 = rnd(rng);
    float g1 = rnd(rng);
    float gr = pt_sqrt(-2.0f * pt_log(pt_max(1e-38f, g0)));
    float gtheta = 2.0f * PT_PI * g1;
    float gx = gr * pt_cos(gtheta), gy = gr * pt_sin(gtheta);
    float pcx = (float)pixelX + (0.5f + PT_AA_SIGMA * gx);
    float pcy = (float)pixelY + (0.5f + PT_AA_SIGMA * gy);

    v4 o4 = mul44(P.CamToWorld, v4{0.0f, 0.0f, 0.0f, 1.0f});
    origin = mk3(o4.x, o4.y, o4.z);
    float uvx = pcx / (float)P.OutputWidth * 2.0f - 1.0f;
    float uvy = pcy / (float)P.OutputHeight * 2.0f - 1.0f;
    v4 d4 = mul44(P.CamInvProj, v4{uvx, uvy, 0.0f, 1.0f});
    v4 w4 = mul44(P.CamToWorld, v4{d4.x, d4.y, d4.z, 0.0f});
    direction = normalize3(mk3(w4.x, w4.y, w4.z));
    if (P.Aperture > 0.0f && P.FocalLength > 0.0f) {
        float sampleLensU = rnd(rng);
        float sampleLensV = rnd(rng);
        float lensU, lensV;
        concentric_sample_disk(sampleLensU, sampleLensV, lensU, lensV);
        float lensRadius = P.Aperture * 0.5f;
        lensU *= lensRadius;
        lensV *= lensRadius;
        float ft = P.FocalLength;
        v3 focalPoint = origin + direction * ft;
        v4 l4 = mul44(P.CamToWorld, v4{lensU, lensV, 0.0f, 1.0f});
        origin = mk3(l4.x, l4.y, l4.z);
        direction = normalize3(focalPoint - origin);
    }
}

// ------------------------------------------------------------------------------------------
// One iteration of the per-pixel program, shared by every schedule.
//
// The reference's PathTrace loop (util/pathtrace.hlsl:10-131) + the sample loop of the kernel entry
// (PathTracer.compute:66-98), cut at the traversal calls: a path alternates between "rays are being traced"
// and path_step(), which (1) adds the NEE of the bounce shaded last time (now that its shadow rays are
// resolved), (2) shades the closest hit that came back -- sky / light / material, NEE preparation, BSDF
// sampling, throughput, Russian roulette -- and (3) when the sample ended, folds it into the pixel and
// either starts the next sample (new camera ray) or writes the pixel.  Between two calls the schedule traces
// up to three rays: env shadow (if hasPending && env.valid), light shadow (if hasPending && light.valid) and
// the next closest-hit ray (if state == PS_TRACE).  Every float and every RNG draw happens in the reference's
// order, whatever the schedule.
// ------------------------------------------------------------------------------------------
enum : uint32_t { PS_TRACE = 0u, PS_ENDING = 1u, PS_DONE = 2u };

struct PathRegs {
    v3 ro, rd;                   // the closest-hit ray in flight (state == PS_TRACE)
    v3 radiance, throughput, color;
    float scatterPdf, maxRoughness;
    uint32_t rng, depth, state, sampleIdx;
    NeeRay env, light;           // pending NEE of the last shaded bounce
    v3 neeOrigin, pendThroughput;
    bool hasPending, green;
};

PT_DEV void path_start_sample(const PTFrameParams& P, uint32_t pixelX, uint32_t pixelY, PathRegs& r, Counters& cn)
{
    cn.paths++;
    r.radiance = mk3(0.0f);
    r.throughput = mk3(1.0f);
    r.depth = 0u;
    r.scatterPdf = 0.0f;
    r.maxRoughness = 0.0f;
    generate_camera_ray(P, pixelX, pixelY, r.rng, r.ro, r.rd);
    r.state = PS_TRACE;
}

// seedRoot / currentSample: RngSeedRoot and CurrentSample of the pass the path belongs to (P's own, or those of pass j of a batch)
PT_DEV void path_init(const PTFrameParams& P, uint32_t seedRoot, uint32_t currentSample, uint32_t pixelX, uint32_t pixelY, uint32_t pixelIndex,
                      PathRegs& r, Counters& cn)
{
    r.rng = pixelIndex * (currentSample + 1u) + seedRoot;                // PathTracer.compute:60
    r.sampleIdx = 0u;
    r.color = mk3(0.0f);
    r.env.valid = 0u; r.light.valid = 0u;
    r.env.dir = mk3(0.0f); r.light.dir = mk3(0.0f);
    r.env.contribution = mk3(0.0f); r.light.contribution = mk3(0.0f);
    r.neeOrigin = mk3(0.0f); r.pendThroughput = mk3(0.0f);
    r.hasPending = false; r.green = false;
    path_start_sample(P, pixelX, pixelY, r, cn);
}

// INLINE_RESOLVE: write the pixel (running mean with AccumulatedOutput) when its last sample ends, as the reference
// kernel does.  Without it the pixel's sample sum stays in r.color and a separate resolve step applies the same formula
// later (pt_wavefront.hip), which lets consecutive passes overlap.
// NeeSink: called once the two NEE rays of a bounce are final (directions, contributions, origin, the throughput they apply
// to), BEFORE the BSDF is sampled.  The wavefront shade kernel writes them to the slot arrays there, so that their 15
// registers are free while sample_brdf runs (pt_wavefront.hip); the other schedules keep them in registers and pass nothing.
struct NoNeeSink { PT_DEV void operator()(PathRegs&) const {} };

// ---- (1) radiance += DirectLight(...) * throughput      (util/pathtrace.hlsl:93, deferred until the shadow rays are in)
PT_DEV void path_apply_pending(PathRegs& r, bool occEnv, bool occLight)
{
    if (r.hasPending) {
        v3 Ld = mk3(0.0f);
        if (r.env.valid == 1u && !occEnv) Ld = Ld + r.env.contribution;
        if (r.light.valid == 1u && !occLight) Ld = Ld + r.light.contribution;
        r.radiance = r.radiance + Ld * r.pendThroughput;
        if (r.green) r.radiance = mk3(0.0f, 1.0f, 0.0f);      // NaN BSDF sample (util/pathtrace.hlsl:100-104)
        r.hasPending = false;
        r.green = false;
    }
}

// ---- (2) shade the closest hit                           (util/pathtrace.hlsl:27-127).  Precondition: r.state == PS_TRACE.
// Reads r.ro, rd, radiance, throughput, scatterPdf, maxRoughness, rng, depth; writes those plus state, hasPending, green,
// env, light, neeOrigin, pendThroughput.
// -DPT_SHADE_TIME (diagnostics build, STATS instantiations only; tools/experiments/shade_time.py): clock ticks >> 4 a WAVE spends in the
// sections of the shading step, through the work counters: attrFetches = hit attributes + light intersection, pixelsRead = miss / light-hit
// branch, materialFetches = material + textures, texelFetches = environment NEE, lightFetches = light NEE, texDescFetches = BSDF sampling + the
// rest, tlasNodeVisits = the whole step, instanceVisits = wave executions of the step
#ifdef PT_SHADE_TIME
#define PT_ST_BEGIN() const unsigned long long _st0 = __builtin_readcyclecounter()
#define PT_ST_MARK(var) unsigned long long var = __builtin_readcyclecounter()
#define PT_ST_ADD(field, a, b) do { if (STATS && (int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(true)) - 1) cn.field += (uint32_t)(((b) - (a)) >> 4); } while (0)
#else
#define PT_ST_MARK(var) do { } while (0)
#define PT_ST_ADD(field, a, b) do { } while (0)
#endif
template <bool STATS, class NeeSink = NoNeeSink>
PT_DEV void path_shade_hit(const DScene& S, const PTFrameParams& P, PathRegs& r, const HitRecord& rec, Counters& cn, NeeSink neeSink = NeeSink())
{
    {
        const uint32_t maxRayBounces = P.MaxRayBounces > 1u ? P.MaxRayBounces : 1u;
        SurfHit hit;
        hit.distance = PT_FAR_PLANE;
        hit.isLight = 0u;
        hit.triIndex = 0u;
        hit.materialIndex = 0;
        hit.position = mk3(0.0f); hit.normal = mk3(0.0f); hit.ffnormal = mk3(0.0f);
        hit.uv = {0.0f, 0.0f};
        PT_ST_MARK(st0);
        if (rec.h.t < PT_FAR_PLANE) {
            if (S.hasTlas) fetch_hit_attributes_tlas(S, r.rd, rec, hit);
            else fetch_hit_attributes<STATS>(S, r.ro, r.rd, rec.h, hit, cn);
        }
        intersect_lights<STATS>(S, r.ro, r.rd, hit, cn);
        PT_ST_MARK(st1);
        PT_ST_ADD(attrFetches, st0, st1);
#ifdef PT_SHADE_TIME
        if (STATS && (int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(true)) - 1) cn.instanceVisits++;
#endif

        if (!(hit.distance < PT_FAR_PLANE)) {
            v4 sky = sample_sky_radiance(S, P, r.rd, r.depth);
            float misWeight = 1.0f;
            if (r.depth > 0u) misWeight = power_heuristic(r.scatterPdf, sky.w);
            if (misWeight > 0.0f) r.radiance = r.radiance + misWeight * mk3(sky.x, sky.y, sky.z) * r.throughput;
            r.state = PS_ENDING;
            PT_ST_MARK(st2);
            PT_ST_ADD(pixelsRead, st1, st2);
        } else if (S.hasLights && hit.isLight) {
            DLight light = load_light(S, (int)hit.triIndex);
#ifndef PT_SHADE_TIME
            if (STATS) cn.lightFetches++;
#endif
            r.radiance = r.radiance + light.emission * r.throughput;
            r.state = PS_ENDING;
        } else {
            PT_ST_MARK(st3);
            Material material = get_material<STATS>(S, hit.materialIndex, r.rd, hit, cn);
            PT_ST_MARK(st4);
            PT_ST_ADD(materialFetches, st3, st4);
            r.maxRoughness = pt_max(r.maxRoughness, material.roughness);
            material.roughness = r.maxRoughness;
            r.radiance = r.radiance + material.emission * r.throughput;
            if (r.depth >= maxRayBounces) {
                r.state = PS_ENDING;
            } else {
                v3 scatterL;
                // util/pathtrace.hlsl:84-85 is ONE boolean expression: (MASK && opacity < cutoff) || (BLEND && RandomFloat() > opacity).
                // HLSL (before HLSL 2021; FXC is in the reference's pipeline, util/bvh.hlsl:133-135, and PathTracer.compute
                // selects neither DXC nor -HV 2021) does not short-circuit && / ||: every operand is evaluated, side effects
                // included, so RandomFloat is drawn on EVERY bounce that reaches this point, whatever the alpha mode.
                const float alphaRnd = rnd(r.rng);
                const bool skip = (material.alphaMode == (float)PT_ALPHA_MODE_MASK && material.opacity < material.alphaCutoff) ||
                                  (material.alphaMode == (float)PT_ALPHA_MODE_BLEND && alphaRnd > material.opacity);
                bool advance = true;
                if (skip) {
                    scatterL = r.rd;                             // rayDepth-- then ++: depth unchanged (util/pathtrace.hlsl:85-88)
                } else {
                    r.neeOrigin = nee_scatter_pos(hit);
                    BsdfShared sh;
                    tint_colors(material, material.eta, sh.F0, sh.Csheen, sh.Cspec0);
                    const Onb ffOnb = make_onb(hit.ffnormal);
                    PT_ST_MARK(st5);
                    nee_prepare_environment(S, P, r.rd, hit, material, r.rng, r.env, sh, ffOnb);
                    PT_ST_MARK(st6);
                    PT_ST_ADD(texelFetches, st5, st6);
                    nee_prepare_light<STATS>(S, r.rd, hit, material, r.neeOrigin, r.rng, r.light, cn, sh, ffOnb);
                    PT_ST_MARK(st7);
                    PT_ST_ADD(lightFetches, st6, st7);
                    r.pendThroughput = r.throughput;
                    r.hasPending = true;
                    neeSink(r);
                    float pdf;
                    v3 f = sample_brdf(material, -r.rd, ffOnb, scatterL, pdf, r.rng, sh);
                    r.scatterPdf = pdf;
                    if (pt_isnan(f.x) || pt_isnan(f.y) || pt_isnan(f.z)) {
                        r.green = true;
                        r.state = PS_ENDING;
                        advance = false;
                    } else if (pdf > 0.0f) {
                        r.throughput = r.throughput * (f / pdf);
                    } else {
                        r.state = PS_ENDING;
                        advance = false;
                    }
                    if (advance) r.depth++;
                    PT_ST_MARK(st8);
                    PT_ST_ADD(texDescFetches, st7, st8);
                }
                if (advance) {
                    r.rd = scatterL;
                    r.ro = hit.position + r.rd * PT_EPSILON;
                    if (P.UseRussianRoulette) {
                        float rrPcont = pt_min(pt_max(r.throughput.x, pt_max(r.throughput.y, r.throughput.z)) + 0.001f, 0.95f);
                        if (rnd(r.rng) >= rrPcont) r.state = PS_ENDING;
                        else r.throughput = r.throughput / rrPcont;
                    }
                }
            }
        }
    }
}

// ---- (3) end of a sample                                 (PathTracer.compute:77-98)
template <bool INLINE_RESOLVE>
PT_DEV void path_end_sample(const PTFrameParams& P, PathRegs& r, uint32_t pixelX, uint32_t pixelY, uint32_t pixelIndex,
                            const float4* __restrict__ accumulated, float4* __restrict__ output, Counters& cn)
{
    if (r.state == PS_ENDING && !r.hasPending) {
        if (P.UseFireflyFilter) {
            float lum = luminance3(r.radiance);
            if (lum > P.MaxFireflyLuminance) r.radiance = r.radiance * (P.MaxFireflyLuminance / lum);
        }
        r.color = r.color + r.radiance;
        r.sampleIdx++;
        const uint32_t numSamples = P.SamplesPerPass > 1 ? (uint32_t)P.SamplesPerPass : 1u;
        if (r.sampleIdx < numSamples) {
            path_start_sample(P, pixelX, pixelY, r, cn);
        } else if (!INLINE_RESOLVE) {
            r.state = PS_DONE;
        } else {
            const float fSamples = (float)numSamples;
            float4 o4;
            if (P.CurrentSample > 0u) {                           // PathTracer.compute:89-94
                float4 a = accumulated[pixelIndex];
                cn.pixelsRead++;
                float cs = (float)P.CurrentSample;
                v3 acc = (r.color + mk3(a.x, a.y, a.z) * cs) / (cs + fSamples);
                o4 = make_float4(acc.x, acc.y, acc.z, 1.0f);
            } else {
                v3 acc = r.color / fSamples;
                o4 = make_float4(acc.x, acc.y, acc.z, 1.0f);
            }
            output[pixelIndex] = o4;
            cn.pixelsWritten++;
            r.state = PS_DONE;
        }
    }
}

template <bool STATS, bool INLINE_RESOLVE = true, class NeeSink = NoNeeSink>
PT_DEV void path_step(const DScene& S, const PTFrameParams& P, PathRegs& r, const HitRecord& rec, bool occEnv, bool occLight,
                      uint32_t pixelX, uint32_t pixelY, uint32_t pixelIndex,
                      const float4* __restrict__ accumulated, float4* __restrict__ output, Counters& cn, NeeSink neeSink = NeeSink())
{
    path_apply_pending(r, occEnv, occLight);
    if (r.state == PS_TRACE) path_shade_hit<STATS, NeeSink>(S, P, r, rec, cn, neeSink);
    path_end_sample<INLINE_RESOLVE>(P, r, pixelX, pixelY, pixelIndex, accumulated, output, cn);
}

// Wave-level helpers (wave64).
PT_DEV uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
PT_DEV uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}
#define PT_NUM_COUNTERS 16
PT_DEV void counters_to_array(const Counters& cn, uint32_t* vals)
{
    vals[0] = cn.paths; vals[1] = cn.closestRays; vals[2] = cn.shadowRays; vals[3] = cn.nodeVisits; vals[4] = cn.triTests;
    vals[5] = cn.attrFetches; vals[6] = cn.materialFetches; vals[7] = cn.lightFetches; vals[8] = cn.texelFetches;
    vals[9] = cn.texDescFetches; vals[10] = cn.pixelsWritten; vals[11] = cn.pixelsRead; vals[12] = cn.maxStack; vals[13] = cn.overflows;
    vals[14] = cn.tlasNodeVisits; vals[15] = cn.instanceVisits;
}

// ------------------------------------------------------------------------------------------
// Resumable CWBVH traversal: the loop of traverse_cwbvh cut into steps, so that a wave can retire finished rays
// and pull new ones into the idle lanes between steps (pt_wavefront.hip, refill trace kernel).  One call of
// ray_step() is exactly one iteration of the reference's outer while(true) (util/bvh.hlsl:141-197):
// [node step | adopt triangle group] -> all triangles of the group -> [pop | finish]; the per-ray operation order,
// hence t/u/v/triIndex and the node/triangle counters, are those of traverse_cwbvh.
// ------------------------------------------------------------------------------------------
struct RayState {
    v3 o, d, invDir;
    uint32_t octinv4;
    uint2 ng, tg;
    uint32_t sp;
    TraceHit hit;
    bool anyHit, overflow;
};

// returns true when the ray is finished before it started (NaN ray: certain miss, see traverse_cwbvh)
PT_DEV bool ray_begin(RayState& r, v3 o, v3 d, bool anyHit)
{
    r.o = o; r.d = d; r.anyHit = anyHit; r.overflow = false;
    r.hit.t = PT_FAR_PLANE; r.hit.u = 0.0f; r.hit.v = 0.0f; r.hit.triIndex = 0u;
    r.sp = 0u;
    r.ng = make_uint2(0u, 0x80000000u);
    r.tg = make_uint2(0u, 0u);
    r.invDir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.octinv4 = (7u - ((d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u))) * 0x1010101u;
    return pt_isnan(o.x) || pt_isnan(o.y) || pt_isnan(o.z) || pt_isnan(d.x) || pt_isnan(d.y) || pt_isnan(d.z);
}

// one outer iteration; returns true when the traversal is complete
#ifdef PT_TRACE_DIAG
#define PT_DIAG_WAVE_COUNT(field) do { if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(true)) - 1) cn.field++; } while (0)
#else
#define PT_DIAG_WAVE_COUNT(field) do { } while (0)
#endif

template <bool STATS, class ST>
PT_DEV bool ray_step(const DScene& S, RayState& r, ST& st, Counters& cn)
{
    PT_DIAG_WAVE_COUNT(attrFetches);          // wave-level outer iterations
    if (r.ng.y > 0x00FFFFFFu) {
        PT_DIAG_WAVE_COUNT(materialFetches);  // wave-level node-block executions
        cwbvh_visit_node<STATS>(S.nodes, 0u, r.o, r.invDir, r.octinv4, r.hit.t, r.ng, r.tg, r.sp, r.overflow, st, cn);
    } else {
        r.tg = r.ng;
        r.ng = make_uint2(0u, 0u);
    }
    while (r.tg.y != 0u) {
        PT_DIAG_WAVE_COUNT(lightFetches);     // wave-level triangle-block executions
        uint32_t triangleIndex = 31u - (uint32_t)__clz((int)r.tg.y);
        uint32_t triAddr = r.tg.x + triangleIndex * 3u;
        intersect_triangle(S.tris, triAddr, r.o, r.d, r.hit);
        if (STATS) cn.triTests++;
        r.tg.y -= 1u << triangleIndex;
        if (r.anyHit && r.hit.t < PT_FAR_PLANE) { r.tg.y = 0u; r.ng.y = 0u; r.sp = 0u; }
    }
    if (r.ng.y <= 0x00FFFFFFu) {
        if (r.sp > 0u) { --r.sp; r.ng = stack_pop(st, r.sp); }
        else {
            if (STATS && r.overflow) cn.overflows++;
            return true;
        }
    }
    return false;
}

// ray_unistep: the same traversal in finer steps, with ONE memory round trip per wave iteration.  Every lane first decides what it does next
// -- test the next triangle of its group, or (pop and) visit the next node -- and only computes an ADDRESS; then all lanes
// request their rows together (3 rows for a triangle, 5 for a node), and after that single wait the node block and the
// triangle block run one after the other under their lane masks.  With the nested loops of ray_step a wave iteration paid a
// node round trip and then one more round trip per triangle of its slowest lane (2.1 on average, at 8 % lane utilisation in
// the triangle block); here a lane with triangles pending simply stays in "triangle mode" for as many iterations as it has
// triangles while its neighbours keep visiting nodes.  The per-ray sequence of operations is that of traverse_cwbvh.
template <bool STATS, class ST>
PT_DEV bool ray_unistep(const DScene& S, RayState& r, ST& st, Counters& cn)
{
    bool finished = false;
    const bool isTri = r.tg.y != 0u;
    bool isNode = false;
    uint32_t triangleIndex = 0u;
    const uint4* p = S.nodes;
    if (isTri) {
        triangleIndex = 31u - (uint32_t)__clz((int)r.tg.y);
        p = (const uint4*)S.tris + (size_t)(r.tg.x + triangleIndex * 3u);
    } else {
        if (r.ng.y <= 0x00FFFFFFu) {
            if (r.sp > 0u) { --r.sp; r.ng = stack_pop(st, r.sp); }
            else finished = true;
        }
        if (!finished) {
            if (r.ng.y > 0x00FFFFFFu) {
                isNode = true;
                p = S.nodes + (size_t)cwbvh_pick_child<STATS>(r.ng, r.octinv4, r.sp, r.overflow, st, cn) * 5;
            } else {
                r.tg = r.ng;                                   // a triangle group came off the stack (util/bvh.hlsl:172-176)
                r.ng = make_uint2(0u, 0u);
            }
        }
    }
    uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0, q2 = q0, q3 = q0, q4 = q0;
    if (isTri || isNode) {
        q0 = p[0]; q1 = p[1]; q2 = p[2];
        if (isNode) { q3 = p[3]; q4 = p[4]; }
    }
    asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x), "+v"(q4.x));      // all rows requested before the first use
    if (isNode) {
        cwbvh_apply_node<STATS>(r.o, r.invDir, r.octinv4, r.hit.t, q0, q1, q2, q3, q4, r.ng, r.tg, cn);
    } else if (isTri) {
        const float4 t0 = make_float4(pt_asfloat(q0.x), pt_asfloat(q0.y), pt_asfloat(q0.z), pt_asfloat(q0.w));
        const float4 t1 = make_float4(pt_asfloat(q1.x), pt_asfloat(q1.y), pt_asfloat(q1.z), pt_asfloat(q1.w));
        const float4 t2 = make_float4(pt_asfloat(q2.x), pt_asfloat(q2.y), pt_asfloat(q2.z), pt_asfloat(q2.w));
        intersect_triangle_rows<false>(t0, t1, t2, 0u, r.o, r.d, r.hit);
        if (STATS) cn.triTests++;
        r.tg.y -= 1u << triangleIndex;
        if (r.anyHit && r.hit.t < PT_FAR_PLANE) finished = true;
    }
    if (finished && STATS && r.overflow) cn.overflows++;
    return finished;
}

// The two halves of ray_unistep as separate phases, so that a wave can decide per iteration WHICH phase to run: the triangle
// block costs ~100 VALU instructions per execution whatever the number of lanes in it, so lanes whose node visit produced
// triangles PARK until enough of them have accumulated (pt_wavefront.hip, PT_WF_TRI_PARK).  A lane's own sequence of operations
// -- hence every result and counter -- is unchanged: it only waits.
template <bool STATS>
PT_DEV bool ray_tri_one(const DScene& S, RayState& r, Counters& cn)                 // precondition: r.tg.y != 0
{
    const uint32_t triangleIndex = 31u - (uint32_t)__clz((int)r.tg.y);
    intersect_triangle(S.tris, r.tg.x + triangleIndex * 3u, r.o, r.d, r.hit);
    if (STATS) cn.triTests++;
    r.tg.y -= 1u << triangleIndex;
    if (r.anyHit && r.hit.t < PT_FAR_PLANE) {
        if (STATS && r.overflow) cn.overflows++;
        return true;
    }
    return false;
}

template <bool STATS, class ST>
PT_DEV bool ray_node_one(const DScene& S, RayState& r, ST& st, Counters& cn)        // precondition: r.tg.y == 0
{
    if (r.ng.y <= 0x00FFFFFFu) {
        if (r.sp > 0u) { --r.sp; r.ng = stack_pop(st, r.sp); }
        else {
            if (STATS && r.overflow) cn.overflows++;
            return true;
        }
    }
    if (r.ng.y > 0x00FFFFFFu) {
        cwbvh_visit_node<STATS>(S.nodes, 0u, r.o, r.invDir, r.octinv4, r.hit.t, r.ng, r.tg, r.sp, r.overflow, st, cn);
    } else {
        r.tg = r.ng;
        r.ng = make_uint2(0u, 0u);
    }
    return false;
}
