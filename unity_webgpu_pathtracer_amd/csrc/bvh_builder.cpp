// bvh_builder.cpp — acceleration-structure builders behind BuildBVH / BuildTLAS.
//
// DERIVED FROM tinybvh v1.6.7 (https://github.com/jbikker/tinybvh, tiny_bvh.h as vendored by the reference at
// Assets/Plugins/Web/tiny_bvh.h), Copyright (c) 2024-2025 Jacco Bikker, MIT License:
//
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//   rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//   persons to whom the Software is furnished to do so, subject to the following conditions: The above copyright notice and
//   this permission notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
//
// Why it is a restatement and not an independent design: BuildBVH's output must be BYTE-IDENTICAL to the reference plugin's
// (the C# host and every downstream fixture consume those bytes), and byte identity of a greedy, float-driven construction
// forces the algorithm, its traversal order and the order of every float operation.  The functions below therefore follow
// the cited tinybvh routines step by step -- with names of our own only where the code is not a direct counterpart -- and
// are licensed under the terms above.  (The builder that is an independent design is bvh_builder_gpu.hip.)
//
// The construction pipeline the reference plugin runs through tinybvh with TINYBVH_NO_SIMD / NO_THREADED_BUILDS
// (Assets/Plugins/Web/plugin.h:9-11):
//
//   BuildBVH  (plugin.cpp:35-40  -> tiny_bvh.h:5787-5795)
//     1. binned-SAH binary BVH, 8 bins, c_trav = c_int = 1          (tiny_bvh.h:2221-2290, 2292-2421)
//     2. depth-first re-layout ("Compact")                           (tiny_bvh.h:3693-3733)
//     3. split leaves down to <= 3 triangles                         (tiny_bvh.h:1987-2016)
//     4. greedy surface-area collapse into an 8-wide tree            (tiny_bvh.h:4935-5007)
//     5. CWBVH encode: octant-ordered children, 8-bit quantised boxes (tiny_bvh.h:5844-5976,
//        Ylitie, Karras, Laine 2017, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs")
//   BuildTLAS (plugin.cpp:111-118 -> tiny_bvh.h:2181-2219, 4572-4615)
//     1. the same binned-SAH builder over the instances' world AABBs
//     2. depth-first conversion to the Aila-Laine two-children-per-node layout
//
// The output is required to be BYTE-IDENTICAL to the reference plugin's (tests/test_bvh_builder.py
// memcmp's it against the reference plugin compiled in place and against committed golden dumps), so every float operation below is
// written in the order the reference evaluates it, float->int conversions use the x86 truncating forms
// the reference's compiled code uses (out-of-range -> "integer indefinite"), and log2f/powf/ceilf/floorf
// come from the same libm.  Compile with -ffp-contract=off.
#include "bvh_builder.h"

#include <cmath>
#include <cstring>
#include <immintrin.h>

namespace ptbvh {

namespace {

constexpr float kFar = 1e30f;
constexpr int kBins = 8;

struct F3 { float x, y, z; };
inline F3 f3(float a) { return {a, a, a}; }
inline float lo(float a, float b) { return a < b ? a : b; }
inline float hi(float a, float b) { return a > b ? a : b; }
inline F3 lo3(F3 a, F3 b) { return {lo(a.x, b.x), lo(a.y, b.y), lo(a.z, b.z)}; }
inline F3 hi3(F3 a, F3 b) { return {hi(a.x, b.x), hi(a.y, b.y), hi(a.z, b.z)}; }
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float comp(const F3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
inline float half_area(F3 v) { return v.x < -kFar ? 0.0f : (v.x * v.y + v.y * v.z + v.z * v.x); }
inline float surface_area(const float* mn, const float* mx)
{
    float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    return ex * ey + ey * ez + ez * ex;
}
// float -> int32 / uint32 as the x86-64 code of the reference performs them (cvttss2si, 32- and 64-bit forms)
inline int32_t trunc_i32(float f) { return _mm_cvttss_si32(_mm_set_ss(f)); }
inline uint32_t trunc_u32(float f) { return (uint32_t)_mm_cvttss_si64(_mm_set_ss(f)); }
inline int32_t clampi(int32_t x, int32_t a, int32_t b) { return x > a ? (x < b ? x : b) : a; }

struct Fragment { F3 bmin; uint32_t prim; F3 bmax; uint32_t clipped; };

inline F3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
inline void st3(float* p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

} // namespace

// ------------------------------------------------------------------------------------------
// binary BVH
// ------------------------------------------------------------------------------------------
struct Bvh2::Impl {
    std::vector<Fragment> frags;
};

Bvh2::Bvh2() : impl(new Impl()) {}
Bvh2::~Bvh2() { delete impl; }

void Bvh2::prepareTriangles(const PTFloat4* verts, uint32_t count)
{
    triCount = count;
    nodes.assign((size_t)count * 2 + 2, Node2{});
    primIdx.resize(count);
    impl->frags.resize(count);
    Node2& root = nodes[0];
    root.leftFirst = 0; root.triCount = count;
    F3 rmin = f3(kFar), rmax = f3(-kFar);
    for (uint32_t i = 0; i < count; i++) {
        const PTFloat4 &a = verts[i * 3], &b = verts[i * 3 + 1], &c = verts[i * 3 + 2];
        F3 v0 = {a.x, a.y, a.z}, v1 = {b.x, b.y, b.z}, v2 = {c.x, c.y, c.z};
        F3 fmin = lo3(v0, lo3(v1, v2));
        F3 fmax = hi3(v0, hi3(v1, v2));
        impl->frags[i] = {fmin, i, fmax, 0};
        rmin = lo3(rmin, fmin);
        rmax = hi3(rmax, fmax);
        primIdx[i] = i;
    }
    st3(root.mn, rmin); st3(root.mx, rmax);
    newNodePtr = 2;
}

void Bvh2::prepareBoxes(const PTBlasInstance* inst, uint32_t count)
{
    triCount = count;
    nodes.assign((size_t)count * 2 + 2, Node2{});
    primIdx.resize(count);
    impl->frags.resize(count);
    Node2& root = nodes[0];
    root.leftFirst = 0; root.triCount = count;
    F3 rmin = f3(kFar), rmax = f3(-kFar);
    for (uint32_t i = 0; i < count; i++) {
        F3 bmin = ld3(inst[i].aabbMin), bmax = ld3(inst[i].aabbMax);
        impl->frags[i] = {bmin, i, bmax, 0};
        rmin = lo3(rmin, bmin);
        rmax = hi3(rmax, bmax);
        primIdx[i] = i;
    }
    st3(root.mn, rmin); st3(root.mx, rmax);
    newNodePtr = 2;
}

// Binned SAH, depth-first with an explicit task stack (left child first).
void Bvh2::build()
{
    const std::vector<Fragment>& fr = impl->frags;
    uint32_t task[256], taskCount = 0, nodeIdx = 0;
    F3 rootExt = sub(ld3(nodes[0].mx), ld3(nodes[0].mn));
    const F3 minDim = {rootExt.x * 1e-20f, rootExt.y * 1e-20f, rootExt.z * 1e-20f};
    F3 bestLMin = f3(0), bestLMax = f3(0), bestRMin = f3(0), bestRMax = f3(0);
    while (true) {
        while (true) {
            Node2& node = nodes[nodeIdx];
            F3 binMin[3][kBins], binMax[3][kBins];
            uint32_t count[3][kBins];
            for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) { binMin[a][i] = f3(kFar); binMax[a][i] = f3(-kFar); count[a][i] = 0; }
            const F3 nmin3 = ld3(node.mn);
            const F3 ext = sub(ld3(node.mx), nmin3);
            const F3 rpd3 = {(float)kBins / ext.x, (float)kBins / ext.y, (float)kBins / ext.z};
            for (uint32_t i = 0; i < node.triCount; i++) {
                const Fragment& f = fr[primIdx[node.leftFirst + i]];
                int32_t bx = clampi(trunc_i32(((f.bmin.x + f.bmax.x) * 0.5f - nmin3.x) * rpd3.x), 0, kBins - 1);
                int32_t by = clampi(trunc_i32(((f.bmin.y + f.bmax.y) * 0.5f - nmin3.y) * rpd3.y), 0, kBins - 1);
                int32_t bz = clampi(trunc_i32(((f.bmin.z + f.bmax.z) * 0.5f - nmin3.z) * rpd3.z), 0, kBins - 1);
                binMin[0][bx] = lo3(binMin[0][bx], f.bmin); binMax[0][bx] = hi3(binMax[0][bx], f.bmax); count[0][bx]++;
                binMin[1][by] = lo3(binMin[1][by], f.bmin); binMax[1][by] = hi3(binMax[1][by], f.bmax); count[1][by]++;
                binMin[2][bz] = lo3(binMin[2][bz], f.bmin); binMax[2][bz] = hi3(binMax[2][bz], f.bmax); count[2][bz]++;
            }
            float splitCost = kFar;
            const float rSAV = 1.0f / surface_area(node.mn, node.mx);
            uint32_t bestAxis = 0, bestPos = 0;
            for (int a = 0; a < 3; a++) {
                if (!((node.mx[a] - node.mn[a]) > comp(minDim, a))) continue;
                F3 lBMin[kBins - 1], rBMin[kBins - 1], lBMax[kBins - 1], rBMax[kBins - 1];
                F3 l1 = f3(kFar), l2 = f3(-kFar), r1 = f3(kFar), r2 = f3(-kFar);
                float ANL[kBins - 1], ANR[kBins - 1];
                uint32_t lN = 0, rN = 0;
                for (int i = 0; i < kBins - 1; i++) {
                    lBMin[i] = l1 = lo3(l1, binMin[a][i]);
                    rBMin[kBins - 2 - i] = r1 = lo3(r1, binMin[a][kBins - 1 - i]);
                    lBMax[i] = l2 = hi3(l2, binMax[a][i]);
                    rBMax[kBins - 2 - i] = r2 = hi3(r2, binMax[a][kBins - 1 - i]);
                    lN += count[a][i];
                    rN += count[a][kBins - 1 - i];
                    ANL[i] = lN == 0 ? kFar : (half_area(sub(l2, l1)) * (float)lN);
                    ANR[kBins - 2 - i] = rN == 0 ? kFar : (half_area(sub(r2, r1)) * (float)rN);
                }
                for (int i = 0; i < kBins - 1; i++) {
                    const float C = ANL[i] + ANR[i];
                    if (C < splitCost) {
                        splitCost = C; bestAxis = (uint32_t)a; bestPos = (uint32_t)i;
                        bestLMin = lBMin[i]; bestRMin = rBMin[i]; bestLMax = lBMax[i]; bestRMax = rBMax[i];
                    }
                }
            }
            splitCost = 1.0f + 1.0f * rSAV * splitCost;                 // c_trav + c_int * rSAV * cost
            const float noSplitCost = (float)node.triCount * 1.0f;
            if (splitCost >= noSplitCost) break;
            // in-place partition of the index range
            uint32_t j = node.leftFirst + node.triCount, src = node.leftFirst;
            const float rpd = comp(rpd3, (int)bestAxis), nmin = comp(nmin3, (int)bestAxis);
            for (uint32_t i = 0; i < node.triCount; i++) {
                const Fragment& f = fr[primIdx[src]];
                int32_t bi = (int32_t)trunc_u32(((comp(f.bmin, (int)bestAxis) + comp(f.bmax, (int)bestAxis)) * 0.5f - nmin) * rpd);
                bi = clampi(bi, 0, kBins - 1);
                if ((uint32_t)bi <= bestPos) src++;
                else { --j; uint32_t t = primIdx[src]; primIdx[src] = primIdx[j]; primIdx[j] = t; }
            }
            const uint32_t leftCount = src - node.leftFirst, rightCount = node.triCount - leftCount;
            if (leftCount == 0 || rightCount == 0 || taskCount == 256) break;
            const uint32_t n = newNodePtr;
            newNodePtr += 2;
            st3(nodes[n].mn, bestLMin); st3(nodes[n].mx, bestLMax);
            nodes[n].leftFirst = node.leftFirst; nodes[n].triCount = leftCount;
            st3(nodes[n + 1].mn, bestRMin); st3(nodes[n + 1].mx, bestRMax);
            nodes[n + 1].leftFirst = j; nodes[n + 1].triCount = rightCount;
            node.leftFirst = n; node.triCount = 0;
            task[taskCount++] = n + 1;
            nodeIdx = n;
        }
        if (taskCount == 0) break;
        nodeIdx = task[--taskCount];
    }
    usedNodes = newNodePtr;
}

// Depth-first re-layout of nodes and of the primitive index list.
void Bvh2::compact()
{
    if (nodes[0].triCount > 0) return;
    std::vector<Node2> tmp(nodes.size());
    std::vector<uint32_t> idx(primIdx.size());
    tmp[0] = nodes[0]; tmp[1] = nodes[1];
    uint32_t newPtr = 2, newIdxPtr = 0, nodeIdx = 0, stack[128], sp = 0;
    while (true) {
        Node2& node = tmp[nodeIdx];
        if (node.triCount > 0) {
            const uint32_t leafStart = newIdxPtr;
            for (uint32_t i = 0; i < node.triCount; i++) idx[newIdxPtr++] = primIdx[node.leftFirst + i];
            node.leftFirst = leafStart;
            if (!sp) break;
            nodeIdx = stack[--sp];
        } else {
            tmp[newPtr] = nodes[node.leftFirst];
            tmp[newPtr + 1] = nodes[node.leftFirst + 1];
            const uint32_t todo1 = newPtr, todo2 = newPtr + 1;
            node.leftFirst = newPtr;
            newPtr += 2;
            nodeIdx = todo1;
            stack[sp++] = todo2;
        }
    }
    newNodePtr = usedNodes = newPtr;
    nodes.swap(tmp);
    primIdx.swap(idx);
}

void Bvh2::splitLeafs(uint32_t maxPrims)
{
    uint32_t stack[64], sp = 0, nodeIdx = 0;
    while (true) {
        if (nodes[nodeIdx].triCount > 0) {
            if (nodes[nodeIdx].triCount > maxPrims) {
                if (newNodePtr + 2 > nodes.size()) nodes.resize(nodes.size() * 2);
                Node2& node = nodes[nodeIdx];
                Node2& left = nodes[newNodePtr];
                Node2& right = nodes[newNodePtr + 1];
                left = node; right = node;
                right.leftFirst = node.leftFirst + maxPrims;
                right.triCount = node.triCount - maxPrims;
                left.triCount = maxPrims;
                node.leftFirst = newNodePtr;
                node.triCount = 0;
                newNodePtr += 2;
            } else {
                if (!sp) break;
                nodeIdx = stack[--sp];
            }
        } else {
            const uint32_t l = nodes[nodeIdx].leftFirst;
            nodeIdx = l;
            stack[sp++] = l + 1;
        }
    }
    usedNodes = newNodePtr;
}

// ------------------------------------------------------------------------------------------
// 8-wide collapse + CWBVH encode
// ------------------------------------------------------------------------------------------
namespace {

struct WideNode {
    float mn[3]; uint32_t firstTri;
    float mx[3]; uint32_t triCount;
    uint32_t child[8];
    uint32_t childCount;
    bool leaf() const { return triCount > 0; }
};

void collapse8(const Bvh2& b, std::vector<WideNode>& w)
{
    w.assign((size_t)b.usedNodes + 2, WideNode{});
    for (uint32_t i = 0; i < b.usedNodes; i++) {
        if (i == 1) continue;
        const Node2& o = b.nodes[i];
        WideNode& n = w[i];
        memcpy(n.mn, o.mn, 12); memcpy(n.mx, o.mx, 12);
        if (o.triCount > 0) { n.triCount = o.triCount; n.firstTri = o.leftFirst; }
        else { n.child[0] = o.leftFirst; n.child[1] = o.leftFirst + 1; n.childCount = 2; }
    }
    uint32_t stack[128], sp = 0, nodeIdx = 0;
    while (true) {
        WideNode& node = w[nodeIdx];
        while (node.childCount < 8) {
            int32_t bestChild = -1;
            float bestSA = 0;
            for (uint32_t i = 0; i < node.childCount; i++) {
                const WideNode& c = w[node.child[i]];
                if (!c.leaf() && node.childCount - 1 + c.childCount <= 8) {
                    const float sa = surface_area(c.mn, c.mx);
                    if (sa > bestSA) { bestChild = (int32_t)i; bestSA = sa; }
                }
            }
            if (bestChild == -1) break;
            const WideNode& c = w[node.child[bestChild]];
            node.child[bestChild] = c.child[0];
            for (uint32_t i = 1; i < c.childCount; i++) node.child[node.childCount++] = c.child[i];
        }
        for (uint32_t i = 0; i < node.childCount; i++) {
            const uint32_t ci = node.child[i];
            if (!w[ci].leaf()) stack[sp++] = ci;
        }
        if (sp == 0) break;
        nodeIdx = stack[--sp];
    }
    WideNode& root = w[0];
    if (root.leaf()) {              // CWBVH needs an interior root: add one level
        w[1] = root;
        root.childCount = 1;
        root.child[0] = 1;
        root.triCount = 0;
    }
}

inline int32_t quant_exponent(float extent)
{
    // (int32_t)(int8_t) ceilf(log2f(extent / 255)) with the x86 float->int8 conversion (cvttss2si, low byte)
    return (int32_t)(int8_t)trunc_i32(ceilf(log2f(extent / 255.0f)));
}

} // namespace

bool Cwbvh::build(const PTFloat4* verts, uint32_t count)
{
    if (!verts || count == 0) return false;
    Bvh2 b;
    b.prepareTriangles(verts, count);
    b.build();
    b.compact();
    b.splitLeafs(3);
    std::vector<WideNode> w;
    collapse8(b, w);

    triCount = count;
    nodeData.assign((size_t)count * 5 + 5, PTFloat4{0, 0, 0, 0});
    triData.assign((size_t)count * 3, PTFloat4{0, 0, 0, 0});
    struct Item { uint32_t node; uint32_t addr; };
    Item stack[256];
    uint32_t sp = 1, nodeDataPtr = 5, triDataPtr = 0;
    stack[0] = {0u, 0u};
    while (sp > 0) {
        const Item it = stack[--sp];
        WideNode& orig = w[it.node];
        const uint32_t addr = it.addr;
        const F3 nodeLo = ld3(orig.mn), nodeHi = ld3(orig.mx);
        const F3 nodeCentroid = {(nodeLo.x + nodeHi.x) * 0.5f, (nodeLo.y + nodeHi.y) * 0.5f, (nodeLo.z + nodeHi.z) * 0.5f};
        // greedy assignment of children to octant-ordered slots
        float cost[8][8];
        int32_t assignment[8];
        bool slotEmpty[8];
        for (int s = 0; s < 8; s++) {
            slotEmpty[s] = true; assignment[s] = -1;
            const F3 ds = {((s >> 2) & 1) ? -1.0f : 1.0f, ((s >> 1) & 1) ? -1.0f : 1.0f, (s & 1) ? -1.0f : 1.0f};
            for (int i = 0; i < 8; i++) {
                if (orig.child[i] == 0) { cost[s][i] = kFar; continue; }
                const WideNode& c = w[orig.child[i]];
                const F3 cc = {(c.mn[0] + c.mx[0]) * 0.5f, (c.mn[1] + c.mx[1]) * 0.5f, (c.mn[2] + c.mx[2]) * 0.5f};
                const F3 dv = sub(cc, nodeCentroid);
                cost[s][i] = dv.x * ds.x + dv.y * ds.y + dv.z * ds.z;
            }
        }
        while (true) {
            float minCost = kFar;
            int32_t ms = -1, mi = -1;
            for (int s = 0; s < 8; s++) for (int i = 0; i < 8; i++)
                if (assignment[i] == -1 && slotEmpty[s] && cost[s][i] < minCost) { minCost = cost[s][i]; ms = s; mi = i; }
            if (ms == -1 && mi == -1) break;
            slotEmpty[ms] = false;
            assignment[mi] = ms;
        }
        for (int i = 0; i < 8; i++) if (assignment[i] == -1) for (int s = 0; s < 8; s++) if (slotEmpty[s]) { slotEmpty[s] = false; assignment[i] = s; break; }
        uint32_t oldChild[8];
        memcpy(oldChild, orig.child, sizeof(oldChild));
        for (int i = 0; i < 8; i++) orig.child[assignment[i]] = oldChild[i];
        // per-axis power-of-two quantisation scale
        const int32_t ex = quant_exponent(nodeHi.x - nodeLo.x);
        const int32_t ey = quant_exponent(nodeHi.y - nodeLo.y);
        const int32_t ez = quant_exponent(nodeHi.z - nodeLo.z);
        const float px = powf(2, (float)ex), py = powf(2, (float)ey), pz = powf(2, (float)ez);
        uint8_t* const qbase = (uint8_t*)&nodeData[addr + 2];
        uint8_t* const meta = ((uint8_t*)&nodeData[addr + 1]) + 8;
        uint32_t internalCount = 0, leafTriCount = 0, childBaseIndex = 0, triangleBaseIndex = 0;
        uint8_t imask = 0;
        for (int i = 0; i < 8; i++) {
            if (orig.child[i] == 0) continue;
            const WideNode& c = w[orig.child[i]];
            qbase[i + 0] = (uint8_t)trunc_i32(floorf((c.mn[0] - nodeLo.x) / px));
            qbase[i + 8] = (uint8_t)trunc_i32(floorf((c.mn[1] - nodeLo.y) / py));
            qbase[i + 16] = (uint8_t)trunc_i32(floorf((c.mn[2] - nodeLo.z) / pz));
            qbase[i + 24] = (uint8_t)trunc_i32(ceilf((c.mx[0] - nodeLo.x) / px));
            qbase[i + 32] = (uint8_t)trunc_i32(ceilf((c.mx[1] - nodeLo.y) / py));
            qbase[i + 40] = (uint8_t)trunc_i32(ceilf((c.mx[2] - nodeLo.z) / pz));
            if (!c.leaf()) {
                const uint32_t childAddr = nodeDataPtr;
                if (internalCount++ == 0) childBaseIndex = childAddr / 5;
                nodeDataPtr += 5;
                imask |= (uint8_t)(1u << i);
                meta[i] = (uint8_t)((1u << 5) | (24u + (uint32_t)i));
                stack[sp++] = {orig.child[i], childAddr};
                continue;
            }
            const uint32_t tcount = c.triCount;
            if (leafTriCount == 0) triangleBaseIndex = triDataPtr;
            const uint32_t unary = tcount == 1 ? 0b001u : tcount == 2 ? 0b011u : 0b111u;
            meta[i] = (uint8_t)((unary << 5) | leafTriCount);
            leafTriCount += tcount;
            for (uint32_t j = 0; j < tcount; j++) {
                const uint32_t tri = b.primIdx[c.firstTri + j];
                const PTFloat4 v0 = verts[tri * 3], v1 = verts[tri * 3 + 1], v2 = verts[tri * 3 + 2];
                triData[triDataPtr + 0] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w};
                triData[triDataPtr + 1] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w};
                PTFloat4 t = v0;
                memcpy(&t.w, &tri, 4);
                triData[triDataPtr + 2] = t;
                triDataPtr += 3;
            }
        }
        const uint8_t exyz[4] = {(uint8_t)ex, (uint8_t)ey, (uint8_t)ez, imask};
        PTFloat4& n0 = nodeData[addr + 0];
        n0.x = nodeLo.x; n0.y = nodeLo.y; n0.z = nodeLo.z;
        memcpy(&n0.w, exyz, 4);
        memcpy(&nodeData[addr + 1].x, &childBaseIndex, 4);
        memcpy(&nodeData[addr + 1].y, &triangleBaseIndex, 4);
    }
    usedBlocks = nodeDataPtr;
    return true;
}

// ------------------------------------------------------------------------------------------
// TLAS
// ------------------------------------------------------------------------------------------
bool Tlas::build(const PTBlasInstance* inst, uint32_t count)
{
    if (!inst || count == 0) return false;
    Bvh2 b;
    b.prepareBoxes(inst, count);
    b.build();
    nodes.assign(b.usedNodes, PTTlasNode{});
    uint32_t newPtr = 0, nodeIdx = 0, stack[128], sp = 0;
    while (true) {
        const Node2& o = b.nodes[nodeIdx];
        const uint32_t idx = newPtr++;
        if (o.triCount > 0) {
            nodes[idx].triCount = o.triCount;
            nodes[idx].firstTri = o.leftFirst;
            if (!sp) break;
            nodeIdx = stack[--sp];
            const uint32_t parent = stack[--sp];
            nodes[parent].right = newPtr;
        } else {
            const Node2& l = b.nodes[o.leftFirst];
            const Node2& r = b.nodes[o.leftFirst + 1];
            memcpy(nodes[idx].lmin, l.mn, 12); memcpy(nodes[idx].rmin, r.mn, 12);
            memcpy(nodes[idx].lmax, l.mx, 12); memcpy(nodes[idx].rmax, r.mx, 12);
            nodes[idx].left = newPtr;
            stack[sp++] = idx;
            stack[sp++] = o.leftFirst + 1;
            nodeIdx = o.leftFirst;
        }
    }
    usedNodes = newPtr;
    indices = b.primIdx;
    return true;
}

} // namespace ptbvh
