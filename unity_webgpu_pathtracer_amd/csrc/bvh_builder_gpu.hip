// bvh_builder_gpu.hip -- a CWBVH builder that runs ON the MI355X, behind PTBuildBVHDevice (include/ptmi_plugin.h).
//
// BuildBVH (bvh_builder.cpp) restates tinybvh's scalar binned-SAH pipeline because its output must be byte-identical to the
// reference plugin's.  This builder sits BESIDE it: same output format (include/ptmi_layouts.h PTCwbvhNode / PTCwbvhTri, the
// format of tiny_bvh.h:5844-5976), same handle table, a different tree -- so its parity bar is not byte equality but hit
// equality: every ray finds the same closest hit (t, primitive) as in the reference-built tree (tests/test_bvh_builder_gpu.py).
//
// Pipeline (every step a kernel; the host only reads back one counter per tree level):
//   1. pt_lbvh_bounds      triangle AABBs + scene centroid bounds (wave reduction + float atomics on ordered-uint keys)
//   2. pt_lbvh_morton      63-bit keys: 3 x 21-bit Morton code of the centroid; the primitive index rides in a second word and
//                          breaks ties in the delta function, so keys are unique (Karras 2012, section 4)
//      rocprim::radix_sort_pairs (a library sort: not part of the render hot path)
//   3. pt_lbvh_hierarchy   binary radix tree, one thread per internal node (Karras 2012, "Maximizing Parallelism in the
//                          Construction of BVHs, Octrees, and k-d Trees", algorithm of figure 4)
//   4. pt_lbvh_fit         bottom-up AABBs and subtree triangle counts; the second thread to arrive at a node continues
//   5. pt_cwbvh_level      one thread per 8-wide node of the current level: greedy surface-area expansion of the binary
//                          subtree into <= 8 children (subtrees of <= 3 triangles become leaves: a radix-tree node covers a
//                          CONTIGUOUS range of the sorted primitives), octant-ordered slot assignment, 8-bit quantisation
//                          against a power-of-two grid, child node / triangle slots claimed with one atomicAdd each
//                          (Ylitie, Karras, Laine 2017, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide
//                          BVHs", sections 3.1-3.3), and the triangle records (e2, e1, v0 | primitive index).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "bvh_builder.h"
#include "ptmi_plugin.h"

namespace ptbvh {

namespace {

#define GPU_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return false; } } while (0)

struct Box { float mn[3], mx[3]; };

__device__ __forceinline__ uint32_t ordered(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// ---- 1. bounds --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_lbvh_bounds(const float4* __restrict__ verts, uint32_t n, Box* __restrict__ prim, uint32_t* __restrict__ sceneKeys)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float c[3] = {0.0f, 0.0f, 0.0f};
    const bool valid = i < n;
    if (valid) {
        const float4 a = verts[3 * i], b = verts[3 * i + 1], d = verts[3 * i + 2];
        Box bx;
        bx.mn[0] = fminf(a.x, fminf(b.x, d.x)); bx.mx[0] = fmaxf(a.x, fmaxf(b.x, d.x));
        bx.mn[1] = fminf(a.y, fminf(b.y, d.y)); bx.mx[1] = fmaxf(a.y, fmaxf(b.y, d.y));
        bx.mn[2] = fminf(a.z, fminf(b.z, d.z)); bx.mx[2] = fmaxf(a.z, fmaxf(b.z, d.z));
        prim[i] = bx;
        for (int k = 0; k < 3; ++k) c[k] = 0.5f * (bx.mn[k] + bx.mx[k]);
    }
    // centroid bounds: wave reduction, then one atomic pair per wave and axis
    for (int k = 0; k < 3; ++k) {
        uint32_t lo = valid ? ordered(c[k]) : 0xFFFFFFFFu, hi = valid ? ordered(c[k]) : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
        }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&sceneKeys[k], lo); atomicMax(&sceneKeys[3 + k], hi); }
    }
}

// ---- 2. Morton keys -----------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread21(uint32_t v)
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void pt_lbvh_morton(const Box* __restrict__ prim, uint32_t n, const uint32_t* __restrict__ sceneKeys,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ ids)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = unordered(sceneKeys[k]), hi = unordered(sceneKeys[3 + k]);
        const float c = 0.5f * (prim[i].mn[k] + prim[i].mx[k]);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        q[k] = min((uint32_t)(t * 2097152.0f), 2097151u);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    ids[i] = i;
}

// ---- 3. radix tree (Karras 2012) ------------------------------------------------------------------
// node references: bit 31 set = leaf (sorted position), clear = internal node index
#define LEAF_BIT 0x80000000u

struct Tree {
    const uint64_t* keys;       // sorted
    uint32_t n;
};

__device__ __forceinline__ int delta(const Tree& t, int i, int j)
{
    if (j < 0 || j >= (int)t.n) return -1;
    const uint64_t a = t.keys[i], b = t.keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));          // equal codes: the sorted position breaks the tie
}

__global__ __launch_bounds__(256) void pt_lbvh_hierarchy(Tree t, uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ parentInternal,
                                                         uint32_t* __restrict__ parentLeaf, uint32_t* __restrict__ first, uint32_t* __restrict__ last)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= (int)t.n - 1) return;
    const int d = (delta(t, i, i + 1) - delta(t, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(t, i, i - d);
    int lmax = 2;
    while (delta(t, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (delta(t, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(t, i, j);
    int s = 0;
    for (int q = (l + 1) / 2; ; q = (q + 1) / 2) {
        if (delta(t, i, i + (s + q) * d) > dnode) s += q;
        if (q <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t lref = lo == gamma ? (LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t rref = hi == gamma + 1 ? (LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = lref; right[i] = rref;
    first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
    if (lref & LEAF_BIT) parentLeaf[gamma] = (uint32_t)i; else parentInternal[gamma] = (uint32_t)i;
    if (rref & LEAF_BIT) parentLeaf[gamma + 1] = (uint32_t)i; else parentInternal[gamma + 1] = (uint32_t)i;
    if (i == 0) parentInternal[0] = 0xFFFFFFFFu;
}

// ---- 4. bottom-up fit -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_lbvh_fit(uint32_t n, const uint32_t* __restrict__ ids, const Box* __restrict__ prim, const uint32_t* __restrict__ left,
                                                   const uint32_t* __restrict__ right, const uint32_t* __restrict__ parentInternal,
                                                   const uint32_t* __restrict__ parentLeaf, Box* nodeBox, uint32_t* visits)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parentLeaf[i];
    while (node != 0xFFFFFFFFu) {
        __threadfence();
        if (atomicAdd(&visits[node], 1u) == 0u) return;            // the first arrival leaves; the second one has both children
        __threadfence();
        Box b;
        const uint32_t refs[2] = {left[node], right[node]};
        for (int k = 0; k < 3; ++k) { b.mn[k] = 1e30f; b.mx[k] = -1e30f; }
        for (int c = 0; c < 2; ++c) {
            // children written by other threads are read past the L1 of this CU (sc1: served from L2 / memory)
            const Box* src = (refs[c] & LEAF_BIT) ? &prim[ids[refs[c] & ~LEAF_BIT]] : &nodeBox[refs[c]];
            for (int k = 0; k < 3; ++k) {
                const float mn = __hip_atomic_load(&src->mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float mx = __hip_atomic_load(&src->mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.mn[k] = fminf(b.mn[k], mn); b.mx[k] = fmaxf(b.mx[k], mx);
            }
        }
        for (int k = 0; k < 3; ++k) {
            __hip_atomic_store(&nodeBox[node].mn[k], b.mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nodeBox[node].mx[k], b.mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        node = parentInternal[node];
    }
}

// ---- 5. 8-wide collapse + CWBVH encode, one tree level per launch --------------------------------------
struct WideArgs {
    const float4* verts;
    const uint32_t* ids;        // sorted position -> primitive
    const Box* prim;
    const Box* nodeBox;
    const uint32_t *left, *right, *first, *last;
    uint32_t n;
    uint32_t* wideRoot;         // [wide node] binary reference it was made from
    uint32_t maxWide;           // capacity of the wide-node pool (nodes, wideRoot)
    uint32_t* counters;         // [0] wide nodes allocated, [1] triangle rows (float4) allocated, [2] pool-overflow flag
    uint4* nodes;               // PTCwbvhNode as 5 x uint4
    float4* tris;               // PTCwbvhTri as 3 x float4
};

__device__ __forceinline__ Box ref_box(const WideArgs& A, uint32_t ref)
{
    return (ref & LEAF_BIT) ? A.prim[A.ids[ref & ~LEAF_BIT]] : A.nodeBox[ref];
}
__device__ __forceinline__ uint32_t ref_count(const WideArgs& A, uint32_t ref)
{
    return (ref & LEAF_BIT) ? 1u : A.last[ref] - A.first[ref] + 1u;
}
__device__ __forceinline__ float half_area(const Box& b)
{
    const float ex = b.mx[0] - b.mn[0], ey = b.mx[1] - b.mn[1], ez = b.mx[2] - b.mn[2];
    return ex * ey + ey * ez + ez * ex;
}

// smallest e with 255 * 2^e >= extent (extent 0 -> a tiny grid): the children then quantise into 0..255
__device__ __forceinline__ int quant_exponent(float extent)
{
    if (!(extent > 1e-36f)) return -120;
    int k;
    const float m = frexpf(extent / 255.0f, &k);                   // extent / 255 = m * 2^k, m in [0.5, 1)
    int e = (m == 0.5f) ? k - 1 : k;
    while (extent / ldexpf(1.0f, e) > 255.0f) ++e;                 // rounding of the division above: at most one step
    return e < -120 ? -120 : (e > 126 ? 126 : e);
}

__global__ __launch_bounds__(64) void pt_cwbvh_level(WideArgs A, uint32_t levelStart, uint32_t levelEnd)
{
    const uint32_t w = levelStart + blockIdx.x * 64u + threadIdx.x;
    if (w >= levelEnd) return;
    // ---- children: greedy expansion of the largest expandable child (a subtree with more than 3 triangles)
    uint32_t child[8];
    uint32_t count;
    const uint32_t rootRef = A.wideRoot[w];
    if ((rootRef & LEAF_BIT) || ref_count(A, rootRef) <= 3u) { child[0] = rootRef; count = 1; }      // tiny scene: a root with one leaf
    else { child[0] = A.left[rootRef]; child[1] = A.right[rootRef]; count = 2; }
    while (count < 8u) {
        int best = -1;
        float bestArea = -1.0f;
        for (uint32_t i = 0; i < count; ++i) {
            if ((child[i] & LEAF_BIT) || ref_count(A, child[i]) <= 3u) continue;
            const float a = half_area(ref_box(A, child[i]));
            if (a > bestArea) { bestArea = a; best = (int)i; }
        }
        if (best < 0) break;
        const uint32_t r = child[best];
        child[best] = A.left[r];
        child[count++] = A.right[r];
    }
    // ---- node box, child boxes
    Box cb[8], nb;
    for (int k = 0; k < 3; ++k) { nb.mn[k] = 1e30f; nb.mx[k] = -1e30f; }
    for (uint32_t i = 0; i < count; ++i) {
        cb[i] = ref_box(A, child[i]);
        for (int k = 0; k < 3; ++k) { nb.mn[k] = fminf(nb.mn[k], cb[i].mn[k]); nb.mx[k] = fmaxf(nb.mx[k], cb[i].mx[k]); }
    }
    // ---- octant-ordered slots: greedy assignment on cost[s][i] = (child centre - node centre) . (+-1, +-1, +-1)_s
    float cost[8][8];
    int slotOf[8], childOf[8];
    for (int s = 0; s < 8; ++s) childOf[s] = -1;
    for (uint32_t i = 0; i < 8u; ++i) slotOf[i] = -1;
    for (int s = 0; s < 8; ++s) {
        const float dx = (s & 4) ? -1.0f : 1.0f, dy = (s & 2) ? -1.0f : 1.0f, dz = (s & 1) ? -1.0f : 1.0f;
        for (uint32_t i = 0; i < count; ++i) {
            const float cx = 0.5f * (cb[i].mn[0] + cb[i].mx[0]) - 0.5f * (nb.mn[0] + nb.mx[0]);
            const float cy = 0.5f * (cb[i].mn[1] + cb[i].mx[1]) - 0.5f * (nb.mn[1] + nb.mx[1]);
            const float cz = 0.5f * (cb[i].mn[2] + cb[i].mx[2]) - 0.5f * (nb.mn[2] + nb.mx[2]);
            cost[s][i] = cx * dx + cy * dy + cz * dz;
        }
    }
    for (uint32_t round = 0; round < count; ++round) {
        float best = 3e30f;
        int bs = -1, bi = -1;
        for (int s = 0; s < 8; ++s) {
            if (childOf[s] >= 0) continue;
            for (uint32_t i = 0; i < count; ++i)
                if (slotOf[i] < 0 && cost[s][i] < best) { best = cost[s][i]; bs = s; bi = (int)i; }
        }
        if (bs < 0) {                                                // NaN costs (degenerate boxes): first free slot
            for (uint32_t i = 0; i < count && bi < 0; ++i) if (slotOf[i] < 0) bi = (int)i;
            for (int s = 0; s < 8 && bs < 0; ++s) if (childOf[s] < 0) bs = s;
        }
        childOf[bs] = bi; slotOf[bi] = bs;
    }
    // ---- count inner children / leaf triangles, claim their storage (inner children and leaf triangles are contiguous, in slot order)
    uint32_t numInner = 0, numTris = 0;
    for (int s = 0; s < 8; ++s) {
        if (childOf[s] < 0) continue;
        const uint32_t r = child[childOf[s]];
        const uint32_t c = ref_count(A, r);
        if (!(r & LEAF_BIT) && c > 3u) numInner++; else numTris += c;
    }
    const uint32_t childBase = numInner ? atomicAdd(&A.counters[0], numInner) : 0u;
    const uint32_t triBase = numTris ? atomicAdd(&A.counters[1], numTris * 3u) : 0u;
    // pool overflow (cannot happen for a well-formed hierarchy: < 2n wide nodes, exactly 3n triangle rows): write nothing past the
    // pools, leave a flag for the host, which fails the build
    if (childBase + numInner > A.maxWide || triBase + numTris * 3u > 3u * A.n) { atomicOr(&A.counters[2], 1u); return; }
    // ---- encode
    const int ex = quant_exponent(nb.mx[0] - nb.mn[0]), ey = quant_exponent(nb.mx[1] - nb.mn[1]), ez = quant_exponent(nb.mx[2] - nb.mn[2]);
    const float px = ldexpf(1.0f, ex), py = ldexpf(1.0f, ey), pz = ldexpf(1.0f, ez);
    uint8_t meta[8], q[48];
    for (int i = 0; i < 8; ++i) meta[i] = 0;
    for (int i = 0; i < 48; ++i) q[i] = 0;
    uint32_t imask = 0, innerSeen = 0, trisSeen = 0;
    for (int s = 0; s < 8; ++s) {
        if (childOf[s] < 0) continue;
        const int i = childOf[s];
        const Box& b = cb[i];
        // floor / ceil against the node's grid; clamped (a child box can never be outside its parent's, but the division rounds)
        q[s + 0] = (uint8_t)fminf(fmaxf(floorf((b.mn[0] - nb.mn[0]) / px), 0.0f), 255.0f);
        q[s + 8] = (uint8_t)fminf(fmaxf(floorf((b.mn[1] - nb.mn[1]) / py), 0.0f), 255.0f);
        q[s + 16] = (uint8_t)fminf(fmaxf(floorf((b.mn[2] - nb.mn[2]) / pz), 0.0f), 255.0f);
        q[s + 24] = (uint8_t)fminf(fmaxf(ceilf((b.mx[0] - nb.mn[0]) / px), 0.0f), 255.0f);
        q[s + 32] = (uint8_t)fminf(fmaxf(ceilf((b.mx[1] - nb.mn[1]) / py), 0.0f), 255.0f);
        q[s + 40] = (uint8_t)fminf(fmaxf(ceilf((b.mx[2] - nb.mn[2]) / pz), 0.0f), 255.0f);
        const uint32_t r = child[i];
        const uint32_t c = ref_count(A, r);
        if (!(r & LEAF_BIT) && c > 3u) {
            imask |= 1u << s;
            meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
            A.wideRoot[childBase + innerSeen] = r;
            innerSeen++;
        } else {
            const uint32_t unary = c == 1u ? 1u : (c == 2u ? 3u : 7u);
            meta[s] = (uint8_t)((unary << 5) | trisSeen);
            const uint32_t firstPos = (r & LEAF_BIT) ? (r & ~LEAF_BIT) : A.first[r];
            for (uint32_t j = 0; j < c; ++j) {
                const uint32_t primIdx = A.ids[firstPos + j];
                const float4 v0 = A.verts[3 * primIdx], v1 = A.verts[3 * primIdx + 1], v2 = A.verts[3 * primIdx + 2];
                float4* out = A.tris + (size_t)triBase + (size_t)(trisSeen + j) * 3u;
                out[0] = make_float4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, 0.0f);
                out[1] = make_float4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, 0.0f);
                out[2] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(primIdx));
            }
            trisSeen += c;
        }
    }
    uint4* out = A.nodes + (size_t)w * 5u;
    out[0] = make_uint4(__float_as_uint(nb.mn[0]), __float_as_uint(nb.mn[1]), __float_as_uint(nb.mn[2]),
                        ((uint32_t)ex & 255u) | (((uint32_t)ey & 255u) << 8) | (((uint32_t)ez & 255u) << 16) | (imask << 24));
    auto pack4 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    out[1] = make_uint4(childBase, triBase, pack4(meta), pack4(meta + 4));
    out[2] = make_uint4(pack4(q + 0), pack4(q + 4), pack4(q + 8), pack4(q + 12));
    out[3] = make_uint4(pack4(q + 16), pack4(q + 20), pack4(q + 24), pack4(q + 28));
    out[4] = make_uint4(pack4(q + 32), pack4(q + 36), pack4(q + 40), pack4(q + 44));
}

struct DevMem {
    std::vector<void*> ptrs;
    ~DevMem() { for (void* p : ptrs) hipFree(p); }
    template <class T> bool alloc(T** out, size_t count, std::string& err)
    {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, count * sizeof(T) + 16);
        if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
        ptrs.push_back(p);
        *out = (T*)p;
        return true;
    }
};

} // namespace

// Builds into `out` (host vectors, the layout BuildBVH produces).  deviceMs: device time from the upload's end to the last
// kernel; returns false with a message on any HIP error.
bool build_cwbvh_device(int device, const PTFloat4* verts, uint32_t n, Cwbvh& out, double* deviceMs, std::string& err)
{
    if (!verts || n == 0) { err = "no triangles"; return false; }
    if (n > 0x7FFFFFF0u / 3u) { err = "too many triangles"; return false; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { err = "no such HIP device (this builder has no CPU fallback; BuildBVH is the CPU builder)"; return false; }
    // the calling thread's current device is left as it was found
    struct DeviceGuard { int prev = -1; ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); } } dguard;
    if (hipGetDevice(&dguard.prev) != hipSuccess) dguard.prev = -1;
    GPU_TRY(hipSetDevice(device));
    hipStream_t stream;
    GPU_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { hipStreamDestroy(s); } } guard{stream};
    DevMem mem;
    float4* dVerts; Box *dPrim, *dNodeBox; uint32_t *dScene, *dIds, *dIdsSorted, *dLeft, *dRight, *dParentI, *dParentL, *dFirst, *dLast, *dVisits, *dWideRoot, *dCounters;
    uint64_t *dKeys, *dKeysSorted; uint4* dNodes; float4* dTris;
    const size_t maxWide = 2 * (size_t)n + 2;                         // a wide node has >= 2 inner children or at least one leaf child of its own: < 2n nodes
    if (!mem.alloc(&dVerts, (size_t)n * 3, err) || !mem.alloc(&dPrim, n, err) || !mem.alloc(&dNodeBox, n, err) || !mem.alloc(&dScene, 8, err) ||
        !mem.alloc(&dIds, n, err) || !mem.alloc(&dIdsSorted, n, err) || !mem.alloc(&dKeys, n, err) || !mem.alloc(&dKeysSorted, n, err) ||
        !mem.alloc(&dLeft, n, err) || !mem.alloc(&dRight, n, err) || !mem.alloc(&dParentI, n, err) || !mem.alloc(&dParentL, n, err) ||
        !mem.alloc(&dFirst, n, err) || !mem.alloc(&dLast, n, err) || !mem.alloc(&dVisits, n, err) || !mem.alloc(&dWideRoot, maxWide, err) ||
        !mem.alloc(&dCounters, 4, err) || !mem.alloc(&dNodes, maxWide * 5, err) || !mem.alloc(&dTris, (size_t)n * 3, err))
        return false;
    GPU_TRY(hipMemcpyAsync(dVerts, verts, (size_t)n * 3 * sizeof(float4), hipMemcpyHostToDevice, stream));
    hipEvent_t e0, e1;
    GPU_TRY(hipEventCreate(&e0));
    GPU_TRY(hipEventCreate(&e1));
    struct EventGuard { hipEvent_t a, b; ~EventGuard() { hipEventDestroy(a); hipEventDestroy(b); } } eguard{e0, e1};
    GPU_TRY(hipEventRecord(e0, stream));
    const uint32_t sceneInit[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    GPU_TRY(hipMemcpyAsync(dScene, sceneInit, sizeof(sceneInit), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipMemsetAsync(dVisits, 0, (size_t)n * 4, stream));
    const uint32_t blocks = (n + 255u) / 256u;
    // a launch that fails (no code object for this device, bad configuration) leaves its outputs untouched: check every one
    hipLaunchKernelGGL(pt_lbvh_bounds, dim3(blocks), dim3(256), 0, stream, dVerts, n, dPrim, dScene);
    GPU_TRY(hipGetLastError());
    hipLaunchKernelGGL(pt_lbvh_morton, dim3(blocks), dim3(256), 0, stream, dPrim, n, dScene, dKeys, dIds);
    GPU_TRY(hipGetLastError());
    {
        size_t tempBytes = 0;
        GPU_TRY(rocprim::radix_sort_pairs(nullptr, tempBytes, dKeys, dKeysSorted, dIds, dIdsSorted, n, 0, 63, stream));
        char* dTemp;
        if (!mem.alloc(&dTemp, tempBytes, err)) return false;
        GPU_TRY(rocprim::radix_sort_pairs(dTemp, tempBytes, dKeys, dKeysSorted, dIds, dIdsSorted, n, 0, 63, stream));
    }
    uint32_t rootRef = LEAF_BIT | 0u;                                 // one triangle: the root is that leaf
    if (n > 1) {
        Tree t{dKeysSorted, n};
        hipLaunchKernelGGL(pt_lbvh_hierarchy, dim3((n - 1 + 255u) / 256u), dim3(256), 0, stream, t, dLeft, dRight, dParentI, dParentL, dFirst, dLast);
        GPU_TRY(hipGetLastError());
        hipLaunchKernelGGL(pt_lbvh_fit, dim3(blocks), dim3(256), 0, stream, n, dIdsSorted, dPrim, dLeft, dRight, dParentI, dParentL, dNodeBox, dVisits);
        GPU_TRY(hipGetLastError());
        rootRef = 0u;
    }
    // level 0 = the root wide node
    const uint32_t countersInit[4] = {1u, 0u, 0u, 0u};
    GPU_TRY(hipMemcpyAsync(dCounters, countersInit, sizeof(countersInit), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipMemcpyAsync(dWideRoot, &rootRef, 4, hipMemcpyHostToDevice, stream));
    WideArgs A{dVerts, dIdsSorted, dPrim, dNodeBox, dLeft, dRight, dFirst, dLast, n, dWideRoot, (uint32_t)maxWide, dCounters, dNodes, dTris};
    uint32_t levelStart = 0, levelEnd = 1, levels = 0;
    while (levelStart < levelEnd) {
        if (levelEnd > maxWide) { err = "wide-node pool exhausted"; return false; }
        hipLaunchKernelGGL(pt_cwbvh_level, dim3((levelEnd - levelStart + 63u) / 64u), dim3(64), 0, stream, A, levelStart, levelEnd);
        GPU_TRY(hipGetLastError());
        uint32_t allocated = 0;
        GPU_TRY(hipMemcpyAsync(&allocated, dCounters, 4, hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
        if (allocated < levelEnd || allocated > maxWide) { err = "wide-node pool exhausted or corrupted (allocated " + std::to_string(allocated) + " of " + std::to_string(maxWide) + ")"; return false; }
        levelStart = levelEnd;
        levelEnd = allocated;
        if (++levels > 4096u) { err = "tree depth out of range"; return false; }
    }
    GPU_TRY(hipEventRecord(e1, stream));
    const uint32_t wideCount = levelEnd;
    {
        // every triangle must have been emitted exactly once (counters[1] = triangle rows written), at least the root node exists
        uint32_t counters[4] = {0, 0, 0, 0};
        GPU_TRY(hipMemcpyAsync(counters, dCounters, sizeof(counters), hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
        if (wideCount < 1u || counters[1] != 3u * n || counters[2] != 0u) {
            err = "device build incomplete: " + std::to_string(wideCount) + " nodes, " + std::to_string(counters[1]) + " triangle rows for " + std::to_string(n) + " triangles";
            return false;
        }
    }
    out.triCount = n;
    out.usedBlocks = wideCount * 5u;
    out.nodeData.assign((size_t)wideCount * 5, PTFloat4{0, 0, 0, 0});
    out.triData.assign((size_t)n * 3, PTFloat4{0, 0, 0, 0});
    GPU_TRY(hipMemcpyAsync(out.nodeData.data(), dNodes, (size_t)wideCount * 80, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(out.triData.data(), dTris, (size_t)n * 48, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    if (deviceMs) { float ms = 0.0f; GPU_TRY(hipEventElapsedTime(&ms, e0, e1)); *deviceMs = ms; }
    return true;
}

} // namespace ptbvh
