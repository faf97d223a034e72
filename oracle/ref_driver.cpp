// oracle/ref_driver.cpp — TEST INFRASTRUCTURE, never shipped, never on the product path.
//
// A thin C driver around the REFERENCE's own tinybvh (Assets/Plugins/Web/tiny_bvh.h, v1.6.7),
// compiled from the header where it lies under /root/reference (see oracle/Makefile: the header
// is found through -I, no reference source is copied into this repository).  It is built with the
// reference plugin's own configuration (Assets/Plugins/Web/plugin.h:9-11: TINYBVH_NO_SIMD,
// NO_THREADED_BUILDS), so BVH::Build is the scalar binned-SAH builder the plugin ships with.
//
// What it is used for:
//   * the "tinybvh CPU traversal" baseline of BASELINE.md §3 (BVH::Intersect / BVH::IsOccluded,
//     tiny_bvh.h:3182-3264 / 3342-3413) timed on the GPU box's host cores (bench.py cpu_baseline,
//     kind = "reference");
//   * cross-checking the oracle's restated CWBVH traversal against an independent reference walker.
//
// Output: oracle/_ref/libref_tinybvh.so (git-ignored, travels to the GPU box with gpurun).
#define TINYBVH_IMPLEMENTATION
#define TINYBVH_NO_SIMD
#define NO_THREADED_BUILDS
#include "tiny_bvh.h"

#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

using namespace tinybvh;

extern "C" {

// Builds a 2-wide BVH exactly as BVH8_CWBVH::Build does before its conversions
// (tiny_bvh.h:5787-5795 -> BuildDefault -> Build under TINYBVH_NO_SIMD).
// `verts` = 3*triCount float4 (borrowed: must outlive the handle).
void* ref_bvh_build(const float* verts, int triCount, double* buildMs)
{
    auto t0 = std::chrono::steady_clock::now();
    BVH* bvh = new BVH();
    bvh->Build((const bvhvec4*)verts, (uint32_t)triCount);
    auto t1 = std::chrono::steady_clock::now();
    if (buildMs) *buildMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return bvh;
}

void ref_bvh_destroy(void* h) { delete (BVH*)h; }

int ref_bvh_used_nodes(void* h) { return (int)((BVH*)h)->usedNodes; }

static inline void make_ray(Ray& r, const float* src)
{
    memset(&r, 0, sizeof(Ray));
    r.O = bvhvec3(src[0], src[1], src[2]);
    r.D = bvhvec3(src[3], src[4], src[5]);
    r.rD = bvhvec3(1.0f / src[3], 1.0f / src[4], 1.0f / src[5]);
    r.hit.t = src[6];
    r.mask = RAY_MASK_INTERSECT_ALL;
}

// rays: n x 8 floats {ox,oy,oz, dx,dy,dz, tmax, kind}; kind 0 = closest hit, 1 = shadow.
// outT/outPrim (may be NULL): per-ray results (t = tmax and prim = 0xFFFFFFFF on a miss;
// shadow rays report t = 0 / prim = 1 when occluded).  Returns wall seconds for the batch.
double ref_bvh_trace(void* h, const float* rays, long long n, int threads, float* outT, unsigned* outPrim)
{
    const BVH* bvh = (const BVH*)h;
    if (threads < 1) threads = 1;
    auto worker = [&](long long b, long long e) {
        for (long long i = b; i < e; i++) {
            const float* src = rays + i * 8;
            Ray r;
            make_ray(r, src);
            if (src[7] != 0.0f) {
                bool occ = bvh->IsOccluded(r);
                if (outT) outT[i] = occ ? 0.0f : src[6];
                if (outPrim) outPrim[i] = occ ? 1u : 0xFFFFFFFFu;
            } else {
                r.hit.prim = 0xFFFFFFFFu;
                bvh->Intersect(r);
                if (outT) outT[i] = r.hit.t;
                if (outPrim) outPrim[i] = r.hit.t < src[6] ? r.hit.prim : 0xFFFFFFFFu;
            }
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (threads == 1) worker(0, n);
    else {
        std::vector<std::thread> pool;
        long long chunk = (n + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
            long long b = t * chunk, e = b + chunk > n ? n : b + chunk;
            if (b < e) pool.emplace_back(worker, b, e);
        }
        for (auto& th : pool) th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

} // extern "C"
