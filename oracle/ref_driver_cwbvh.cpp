// oracle/ref_driver_cwbvh.cpp — TEST INFRASTRUCTURE, never shipped, never on the product path.
//
// Second driver around the REFERENCE's tinybvh (Assets/Plugins/Web/tiny_bvh.h, v1.6.7), compiled from the header where
// it lies under /root/reference (oracle/Makefile: reached through -I, nothing copied).  Unlike ref_driver.cpp this TU is
// built WITHOUT TINYBVH_NO_SIMD and with -mavx2 -mfma -mlzcnt -mpopcnt, because the reference only compiles its own CWBVH
// walker, BVH8_CWBVH::Intersect (tiny_bvh.h:7006-7114: scalar code, the CPU twin of util/bvh.hlsl:126-199), inside the
// BVH_USEAVX section (tiny_bvh.h:6309, :6990).  The tree is built through the SCALAR pipeline of BVH8_CWBVH::Build
// (tiny_bvh.h:5787-5795) with BVH::Build in place of BuildDefault -- BuildDefault would pick the "nearly identical" AVX
// binned builder in this TU (tiny_bvh.h:1816-1830) -- so nodes and triangles are byte-identical to what the reference
// plugin (TINYBVH_NO_SIMD, plugin.h:9-11) hands to the shader; tests/test_oracle.py checks that with memcmp.
//
// The recipe compiles this TU with -ffp-contract=off: gcc's default (-ffp-contract=fast) would fuse the dot and cross
// products of MOLLER_TRUMBORE_TEST (tiny_bvh.h:1643-1655) into fmas under -mfma, while the shader contract of this project
//  evaluates them as separate IEEE operations.  With contraction off the reference's triangle arithmetic is
// operation for operation that of util/bvh.hlsl:23-59, so (t, u, v) of a hit must agree BIT FOR BIT with the restatement.
//
// What it is used for:
//   * pinning the oracle's RayIntersectBvh restatement (oracle/pt_oracle.cpp) against the reference's own CWBVH walker:
//     primitive, t, u, v per ray (tests/test_oracle.py, golden vectors tests/golden/trace_cwbvh_*.npz);
//   * the second CPU baseline leg of SURVEY.md 8(d): BVH8_CWBVH::Intersect over the identical ray list (bench.py).
//
// Output: oracle/_ref/libref_cwbvh.so (git-ignored, travels to the GPU box with gpurun).
#define TINYBVH_IMPLEMENTATION
#define NO_THREADED_BUILDS
#include "tiny_bvh.h"

#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

using namespace tinybvh;

extern "C" {

// `verts` = 3*triCount float4 (borrowed: must outlive the handle)
void* ref_cwbvh_build(const float* verts, int triCount, double* buildMs)
{
    auto t0 = std::chrono::steady_clock::now();
    BVH8_CWBVH* c = new BVH8_CWBVH();
    const bvhvec4slice slice((const bvhvec4*)verts, (uint32_t)triCount * 3u, sizeof(bvhvec4));
    c->bvh8.bvh.context = c->bvh8.context = c->context;       // tiny_bvh.h:5789
    c->bvh8.bvh.Build(slice);                                 // :5790 with the scalar builder (what BuildDefault is under TINYBVH_NO_SIMD)
    c->bvh8.bvh.Compact();                                    // :5791
    c->bvh8.bvh.SplitLeafs(3);                                // :5792
    c->bvh8.ConvertFrom(c->bvh8.bvh, false);                  // :5793
    c->ConvertFrom(c->bvh8, true);                            // :5794
    auto t1 = std::chrono::steady_clock::now();
    if (buildMs) *buildMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return c;
}

void ref_cwbvh_destroy(void* h) { delete (BVH8_CWBVH*)h; }

// borrowed pointers to the node / triangle arrays, sizes in bytes as plugin.cpp:60-70 reports them
void ref_cwbvh_data(void* h, const void** nodes, long long* nodeBytes, const void** tris, long long* triBytes)
{
    const BVH8_CWBVH* c = (const BVH8_CWBVH*)h;
    *nodes = c->bvh8Data; *nodeBytes = (long long)c->usedBlocks * 16;
    *tris = c->bvh8Tris; *triBytes = (long long)c->triCount * 3 * 16;
}

// rays: n x 8 floats {ox,oy,oz, dx,dy,dz, tmax, kind} (kind is ignored: BVH8_CWBVH::IsOccluded is FALLBACK_SHADOW_QUERY,
// i.e. Intersect + a comparison, tiny_bvh.h:1353).  out (may be NULL): per ray {t, u, v, prim bits}; prim = 0xFFFFFFFF on a miss.
// Returns wall seconds for the batch.
double ref_cwbvh_trace(void* h, const float* rays, long long n, int threads, float* out)
{
    const BVH8_CWBVH* c = (const BVH8_CWBVH*)h;
    if (threads < 1) threads = 1;
    auto worker = [&](long long b, long long e) {
        for (long long i = b; i < e; i++) {
            const float* src = rays + i * 8;
            Ray r;
            memset(&r, 0, sizeof(Ray));
            r.O = bvhvec3(src[0], src[1], src[2]);
            r.D = bvhvec3(src[3], src[4], src[5]);
            r.rD = bvhvec3(1.0f / src[3], 1.0f / src[4], 1.0f / src[5]);
            r.hit.t = src[6];
            r.hit.prim = 0xFFFFFFFFu;
            r.mask = RAY_MASK_INTERSECT_ALL;
            c->Intersect(r);
            if (out) {
                const bool hit = r.hit.t < src[6];
                out[i * 4 + 0] = r.hit.t;
                out[i * 4 + 1] = hit ? r.hit.u : 0.0f;
                out[i * 4 + 2] = hit ? r.hit.v : 0.0f;
                const uint32_t p = hit ? r.hit.prim : 0xFFFFFFFFu;
                memcpy(&out[i * 4 + 3], &p, 4);
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
