// pt_api.hip — the PT* render entry points of include/ptmi_plugin.h (Part 2).
//
// They replace what the reference does through Unity's ComputeShader API: buffer uploads
// (BVHScene.cs:640-667 ComputeBuffer.SetData), per-frame uniforms + DispatchCompute
// (PathTracer.cs:226-252) and the ping-pong frame bookkeeping (PathTracer.cs:246-247, 268-272).
// There is NO CPU fallback: without a HIP device PTCreate fails with PT_ERR_NO_DEVICE.
#include "pt_launch.h"
hipError_t pt_launch_wavefront_b(const DScene& S, const PTFrameParams& P, const PTBatch& batch, const float4* accumulated, float4* output,
                                 const PTTileMap& tm, const PTWfBuffers& B, unsigned long long* gstats, bool fullStats,
                                 hipStream_t stream, hipEvent_t orderAfter, bool zeroOutputFirst,
                                 uint32_t* launchesOut, int traceVariant, uint32_t iterationsOverride);

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// Consecutive passes run on up to PT_WF_SETS streams so that one pass's launch tails are filled by its neighbours' kernels.
// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them the context stream's) and
// streams that share a queue serialise: more sets than queues is SLOWER than fewer sets.  The library never
// touches the host's environment: it reads the variable the host chose (export GPU_MAX_HW_QUEUES=16 before
// the first HIP call; bench.py, host/pt_host.c and the Python mirror do) and sizes the default number of passes in flight
// to it: >= 16 queues -> 12 sets, >= 8 -> 6, otherwise 3.  PTSetPassesInFlight overrides.
uint32_t default_passes_in_flight()
{
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const long q = e ? strtol(e, nullptr, 10) : 4;
    uint32_t n = q >= 16 ? 12u : (q >= 8 ? 6u : 3u);
    return n < (uint32_t)PT_WF_SETS ? n : (uint32_t)PT_WF_SETS;
}

// roctx ranges around every pass / scene upload (rocprofv3 --marker-trace shows them; the reference wraps its dispatch in
// _cmd.BeginSample / EndSample("Path Tracer"), PathTracer.cs:226,252).  Resolved lazily: a host on a runtime-only ROCm install
// without librocprofiler-sdk-roctx still loads the plugin, and the ranges are no-ops.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const RoctxApi& roctx_api() { static RoctxApi api; return api; }

thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;           // allocation capacity (a smaller scene re-uses it)
    size_t used = 0;            // bytes of the scene currently uploaded
};

struct EventPair { hipEvent_t start, stop; uint32_t launches; };

struct RoctxRange {
    explicit RoctxRange(const char* name) { if (roctx_api().push) roctx_api().push(name); }
    ~RoctxRange() { if (roctx_api().pop) roctx_api().pop(); }
};

} // namespace

struct PTContext {
    int device = 0;
    hipStream_t stream = nullptr;
    DeviceBuffer nodes, tris, attrs, materials, lights, lightConst, tex, tlas, tlasBfs, instances, instByLeaf, envTex, envCdf;
    void* batchScratch = nullptr;               // PTRenderPassBatchTo under the megakernel: the frame the odd passes of a batch write
    size_t batchScratchBytes = 0;
    void* present = nullptr;                    // PTPresentToHost staging frame
    size_t presentBytes = 0;
    DScene scene = {};
    bool hasScene = false;
    float4* frames[2] = {nullptr, nullptr};
    uint32_t frameW = 0, frameH = 0;
    int cur = 0;
    int rank = 0, world = 1;
    int statsLevel = 0;
    int schedule = -1;                          // -1 auto (default), 0 megakernel, 1 wavefront + refill trace, 2, 3: see PTSetSchedule
    unsigned long long* dStats = nullptr;       // 16 counters, PTStats order
    bool profiling = false;
    std::vector<EventPair> pending;             // recorded, not yet read
    std::vector<EventPair> freeEvents;
    PTTimings timings = {};
    // wavefront schedules: PT_WF_SETS path-state sets, each with its own stream, so consecutive passes overlap
    struct WfSet {
        PTWfBuffers wf = {};
        void* arena = nullptr;
        size_t arenaBytes = 0;
        hipStream_t stream = nullptr;
        hipEvent_t callEv = nullptr, done = nullptr;
    } sets[PT_WF_SETS];
    uint32_t nextSet = 0;
    uint32_t residentWaves = 0;                 // CUs x 4 SIMDs x 8 waves (device property, read once)
    uint32_t subFrames = 1;                     // launch chains a pass is cut into (PTSetSubFrames)
    uint32_t numSets = 0;                       // passes in flight = state sets in use (PTSetPassesInFlight); carved on first use
    uint32_t wfIterations = 0;                  // 0 = automatic
    uint32_t shadeSorted = PT_WF_SHADE_SORTED;  // PT_WF_SHADE_SORTED in the environment overrides (experiments)
    uint32_t tailCut = PT_WF_TAIL_CUT;          // wavefront tail cut (pt_wavefront.hip tail_cut); PT_WF_TAIL_CUT in the environment overrides (experiments)
};

namespace {

int upload(PTContext* c, DeviceBuffer& b, const void* src, size_t bytes)
{
    if (b.ptr && b.bytes < bytes) { hipFree(b.ptr); b.ptr = nullptr; b.bytes = 0; }
    b.used = bytes;
    if (bytes == 0) return PT_OK;
    if (!b.ptr) {
        HIP_TRY(hipMalloc(&b.ptr, bytes));
        b.bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, c->stream));
    return PT_OK;
}

int ensure_frames(PTContext* c, uint32_t w, uint32_t h)
{
    if (c->frames[0] && c->frameW == w && c->frameH == h) return PT_OK;
    for (int i = 0; i < 2; i++) if (c->frames[i]) { hipFree(c->frames[i]); c->frames[i] = nullptr; }
    size_t bytes = (size_t)w * h * sizeof(float4);
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipMalloc((void**)&c->frames[i], bytes));
        HIP_TRY(hipMemsetAsync(c->frames[i], 0, bytes, c->stream));
    }
    c->frameW = w; c->frameH = h;
    c->cur = 0;                         // PrepareRenderTexture re-created the targets -> Reset() (PathTracer.cs:211-215)
    return PT_OK;
}

// one arena for all slot-indexed arrays of a wavefront state set; re-carved when the slot count changes
int ensure_wavefront(PTContext* c, PTContext::WfSet& set, uint32_t numSlots, uint32_t maxIterations)
{
    if (!set.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&set.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&set.callEv, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&set.done, hipEventDisableTiming));
    }
    const bool needTlas = c->scene.hasTlas != 0u;
    set.wf.tailCut = c->tailCut;
    set.wf.shadeSorted = c->shadeSorted;
    const bool needRec = c->shadeSorted == 2u;
    if (set.wf.flags && set.wf.numSlots == numSlots && set.wf.maxIterations >= maxIterations && (!needTlas || set.wf.tlasSpill) && (!needRec || set.wf.recPlanes)) return PT_OK;
    const size_t n = numSlots;
    const uint32_t numRows = 4u * (numSlots >> 6);
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // trace waves own 64 or 128 consecutive slots (pt_wf_wide_ranges); a wave addresses 64 slab rows and PT_WF_SUSPEND records
    if (c->residentWaves == 0u) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        c->residentWaves = (uint32_t)prop.multiProcessorCount * 4u * 8u;
    }
    const uint32_t residentWaves = c->residentWaves;
    // the HAS_TLAS refill kernel always uses PT_WF_RANGE = 128 slots per wave; the plain one 64 or 128 (pt_wf_wide_ranges)
    const size_t traceWaves = pt_wf_max_trace_waves(numSlots, residentWaves);
    const size_t spillBytes = traceWaves * 64u * (size_t)(needTlas ? PT_WF_SPILL_ROW_ENTRIES : PT_BVH_STACK_SIZE - PT_WF_LDS_STACK) * sizeof(uint2);
    const size_t suspBytes = traceWaves * (size_t)(PT_WF_SUSPEND ? PT_WF_SUSPEND : 1u) * 6 * sizeof(uint4);
    const size_t tlasSpillBytes = needTlas ? traceWaves * 64u * (size_t)PT_BVH_STACK_SIZE * sizeof(uint32_t) : 0;
    size_t total = align(n * 4) * 2 + align(n * 16) * PT_F4_COUNT + align(n * 2) + align((size_t)maxIterations * 4 * PT_WF_ALIVE_SHARDS) +
                   align((size_t)numRows * 16 * 8) + align(8 * 32 * 4) + (needRec ? align(n * 16) * 10 + align(n * 4) + align(64) : 0) + align(spillBytes) + align(suspBytes) + align(traceWaves * 4) + align(tlasSpillBytes);
    // The arena only ever grows: a different slot count (another batch size, another frame size) that fits is a new carving of
    // the same memory -- launches already enqueued on the set's stream keep the pointers they were given and finish first
    // (stream order), so neither a synchronisation nor an allocation lands in a caller's timed region.
    if (set.arena && set.arenaBytes < total) {
        HIP_TRY(hipStreamSynchronize(set.stream));
        hipFree(set.arena); set.arena = nullptr; set.arenaBytes = 0;
    }
    if (!set.arena) {
        HIP_TRY(hipMalloc(&set.arena, total));
        set.arenaBytes = total;
    }
    char* p = (char*)set.arena;
    auto carve = [&](size_t bytes) { char* q = p; p += align(bytes); return q; };
    PTWfBuffers& B = set.wf;
    B.flags = (uint32_t*)carve(n * 4);
    B.rng = (uint32_t*)carve(n * 4);
    float4* dummy = nullptr;
    float4** arrs[PT_F4_COUNT] = {&B.ray[0], &dummy, &B.ray[1], &dummy, &B.ray[2], &dummy, &B.rad, &B.thr, &B.color, &B.envC, &B.lightC, &B.pthr, &B.hit, &B.hit2, &B.pixsum};   // PT_F4_* order; a ray-record array spans two strides
    for (auto a : arrs) *a = (float4*)carve(n * 16);
    B.f4base = B.ray[0];
    B.f4stride = (uint32_t)(align(n * 16) / 16);
    B.occl = (uint8_t*)carve(n * 2);
    B.aliveFlags = (uint32_t*)carve((size_t)maxIterations * 4 * PT_WF_ALIVE_SHARDS);
    B.statRows = (unsigned long long*)carve((size_t)numRows * 16 * 8);
    B.chunkHeads = (uint32_t*)carve(8 * 32 * 4);
    B.stackSpill = (uint2*)carve(spillBytes);
    B.susp = (uint4*)carve(suspBytes);
    B.suspCount = (uint32_t*)carve(traceWaves * 4);
    B.recPlanes = nullptr; B.recDest = nullptr; B.recCount = nullptr;
    if (needRec) {
        B.recPlanes = (float4*)carve(n * 16);
        for (int k = 1; k < 10; ++k) carve(n * 16);
        B.recDest = (uint32_t*)carve(n * 4);
        B.recCount = (uint32_t*)carve(64);
    }
    B.tlasSpill = needTlas ? (uint32_t*)carve(tlasSpillBytes) : nullptr;
    B.residentWaves = residentWaves;
    B.tailCut = c->tailCut;
    B.numSlots = numSlots;
    B.numStatRows = numRows;
    B.maxIterations = maxIterations;
    HIP_TRY(hipMemsetAsync(B.statRows, 0, (size_t)numRows * 16 * 8, set.stream));
    return PT_OK;
}

// One host pass over everything the kernels will index with scene data (D3D / WebGPU robust buffer access returns zeros for
// an out-of-range read; a HIP kernel faults the GPU): triangle -> material, material -> texture descriptor -> texels,
// CWBVH node -> child nodes / triangle rows, triangle row -> attribute record, TLAS node -> nodes / instances, instance ->
// BLAS offsets.  Costs a few milliseconds for a 250k-triangle scene, once per PTSetScene.
bool validate_scene(const PTSceneDesc& s, std::string& why)
{
    const uint64_t nodeCount = s.bvhNodesBytes / 80u, triRows = s.bvhTrisBytes / 16u, attrCount = s.triAttrsBytes / 128u;
    const bool textures = (s.features & PT_FEATURE_HAS_TEXTURES) != 0, tlasOn = (s.features & PT_FEATURE_HAS_TLAS) != 0;
    auto bad = [&](const std::string& m) { why = m; return false; };
    // materials -> textures (util/material.hlsl:8-82 reads the five slots as float indices; < 0 = none)
    const float* mats = (const float*)s.materials;
    for (uint32_t m = 0; m < s.materialCount; ++m) {
        const int slots[4] = {22, 23, 25, 26};                            // baseColor, metallicRoughness, emission, occlusion (normal map is unused)
        for (int k : slots) {
            const float f = mats[(size_t)m * 32 + k];
            if (!textures || f < 0.0f) continue;
            if (!(f < 1.0e9f)) return bad("material " + std::to_string(m) + ": texture index is not a finite number");
            const uint64_t t = (uint64_t)f;
            if (4 * t + 3 >= s.textureDataUints) return bad("material " + std::to_string(m) + ": texture index " + std::to_string(t) + " has no descriptor");
            const uint32_t* d = s.textureData + 4 * t;
            if (d[0] == 0 || d[1] == 0 || (uint64_t)d[2] + (uint64_t)d[0] * d[1] > s.textureDataUints)
                return bad("texture " + std::to_string(t) + ": descriptor {w, h, offset} reaches outside textureData");
        }
    }
    // triangles -> materials
    const uint8_t* attrs = (const uint8_t*)s.triAttrs;
    if (!tlasOn)
        for (uint64_t i = 0; i < attrCount; ++i) {
            uint32_t mi;
            memcpy(&mi, attrs + i * 128 + 120, 4);
            if (mi >= s.materialCount) return bad("triangle " + std::to_string(i) + ": materialIndex " + std::to_string(mi) + " >= materialCount");
        }
    // CWBVH: children, triangle rows, primitive indices -- decoded EXACTLY as the kernels decode them (cwbvh_node_hitmask,
    // pt_device.h): a child is inner iff bits 3 and 4 of its meta byte are both set, whatever the upper bits say; a leaf's
    // triangle bits are (meta >> 5) & 7 shifted to bit (meta & 31).  Every BLAS is walked from its root with a visited map, so a
    // child pointer that points back (a cycle: traversal would never end) or sideways (two parents) is refused, and only what
    // a ray can reach has to be in range.  With HAS_TLAS the arrays hold several BLASes back to back; every distinct
    // (bvhOffset, triOffset, triAttributeOffset) triple is walked once.
    const uint8_t* nodes = (const uint8_t*)s.bvhNodes;
    const float* tris = (const float*)s.bvhTris;
    std::vector<uint8_t> seen(nodeCount, 0);
    auto walk_blas = [&](uint64_t nodeOff, uint64_t triOff, uint64_t attrOff, const std::string& who) -> bool {
        std::vector<uint32_t> todo(1, 0u);
        std::vector<uint64_t> touched;
        bool ok = true;
        while (ok && !todo.empty()) {
            const uint32_t rel = todo.back();
            todo.pop_back();
            const uint64_t n = nodeOff + rel;
            if (n >= nodeCount) { ok = bad(who + ": child index past the node array"); break; }
            if (seen[n]) { ok = bad(who + ": node " + std::to_string(n) + " is reachable twice (the node graph is not a tree)"); break; }
            seen[n] = 1;
            touched.push_back(n);
            const uint8_t* p = nodes + n * 80;
            uint32_t childBase, triBase;
            memcpy(&childBase, p + 16, 4);
            memcpy(&triBase, p + 20, 4);
            const uint8_t imask = p[15];
            uint32_t inner = 0;
            for (int k = 0; k < 8 && ok; ++k) {
                const uint32_t m = p[24 + k];
                if (m == 0) continue;
                if ((m & 0x18u) == 0x18u) {
                    // the kernel shifts (meta >> 5) & 7 to bit 24 + slot: anything but 1 sets the hit bit of ANOTHER slot
                    if ((m >> 5) != 1u) { ok = bad(who + ": node " + std::to_string(n) + " has an inner child whose meta byte is not (1 << 5) | (24 + slot)"); break; }
                    if ((uint64_t)childBase + inner >= 0xFFFFFFFFull) { ok = bad(who + ": child index overflow"); break; }
                    todo.push_back(childBase + inner);
                    inner++;
                    continue;
                }
                const uint32_t first = m & 31u, bits = (m >> 5) & 7u;
                if (bits == 0u) continue;
                const uint32_t top = first + (31u - (uint32_t)__builtin_clz(bits));        // highest triangle bit of the 24-bit triangle mask
                if (top >= 24u) { ok = bad(who + ": node " + std::to_string(n) + " has a leaf whose triangle bits leave the 24-bit mask"); break; }
                for (uint32_t q = first; q <= top; ++q) {
                    if (!((bits >> (q - first)) & 1u)) continue;
                    const uint64_t row = triOff + triBase + (uint64_t)q * 3u;
                    if (row + 2 >= triRows) { ok = bad(who + ": triangle rows past the triangle array"); break; }
                    uint32_t prim;
                    memcpy(&prim, tris + (row + 2) * 4 + 3, 4);
                    if (attrOff + prim >= attrCount) { ok = bad(who + ": primitive index " + std::to_string(prim) + " has no attribute record"); break; }
                }
            }
            if (ok && inner != (uint32_t)__builtin_popcount(imask)) ok = bad(who + ": node " + std::to_string(n) + ": imask does not match its inner children");
        }
        if (tlasOn) for (uint64_t n : touched) seen[n] = 0;          // BLASes may share nodes between instances, never within one
        return ok;
    };
    if (!tlasOn) {
        if (!walk_blas(0, 0, 0, "BVH")) return false;
    } else {
        if (s.tlasIndexOffset < 16u) return bad("tlasIndexOffset < 16: HAS_TLAS needs at least one TLAS node");
        const uint64_t tlasNodes = s.tlasIndexOffset / 16u, indices = s.tlasDataFloats - s.tlasIndexOffset;
        const uint32_t* T = (const uint32_t*)s.tlasData;
        // the 2-wide TLAS is walked from node 0 as the kernel walks it (tlas.hlsl:246-331): children in range, no node twice
        {
            std::vector<uint8_t> seenT(tlasNodes, 0);
            std::vector<uint32_t> todo(1, 0u);
            while (!todo.empty()) {
                const uint32_t n = todo.back();
                todo.pop_back();
                if (n >= tlasNodes) return bad("TLAS child index " + std::to_string(n) + " past the TLAS nodes");
                if (seenT[n]) return bad("TLAS node " + std::to_string(n) + " is reachable twice (the TLAS is not a tree)");
                seenT[n] = 1;
                const uint32_t left = T[(size_t)n * 16 + 3], right = T[(size_t)n * 16 + 7], count = T[(size_t)n * 16 + 11], first = T[(size_t)n * 16 + 15];
                if (count == 0) { todo.push_back(left); todo.push_back(right); }
                else if ((uint64_t)first + count > indices) return bad("TLAS node " + std::to_string(n) + ": instance range past the index list");
            }
        }
        for (uint64_t i = 0; i < indices; ++i)
            if (T[s.tlasIndexOffset + i] >= s.instanceCount) return bad("TLAS index " + std::to_string(i) + " >= instanceCount");
        const uint8_t* inst = (const uint8_t*)s.gpuInstances;
        std::vector<std::array<int32_t, 3>> walked;                  // distinct BLASes already validated
        for (uint32_t i = 0; i < s.instanceCount; ++i) {
            int32_t off[4];
            memcpy(off, inst + (size_t)i * 144 + 128, 16);
            if (off[0] < 0 || (uint64_t)off[0] >= nodeCount || off[1] < 0 || (uint64_t)off[1] > triRows || off[2] < 0 || (uint64_t)off[2] > attrCount ||
                off[3] < 0 || (uint32_t)off[3] >= s.materialCount)
                return bad("instance " + std::to_string(i) + ": bvhOffset / triOffset / triAttributeOffset / materialIndex out of range");
            const std::array<int32_t, 3> key = {off[0], off[1], off[2]};
            bool done = false;
            for (const auto& k : walked) if (k == key) { done = true; break; }
            if (done) continue;
            if (!walk_blas((uint64_t)off[0], (uint64_t)off[1], (uint64_t)off[2], "instance " + std::to_string(i))) return false;
            walked.push_back(key);
        }
    }
    return true;
}

int drain_events(PTContext* c)
{
    for (auto& ep : c->pending) {
        HIP_TRY(hipEventSynchronize(ep.stop));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ep.start, ep.stop));
        c->timings.passes++;
        c->timings.kernelMsTotal += ms;
        c->timings.kernelMsLast = ms;
        c->timings.kernelLaunches += ep.launches;
        c->freeEvents.push_back(ep);
    }
    c->pending.clear();
    return PT_OK;
}

// "Versioning" (include/ptmi_plugin.h): copy min(structSize, sizeof) bytes of the host's struct into a zeroed one of ours
int import_params(const PTFrameParams* in, PTFrameParams& p)
{
    if (!in) return fail(PT_ERR_INVALID_ARG, "params == NULL");
    if (in->structSize < PT_FRAME_PARAMS_MIN_SIZE || in->structSize > 4096u)
        return fail(PT_ERR_INVALID_ARG, "PTFrameParams.structSize is not set (must be sizeof(PTFrameParams) of the host's header)");
    memset(&p, 0, sizeof(p));
    memcpy(&p, in, in->structSize < sizeof(p) ? in->structSize : sizeof(p));
    p.structSize = (uint32_t)sizeof(p);
    return PT_OK;
}

int validate_params(const PTFrameParams* p)
{
    if (!p) return fail(PT_ERR_INVALID_ARG, "params == NULL");
    if (p->OutputWidth == 0 || p->OutputHeight == 0) return fail(PT_ERR_INVALID_ARG, "OutputWidth/OutputHeight == 0");
    if ((uint64_t)p->OutputWidth * p->OutputHeight > 0x7FFFFFFFull / 4) return fail(PT_ERR_INVALID_ARG, "frame too large");
    return PT_OK;
}

// auto (-1): scenes whose whole BVH is a handful of nodes (Cornell box: 1 node) have no traversal to speak of; the
// wavefront's per-iteration path-state traffic then costs more than it buys (measured 8.7 vs 10.5 Grays/s)
int effective_schedule(const PTContext* c)
{
    if (c->schedule >= 0) return c->schedule;
    return (c->nodes.used <= 80u * 16u && !c->scene.hasTlas) ? 0 : 1;
}

// `batch`: the passes a launch sequence renders together (count >= 1; batch.seedRoot[0] / currentSample[0] = those of *hostParams)
int render_to(PTContext* c, const PTFrameParams* hostParams, float4* dOut, const float4* dAcc, const PTBatch* hostBatch = nullptr)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    RoctxRange range("PT pass (enqueue)");
    PTFrameParams params;
    int rc = import_params(hostParams, params);
    if (rc) return rc;
    const PTFrameParams* p = &params;
    if ((rc = validate_params(p))) return rc;
    PTBatch batch = {};
    if (hostBatch) batch = *hostBatch;
    else { batch.count = 1u; batch.seedRoot[0] = p->RngSeedRoot; batch.currentSample[0] = p->CurrentSample; }
    if (!c->hasScene) return fail(PT_ERR_NO_SCENE, "PTSetScene has not been called");
    if (!dOut) return fail(PT_ERR_INVALID_ARG, "output buffer == NULL");
    if (p->CurrentSample > 0 && !dAcc) return fail(PT_ERR_INVALID_ARG, "CurrentSample > 0 needs an accumulated frame");
    HIP_TRY(hipSetDevice(c->device));

    const PTTileMap tm = pt_make_tile_map(*p, c->rank, c->world);

    EventPair ep = {};
    if (c->profiling) {
        if (c->pending.size() >= 4096) { rc = drain_events(c); if (rc) return rc; }
        if (!c->freeEvents.empty()) { ep = c->freeEvents.back(); c->freeEvents.pop_back(); }
        else { HIP_TRY(hipEventCreate(&ep.start)); HIP_TRY(hipEventCreate(&ep.stop)); }
    }
    uint32_t launches = 0;
    const int schedule = effective_schedule(c);
    switch (schedule) {
    case 1:
    case 2:
    case 3:
    case 4: {
        const uint32_t spp = p->SamplesPerPass > 1 ? (uint32_t)p->SamplesPerPass : 1u;
        const uint32_t bounces = p->MaxRayBounces > 1u ? p->MaxRayBounces : 1u;
        if (spp > 4095u || bounces > 8191u) return fail(PT_ERR_UNSUPPORTED, "wavefront schedules pack SamplesPerPass <= 4095 and MaxRayBounces <= 8191");
        const uint64_t maxIt = (uint64_t)spp * (bounces + 2u) + 4u;
        // A pass may be cut into SUB-FRAMES (PTSetSubFrames): interleaved subsets of the context's 16x16 blocks, each with its own
        // launch sequence on its own state set and stream, all writing the same output frame.  To the kernels a sub-frame is
        // tile ownership (rank + world * j of world * S); to the machine it is what a second pass in flight is -- other kernels to
        // fill the ramp-up and drain of every launch -- without the host having to run ahead of the frame it shows.
        uint32_t S = c->subFrames < 1u ? 1u : c->subFrames;
        if (S > c->numSets) S = c->numSets;
        const bool zeroOnce = c->world > 1 && S > 1u;
        if (zeroOnce) HIP_TRY(hipMemsetAsync(dOut, 0, (size_t)p->OutputWidth * p->OutputHeight * sizeof(float4), c->stream));
        for (uint32_t j = 0; j < S; ++j) {
            const PTTileMap tmj = S == 1u ? tm : pt_make_tile_map(*p, c->rank + c->world * (int)j, c->world * (int)S);
            if (c->nextSet >= c->numSets) c->nextSet = 0u;
            PTContext::WfSet& set = c->sets[c->nextSet];
            c->nextSet = (c->nextSet + 1u) % c->numSets;
            // every set IN USE is carved on the FIRST pass of a frame size (a no-op afterwards): a caller that times passes after a
            // short warm-up must not find the allocation of sets it has not reached yet inside its timed region.  Sets beyond
            // PTSetPassesInFlight are never allocated (0.6 GB each at 1080p).
            const uint32_t slotsPerPass = pt_num_slots(tmj);
            if ((uint64_t)slotsPerPass * batch.count > 0x3FFFFFFFull) return fail(PT_ERR_UNSUPPORTED, "batch too large: passes x owned pixels exceeds 2^30 slots");
            if (j == 0u)
                for (uint32_t k = 0; k < c->numSets; ++k) {
                    if ((rc = ensure_wavefront(c, c->sets[k], slotsPerPass * batch.count, (uint32_t)(maxIt > 65536u ? 65536u : maxIt)))) return rc;
                    c->sets[k].wf.slotsPerPass = slotsPerPass;
                }
            // the launch chain runs on the set's own stream; only its resolve (which reads `accumulated` and writes `output`)
            // is ordered after what the caller has enqueued on the context stream so far, the previous pass included
            HIP_TRY(hipEventRecord(set.callEv, c->stream));
            if (c->profiling && j == 0u) HIP_TRY(hipEventRecord(ep.start, set.stream));
            uint32_t n = 0;
            // the default schedule's kernels (refill trace + shade) come from the translation unit built without the post-RA scheduler,
            // HAS_TLAS, the fused persistent kernel and the other schedules from the one built with it (csrc/Makefile)
            const bool tuA = schedule == 1 && c->scene.hasTlas == 0u;
            HIP_TRY((tuA ? pt_launch_wavefront : pt_launch_wavefront_b)(c->scene, *p, batch, dAcc, dOut, tmj, set.wf, c->dStats, c->statsLevel > 0, set.stream, set.callEv,
                                        c->world > 1 && !zeroOnce, &n, schedule == 4 ? 4 : (schedule == 1 ? 2 : (schedule == 2 ? 1 : 0)), c->wfIterations));
            launches += n;
            if (c->profiling && j + 1u == S) HIP_TRY(hipEventRecord(ep.stop, set.stream));
            HIP_TRY(hipEventRecord(set.done, set.stream));
            HIP_TRY(hipStreamWaitEvent(c->stream, set.done, 0));          // consumers of the context stream see the finished frame
        }
        break;
    }
    case 0:
    default:
        if (batch.count != 1u) return fail(PT_ERR_UNSUPPORTED, "internal: the megakernel renders one pass per launch");
        // pixels this context does not own must read as exact zeros (sum over ranks == single-GPU frame)
        if (c->world > 1)
            HIP_TRY(hipMemsetAsync(dOut, 0, (size_t)p->OutputWidth * p->OutputHeight * sizeof(float4), c->stream));
        if (c->profiling) HIP_TRY(hipEventRecord(ep.start, c->stream));
        HIP_TRY(pt_launch_megakernel(c->scene, *p, dAcc, dOut, tm, c->dStats, c->statsLevel > 0, c->stream));
        if (c->profiling) HIP_TRY(hipEventRecord(ep.stop, c->stream));
        launches = 1;
        break;
    }
    if (c->profiling) {
        ep.launches = launches;
        c->pending.push_back(ep);
    }
    return PT_OK;
}

} // namespace

extern "C" {

PT_API int PTCreate(int deviceIndex, PTContext** outCtx)
{
    if (!outCtx) return fail(PT_ERR_INVALID_ARG, "outCtx == NULL");
    *outCtx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(PT_ERR_NO_DEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                          "); this library has no CPU fallback");
    if (deviceIndex < 0 || deviceIndex >= count) return fail(PT_ERR_INVALID_ARG, "deviceIndex out of range");
    HIP_TRY(hipSetDevice(deviceIndex));
    PTContext* c = new PTContext();
    c->device = deviceIndex;
    c->numSets = default_passes_in_flight();
    if (const char* e = getenv("PT_WF_SHADE_SORTED")) c->shadeSorted = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("PT_WF_TAIL_CUT")) c->tailCut = (uint32_t)strtoul(e, nullptr, 10);
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return fail(PT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(se)); }
    se = hipMalloc((void**)&c->dStats, 16 * sizeof(unsigned long long));
    if (se != hipSuccess) { hipStreamDestroy(c->stream); delete c; return fail(PT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(se)); }
    hipMemsetAsync(c->dStats, 0, 16 * sizeof(unsigned long long), c->stream);
    *outCtx = c;
    return PT_OK;
}

PT_API int PTDestroy(PTContext* c)
{
    if (!c) return PT_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto* b : {&c->nodes, &c->tris, &c->attrs, &c->materials, &c->lights, &c->lightConst, &c->tex, &c->tlas, &c->tlasBfs, &c->instances, &c->instByLeaf, &c->envTex, &c->envCdf}) if (b->ptr) hipFree(b->ptr);
    for (int i = 0; i < 2; i++) if (c->frames[i]) hipFree(c->frames[i]);
    if (c->dStats) hipFree(c->dStats);
    if (c->present) hipFree(c->present);
    if (c->batchScratch) hipFree(c->batchScratch);
    for (auto& set : c->sets) {
        if (set.stream) { hipStreamSynchronize(set.stream); hipStreamDestroy(set.stream); }
        if (set.arena) hipFree(set.arena);
        if (set.callEv) hipEventDestroy(set.callEv);
        if (set.done) hipEventDestroy(set.done);
    }
    for (auto& ep : c->pending) { hipEventDestroy(ep.start); hipEventDestroy(ep.stop); }
    for (auto& ep : c->freeEvents) { hipEventDestroy(ep.start); hipEventDestroy(ep.stop); }
    hipStreamDestroy(c->stream);
    delete c;
    return PT_OK;
}

namespace { int set_scene(PTContext* c, const PTSceneDesc* hostScene, bool validate); }
PT_API int PTSetScene(PTContext* c, const PTSceneDesc* hostScene) { return set_scene(c, hostScene, true); }

} // extern "C"
namespace {
int set_scene(PTContext* c, const PTSceneDesc* hostScene, bool validate)
{
    if (!c || !hostScene) return fail(PT_ERR_INVALID_ARG, "ctx/scene == NULL");
    if (hostScene->structSize < PT_SCENE_DESC_MIN_SIZE || hostScene->structSize > 4096u)
        return fail(PT_ERR_INVALID_ARG, "PTSceneDesc.structSize is not set (must be sizeof(PTSceneDesc) of the host's header)");
    PTSceneDesc sceneCopy;
    memset(&sceneCopy, 0, sizeof(sceneCopy));
    memcpy(&sceneCopy, hostScene, hostScene->structSize < sizeof(sceneCopy) ? hostScene->structSize : sizeof(sceneCopy));
    const PTSceneDesc* s = &sceneCopy;
    RoctxRange range("PTSetScene (validate + upload)");
    const bool tlasOn = (s->features & PT_FEATURE_HAS_TLAS) != 0;
    if (tlasOn && (!s->tlasData || s->tlasDataFloats < 16 || !s->gpuInstances || s->instanceCount == 0 ||
                   s->tlasIndexOffset >= s->tlasDataFloats || s->tlasIndexOffset % 16 != 0))
        return fail(PT_ERR_INVALID_ARG, "HAS_TLAS needs tlasData (nodes + indices), tlasIndexOffset and gpuInstances");
    if (!s->bvhNodes || s->bvhNodesBytes < 80 || s->bvhNodesBytes % 80) return fail(PT_ERR_INVALID_ARG, "bvhNodes must be a non-empty multiple of 80 bytes");
    if ((uint64_t)s->bvhNodesBytes >= (1ull << 32)) return fail(PT_ERR_INVALID_ARG, "bvhNodes of 4 GiB and more are not supported (the kernels address nodes with 32-bit byte offsets)");
    if (!s->bvhTris || s->bvhTrisBytes % 48) return fail(PT_ERR_INVALID_ARG, "bvhTris must be a multiple of 48 bytes");
    if (!s->triAttrs || s->triAttrsBytes % 128) return fail(PT_ERR_INVALID_ARG, "triAttrs must be a multiple of 128 bytes");
    if (!s->materials || s->materialCount == 0) return fail(PT_ERR_INVALID_ARG, "materials missing");
    if ((s->features & PT_FEATURE_HAS_LIGHTS) && (!s->lights || s->lightCount == 0)) return fail(PT_ERR_INVALID_ARG, "HAS_LIGHTS without lights");
    if ((s->features & PT_FEATURE_HAS_TEXTURES) && (!s->textureData || s->textureDataUints == 0)) return fail(PT_ERR_INVALID_ARG, "HAS_TEXTURES without texture data");
    const bool envOn = (s->features & PT_FEATURE_HAS_ENVIRONMENT_TEXTURE) != 0;
    if (envOn && (!s->envTexture || s->envWidth == 0 || s->envHeight == 0 || (uint64_t)s->envWidth * s->envHeight > 0x7FFFFFFFull))
        return fail(PT_ERR_INVALID_ARG, "HAS_ENVIRONMENT_TEXTURE needs envTexture, envWidth, envHeight");
    if (validate) {
        std::string why;
        if (!validate_scene(*s, why)) return fail(PT_ERR_INVALID_ARG, "scene refused: " + why);
    }
    HIP_TRY(hipSetDevice(c->device));
    for (auto& set : c->sets) if (set.stream) HIP_TRY(hipStreamSynchronize(set.stream));   // no pass may still read the old scene
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc;
    if ((rc = upload(c, c->nodes, s->bvhNodes, s->bvhNodesBytes))) return rc;
    if ((rc = upload(c, c->tris, s->bvhTris, s->bvhTrisBytes))) return rc;
    if ((rc = upload(c, c->attrs, s->triAttrs, s->triAttrsBytes))) return rc;
    if ((rc = upload(c, c->materials, s->materials, (size_t)s->materialCount * 128))) return rc;
    const bool lights = (s->features & PT_FEATURE_HAS_LIGHTS) != 0;
    const bool textures = (s->features & PT_FEATURE_HAS_TEXTURES) != 0;
    if (lights && (rc = upload(c, c->lights, s->lights, (size_t)s->lightCount * 64))) return rc;
    if (textures && (rc = upload(c, c->tex, s->textureData, (size_t)s->textureDataUints * 4))) return rc;
    if (tlasOn && (rc = upload(c, c->tlas, s->tlasData, (size_t)s->tlasDataFloats * 4))) return rc;
    if (tlasOn && (rc = upload(c, c->instances, s->gpuInstances, (size_t)s->instanceCount * 144))) return rc;
    std::vector<float> byLeaf, bfs;
    if (tlasOn) {
        // what entering an instance reads (util/tlas.hlsl:129-147: TLASData[TLASIndexOffset + k] -> GPUInstance) as ONE record per
        // index slot k, in the order the TLAS leaves list them: the kernels save a dependent fetch per instance entry
        const uint64_t indices = s->tlasDataFloats - s->tlasIndexOffset;
        const uint32_t* T = (const uint32_t*)s->tlasData;
        const float* inst = (const float*)s->gpuInstances;
        byLeaf.assign((size_t)indices * 24, 0.0f);
        for (uint64_t k = 0; k < indices; ++k) {
            const uint32_t idx = T[s->tlasIndexOffset + k];                 // validated above: < instanceCount
            memcpy(&byLeaf[k * 24], inst + (size_t)idx * 36 + 16, 20 * sizeof(float));      // worldToLocal + the offsets row
            memcpy(&byLeaf[k * 24 + 20], &idx, 4);
        }
        if ((rc = upload(c, c->instByLeaf, byLeaf.data(), byLeaf.size() * sizeof(float)))) return rc;
        // the nodes reachable from node 0, renumbered breadth-first: the top of the tree -- what nearly every ray visits -- becomes
        // the FIRST nodes of the array, which pt_wf_trace_refill_tlas copies into LDS.  Children indices are rewritten, every other
        // word (boxes, instance count, first index slot) is copied, so a walk visits the same nodes in the same order.
        const uint32_t tlasNodes = s->tlasIndexOffset / 16u;
        std::vector<uint32_t> order(1, 0u), renum(tlasNodes, 0xFFFFFFFFu);
        renum[0] = 0u;
        for (size_t i = 0; i < order.size(); ++i) {
            const uint32_t* n = T + (size_t)order[i] * 16;
            if (n[11] != 0u) continue;                                       // a leaf: instance count > 0
            for (uint32_t child : {n[3], n[7]})                              // < tlasNodes when the scene was validated; otherwise left alone
                if (child < tlasNodes && renum[child] == 0xFFFFFFFFu) { renum[child] = (uint32_t)order.size(); order.push_back(child); }
        }
        bfs.resize(order.size() * 16);
        for (size_t i = 0; i < order.size(); ++i) {
            uint32_t* d = (uint32_t*)&bfs[i * 16];
            memcpy(d, T + (size_t)order[i] * 16, 64);
            if (d[11] == 0u) { if (d[3] < tlasNodes) d[3] = renum[d[3]]; if (d[7] < tlasNodes) d[7] = renum[d[7]]; }
        }
        if ((rc = upload(c, c->tlasBfs, bfs.data(), bfs.size() * sizeof(float)))) return rc;
    }
    std::vector<float> cdf;
    float cdfSum = 0.0f;
    if (envOn) {
        // OnEnvTexReadback (PathTracer.cs:297-306): running fp32 sum of Color.grayscale = 0.299 r + 0.587 g + 0.114 b
        const size_t n = (size_t)s->envWidth * s->envHeight;
        cdf.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const float* px = s->envTexture + 4 * i;
            cdfSum += 0.299f * px[0] + 0.587f * px[1] + 0.114f * px[2];
            cdf[i] = cdfSum;
        }
        if ((rc = upload(c, c->envTex, s->envTexture, n * 16))) return rc;
        if ((rc = upload(c, c->envCdf, cdf.data(), n * 4))) return rc;
    }
    if (lights) {                                  // room for the per-light constants, filled below once the scene view is complete
        DeviceBuffer& b = c->lightConst;
        const size_t bytes = (size_t)s->lightCount * 64;
        if (b.ptr && b.bytes < bytes) { hipFree(b.ptr); b.ptr = nullptr; b.bytes = 0; }
        if (!b.ptr) { HIP_TRY(hipMalloc(&b.ptr, bytes)); b.bytes = bytes; }
        b.used = bytes;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));      // inputs are borrowed for the duration of the call only
    c->scene.nodes = (const uint4*)c->nodes.ptr;
    c->scene.tris = (const float4*)c->tris.ptr;
    c->scene.attrs = (const float4*)c->attrs.ptr;
    c->scene.materials = (const float4*)c->materials.ptr;
    c->scene.lights = lights ? (const float4*)c->lights.ptr : nullptr;
    c->scene.tex = textures ? (const uint32_t*)c->tex.ptr : nullptr;
    c->scene.lightCount = lights ? (int32_t)s->lightCount : 0;
    c->scene.materialCount = s->materialCount;
    c->scene.hasLights = lights ? 1u : 0u;
    c->scene.hasTextures = textures ? 1u : 0u;
    c->scene.tlas = tlasOn ? (const float*)c->tlas.ptr : nullptr;
    c->scene.instances = tlasOn ? (const float4*)c->instances.ptr : nullptr;
    c->scene.instByLeaf = tlasOn ? (const float4*)c->instByLeaf.ptr : nullptr;
    c->scene.tlasBfs = tlasOn ? (const float*)c->tlasBfs.ptr : nullptr;
    c->scene.tlasNodeCount = tlasOn ? (uint32_t)(bfs.size() / 16) : 0u;
    c->scene.tlasIndexOffset = tlasOn ? s->tlasIndexOffset : 0u;
    c->scene.hasTlas = tlasOn ? 1u : 0u;
    c->scene.envTex = envOn ? (const float4*)c->envTex.ptr : nullptr;
    c->scene.envCdf = envOn ? (const float*)c->envCdf.ptr : nullptr;
    c->scene.envW = envOn ? (int32_t)s->envWidth : 0;
    c->scene.envH = envOn ? (int32_t)s->envHeight : 0;
    c->scene.envCdfSum = cdfSum;
    c->scene.hasEnvTex = envOn ? 1u : 0u;
    c->scene.lightConst = lights ? (const float4*)c->lightConst.ptr : nullptr;
    if (lights) {
        HIP_TRY(pt_launch_derive_lights(c->scene, (float4*)c->lightConst.ptr, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->hasScene = true;
    return PT_OK;
}
} // namespace
extern "C" {

PT_API int PTSetTileOwnership(PTContext* c, int rank, int worldSize)
{
    if (!c || worldSize < 1 || rank < 0 || rank >= worldSize) return fail(PT_ERR_INVALID_ARG, "bad rank/worldSize");
    c->rank = rank;
    c->world = worldSize;
    return PT_OK;
}

PT_API int PTRenderPass(PTContext* c, const PTFrameParams* hostParams)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    PTFrameParams params;
    int rc = import_params(hostParams, params);
    if (rc) return rc;
    const PTFrameParams* p = &params;
    if ((rc = validate_params(p))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = ensure_frames(c, p->OutputWidth, p->OutputHeight))) return rc;
    return render_to(c, p, c->frames[c->cur], c->frames[1 - c->cur]);
}

PT_API int PTRenderPassBatchTo(PTContext* c, const PTFrameParams* hostParams, int count, void* dOutput, const void* dAccumulated);

PT_API int PTRenderPassBatch(PTContext* c, const PTFrameParams* hostParams, int count)
{
    if (!c || !hostParams) return fail(PT_ERR_INVALID_ARG, "ctx/params == NULL");
    PTFrameParams params;
    int rc = import_params(hostParams, params);
    if (rc || (rc = validate_params(&params))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = ensure_frames(c, params.OutputWidth, params.OutputHeight))) return rc;
    return PTRenderPassBatchTo(c, hostParams, count, c->frames[c->cur], c->frames[1 - c->cur]);
}

PT_API int PTFlipFrames(PTContext* c) { if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL"); c->cur = 1 - c->cur; return PT_OK; }
PT_API int PTResetFrames(PTContext* c) { if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL"); c->cur = 0; return PT_OK; }

PT_API int PTRenderPassTo(PTContext* c, const PTFrameParams* p, void* dOutput, const void* dAccumulated)
{
    return render_to(c, p, (float4*)dOutput, (const float4*)dAccumulated);
}

PT_API int PTRenderPassBatchTo(PTContext* c, const PTFrameParams* hostParams, int count, void* dOutput, const void* dAccumulated)
{
    if (!c || !hostParams) return fail(PT_ERR_INVALID_ARG, "ctx/params == NULL");
    if (count < 1 || count > PT_MAX_BATCH) return fail(PT_ERR_INVALID_ARG, "count outside 1.." + std::to_string(PT_MAX_BATCH));
    // the host's array has the stride of ITS header (structSize of the first element)
    const uint32_t stride = hostParams->structSize;
    if (stride < PT_FRAME_PARAMS_MIN_SIZE || stride > 4096u) return fail(PT_ERR_INVALID_ARG, "PTFrameParams.structSize is not set");
    PTFrameParams first, other;
    int rc = import_params(hostParams, first);
    if (rc) return rc;
    PTBatch batch = {};
    batch.count = (uint32_t)count;
    for (int j = 0; j < count; ++j) {
        const PTFrameParams* hp = (const PTFrameParams*)((const char*)hostParams + (size_t)j * stride);
        if (hp->structSize != stride) return fail(PT_ERR_INVALID_ARG, "every PTFrameParams of a batch must carry the same structSize");
        if ((rc = import_params(hp, other))) return rc;
        batch.seedRoot[j] = other.RngSeedRoot;
        batch.currentSample[j] = other.CurrentSample;
        other.RngSeedRoot = first.RngSeedRoot;
        other.CurrentSample = first.CurrentSample;
        if (memcmp(&other, &first, sizeof(first)) != 0)
            return fail(PT_ERR_INVALID_ARG, "the passes of a batch may differ in RngSeedRoot and CurrentSample only (pass " + std::to_string(j) + " differs elsewhere)");
    }
    if (count == 1) return render_to(c, hostParams, (float4*)dOutput, (const float4*)dAccumulated);
    if (effective_schedule(c) == 0) {
        // the megakernel writes pixels itself: run the passes one by one, ping-ponging between dOutput and a scratch frame so that the
        // last pass lands in dOutput
        if ((rc = validate_params(&first))) return rc;
        HIP_TRY(hipSetDevice(c->device));
        const size_t bytes = (size_t)first.OutputWidth * first.OutputHeight * sizeof(float4);
        if (c->batchScratchBytes < bytes) {
            if (c->batchScratch) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->batchScratch); c->batchScratch = nullptr; c->batchScratchBytes = 0; }
            HIP_TRY(hipMalloc(&c->batchScratch, bytes));
            c->batchScratchBytes = bytes;
        }
        const float4* acc = (const float4*)dAccumulated;
        for (int j = 0; j < count; ++j) {
            PTFrameParams pj = first;
            pj.RngSeedRoot = batch.seedRoot[j];
            pj.CurrentSample = batch.currentSample[j];
            float4* out = ((count - 1 - j) & 1) ? (float4*)c->batchScratch : (float4*)dOutput;
            if ((rc = render_to(c, &pj, out, acc))) return rc;
            acc = out;
        }
        return PT_OK;
    }
    return render_to(c, hostParams, (float4*)dOutput, (const float4*)dAccumulated, &batch);
}

PT_API int PTSynchronize(PTContext* c)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

PT_API int PTReadback(PTContext* c, float* dst, uint64_t dstFloats)
{
    if (!c || !dst) return fail(PT_ERR_INVALID_ARG, "ctx/dst == NULL");
    if (!c->frames[0]) return fail(PT_ERR_INVALID_ARG, "no frame rendered yet");
    uint64_t need = (uint64_t)c->frameW * c->frameH * 4;
    if (dstFloats < need) return fail(PT_ERR_INVALID_ARG, "destination too small");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, c->frames[c->cur], need * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

namespace {
struct ScopedDev {                      // device scratch that is freed on every exit path
    void* p = nullptr;
    ~ScopedDev() { if (p) hipFree(p); }
};
}

PT_API int PTProcessMeshes(PTContext* c, const PTMeshDesc* meshes, uint32_t meshCount, uint32_t totalTriangles,
                           float* outVertexPositions, void* outTriangleAttributes)
{
    if (!c || !meshes || meshCount == 0 || totalTriangles == 0 || !outVertexPositions || !outTriangleAttributes)
        return fail(PT_ERR_INVALID_ARG, "ctx/meshes/outputs == NULL or nothing to process");
    for (uint32_t i = 0; i < meshCount; ++i) {
        const PTMeshDesc& m = meshes[i];
        if (!m.vertexBuffer || m.VertexStride == 0) return fail(PT_ERR_INVALID_ARG, "mesh without vertex buffer / stride");
        if ((m.VertexStride | m.PositionOffset | m.NormalOffset | m.TangentOffset | m.UVOffset) & 3u)
            return fail(PT_ERR_INVALID_ARG, "vertex stride and attribute offsets must be multiples of 4 (ByteAddressBuffer loads)");
        if ((uint64_t)m.OutputTriangleStart + m.TriangleCount > totalTriangles) return fail(PT_ERR_INVALID_ARG, "mesh writes past totalTriangles");
        // every index the kernel will read must stay inside the buffers the caller handed over
        uint64_t maxIndex;
        if (m.indexBuffer) {
            const bool wide = (m.flags & PT_MESH_HAS_32_BIT_INDICES) != 0;
            const uint64_t need = (uint64_t)m.TriangleCount * (wide ? 12u : 6u);
            if (m.indexBufferBytes < need) return fail(PT_ERR_INVALID_ARG, "index buffer too small for TriangleCount");
            maxIndex = 0;
            for (uint64_t k = 0; k < (uint64_t)m.TriangleCount * 3u; ++k) {
                const uint64_t v = wide ? ((const uint32_t*)m.indexBuffer)[k] : ((const uint16_t*)m.indexBuffer)[k];
                if (v > maxIndex) maxIndex = v;
            }
        } else maxIndex = m.TriangleCount ? (uint64_t)m.TriangleCount * 3u - 1u : 0u;
        uint32_t far = m.PositionOffset + 12u;
        if ((m.flags & PT_MESH_HAS_NORMALS) && m.NormalOffset + 12u > far) far = m.NormalOffset + 12u;
        if ((m.flags & PT_MESH_HAS_TANGENTS) && m.TangentOffset + 12u > far) far = m.TangentOffset + 12u;
        if ((m.flags & PT_MESH_HAS_UVS) && m.UVOffset + 8u > far) far = m.UVOffset + 8u;
        if (m.TriangleCount && maxIndex * m.VertexStride + far > m.vertexBufferBytes) return fail(PT_ERR_INVALID_ARG, "vertex buffer too small for the indices used");
    }
    HIP_TRY(hipSetDevice(c->device));
    ScopedDev dPos, dAttr;
    const size_t posBytes = (size_t)totalTriangles * 3 * sizeof(float4), attrBytes = (size_t)totalTriangles * 128;
    HIP_TRY(hipMalloc(&dPos.p, posBytes));
    HIP_TRY(hipMalloc(&dAttr.p, attrBytes));
    HIP_TRY(hipMemsetAsync(dPos.p, 0, posBytes, c->stream));
    HIP_TRY(hipMemsetAsync(dAttr.p, 0, attrBytes, c->stream));
    for (uint32_t i = 0; i < meshCount; ++i) {
        const PTMeshDesc& m = meshes[i];
        if (m.TriangleCount == 0) continue;
        ScopedDev dVb, dIb;
        // the 16-bit path reads two whole words around the last index: pad the staged copy
        const size_t ibBytes = m.indexBuffer ? (((size_t)m.indexBufferBytes + 3u) & ~(size_t)3u) + 8u : 0u;
        HIP_TRY(hipMalloc(&dVb.p, m.vertexBufferBytes));
        HIP_TRY(hipMemcpyAsync(dVb.p, m.vertexBuffer, m.vertexBufferBytes, hipMemcpyHostToDevice, c->stream));
        if (m.indexBuffer) {
            HIP_TRY(hipMalloc(&dIb.p, ibBytes));
            HIP_TRY(hipMemsetAsync(dIb.p, 0, ibBytes, c->stream));
            HIP_TRY(hipMemcpyAsync(dIb.p, m.indexBuffer, m.indexBufferBytes, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(pt_launch_process_mesh(m, dVb.p, dIb.p, (float4*)dPos.p, (float4*)dAttr.p, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));      // the staged buffers go out of scope
    }
    HIP_TRY(hipMemcpyAsync(outVertexPositions, dPos.p, posBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(outTriangleAttributes, dAttr.p, attrBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

PT_API int PTCopyTextureData(PTContext* c, const PTTextureDesc* textures, uint32_t count, uint32_t* outTextureData, uint64_t outUints)
{
    if (!c || !textures || count == 0 || !outTextureData) return fail(PT_ERR_INVALID_ARG, "ctx/textures/output == NULL");
    uint64_t total = (uint64_t)count * 4u;
    for (uint32_t i = 0; i < count; ++i) {
        if (!textures[i].texels || textures[i].width == 0 || textures[i].height == 0) return fail(PT_ERR_INVALID_ARG, "empty texture");
        total += (uint64_t)textures[i].width * textures[i].height;
    }
    if (total > 0xFFFFFFFFull) return fail(PT_ERR_INVALID_ARG, "texture data exceeds the 32-bit offsets of the descriptor");
    if (outUints < total) return fail(PT_ERR_INVALID_ARG, "outTextureData too small");
    HIP_TRY(hipSetDevice(c->device));
    ScopedDev dData;
    HIP_TRY(hipMalloc(&dData.p, total * 4u));
    uint32_t descriptorOffset = 0u, dataOffset = count * 4u;            // texel data starts after the descriptors (BVHScene.cs:388-389)
    for (uint32_t i = 0; i < count; ++i) {
        const PTTextureDesc& t = textures[i];
        const size_t bytes = (size_t)t.width * t.height * sizeof(float4);
        ScopedDev dTex;
        HIP_TRY(hipMalloc(&dTex.p, bytes));
        HIP_TRY(hipMemcpyAsync(dTex.p, t.texels, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(pt_launch_copy_texture((const float4*)dTex.p, t.width, t.height, dataOffset, descriptorOffset, t.hasAlpha, (uint32_t*)dData.p, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        dataOffset += t.width * t.height;
        descriptorOffset += 4u;
    }
    HIP_TRY(hipMemcpyAsync(outTextureData, dData.p, total * 4u, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

PT_API int PTPresent(PTContext* c, const PTPresentParams* q, const void* dSrc, void* dDst)
{
    if (!c || !q || !dDst) return fail(PT_ERR_INVALID_ARG, "ctx/params/dst == NULL");
    if (q->OutputWidth == 0 || q->OutputHeight == 0 || (uint64_t)q->OutputWidth * q->OutputHeight > 0x7FFFFFFFull / 4)
        return fail(PT_ERR_INVALID_ARG, "bad presentation size");
    if (!dSrc) {
        if (!c->frames[0]) return fail(PT_ERR_INVALID_ARG, "no frame rendered yet");
        if (q->OutputWidth != c->frameW || q->OutputHeight != c->frameH) return fail(PT_ERR_INVALID_ARG, "presentation size differs from the rendered frame");
        dSrc = c->frames[c->cur];
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(pt_launch_present(*q, (const float4*)dSrc, (float4*)dDst, c->stream));
    return PT_OK;
}

PT_API int PTPresentToHost(PTContext* c, const PTPresentParams* q, float* dst, uint64_t dstFloats)
{
    if (!c || !q || !dst) return fail(PT_ERR_INVALID_ARG, "ctx/params/dst == NULL");
    const uint64_t need = (uint64_t)q->OutputWidth * q->OutputHeight * 4;
    if (dstFloats < need) return fail(PT_ERR_INVALID_ARG, "destination too small");
    HIP_TRY(hipSetDevice(c->device));
    if (c->presentBytes < need * sizeof(float)) {
        if (c->present) { hipFree(c->present); c->present = nullptr; c->presentBytes = 0; }
        HIP_TRY(hipMalloc(&c->present, need * sizeof(float)));
        c->presentBytes = need * sizeof(float);
    }
    int rc = PTPresent(c, q, nullptr, c->present);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, c->present, need * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

PT_API void* PTGetFramePointer(PTContext* c, int which)
{
    if (!c) return nullptr;
    if (which < 0) return c->frames[c->cur];
    return which < 2 ? c->frames[which] : nullptr;
}

PT_API void* PTGetStream(PTContext* c) { return c ? (void*)c->stream : nullptr; }

PT_API int PTSetStatsLevel(PTContext* c, int level)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    c->statsLevel = level > 0 ? 1 : 0;
    return PT_OK;
}

PT_API int PTGetStats(PTContext* c, PTStats* out)
{
    if (!c || !out) return fail(PT_ERR_INVALID_ARG, "ctx/out == NULL");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[16];
    HIP_TRY(hipMemcpyAsync(h, c->dStats, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    static_assert(sizeof(PTStats) == 16 * 8, "PTStats is 16 counters");
    memcpy(out, h, sizeof(PTStats));
    return PT_OK;
}

PT_API int PTResetStats(PTContext* c)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->dStats, 0, 16 * sizeof(unsigned long long), c->stream));
    return PT_OK;
}

PT_API int PTSetProfiling(PTContext* c, int enabled)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    c->profiling = enabled != 0;
    return PT_OK;
}

PT_API int PTGetTimings(PTContext* c, PTTimings* out)
{
    if (!c || !out) return fail(PT_ERR_INVALID_ARG, "ctx/out == NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_events(c);
    if (rc) return rc;
    *out = c->timings;
    return PT_OK;
}

PT_API int PTResetTimings(PTContext* c)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    int rc = drain_events(c);
    if (rc) return rc;
    c->timings = PTTimings{};
    return PT_OK;
}

PT_API int PTGetSchedule(PTContext* c)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    return effective_schedule(c);
}

PT_API int PTSetSchedule(PTContext* c, int schedule)
{
    if (!c) return fail(PT_ERR_INVALID_ARG, "ctx == NULL");
    if (schedule < -1 || schedule > 4)
        return fail(PT_ERR_UNSUPPORTED, "unknown schedule (0 = megakernel, 1 = wavefront + refill trace, 2 = wavefront, one ray per lane, 3 = wavefront + persistent dynamic-chunk trace, 4 = fused persistent wavefront)");
    c->schedule = schedule;
    return PT_OK;
}

PT_API int PTSetPassesInFlight(PTContext* c, int passes)
{
    if (!c || passes < 0 || passes > PT_WF_SETS) return fail(PT_ERR_INVALID_ARG, "ctx == NULL or passes outside 0.." + std::to_string(PT_WF_SETS));
    HIP_TRY(hipSetDevice(c->device));
    for (auto& set : c->sets) if (set.stream) HIP_TRY(hipStreamSynchronize(set.stream));
    c->numSets = passes == 0 ? default_passes_in_flight() : (uint32_t)passes;
    c->nextSet = 0u;
    // sets that are no longer used give their memory back
    for (uint32_t k = c->numSets; k < (uint32_t)PT_WF_SETS; ++k) {
        PTContext::WfSet& set = c->sets[k];
        if (set.arena) { hipFree(set.arena); set.arena = nullptr; set.arenaBytes = 0; set.wf = PTWfBuffers{}; }
    }
    return PT_OK;
}

PT_API int PTSetSubFrames(PTContext* c, int subFrames)
{
    if (!c || subFrames < 1 || subFrames > PT_WF_SETS) return fail(PT_ERR_INVALID_ARG, "ctx == NULL or subFrames outside 1.." + std::to_string(PT_WF_SETS));
    c->subFrames = (uint32_t)subFrames;
    return PT_OK;
}

PT_API int PTGetPassesInFlight(PTContext* c) { return c ? (int)c->numSets : fail(PT_ERR_INVALID_ARG, "ctx == NULL"); }

PT_API int PTSetWavefrontIterations(PTContext* c, int iterations)
{
    if (!c || iterations < 0) return fail(PT_ERR_INVALID_ARG, "ctx == NULL or iterations < 0");
    c->wfIterations = (uint32_t)iterations;
    return PT_OK;
}

// ------------------------------------------------------------------------------------------
// tile pack / unpack and the one-process multi-device group (no reference counterpart)
// ------------------------------------------------------------------------------------------
PT_API int PTGetOwnedTileSlots(PTContext* c, const PTFrameParams* hostParams, uint64_t* out)
{
    if (!c || !out) return fail(PT_ERR_INVALID_ARG, "ctx/out == NULL");
    PTFrameParams p;
    int rc = import_params(hostParams, p);
    if (rc || (rc = validate_params(&p))) return rc;
    *out = pt_num_slots(pt_make_tile_map(p, c->rank, c->world));
    return PT_OK;
}

PT_API int PTPackOwnedTiles(PTContext* c, const PTFrameParams* hostParams, const void* dFrame, void* dPacked)
{
    if (!c || !dFrame || !dPacked) return fail(PT_ERR_INVALID_ARG, "ctx/frame/packed == NULL");
    PTFrameParams p;
    int rc = import_params(hostParams, p);
    if (rc || (rc = validate_params(&p))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(pt_launch_pack_tiles(pt_make_tile_map(p, c->rank, c->world), p.OutputWidth, (const float4*)dFrame, (float4*)dPacked, c->stream));
    return PT_OK;
}

PT_API int PTUnpackTiles(PTContext* c, const PTFrameParams* hostParams, int rank, int world, const void* dPacked, void* dFrame)
{
    if (!c || !dFrame || !dPacked) return fail(PT_ERR_INVALID_ARG, "ctx/frame/packed == NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(PT_ERR_INVALID_ARG, "bad rank/worldSize");
    PTFrameParams p;
    int rc = import_params(hostParams, p);
    if (rc || (rc = validate_params(&p))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(pt_launch_unpack_tiles(pt_make_tile_map(p, rank, world), p.OutputWidth, (const float4*)dPacked, (float4*)dFrame, c->stream));
    return PT_OK;
}

} // extern "C"

struct PTGroup {
    std::vector<PTContext*> ctx;
    std::vector<float4*> packed;        // on device i: its owned tiles, dense
    std::vector<float4*> staged;        // on the root device: the same, after the peer copy (staged[0] == packed[0])
    std::vector<hipEvent_t> arrived;    // recorded on ctx[i]->stream after the peer copy
    std::vector<hipEvent_t> unpacked;   // recorded on the root's stream after staged[i] was scattered into the assembled frame
    std::vector<char> unpackedValid;
    std::vector<size_t> packedBytes;
    float4* assembled = nullptr;
    uint32_t frameW = 0, frameH = 0;
};

namespace {

// runs fn(i) for every device of the group, device 0 on the calling thread and the others on one thread each (a pass is
// ~160 launches per device: enqueued serially, eight devices would be host-bound); the first error is re-raised here
template <class F>
int for_each_device(PTGroup* g, F fn)
{
    const size_t n = g->ctx.size();
    std::vector<int> rcs(n, PT_OK);
    std::vector<std::string> msgs(n);
    std::vector<std::thread> workers;
    for (size_t i = 1; i < n; ++i)
        workers.emplace_back([&, i] { rcs[i] = fn((int)i); if (rcs[i]) msgs[i] = g_lastError; });
    rcs[0] = fn(0);
    if (rcs[0]) msgs[0] = g_lastError;
    for (auto& w : workers) w.join();
    for (size_t i = 0; i < n; ++i)
        if (rcs[i]) return fail(rcs[i], "device " + std::to_string(g->ctx[i]->device) + " (rank " + std::to_string(i) + "): " + msgs[i]);
    return PT_OK;
}

int group_ensure_buffers(PTGroup* g, const PTFrameParams& p)
{
    PTContext* root = g->ctx[0];
    const size_t frameBytes = (size_t)p.OutputWidth * p.OutputHeight * sizeof(float4);
    if (!g->assembled || g->frameW != p.OutputWidth || g->frameH != p.OutputHeight) {
        HIP_TRY(hipSetDevice(root->device));
        if (g->assembled) { HIP_TRY(hipStreamSynchronize(root->stream)); hipFree(g->assembled); g->assembled = nullptr; }
        HIP_TRY(hipMalloc((void**)&g->assembled, frameBytes));
        HIP_TRY(hipMemsetAsync(g->assembled, 0, frameBytes, root->stream));
        g->frameW = p.OutputWidth; g->frameH = p.OutputHeight;
    }
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const size_t need = (size_t)pt_num_slots(pt_make_tile_map(p, (int)i, (int)g->ctx.size())) * sizeof(float4);
        if (g->packedBytes[i] >= need) continue;
        PTContext* c = g->ctx[i];
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (g->packed[i]) hipFree(g->packed[i]);
        HIP_TRY(hipMalloc((void**)&g->packed[i], need));
        if (i > 0) {
            HIP_TRY(hipSetDevice(root->device));
            HIP_TRY(hipStreamSynchronize(root->stream));
            if (g->staged[i]) hipFree(g->staged[i]);
            HIP_TRY(hipMalloc((void**)&g->staged[i], need));
        } else g->staged[0] = g->packed[0];
        g->packedBytes[i] = need;
    }
    return PT_OK;
}

} // namespace

extern "C" {

PT_API int PTCreateMulti(const int* devices, int n, PTGroup** out)
{
    if (!out) return fail(PT_ERR_INVALID_ARG, "outGroup == NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(PT_ERR_INVALID_ARG, "deviceIndices == NULL or deviceCount outside 1..64");
    PTGroup* g = new PTGroup();
    for (int i = 0; i < n; ++i) {
        PTContext* c = nullptr;
        int rc = PTCreate(devices[i], &c);
        if (rc == PT_OK) rc = PTSetTileOwnership(c, i, n);
        if (rc != PT_OK) {
            const std::string msg = g_lastError;
            if (c) PTDestroy(c);
            for (PTContext* d : g->ctx) PTDestroy(d);
            delete g;
            return fail(rc, msg);
        }
        g->ctx.push_back(c);
    }
    g->packed.assign(n, nullptr);
    g->staged.assign(n, nullptr);
    g->packedBytes.assign(n, 0);
    g->arrived.assign(n, nullptr);
    g->unpacked.assign(n, nullptr);
    g->unpackedValid.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        hipSetDevice(g->ctx[0]->device);
        if (hipEventCreateWithFlags(&g->unpacked[i], hipEventDisableTiming) != hipSuccess) { PTGroupDestroy(g); return fail(PT_ERR_HIP, "hipEventCreate failed"); }
        hipSetDevice(g->ctx[i]->device);
        if (hipEventCreateWithFlags(&g->arrived[i], hipEventDisableTiming) != hipSuccess) { PTGroupDestroy(g); return fail(PT_ERR_HIP, "hipEventCreate failed"); }
        // direct xGMI copies into the root's staging buffers; "already enabled" / "not supported" leave the staged path
        if (g->ctx[i]->device != g->ctx[0]->device) { (void)hipDeviceEnablePeerAccess(g->ctx[0]->device, 0); (void)hipGetLastError(); }
    }
    *out = g;
    return PT_OK;
}

PT_API int PTGroupDestroy(PTGroup* g)
{
    if (!g) return PT_OK;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        PTContext* c = g->ctx[i];
        hipSetDevice(c->device);
        hipStreamSynchronize(c->stream);
        if (g->packed[i]) hipFree(g->packed[i]);
        if (i > 0 && g->staged[i]) { hipSetDevice(g->ctx[0]->device); hipFree(g->staged[i]); }
        if (g->arrived[i]) hipEventDestroy(g->arrived[i]);
        if (g->unpacked[i]) hipEventDestroy(g->unpacked[i]);
    }
    if (g->assembled) { hipSetDevice(g->ctx[0]->device); hipStreamSynchronize(g->ctx[0]->stream); hipFree(g->assembled); }
    for (PTContext* c : g->ctx) PTDestroy(c);
    delete g;
    return PT_OK;
}

PT_API int PTGroupSize(PTGroup* g) { return g ? (int)g->ctx.size() : fail(PT_ERR_INVALID_ARG, "group == NULL"); }
PT_API PTContext* PTGroupGetContext(PTGroup* g, int i) { return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[i] : nullptr; }

PT_API int PTGroupSetScene(PTGroup* g, const PTSceneDesc* scene)
{
    if (!g) return fail(PT_ERR_INVALID_ARG, "group == NULL");
    // the index validation walks the whole scene on the host: once for the group, not once per device
    int rc = set_scene(g->ctx[0], scene, true);
    if (rc) return rc;
    return for_each_device(g, [&](int i) { return i == 0 ? PT_OK : set_scene(g->ctx[i], scene, false); });
}

PT_API int PTGroupRenderPassBatch(PTGroup* g, const PTFrameParams* hostParams, int count);
PT_API int PTGroupRenderPass(PTGroup* g, const PTFrameParams* hostParams) { return PTGroupRenderPassBatch(g, hostParams, 1); }

PT_API int PTGroupRenderPassBatch(PTGroup* g, const PTFrameParams* hostParams, int count)
{
    if (!g) return fail(PT_ERR_INVALID_ARG, "group == NULL");
    PTFrameParams p;
    int rc = import_params(hostParams, p);
    if (rc || (rc = validate_params(&p))) return rc;
    if ((rc = group_ensure_buffers(g, p))) return rc;
    PTContext* root = g->ctx[0];
    const int world = (int)g->ctx.size();
    rc = for_each_device(g, [&](int i) -> int {
        PTContext* c = g->ctx[i];
        int r = count == 1 ? PTRenderPass(c, hostParams) : PTRenderPassBatch(c, hostParams, count);   // owned tiles into the device's own ping-pong frames
        if (r) return r;
        // staged[i] (and packed[0], which the root unpacks in place) may only be overwritten once the root has scattered the
        // previous pass's tiles out of it: a straggling root must not see tiles of pass k+1 in the assembled frame of pass k
        if (g->unpackedValid[i]) HIP_TRY(hipStreamWaitEvent(c->stream, g->unpacked[i], 0));
        HIP_TRY(pt_launch_pack_tiles(pt_make_tile_map(p, i, world), p.OutputWidth, c->frames[c->cur], g->packed[i], c->stream));
        if (i > 0) HIP_TRY(hipMemcpyPeerAsync(g->staged[i], root->device, g->packed[i], c->device, g->packedBytes[i], c->stream));
        HIP_TRY(hipEventRecord(g->arrived[i], c->stream));
        return PT_OK;
    });
    if (rc) return rc;
    HIP_TRY(hipSetDevice(root->device));
    for (int i = 0; i < world; ++i) {
        if (i > 0) HIP_TRY(hipStreamWaitEvent(root->stream, g->arrived[i], 0));
        HIP_TRY(pt_launch_unpack_tiles(pt_make_tile_map(p, i, world), p.OutputWidth, g->staged[i], g->assembled, root->stream));
        HIP_TRY(hipEventRecord(g->unpacked[i], root->stream));
        g->unpackedValid[i] = 1;
    }
    return PT_OK;
}

PT_API int PTGroupFlipFrames(PTGroup* g)
{
    if (!g) return fail(PT_ERR_INVALID_ARG, "group == NULL");
    for (PTContext* c : g->ctx) PTFlipFrames(c);
    return PT_OK;
}

PT_API int PTGroupResetFrames(PTGroup* g)
{
    if (!g) return fail(PT_ERR_INVALID_ARG, "group == NULL");
    for (PTContext* c : g->ctx) PTResetFrames(c);
    return PT_OK;
}

PT_API int PTGroupSynchronize(PTGroup* g)
{
    if (!g) return fail(PT_ERR_INVALID_ARG, "group == NULL");
    for (size_t i = g->ctx.size(); i-- > 0;) { int rc = PTSynchronize(g->ctx[i]); if (rc) return rc; }   // the root last: it waits for the others
    return PT_OK;
}

PT_API int PTGroupReadback(PTGroup* g, float* dst, uint64_t dstFloats)
{
    if (!g || !dst) return fail(PT_ERR_INVALID_ARG, "group/dst == NULL");
    if (!g->assembled) return fail(PT_ERR_INVALID_ARG, "no frame rendered yet");
    const uint64_t need = (uint64_t)g->frameW * g->frameH * 4;
    if (dstFloats < need) return fail(PT_ERR_INVALID_ARG, "destination too small");
    PTContext* root = g->ctx[0];
    HIP_TRY(hipSetDevice(root->device));
    HIP_TRY(hipMemcpyAsync(dst, g->assembled, need * sizeof(float), hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipStreamSynchronize(root->stream));
    return PT_OK;
}

PT_API void* PTGroupGetAssembledFrame(PTGroup* g) { return g ? (void*)g->assembled : nullptr; }

PT_API int PTGroupGetStats(PTGroup* g, PTStats* out)
{
    if (!g || !out) return fail(PT_ERR_INVALID_ARG, "group/out == NULL");
    uint64_t acc[16] = {};
    for (PTContext* c : g->ctx) {
        PTStats st;
        int rc = PTGetStats(c, &st);
        if (rc) return rc;
        const uint64_t* v = (const uint64_t*)&st;
        for (int k = 0; k < 16; ++k) { if (k == 12) acc[k] = v[k] > acc[k] ? v[k] : acc[k]; else acc[k] += v[k]; }
    }
    memcpy(out, acc, sizeof(PTStats));
    return PT_OK;
}

PT_API int PTGroupResetStats(PTGroup* g)
{
    if (!g) return fail(PT_ERR_INVALID_ARG, "group == NULL");
    for (PTContext* c : g->ctx) { int rc = PTResetStats(c); if (rc) return rc; }
    return PT_OK;
}

PT_API const char* PTGetLastError(void) { return g_lastError.c_str(); }
PT_API int PTGetVersion(void) { return (0 << 16) | 1; }

} // extern "C"
