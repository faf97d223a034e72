// plugin_exports.cpp — Part 1 of include/ptmi_plugin.h: the acceleration-structure exports the Unity C#
// host binds (Assets/Scripts/util/TinyBVH.cs:15-50), replacing Assets/Plugins/Web/plugin.cpp.
//
// Same handle semantics as the reference: a handle is the index of the first free slot of a global
// table (plugin.cpp:8-21, 90-103), Destroy* frees and nulls the slot, invalid handles read as 0 / NULL.
// Differences, all on the safe side: degenerate input returns -1 instead of exit(1)
// (tiny_bvh.h:1615-1618), and "bool" results are full-width ints.
#include <chrono>
#include <cstddef>
#include <deque>
#include <string>
#include "bvh_builder.h"
#include "ptmi_plugin.h"

namespace {
std::deque<ptbvh::Cwbvh*> g_bvhs;
std::deque<ptbvh::Tlas*> g_tlases;

template <class T> int add_slot(std::deque<T*>& table, T* obj)
{
    for (size_t i = 0; i < table.size(); ++i)
        if (table[i] == nullptr) { table[i] = obj; return (int)i; }
    table.push_back(obj);
    return (int)table.size() - 1;
}
template <class T> T* get_slot(std::deque<T*>& table, int index)
{
    return (index >= 0 && index < (int)table.size()) ? table[index] : nullptr;
}
template <class T> void free_slot(std::deque<T*>& table, int index)
{
    if (index >= 0 && index < (int)table.size() && table[index]) { delete table[index]; table[index] = nullptr; }
}
} // namespace

extern "C" {

PT_API int BuildBVH(const PTFloat4* vertices, int triangleCount)
{
    if (!vertices || triangleCount <= 0) return -1;
    ptbvh::Cwbvh* bvh = new ptbvh::Cwbvh();
    const auto t0 = std::chrono::steady_clock::now();
    if (!bvh->build(vertices, (uint32_t)triangleCount)) { delete bvh; return -1; }
    bvh->buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return add_slot(g_bvhs, bvh);
}

// The builder that runs on the MI355X (bvh_builder_gpu.hip).  Same handle table as BuildBVH: every Get* / DestroyBVH export
// works on the handle.  -1 on degenerate input or when there is no such HIP device (no CPU fallback: BuildBVH is the CPU path).
namespace { std::string g_buildError; }
PT_API int PTBuildBVHDevice(int deviceIndex, const PTFloat4* vertices, int triangleCount)
{
    if (!vertices || triangleCount <= 0) { g_buildError = "vertices == NULL or triangleCount <= 0"; return -1; }
    ptbvh::Cwbvh* bvh = new ptbvh::Cwbvh();
    if (!ptbvh::build_cwbvh_device(deviceIndex, vertices, (uint32_t)triangleCount, *bvh, &bvh->buildMs, g_buildError)) { delete bvh; return -1; }
    g_buildError.clear();
    return add_slot(g_bvhs, bvh);
}
PT_API const char* PTGetBVHBuildError(void) { return g_buildError.c_str(); }
PT_API double PTGetBVHBuildMs(int index)
{
    ptbvh::Cwbvh* b = get_slot(g_bvhs, index);
    return b ? b->buildMs : -1.0;
}
PT_API void DestroyBVH(int index) { free_slot(g_bvhs, index); }
PT_API int IsBVHReady(int index) { return get_slot(g_bvhs, index) != nullptr; }
PT_API void* GetBVH(int index) { return get_slot(g_bvhs, index); }
PT_API void* GetBVHPtr(int index) { return get_slot(g_bvhs, index); }
PT_API int GetCWBVHNodesSize(int index)
{
    ptbvh::Cwbvh* b = get_slot(g_bvhs, index);
    return b ? (int)(b->usedBlocks * 16u) : 0;
}
PT_API int GetCWBVHTrisSize(int index)
{
    ptbvh::Cwbvh* b = get_slot(g_bvhs, index);
    return b ? (int)(b->triCount * 3u * 16u) : 0;
}
PT_API int GetCWBVHData(int index, PTFloat4** bvhNodes, PTFloat4** bvhTris)
{
    ptbvh::Cwbvh* b = get_slot(g_bvhs, index);
    if (!b || !bvhNodes || !bvhTris || b->nodeData.empty() || b->triData.empty()) return 0;
    *bvhNodes = b->nodeData.data();
    *bvhTris = b->triData.data();
    return 1;
}

PT_API int BuildTLAS(const PTBlasInstance* instances, int instanceCount)
{
    if (!instances || instanceCount <= 0) return -1;
    ptbvh::Tlas* t = new ptbvh::Tlas();
    if (!t->build(instances, (uint32_t)instanceCount)) { delete t; return -1; }
    return add_slot(g_tlases, t);
}
PT_API void DestroyTLAS(int index) { free_slot(g_tlases, index); }
PT_API int IsTLASReady(int index) { return get_slot(g_tlases, index) != nullptr; }
PT_API int GetTLASNodesSize(int index)
{
    ptbvh::Tlas* t = get_slot(g_tlases, index);
    return t ? (int)(t->usedNodes * 64u) : 0;
}
PT_API int GetTLASData(int index, PTFloat4** tlasNodes, uint32_t** tlasIndices)
{
    ptbvh::Tlas* t = get_slot(g_tlases, index);
    if (!t || !tlasNodes || !tlasIndices || t->nodes.empty()) return 0;
    *tlasNodes = (PTFloat4*)t->nodes.data();
    *tlasIndices = t->indices.data();
    return 1;
}

} // extern "C"
