// bvh_builder.h — acceleration-structure builders behind BuildBVH / BuildTLAS (see bvh_builder.cpp).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "ptmi_layouts.h"

namespace ptbvh {

// Binary BVH node, 32 bytes: bounds + (leftFirst, triCount); leaf iff triCount > 0.
struct Node2 {
    float mn[3]; uint32_t leftFirst;
    float mx[3]; uint32_t triCount;
};

class Bvh2 {
public:
    Bvh2();
    ~Bvh2();
    Bvh2(const Bvh2&) = delete;
    Bvh2& operator=(const Bvh2&) = delete;
    void prepareTriangles(const PTFloat4* verts, uint32_t triCount);
    void prepareBoxes(const PTBlasInstance* instances, uint32_t count);
    void build();                       // binned SAH
    void compact();                     // depth-first re-layout
    void splitLeafs(uint32_t maxPrims);
    std::vector<Node2> nodes;           // node 1 is unused (children are allocated in pairs from 2)
    std::vector<uint32_t> primIdx;
    uint32_t triCount = 0, usedNodes = 0, newNodePtr = 2;
private:
    struct Impl;
    Impl* impl;
};

// 8-wide compressed BVH in the layout of include/ptmi_layouts.h (PTCwbvhNode / PTCwbvhTri).
class Cwbvh {
public:
    bool build(const PTFloat4* verts, uint32_t triCount);
    std::vector<PTFloat4> nodeData;     // usedBlocks float4s are valid (5 per node)
    std::vector<PTFloat4> triData;      // triCount * 3 float4s
    uint32_t usedBlocks = 0, triCount = 0;
    double buildMs = 0.0;               // BuildBVH: host wall time of build(); PTBuildBVHDevice: device time of the kernels
};

// bvh_builder_gpu.hip: LBVH + 8-wide collapse + CWBVH encode on the MI355X (same output format, a different tree)
bool build_cwbvh_device(int device, const PTFloat4* verts, uint32_t triCount, Cwbvh& out, double* deviceMs, std::string& err);

// Two-level top: 2-wide BVH over instance AABBs in the Aila-Laine layout (PTTlasNode).
class Tlas {
public:
    bool build(const PTBlasInstance* instances, uint32_t count);
    std::vector<PTTlasNode> nodes;
    std::vector<uint32_t> indices;
    uint32_t usedNodes = 0;
};

} // namespace ptbvh
