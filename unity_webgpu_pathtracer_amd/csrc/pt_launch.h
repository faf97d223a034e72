// pt_launch.h — host-visible launcher declarations shared by pt_kernels.hip and pt_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_device.h"

// Which 16x16-pixel blocks this context renders (PTSetTileOwnership) and how much of the frame the
// dispatch covers (PTFrameParams.DispatchGroupsX/Y, PathTracer.cs:203-208).
struct PTTileMap {
    uint32_t rank, world;
    uint32_t blocksX, blocksY;     // ceil(coverW/16), ceil(coverH/16)
    uint32_t coverW, coverH;       // pixels with x < coverW && y < coverH are rendered
};

hipError_t pt_launch_megakernel(const DScene& S, const PTFrameParams& P, const float4* accumulated, float4* output,
                                const PTTileMap& tm, unsigned long long* gstats, bool fullStats, hipStream_t stream);
