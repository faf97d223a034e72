// gather16.hip — what does a divergent 16-byte-per-lane gather cost on the MI355X vector-memory path (TA / TCP / TD)?
// Each lane chases its own pseudo-random chain of 16-byte rows through a table (so loads are dependent per lane, like BVH
// traversal), with LOADS_PER_STEP independent row requests per step.  Variants:
//   active = 64 / 32 / 16 / 8 lanes per wave (the others idle: does an instruction with few active lanes cost less?)
//   shape  = "lane": every lane its own random 80-byte record, 5 row loads (the traversal's node fetch)
//            "quad": the 4 lanes of a quad load 4 consecutive rows of ONE record (64 contiguous bytes), 5 instructions
//                    fetch 4 records + row 4 per lane (the quad-cooperative fetch), same bytes per lane
//            "line": like "lane" but records 128-byte aligned
// Output: JSON lines with wave-loads/s, lane-requests/s, cycles per wave-load and per lane-request per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int STEPS = 512;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(64, 8) void k_gather(const uint4* __restrict__ table, uint32_t numRecords, uint32_t strideRows, uint32_t activeLanes, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t state = hash(blockIdx.x * 64u + lane + 1u);
    uint32_t acc = 0;
    if (lane < activeLanes) {
        for (int s = 0; s < STEPS; ++s) {
            const uint32_t rec = state % numRecords;
            uint4 r0, r1, r2, r3, r4;
            if (SHAPE == 1) {
                // quad-cooperative: lane j of the quad loads row j of the records of lanes 0..3 of the quad, then row 4 of its own
                const uint32_t j = lane & 3u;
                const uint32_t recA = __shfl(rec, (lane & ~3u) + 0, 64), recB = __shfl(rec, (lane & ~3u) + 1, 64);
                const uint32_t recC = __shfl(rec, (lane & ~3u) + 2, 64), recD = __shfl(rec, (lane & ~3u) + 3, 64);
                r0 = table[(size_t)recA * strideRows + j];
                r1 = table[(size_t)recB * strideRows + j];
                r2 = table[(size_t)recC * strideRows + j];
                r3 = table[(size_t)recD * strideRows + j];
                r4 = table[(size_t)rec * strideRows + 4];
            } else {
                const uint4* p = table + (size_t)rec * strideRows;
                r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3]; r4 = p[4];
            }
            asm volatile("" : "+v"(r0.x), "+v"(r1.x), "+v"(r2.x), "+v"(r3.x), "+v"(r4.x));
            const uint32_t mix = r0.x ^ r1.y ^ r2.z ^ r3.w ^ r4.x;
            acc += mix;
            state = hash(state + (mix & 1u));          // next record depends on the loaded data (a dependent chain)
        }
    }
    if (acc == 0x12345u) out[blockIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t numRecords = 45000;                      // Sponza-class: 44,835 CWBVH nodes
    std::vector<uint32_t> host((size_t)numRecords * 8 * 4);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (uint32_t)rand();
    uint4* table; uint32_t* out;
    CHECK(hipMalloc(&table, host.size() * 4));
    CHECK(hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = cus * 32 * 4;                        // 4 rounds of a full chip of single-wave workgroups
    struct V { const char* name; int shape; uint32_t stride; };
    const V variants[] = {{"lane_80B", 0, 5}, {"line_128B", 0, 8}, {"quad_80B", 1, 5}, {"quad_128B", 1, 8}};
    printf("[\n");
    bool first = true;
    for (const V& v : variants)
        for (uint32_t active : {64u, 32u, 16u, 8u}) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (v.shape == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(64), 0, 0, table, numRecords, v.stride, active, out);
                else hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(64), 0, 0, table, numRecords, v.stride, active, out);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double waveLoads = (double)blocks * STEPS * 5.0;
            const double laneReq = waveLoads * active;
            const double cuCycles = best * 1e-3 * 2.4e9 * cus;
            printf("%s {\"variant\": \"%s\", \"active_lanes\": %u, \"ms\": %.3f, \"wave_loads_per_s\": %.3e, \"lane_requests_per_s\": %.3e, "
                   "\"cu_cycles_per_wave_load\": %.2f, \"cu_cycles_per_lane_request\": %.3f, \"records_per_s\": %.3e}",
                   first ? "" : ",\n", v.name, active, best, waveLoads / (best * 1e-3), laneReq / (best * 1e-3), cuCycles / waveLoads, cuCycles / laneReq,
                   (double)blocks * STEPS * active / (best * 1e-3));
            first = false;
        }
    printf("\n]\n");
    return 0;
}
