// valu_issue.hip — measures the wave64 VALU issue rate of the MI355X (gfx950): wave-instructions per second, chip-wide,
// for independent and for dependent v_fma_f32 streams at 1 / 2 / 4 / 8 waves per SIMD.
//
// Why: profiles/summarize.py and bench.py price the render kernels' SQ_INSTS_VALU against "the VALU issue peak".
// Round 1 hard-coded 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles; /opt/skills/guides/MI355X_MICROARCH.md says a wave64
// v_fma_f32 issues in 2 cycles when more than one wave shares a SIMD and in 4 for one wave alone.  This settles it by
// measurement; the output is kept as profiles/<tag>_valu_issue.json and read by both scripts.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip        Run: ./valu_issue > out.json
//
// Occupancy is pinned with LDS: a 256-thread workgroup (one wave per SIMD) declares 160 KiB / w of LDS, so exactly w
// workgroups fit a CU = w waves per SIMD; the grid is 16 x (256 x w) workgroups, so every CU stays full until the tail.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;      // loop trips per wave
constexpr int PER_ITER = 64;     // v_fma_f32 per trip

// 16 independent accumulators, 4 rounds per trip
__global__ __launch_bounds__(256) void k_indep(float* out, float b, float c)
{
    extern __shared__ float lds[];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < PER_ITER / 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) { lds[threadIdx.x] = s; out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255]; }
}

// ONE accumulator: every v_fma_f32 depends on the previous one
__global__ __launch_bounds__(256) void k_dep(float* out, float b, float c)
{
    extern __shared__ float lds[];
    float a = (float)threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < PER_ITER; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    if (a == 12345.678f) { lds[threadIdx.x] = a; out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255]; }
}

// 8 independent register PAIRS, v_pk_fma_f32 (two fp32 FMAs per lane per instruction): does the packed form issue at the
// plain rate (= 2x the flops) or at half of it?
__global__ __launch_bounds__(256) void k_pk(float* out, float b, float c)
{
    extern __shared__ float lds[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    const f2 bb = {b, b}, cc = {c, c};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)(threadIdx.x + i), (float)(threadIdx.x + 2 * i)};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < PER_ITER / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(bb), "v"(cc));
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) { lds[threadIdx.x] = s; out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255]; }
}

// the traversal's own mix: v_cvt_f32_ubyteN + v_fma_f32 + v_max/min (independent per "child"), 6 cvt + 6 fma + 4 minmax per group
__global__ __launch_bounds__(256) void k_mix(float* out, float b, float c, unsigned q)
{
    extern __shared__ float lds[];
    float acc = 0.0f;
    unsigned qq = q + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float f0, f1, f2, f3, f4, f5;
            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f0) : "v"(qq));
            asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f1) : "v"(qq));
            asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f2) : "v"(qq));
            asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(f3) : "v"(qq));
            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f4) : "v"(qq));
            asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f5) : "v"(qq));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f5) : "v"(b), "v"(c));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f4), "v"(f5));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(f0) : "v"(c));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(f3) : "v"(b));
            acc += f0 - f3;           // 2 more VALU (v_sub + v_add) per group, dependent on acc only through the add
        }
    }
    if (acc == 12345.678f) { lds[threadIdx.x] = acc; out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255]; }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, 64u << 20));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_max\": %d, \"iters\": %d, \"valu_per_iter\": {\"indep\": %d, \"dep\": %d, \"mix\": %d, \"pk\": %d},\n \"results\": [\n",
           prop.gcnArchName, cus, prop.clockRate / 1000, ITERS, PER_ITER, PER_ITER, 4 * 18, PER_ITER);
    const char* names[4] = {"indep", "dep", "mix", "pk_fma"};
    bool first = true;
    for (int kind = 0; kind < 4; ++kind) {
        for (int w : {1, 2, 4, 8}) {
            const size_t ldsBytes = (160u * 1024u) / w;
            const void* fn = kind == 0 ? (const void*)k_indep : kind == 1 ? (const void*)k_dep : kind == 2 ? (const void*)k_mix : (const void*)k_pk;
            CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
            const int blocks = 16 * cus * w;
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (kind == 0) hipLaunchKernelGGL(k_indep, dim3(blocks), dim3(256), ldsBytes, 0, out, 1.0000001f, 1e-9f);
                else if (kind == 1) hipLaunchKernelGGL(k_dep, dim3(blocks), dim3(256), ldsBytes, 0, out, 1.0000001f, 1e-9f);
                else if (kind == 3) hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), ldsBytes, 0, out, 1.0000001f, 1e-9f);
                else hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), ldsBytes, 0, out, 1.0000001f, 1e-9f, 0x01020304u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double perIter = kind == 2 ? 4.0 * 18.0 : (double)PER_ITER;
            const double waveInsts = (double)blocks * 4.0 * ITERS * perIter;
            const double rate = waveInsts / (best * 1e-3);
            const double cyclesPerInst = (double)cus * 4.0 * 2.4e9 / rate;       // per SIMD, at the nominal 2.4 GHz
            printf("%s  {\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_s\": %.4e, \"simd_cycles_per_wave_inst_at_2.4GHz\": %.3f}",
                   first ? "" : ",\n", names[kind], w, best, rate, cyclesPerInst);
            first = false;
        }
    }
    printf("\n ]}\n");
    return 0;
}
