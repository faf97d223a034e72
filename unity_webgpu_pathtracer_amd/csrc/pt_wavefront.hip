// pt_wavefront.hip — schedules 1-3: the same per-pixel program as the megakernel, cut at the traversal calls.
//
// Why: in one fused kernel the Disney-BSDF shading code sets the register budget (185 VGPRs -> 2 waves/SIMD, or
// 128 with spills -> 4), and lanes that shade wait for lanes that traverse and vice versa (measured VALU lane
// utilisation 19 %).  Here every pixel owns a SLOT of path state in HBM and a pass is
//     init                     camera ray of the first sample into every slot
//     N x [ trace ; shade ]    trace: lean CWBVH traversal only, 8 waves/SIMD, rays compacted into lanes per wave;
//                              shade: path_step() of pt_device.h (apply NEE, shade, roulette, next sample)
//     cleanup                  lanes whose pixel is still alive after N iterations run it to completion megakernel-style
//     resolve                  out = (sum + accumulated * CurrentSample) / (CurrentSample + spp)   (PathTracer.compute:89-98)
// Samples of a pixel stay sequential (one RNG chain per pixel per pass, PathTracer.compute:60,66), so a slot carries one
// path at a time and a pass needs about SamplesPerPass x (bounces + 2) iterations.  N is that bound; pixels that need
// more (alpha-skips add iterations without adding depth) are finished by the cleanup kernel, so a pass is a fixed,
// HOST-SYNC-FREE sequence of launches.  Because the pixel write is decoupled (resolve), pass k+1 can start tracing on a
// second state set and stream while pass k is still draining; only the tiny resolve kernels are ordered.
//
// There are NO queues and NO global atomics in schedule 1: rays are found by scanning the slot-indexed flag words,
// results go to slot-indexed arrays, work counters accumulate in per-wave rows (plain read-modify-write: one wave owns
// a row) and are folded once per pass.  Every fp32 value and RNG draw is produced by the same device functions in the
// same per-path order as in schedule 0, so frames and counters are bit-identical between schedules and to the oracle.
#include "pt_device.h"
#include "pt_launch.h"
#include <type_traits>

namespace {

#define PT_WF_CHUNK 64u         // slots per work chunk of the persistent trace kernel (= one scan window per ray kind)
#define PT_WF_SHARDS 8u         // chunk counters, one per XCD

// flags word: [1:0] state | [2] hasPending | [4:3] env.valid | [5] light.valid | [6] green | [18:7] sampleIdx | [31:19] depth
PT_DEV uint32_t pack_flags(const PathRegs& r)
{
    return (r.state & 3u) | ((r.hasPending ? 1u : 0u) << 2) | ((r.env.valid & 3u) << 3) | ((r.light.valid & 1u) << 5) |
           ((r.green ? 1u : 0u) << 6) | ((r.sampleIdx & 0xFFFu) << 7) | ((r.depth & 0x1FFFu) << 19);
}
PT_DEV uint32_t fl_state(uint32_t f) { return f & 3u; }
PT_DEV bool fl_pending(uint32_t f) { return (f >> 2) & 1u; }
PT_DEV uint32_t fl_env(uint32_t f) { return (f >> 3) & 3u; }
PT_DEV uint32_t fl_light(uint32_t f) { return (f >> 5) & 1u; }

PT_DEV float4 f4(v3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
PT_DEV v3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }

// float4 state array k of the set (pt_launch.h PT_F4_*), addressed from ONE base pointer
PT_DEV float4* f4_array(const PTWfBuffers& B, uint32_t k) { return B.f4base + (size_t)k * B.f4stride; }
// the ray of (slot, kind): bounce ray (ro, rd), environment NEE (neeO, envD) or light NEE (neeO, lightD)
PT_DEV void fetch_ray(const PTWfBuffers& B, uint32_t slot, uint32_t kind, v3& o, v3& d)
{
    const float4* rec = f4_array(B, 2u * kind) + 2u * (size_t)slot;          // PT_F4_RAY0 / 1 / 2: one 32-byte record, one sector
    o = xyz(rec[0]);
    d = xyz(rec[1]);
}

PT_DEV void store_path(const PTWfBuffers& B, uint32_t slot, const PathRegs& r, bool writeNee)
{
    B.flags[slot] = pack_flags(r);
    B.rng[slot] = r.rng;
    B.ray[0][2u * slot] = f4(r.ro, r.scatterPdf);
    B.ray[0][2u * slot + 1u] = f4(r.rd, r.maxRoughness);
    B.rad[slot] = f4(r.radiance, 0.0f);
    B.thr[slot] = f4(r.throughput, 0.0f);
    B.color[slot] = f4(r.color, 0.0f);
    if (writeNee) {
        B.ray[1][2u * slot] = f4(r.neeOrigin, 0.0f);
        B.ray[1][2u * slot + 1u] = f4(r.env.dir, 0.0f);
        B.ray[2][2u * slot] = f4(r.neeOrigin, 0.0f);
        B.ray[2][2u * slot + 1u] = f4(r.light.dir, 0.0f);
        B.envC[slot] = f4(r.env.contribution, 0.0f);
        B.lightC[slot] = f4(r.light.contribution, 0.0f);
        B.pthr[slot] = f4(r.pendThroughput, 0.0f);
    }
}

// Tail cut: B.aliveFlags[i] = number of slots still alive after shade launch i (zeroed at the start of a pass).  Once that
// drops below B.tailCut, every later trace / shade launch of the pass returns at its first instruction and pt_wf_cleanup runs
// the stragglers to completion: the last ~20 of the 52 iterations of a pass used to cost ~170 us each (the latency of the
// longest surviving ray, three launches per iteration) for a handful of rays.  Wave-uniform, no host synchronisation.
// The count is kept in PT_WF_ALIVE_SHARDS words per iteration (a wave adds to word blockIdx % SHARDS): 32,000 atomics per
// launch on ONE word serialise in one L2 channel and doubled the time of a pass.
PT_DEV bool tail_cut(const PTWfBuffers& B, uint32_t iteration)
{
    if (iteration == 0u || B.tailCut == 0u) return false;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t part = lane < PT_WF_ALIVE_SHARDS ? B.aliveFlags[(iteration - 1u) * PT_WF_ALIVE_SHARDS + lane] : 0u;
    return wave_sum_u32(part) < B.tailCut;
}

// per-wave counter rows: one wave owns a row, so a plain read-modify-write is race-free across launches of one stream
template <bool STATS>
PT_DEV void flush_counters(const Counters& cn, unsigned long long* rows, uint32_t row, uint32_t lane)
{
    uint32_t vals[PT_NUM_COUNTERS];
    counters_to_array(cn, vals);
    unsigned long long* p = rows + (size_t)row * 16u;
#pragma unroll
    for (int i = 0; i < PT_NUM_COUNTERS; ++i) {
        if (!STATS && (i >= 3 && i <= 9)) continue;
        if (!STATS && i >= 12) continue;
        if (i == 12) {
            uint32_t m = wave_max_u32(vals[i]);
            if (lane == 0 && m > p[i]) p[i] = m;
        } else {
            uint32_t s = wave_sum_u32(vals[i]);
            if (lane == 0 && s) p[i] += s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// init: every owned pixel starts its first sample (camera ray in the slot, state = TRACE)
// ------------------------------------------------------------------------------------------
// the pixel slot and the pass (of a batch) a slot belongs to
PT_DEV uint32_t pixel_slot_of(const PTWfBuffers& B, uint32_t slot, uint32_t& pass)
{
    pass = 0u;
    if (slot >= B.slotsPerPass) { pass = slot / B.slotsPerPass; slot -= pass * B.slotsPerPass; }
    return slot;
}

__global__ __launch_bounds__(256) void pt_wf_init(PTFrameParams P, PTBatch batch, PTTileMap tm, PTWfBuffers B)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0u && threadIdx.x < PT_WF_SHARDS) B.chunkHeads[threadIdx.x * 32u] = 0u;
    if (slot >= B.numSlots) return;
    uint32_t px, py, pass;
    Counters cn = {};
    if (!pt_slot_to_pixel(tm, pixel_slot_of(B, slot, pass), px, py)) B.flags[slot] = PS_DONE;
    else {
        PathRegs r;
        uint32_t seedRoot, currentSample;
        pt_batch_pick(batch, pass, seedRoot, currentSample);
        path_init(P, seedRoot, currentSample, px, py, py * P.OutputWidth + px, r, cn);
        store_path(B, slot, r, false);
    }
    // every lane of the wave takes part in the reduction (a lane that had returned would be read as garbage)
    flush_counters<false>(cn, B.statRows, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63u);
}

// ------------------------------------------------------------------------------------------
// trace: one lane per (slot, kind).  kind 0 = closest hit of the bounce ray, 1 = environment NEE, 2 = light NEE
// ------------------------------------------------------------------------------------------
#ifndef PT_WF_TRACE_MIN_WAVES
#define PT_WF_TRACE_MIN_WAVES 8
#endif

template <bool STATS, bool TLAS>
__global__ __launch_bounds__(256, 4) void pt_wf_trace(DScene S, PTWfBuffers B, uint32_t iteration)
{
    __shared__ uint2 s_stack[PT_LDS_STACK][256];
    if (tail_cut(B, iteration)) return;
    const uint32_t nb = B.numSlots >> 8;
    const uint32_t vb = blockIdx.x;
    const uint32_t kind = vb / nb;
    const uint32_t slot = (vb % nb) * 256u + threadIdx.x;
    const uint32_t f = B.flags[slot];
    bool valid;
    if (kind == 0u) valid = fl_state(f) == PS_TRACE;
    else if (kind == 1u) valid = fl_pending(f) && fl_env(f) != 0u;
    else valid = fl_pending(f) && fl_light(f) != 0u;
    Counters cn = {};
    if (valid) {
        v3 o, d;
        fetch_ray(B, slot, kind, o, d);
        TravStack st;
        st.lds = PT_LDS_U2(&s_stack[0][threadIdx.x]);
        st.stride = 256u;
        HitRecord h;
        h.h.t = PT_FAR_PLANE; h.h.u = 0.0f; h.h.v = 0.0f; h.h.triIndex = 0u;
        h.pos = mk3(0.0f); h.inst = 0u;
        bool occluded;
        if (TLAS) occluded = traverse_tlas<STATS>(S, o, d, kind != 0u, h, st, cn);
        else { traverse_cwbvh<STATS>(S, o, d, kind != 0u, h.h, st, cn); occluded = h.h.t < PT_FAR_PLANE; }
        if (kind == 0u) {
            B.hit[slot] = make_float4(h.h.t, h.h.u, h.h.v, pt_asfloat(h.h.triIndex));
            if (TLAS) B.hit2[slot] = make_float4(h.pos.x, h.pos.y, h.pos.z, pt_asfloat(h.inst));
            cn.closestRays++;
        } else { B.occl[(size_t)(kind - 1u) * B.numSlots + slot] = occluded ? 1 : 0; cn.shadowRays++; }
    }
    flush_counters<STATS>(cn, B.statRows, (B.numSlots >> 6) + vb * 4u + (threadIdx.x >> 6), threadIdx.x & 63u);
}

// ------------------------------------------------------------------------------------------
// trace with refill (schedule 1): persistent-threads ray scheduling inside each wave, no atomics.
//
// A wave owns PT_WF_RANGE consecutive slots = 3 * PT_WF_RANGE potential rays, enumerated kind-major (all bounce
// rays of the range, then the environment shadow rays, then the light shadow rays: neighbours in the enumeration are
// neighbours on screen and of one kind, i.e. as coherent as this path tracer gets).  The wave keeps 64 resumable
// traversals in flight; whenever PT_WF_REFILL or more lanes have retired their ray it scans the next 64 candidates
// (one coalesced read of the flag words), ballots which exist, and compacts them into the idle lanes through a
// 64-entry LDS exchange (rank-of-set-bit on both masks).  Rays of very different length therefore never hold a wave
// hostage, and the enumeration order -- hence every result -- is independent of timing.
// ------------------------------------------------------------------------------------------
#ifndef PT_WF_RANGE
#define PT_WF_RANGE 128u        // largest slots-per-wave the launcher may pick (power of two): two 8x8 tiles x 3 kinds = 384 candidate rays.
                                // Full 1080p frame, 64 / 128 / 256: 4,590 / 4,700 / 4,690 Mrays/s; small launches use 64 (pt_launch_wavefront)
#endif
static_assert(PT_WF_RANGE >= 64u && (PT_WF_RANGE & (PT_WF_RANGE - 1u)) == 0u,
              "one counter row per trace wave: pt_api.hip sizes statRows for 4 x numSlots/64 rows; the main trace launch uses rows numSlots/64 + wave, the tail launch rows 2 x numSlots/64 + wave");
#ifndef PT_WF_REFILL
#define PT_WF_REFILL 16u        // refill when at least this many lanes are idle
#endif
#ifndef PT_WF_FLAT
#define PT_WF_FLAT 0            // 0: ray_step (node visit + all its triangles per wave iteration), 2: ray_unistep (one triangle OR one node visit, one memory round trip)
#endif
#ifndef PT_WF_TRI_PARK
#define PT_WF_TRI_PARK 4u       // > 0: two-phase wave iteration (ray_tri_one / ray_node_one); lanes with triangles pending wait until
                                // that many lanes do.  0: ray_step's nested loops.  2 ... 12 measure the same (+6 %), 16: +3 %, 24: -2 %
#endif
#ifndef PT_WF_ROOT_SCALAR
#define PT_WF_ROOT_SCALAR 0     // 1: main refill launch reads the root node once per wave through the scalar path and visits it when a ray is taken.
                                // Measured: 4,918 / 4,938 against 5,090 / 5,063 Mrays/s -- 4.5e8 lane requests and one dependent fetch per ray saved, but the
                                // node test then runs in the refill block at 25-50 % of the lanes (round 2's LDS copy of the root lost the same way)
#endif
#ifndef PT_WF_HIT_REGS
#define PT_WF_HIT_REGS 1        // main refill launch (with PT_WF_TRI_PARK): the hit's (u, v, triIndex) stay in registers until the ray ends
#endif
#if PT_WF_FLAT == 2
#define PT_WF_STEP ray_unistep
#else
#define PT_WF_STEP ray_step
#endif

// LDS words the lanes of a wave exchange through, as LDS-address-space pointers: through a generic `volatile uint32_t*` every
// access is a FLAT instruction (the compiler does not infer the address space of a volatile access) -- it travels the
// texture-address path and counts against both memory counters.
typedef volatile __attribute__((address_space(3))) uint32_t* pt_lds_u32;
typedef volatile __attribute__((address_space(3))) uint16_t* pt_lds_u16;
// number of set bits of a wave mask below this lane (v_mbcnt: no per-lane 64-bit "lanes below me" mask to keep in registers)
PT_DEV uint32_t rank_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Result stores of the trace kernels.  The empty asm pins the address arithmetic (and the constant miss record) to the
// store instead of letting it be hoisted into registers that live across the whole traversal loop (64-VGPR budget).
PT_DEV void store_miss(const PTWfBuffers& B, uint32_t slot)
{
    float far = PT_FAR_PLANE;
    asm volatile("" : "+v"(far), "+v"(slot));
    f4_array(B, PT_F4_HIT)[slot] = make_float4(far, 0.0f, 0.0f, 0.0f);
}
PT_DEV void store_occlusion(const PTWfBuffers& B, uint32_t kind, uint32_t slot, bool occluded)
{
    asm volatile("" : "+v"(kind), "+v"(slot));
    B.occl[(size_t)(kind - 1u) * B.numSlots + slot] = occluded ? 1 : 0;
}

PT_DEV bool ray_exists(uint32_t f, uint32_t kind)
{
    if (kind == 0u) return fl_state(f) == PS_TRACE;
    if (kind == 1u) return fl_pending(f) && fl_env(f) != 0u;
    return fl_pending(f) && fl_light(f) != 0u;
}

// ---- suspension (PT_WF_SUSPEND > 0) -----------------------------------------------------------------------------------
// A launch only holds 5-6 rays per lane of the chip, and ray lengths spread over an order of magnitude, so a wave spends
// most of its iterations DRAINING: measured on the Sponza-class pass, 54 % of all wave iterations ran after the wave's
// range was exhausted and 32 % with <= 16 active lanes, each costing the same ~450 VALU instructions and memory round
// trips as a full one.  So a wave whose range is exhausted stops as soon as PT_WF_SUSPEND or fewer rays are left: it
// writes those rays' traversal state (node group, triangle group, stack, t: 96 bytes) to a record array and exits, and a
// TAIL launch of the same kernel packs the records of PT_WF_TAIL_GROUP consecutive waves into full waves and resumes
// them.  A ray resumes exactly where it stopped, so results and counters are unchanged; late, sparse iterations (a
// handful of rays per 64-slot range) become a cheap scan + an 8:1 compacted tail.
// PT_WF_SUSPEND (pt_launch.h, default 16; 0 = off)
#ifndef PT_WF_TAIL_GROUP
#define PT_WF_TAIL_GROUP 8u     // source waves per tail wave
#endif
#define PT_WF_SUSP_STACK_ROWS ((PT_WF_LDS_STACK + 1u) / 2u)       // two stack entries per uint4 row
#define PT_WF_SUSP_ROWS (2u + PT_WF_SUSP_STACK_ROWS)             // uint4 rows per record (6 = 96 bytes with the default 8-entry LDS stack)
static_assert(PT_WF_SUSP_ROWS <= 6u, "pt_api.hip sizes the record array for 6 rows per record");

// A record holds the LDS part of the stack; a ray whose stack reaches into the HBM slab is not suspended (it is finished first).
template <class ST>
PT_DEV void suspend_ray(const PTWfBuffers& B, uint32_t rec, uint32_t slot, uint32_t kind, const RayState& r, ST& st)
{
    uint4* p = B.susp + (size_t)rec * PT_WF_SUSP_ROWS;
    p[0] = make_uint4(slot | (kind << 30), r.sp | (r.overflow ? 0x100u : 0u), r.ng.x, r.ng.y);
    p[1] = make_uint4(r.tg.x, r.tg.y, pt_asuint(r.hit.t), 0u);
#pragma unroll
    for (uint32_t e = 0; e < PT_WF_SUSP_STACK_ROWS; ++e) {
        const pt_u2 x = st.lds[(2u * e) * st.stride];
        const pt_u2 y = (2u * e + 1u < PT_WF_LDS_STACK) ? (pt_u2)st.lds[(2u * e + 1u) * st.stride] : pt_u2{0u, 0u};
        p[2u + e] = make_uint4(x.x, x.y, y.x, y.y);
    }
}

// RANGE: slots per wave (64 or 128; pt_launch_wavefront picks by the size of the launch)
template <bool STATS, bool TAIL, uint32_t RANGE>
__global__ __launch_bounds__(64, PT_WF_TRACE_MIN_WAVES) void pt_wf_trace_refill(DScene S, PTWfBuffers B, uint32_t iteration)
{
    __shared__ uint2 s_stack[PT_WF_LDS_STACK][64];
    if (tail_cut(B, iteration)) return;                               // main and tail launch of an iteration decide alike
    __shared__ uint32_t s_xchg[TAIL ? PT_WF_TAIL_GROUP * (PT_WF_SUSPEND ? PT_WF_SUSPEND : 1u) : 64u];
    // the wave's counter row is parked in LDS until the end: kept in a register it is the one value the compiler spilled to
    // scratch, and ANY scratch costs this kernel its occupancy (see TravStackT)
    __shared__ uint32_t s_gw;
    const uint32_t lane = threadIdx.x;
    const uint32_t gw = blockIdx.x;
    const uint32_t numWaves = (B.numSlots + RANGE - 1u) / RANGE;
    const uint32_t slotBase = gw * RANGE;
    pt_lds_u32 xchg = (pt_lds_u32)&s_xchg[0];
    if (lane == 0u) s_gw = gw;

    // candidates: the main launch scans the flag words of its slot range, kind-major; the tail launch walks the records that
    // its PT_WF_TAIL_GROUP source waves left behind (their indices, packed, in s_xchg)
    uint32_t nItems = 3u * RANGE;
    if (TAIL) {
        const uint32_t s0 = gw * PT_WF_TAIL_GROUP;
        uint32_t c = 0u;
        if (lane < PT_WF_TAIL_GROUP && s0 + lane < numWaves) c = B.suspCount[s0 + lane];
        uint32_t incl = c;                                            // inclusive prefix over lanes 0..PT_WF_TAIL_GROUP-1
#pragma unroll
        for (uint32_t off = 1u; off < PT_WF_TAIL_GROUP; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += o; }
        nItems = (uint32_t)__shfl((int)incl, (int)PT_WF_TAIL_GROUP - 1, 64);
        constexpr uint32_t kT = PT_WF_SUSPEND ? PT_WF_SUSPEND : 1u;
#pragma unroll
        for (uint32_t r = 0; r < (PT_WF_TAIL_GROUP * kT + 63u) / 64u; ++r) {
            const uint32_t q = r * 64u + lane, i = q / kT, k = q % kT;
            const uint32_t ci = (uint32_t)__shfl((int)c, (int)(i & 63u), 64), pi = (uint32_t)__shfl((int)(incl - c), (int)(i & 63u), 64);
            if (i < PT_WF_TAIL_GROUP && k < ci) xchg[pi + k] = (s0 + i) * kT + k;
        }
        __builtin_amdgcn_wave_barrier();
        if (nItems == 0u) return;                                     // wave-uniform; nothing was counted
    }

    Counters cn = {};
    TravStackT<PT_WF_LDS_STACK, true> st;
    st.lds = PT_LDS_U2(&s_stack[0][lane]);
    st.stride = 64u;
    st.gbase = B.stackSpill;
    st.gwave = PT_LDS_WORD(s_gw);                                                 // slab row = wave * 64 + lane < numSlots
#if PT_WF_ROOT_SCALAR
    uint4 root0 = make_uint4(0u, 0u, 0u, 0u), root1 = root0, root2 = root0, root3 = root0, root4 = root0;
    if (!TAIL) {
        const float4 r0 = pt_uniform_load((const float4*)S.nodes, 0), r1 = pt_uniform_load((const float4*)S.nodes, 1), r2 = pt_uniform_load((const float4*)S.nodes, 2),
                     r3 = pt_uniform_load((const float4*)S.nodes, 3), r4 = pt_uniform_load((const float4*)S.nodes, 4);
        root0 = make_uint4(pt_asuint(r0.x), pt_asuint(r0.y), pt_asuint(r0.z), pt_asuint(r0.w));
        root1 = make_uint4(pt_asuint(r1.x), pt_asuint(r1.y), pt_asuint(r1.z), pt_asuint(r1.w));
        root2 = make_uint4(pt_asuint(r2.x), pt_asuint(r2.y), pt_asuint(r2.z), pt_asuint(r2.w));
        root3 = make_uint4(pt_asuint(r3.x), pt_asuint(r3.y), pt_asuint(r3.z), pt_asuint(r3.w));
        root4 = make_uint4(pt_asuint(r4.x), pt_asuint(r4.y), pt_asuint(r4.z), pt_asuint(r4.w));
    }
#endif
    RayState rs;
    rs.sp = 0u; rs.anyHit = false; rs.overflow = false;
    bool have = false;
    uint32_t mySlot = 0u, myKind = 0u;
    uint32_t cursor = 0u, nSuspended = 0u;

    while (true) {
        uint32_t nIdle = (uint32_t)__popcll(__ballot(!have));
        // ---- refill: compact the next candidates into the idle lanes
        while (cursor < nItems && (nIdle >= PT_WF_REFILL || nIdle == 64u)) {
            const unsigned long long idle = __ballot(!have);
            const uint32_t rankI = rank_below(idle);
            if (!TAIL) {
                const uint32_t item = cursor + lane;
                const uint32_t kind = item / RANGE;
                const uint32_t slot = slotBase + (item & (RANGE - 1u));
                bool valid = item < nItems && slot < B.numSlots;
                if (valid) valid = ray_exists(B.flags[slot], kind);
                const unsigned long long V = __ballot(valid);
                const uint32_t nV = (uint32_t)__popcll(V);
                const uint32_t take = nIdle < nV ? nIdle : nV;
                const uint32_t rankV = rank_below(V);
                uint32_t consumed = 64u;
                if (take < nV) consumed = (uint32_t)__ffsll((long long)__ballot(valid && rankV == take)) - 1u;   // first ray NOT taken
                if (valid && rankV < take) xchg[rankV] = item;
                __builtin_amdgcn_wave_barrier();
                if (!have && rankI < take) {
                    const uint32_t it = xchg[rankI];
                    myKind = it / RANGE;
                    mySlot = slotBase + (it & (RANGE - 1u));
                    v3 o, d;
                    fetch_ray(B, mySlot, myKind, o, d);
                    if (myKind == 0u) cn.closestRays++; else cn.shadowRays++;
                    if (ray_begin(rs, o, d, myKind != 0u)) {
                        // NaN ray: certain miss
                        if (myKind == 0u) store_miss(B, mySlot);
                        else store_occlusion(B, myKind, mySlot, false);
                    } else {
                        have = true;
#if PT_WF_ROOT_SCALAR
                        // Every ray's first node visit is the ROOT (ray_begin leaves the node group (0, bit 31): child 0 of base 0): the
                        // same five rows for every lane.  They are read once per wave through the scalar path (root0..root4) and the
                        // visit happens here, at refill time: ~9 x 10^7 rays x 5 rows per pass leave the vector-memory path and every
                        // ray's chain of dependent fetches is one shorter.  Same operations per ray: cwbvh_pick_child (nothing to push:
                        // the group holds one child) + cwbvh_apply_node.
                        rs.ng.y = 0u;                                                       // the root taken off its node group
                        cwbvh_apply_node<STATS>(rs.o, rs.invDir, rs.octinv4, rs.hit.t, root0, root1, root2, root3, root4, rs.ng, rs.tg, cn);
#endif
                    }
                }
                __builtin_amdgcn_wave_barrier();
                cursor += consumed;
            } else {
                const uint32_t left = nItems - cursor;
                const uint32_t take = nIdle < left ? nIdle : left;
                if (!have && rankI < take) {
                    const uint4* p = B.susp + (size_t)xchg[cursor + rankI] * PT_WF_SUSP_ROWS;
                    const uint4 a = p[0], b = p[1];
                    uint4 er[PT_WF_SUSP_STACK_ROWS];
#pragma unroll
                    for (uint32_t e = 0; e < PT_WF_SUSP_STACK_ROWS; ++e) er[e] = p[2u + e];
                    mySlot = a.x & 0x3FFFFFFFu;
                    myKind = a.x >> 30;
                    v3 o, d;
                    fetch_ray(B, mySlot, myKind, o, d);
                    ray_begin(rs, o, d, myKind != 0u);                // same invDir / octinv4 as when the ray started
                    rs.sp = a.y & 0xFFu;
                    rs.overflow = (a.y & 0x100u) != 0u;
                    rs.ng = make_uint2(a.z, a.w);
                    rs.tg = make_uint2(b.x, b.y);
                    rs.hit.t = pt_asfloat(b.z);
#pragma unroll
                    for (uint32_t e = 0; e < PT_WF_SUSP_STACK_ROWS; ++e) {
                        st.lds[(2u * e) * 64u] = pt_u2{er[e].x, er[e].y};
                        if (2u * e + 1u < PT_WF_LDS_STACK) st.lds[(2u * e + 1u) * 64u] = pt_u2{er[e].z, er[e].w};
                    }
                    have = true;
                }
                cursor += take;
            }
            nIdle = (uint32_t)__popcll(__ballot(!have));
        }
        if (nIdle == 64u) break;                                      // candidates exhausted and nothing in flight
        // ---- traverse until enough lanes have retired; once the candidates are exhausted, until PT_WF_SUSPEND or fewer rays
        //      are left (main launch) or all have finished (tail launch)
        const bool exhausted = cursor >= nItems;
        const uint32_t stopAt = !exhausted ? PT_WF_REFILL : ((!TAIL && PT_WF_SUSPEND > 0u) ? 64u - PT_WF_SUSPEND : 64u);
        while (nIdle < stopAt) {
#ifdef PT_TRACE_HIST
            // diagnostics build: histogram of active lanes per wave iteration, reported through the shading counters
            // (attrFetches: 1-8 lanes, materialFetches: 9-16, lightFetches: 17-32, texelFetches: 33-48, texDescFetches: 49-64;
            //  tlasNodeVisits: iterations after the wave's candidates were exhausted)
            {
                const uint32_t nAct = (uint32_t)__popcll(__ballot(have));
                if (lane == 0u) {
                    if (nAct <= 8u) cn.attrFetches++; else if (nAct <= 16u) cn.materialFetches++; else if (nAct <= 32u) cn.lightFetches++;
                    else if (nAct <= 48u) cn.texelFetches++; else cn.texDescFetches++;
                    if (exhausted) cn.tlasNodeVisits++;
                }
            }
#endif
#if PT_WF_TRI_PARK > 0
            // Two phases per wave iteration instead of ray_step's nested loops: (1) every lane with a triangle pending tests ONE
            // (the block runs when PT_WF_TRI_PARK lanes want it, or when nobody can do anything else), (2) every lane without a
            // triangle pending -- including those that have just tested their last one -- pops and visits its next node.  A lane
            // with k triangles spends k - 1 extra iterations in phase 1 while its neighbours keep visiting nodes; the triangle
            // block runs once per iteration at ~3x the lane utilisation of the nested loop (2.1 executions at 8 %).
            {
                const bool wantTri = have && rs.tg.y != 0u;
                const uint32_t nT = (uint32_t)__popcll(__ballot(wantTri));
                const uint32_t nN = (uint32_t)__popcll(__ballot(have && !wantTri));
                bool fin = false;
                if ((nT >= PT_WF_TRI_PARK || nN == 0u) && wantTri) {
                    const float tBefore = rs.hit.t;
                    fin = ray_tri_one<STATS>(S, rs, cn);
                    // TAIL: the hit record goes to memory when a test improved it (a resumed ray has only t in registers).
                    // main launch (PT_WF_HIT_REGS): (u, v, triIndex) stay in registers and are written once, when the ray finishes or is suspended
                    if ((TAIL || !PT_WF_HIT_REGS) && myKind == 0u && rs.hit.t < tBefore) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                }
                if (have && !fin && rs.tg.y == 0u) fin = ray_node_one<STATS>(S, rs, st, cn);
                if (fin) {
                    if (myKind == 0u) {
                        if (!(rs.hit.t < PT_FAR_PLANE)) store_miss(B, mySlot);
                        else if (!TAIL && PT_WF_HIT_REGS) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                    }
                    else store_occlusion(B, myKind, mySlot, rs.hit.t < PT_FAR_PLANE);
                    have = false;
                }
            }
#else
            if (have) {
                // the hit record goes to memory whenever a step improved it, so (u, v, triIndex) need no registers across
                // iterations (64 VGPRs without a spill = no scratch, see TravStackT); a ray that never hits writes the miss
                const float tBefore = rs.hit.t;
                const bool fin = PT_WF_STEP<STATS>(S, rs, st, cn);
                if (myKind == 0u && rs.hit.t < tBefore) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                if (fin) {
                    if (myKind == 0u) { if (!(rs.hit.t < PT_FAR_PLANE)) store_miss(B, mySlot); }
                    else store_occlusion(B, myKind, mySlot, rs.hit.t < PT_FAR_PLANE);
                    have = false;
                }
            }
#endif
            nIdle = (uint32_t)__popcll(__ballot(!have));
        }
        if (!TAIL && PT_WF_SUSPEND > 0u && exhausted && nIdle < 64u) {
            // a ray whose stack reaches into the HBM slab stays (the record only holds the LDS entries): finish those first
            if (__ballot(have && rs.sp > PT_WF_LDS_STACK) != 0ull) {
                while (__ballot(have && rs.sp > PT_WF_LDS_STACK) != 0ull) {
                    if (have) {
                        const float tBefore = rs.hit.t;
#if PT_WF_TRI_PARK > 0
                        // a lane may arrive with triangles pending: one triangle OR one node visit, whichever is next for it
                        const bool fin = rs.tg.y != 0u ? ray_tri_one<STATS>(S, rs, cn) : ray_node_one<STATS>(S, rs, st, cn);
#else
                        const bool fin = PT_WF_STEP<STATS>(S, rs, st, cn);
#endif
                        if ((TAIL || !PT_WF_HIT_REGS || PT_WF_TRI_PARK == 0u) && myKind == 0u && rs.hit.t < tBefore) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                        if (fin) {
                            if (myKind == 0u) {
                                if (!(rs.hit.t < PT_FAR_PLANE)) store_miss(B, mySlot);
                                else if (!TAIL && PT_WF_HIT_REGS && PT_WF_TRI_PARK > 0u) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                            } else store_occlusion(B, myKind, mySlot, rs.hit.t < PT_FAR_PLANE);
                            have = false;
                        }
                    }
                }
            }
            const unsigned long long act = __ballot(have);
            // a suspended bounce ray leaves its best hit so far in the hit array (the tail launch writes there only when it improves it)
            if (PT_WF_HIT_REGS && PT_WF_TRI_PARK > 0u && have && myKind == 0u && rs.hit.t < PT_FAR_PLANE)
                f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
            if (have) suspend_ray(B, *PT_LDS_WORD(s_gw) * PT_WF_SUSPEND + rank_below(act), mySlot, myKind, rs, st);
            have = false;
            nSuspended = (uint32_t)__popcll(act);
            break;
        }
    }
    if (!TAIL && PT_WF_SUSPEND > 0u && lane == 0u) B.suspCount[*PT_LDS_WORD(s_gw)] = nSuspended;     // every main wave reports, zero included
    __builtin_amdgcn_wave_barrier();
    flush_counters<STATS>(cn, B.statRows, (TAIL ? 2u : 1u) * (B.numSlots >> 6) + *PT_LDS_WORD(s_gw), lane);
}

// ------------------------------------------------------------------------------------------
// HAS_TLAS through the refill scheduler (util/tlas.hlsl:129-332).  The two-level walk of a ray is cut so that the hot loop of
// the wave contains ONE kind of step, the CWBVH iteration of the instance a lane is inside (ray_step with the instance's
// node / triangle / attribute offsets and the tmin = 0 triangle test); whenever a lane's instance is finished it runs the
// instance epilogue (world-space position and distance, tlas.hlsl:208-229) and walks the 2-wide TLAS -- a short, rarely taken
// loop -- to the next instance whose box it hits.  Per lane: the BLAS stack in LDS (8 entries, deeper ones in the HBM slab) plus
// a second, narrow LDS stack for TLAS node indices; the world-space ray is re-read from the slot arrays when an instance is
// entered.  Same per-ray operation order as traverse_tlas, so frames and all counters stay bit-identical.
// ------------------------------------------------------------------------------------------
#ifndef PT_WF_TLAS_LDS_STACK
#define PT_WF_TLAS_LDS_STACK 8u
#endif

#ifndef PT_WF_TLAS_STEPS
#define PT_WF_TLAS_STEPS 0xFFFFFFFFu // TLAS-walk steps a lane between instances takes per wave iteration (0xFFFFFFFF: until it is inside the next instance).
                                     // 200-instance scene, 1080p / 8 spp: unbounded 2,076 Mrays/s, 1 step 1,971, 2 steps 1,889 (round 3)
#endif
#ifndef PT_WF_TLAS_CONT
#define PT_WF_TLAS_CONT 16u          // > 1: a lane keeps walking the TLAS within one wave iteration only while at least this many lanes walk with it
                                     // (0 / 4 / 8 / 16 / 24 / 32: 2,070 / 2,091 / 2,124 / 2,130 / 2,097 / 2,049 Mrays/s on the 200-instance scene)
#endif
#ifndef PT_WF_TLAS_QUORUM
#define PT_WF_TLAS_QUORUM 0u         // > 0: the TLAS-walk block runs when that many lanes want it, or when no lane is inside an instance (8: 1,920, 16: 1,892)
#endif
#ifndef PT_WF_TLAS_MIN_WAVES
#define PT_WF_TLAS_MIN_WAVES 6      // 80 VGPRs, no scratch (without the SLP vectorizer); 5 waves: -3 %
#endif
// WAVES = 1: one-wave workgroups, TLAS nodes fetched from HBM / L2 (round 2).
// WAVES > 1 (round 3, the default for TLAS trees below 65,536 nodes): the WAVES waves of a workgroup share ONE copy of the top of
// the TLAS in LDS -- the first PT_WF_TLAS_CACHE_NODES nodes of S.tlasBfs, the breadth-first renumbering PTSetScene makes of the
// reference's node array (same nodes, same children, same visiting order: only the indices differ, and nothing observable
// depends on them).  The TLAS walk is a chain of ~13 dependent 64-byte fetches per ray at a quarter of the lanes (57 % of this
// kernel's time when they come from memory); from LDS a step costs a ds_read_b128 x 4 instead of a vector-memory round trip
// and takes 52 of the ray's ~145 sixteen-byte lane requests off the texture-address path.  To pay for the 25.6 KB copy at the same
// 24 waves per CU the per-wave LDS shrinks: PT_WF_TLAS_BLAS_LDS_STACK (4) CWBVH stack entries (deeper ones in the HBM slab, as
// before) and 16-bit TLAS stack entries.  Each wave still owns its own slot range and never waits for another after the
// one barrier that follows the copy.
template <bool STATS, uint32_t WAVES>
__global__ __launch_bounds__(64 * WAVES, PT_WF_TLAS_MIN_WAVES) void pt_wf_trace_refill_tlas(DScene S, PTWfBuffers B, uint32_t iteration)
{
    constexpr bool kCached = WAVES > 1u;
    constexpr uint32_t kBlasLds = kCached ? (uint32_t)PT_WF_TLAS_BLAS_LDS_STACK : (uint32_t)PT_WF_LDS_STACK;
    typedef typename std::conditional<kCached, uint16_t, uint32_t>::type tentry_t;
    __shared__ uint2 s_stack[WAVES][kBlasLds][64];
    if (tail_cut(B, iteration)) return;
    __shared__ tentry_t s_tstack[WAVES][PT_WF_TLAS_LDS_STACK][64];
    __shared__ uint32_t s_xchg[WAVES][64];
    __shared__ uint32_t s_gw[WAVES];
    __shared__ float4 s_tcache[kCached ? PT_WF_TLAS_CACHE_NODES * 4u : 1u];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = kCached ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;     // wave-uniform: LDS bases stay scalar
    const uint32_t gw = blockIdx.x * WAVES + wv;
    // WAVES = 1: the wave scans its own PT_WF_RANGE slots.  WAVES > 1: the workgroup's waves SHARE its WAVES x PT_WF_RANGE slots, in
    // chunks of PT_WF_TLAS_CHUNK slots handed out by an LDS counter -- a wave whose chunk is scanned takes the next one while its
    // rays are still in flight, so the waves of a workgroup run out of work together (with a fixed range per wave the workgroup
    // held its LDS and wave slots until its SLOWEST range was done: measured 2.9x slower, 8 waves without the LDS copy).
    constexpr uint32_t kChunk = kCached ? PT_WF_TLAS_CHUNK : PT_WF_RANGE;
    __shared__ uint32_t s_next;
    const uint32_t wgBase = blockIdx.x * WAVES * PT_WF_RANGE;
    uint32_t slotBase = gw * PT_WF_RANGE;
    pt_lds_u32 xchg = (pt_lds_u32)&s_xchg[wv][0];
    typedef typename std::conditional<kCached, pt_lds_u16, pt_lds_u32>::type tlds_t;
    tlds_t tlds = (tlds_t)&s_tstack[wv][0][lane];
    if (lane == 0u) s_gw[wv] = gw;
    const uint32_t nItems = 3u * kChunk;
    const float* T = S.tlasBfs;
    uint32_t cachedNodes = 0u;
    if (kCached) {
        cachedNodes = S.tlasNodeCount < PT_WF_TLAS_CACHE_NODES ? S.tlasNodeCount : PT_WF_TLAS_CACHE_NODES;
        for (uint32_t i = threadIdx.x; i < cachedNodes * 4u; i += 64u * WAVES) s_tcache[i] = ((const float4*)T)[i];
        if (threadIdx.x == 0u) s_next = 0u;
        __syncthreads();
    }

    Counters cn = {};
    TravStackT<kBlasLds, true> st;
    st.lds = PT_LDS_U2(&s_stack[wv][0][lane]);
    st.stride = 64u;
    st.gbase = B.stackSpill;
    st.gwave = PT_LDS_WORD(s_gw[wv]);
    RayState rs;
    rs.sp = 0u; rs.anyHit = false; rs.overflow = false;
    bool have = false, inBlas = false, needPop = false, hitFound = false;
    uint32_t mySlot = 0u, myKind = 0u, cursor = kCached ? nItems : 0u;       // WAVES > 1: no chunk yet
    bool more = kCached;                                                      // chunks may remain (WAVES > 1)
    v3 O = mk3(0.0f), rD = mk3(0.0f);
    uint32_t nodeIndex = 0u, tsp = 0u, nextInst = 0u, instLeft = 0u;
    uint32_t nodeOffset = 0u, triOffset = 0u, attrOffset = 0u, instIndex = 0u;

    // The empty asm statements keep the LDS access and the HBM-slab access of an entry in their own branches.  Without them the
    // compiler sinks the two into ONE access through a selected pointer -- a FLAT load / store (round 2's kernel had two of each in
    // its walk loop): a flat access counts against both the vector-memory and the LDS counter, and one that resolves to LDS
    // still travels the texture-address path.
    auto tpush = [&](uint32_t v) {
        if (tsp < PT_WF_TLAS_LDS_STACK) { tlds[tsp * 64u] = (tentry_t)v; asm volatile("" ::: "memory"); }
        else if (tsp < PT_BVH_STACK_SIZE) B.tlasSpill[spill_row(PT_LDS_WORD(s_gw[wv])) * PT_BVH_STACK_SIZE + tsp] = v;
        tsp++;
    };
    auto tpop = [&]() -> uint32_t {
        --tsp;
        if (tsp < PT_WF_TLAS_LDS_STACK) { uint32_t v = tlds[tsp * 64u]; asm volatile("" : "+v"(v)); return v; }
        if (tsp < PT_BVH_STACK_SIZE) return B.tlasSpill[spill_row(PT_LDS_WORD(s_gw[wv])) * PT_BVH_STACK_SIZE + tsp];
        return 0u;
    };

    while (true) {
        uint32_t nIdle = (uint32_t)__popcll(__ballot(!have));
        while ((kCached || cursor < nItems) && (nIdle >= PT_WF_REFILL || nIdle == 64u)) {
            if (kCached && cursor >= nItems) {
                if (!more) break;
                uint32_t c = 0u;
                if (lane == 0u) c = atomicAdd(&s_next, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= WAVES * PT_WF_RANGE / kChunk || wgBase + c * kChunk >= B.numSlots) { more = false; break; }
                slotBase = wgBase + c * kChunk;
                cursor = 0u;
            }
            const unsigned long long idle = __ballot(!have);
            const uint32_t rankI = rank_below(idle);
            const uint32_t item = cursor + lane;
            const uint32_t kind = item / kChunk;
            const uint32_t slot = slotBase + (item & (kChunk - 1u));
            bool valid = item < nItems && slot < B.numSlots;
            if (valid) valid = ray_exists(B.flags[slot], kind);
            const unsigned long long V = __ballot(valid);
            const uint32_t nV = (uint32_t)__popcll(V);
            const uint32_t take = nIdle < nV ? nIdle : nV;
            const uint32_t rankV = rank_below(V);
            uint32_t consumed = 64u;
            if (take < nV) consumed = (uint32_t)__ffsll((long long)__ballot(valid && rankV == take)) - 1u;
            if (valid && rankV < take) xchg[rankV] = item;
            __builtin_amdgcn_wave_barrier();
            if (!have && rankI < take) {
                const uint32_t it = xchg[rankI];
                myKind = it / kChunk;
                mySlot = slotBase + (it & (kChunk - 1u));
                v3 d;
                fetch_ray(B, mySlot, myKind, O, d);
                if (myKind == 0u) cn.closestRays++; else cn.shadowRays++;
                if (pt_isnan(O.x) || pt_isnan(O.y) || pt_isnan(O.z) || pt_isnan(d.x) || pt_isnan(d.y) || pt_isnan(d.z)) {
                    if (myKind == 0u) store_miss(B, mySlot);              // NaN ray: certain miss (traverse_tlas)
                    else store_occlusion(B, myKind, mySlot, false);
                } else {
                    const v3 D = normalize3(d);                           // tlas.hlsl:238-240
                    rD = mk3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
                    rs.hit.t = PT_FAR_PLANE;
                    rs.anyHit = myKind != 0u;
                    nodeIndex = 0u; tsp = 0u; instLeft = 0u; needPop = false; inBlas = false;
                    have = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
            cursor += consumed;
            nIdle = (uint32_t)__popcll(__ballot(!have));
        }
        if (nIdle == 64u) break;
        const uint32_t stopAt = (kCached ? more : cursor < nItems) ? PT_WF_REFILL : 64u;
        do {
#ifdef PT_TLAS_DIAG
            // diagnostics builds (tools/experiments/tlas_diag.sh): wave-level event counts / clock ticks through the two pixel counters
            const bool dl0 = lane == (uint32_t)(__ffsll((long long)__ballot(true)) - 1);
            if (PT_TLAS_DIAG == 1 && dl0) cn.pixelsWritten++;                                   // outer wave iterations
            const unsigned long long dT0 = __builtin_readcyclecounter();
#endif
            // ---- a lane that is between instances (its instance epilogue is done) walks the TLAS to the next instance whose box it hits.
            //      Per ray of the 200-instance scene: 13.0 TLAS nodes + 5.9 instance entries + 9.6 BLAS nodes + 3.0 triangles = 37
            //      steps against 18.6 for the same geometry baked flat.  Measured with the -DPT_TLAS_DIAG builds (tools/experiments/
            //      tlas_diag.py): this block is 57 % of the loop's time; it runs in 45 % of the wave iterations, 11 inner steps per
            //      execution (until its slowest lane is inside an instance) at 15.5 of 64 lanes, 80.8 M wave-steps per pass against 16.2 M
            //      executions of the CWBVH step below (37.8 lanes).  Neither the chain's length (one fetch per instance entry instead
            //      of two, instByLeaf: no change) nor its scheduling (one TLAS step per wave iteration, a quorum for entering the block,
            //      triangle parking: all slower) is the lever; letting the last few walkers wait for company (PT_WF_TLAS_CONT) is
            //      worth 3 %.  What is left is the work itself: 19 TLAS steps per ray at a quarter of the lanes.
#ifndef PT_WF_TLAS_SPLIT_ENTRY
#define PT_WF_TLAS_SPLIT_ENTRY 1     // 1: the walk loop only visits TLAS nodes; lanes that reached a leaf (or still have instances of their leaf left) enter
                                     // their instance together in ONE block after it.  0 (round 2): the ~130-instruction entry path sits inside the walk loop and runs
                                     // in most of its ~11 steps per execution for one or two lanes each
#endif
            const bool walkT = have && !inBlas && (!PT_WF_TLAS_SPLIT_ENTRY || instLeft == 0u);
#if PT_WF_TLAS_QUORUM > 0
            const uint32_t nWalk = (uint32_t)__popcll(__ballot(walkT)), nIn = (uint32_t)__popcll(__ballot(have && inBlas));
            const bool runWalk = nWalk >= PT_WF_TLAS_QUORUM || nIn == 0u;
#else
            const bool runWalk = true;
#endif
#ifdef PT_TLAS_DIAG
            if (PT_TLAS_DIAG == 1 && dl0 && __ballot(walkT && runWalk) != 0ull) cn.pixelsRead++;      // executions of the TLAS-walk block
            if (PT_TLAS_DIAG == 3 && walkT) cn.pixelsRead++;                                        // lanes entering the walk block
#endif
            // enter the next instance of the current TLAS leaf (tlas.hlsl:129-147)
            // ONE fetch: the record PTSetScene laid out per TLAS index slot (worldToLocal, offsets, instance index) instead of
            // TLASData[TLASIndexOffset + k] -> instance record (two dependent fetches; same values)
            auto enter_instance = [&]() {
                const float4* ip = S.instByLeaf + (size_t)nextInst * 6;
                const float4 w0 = ip[0], w1 = ip[1], w2 = ip[2], w3 = ip[3], ints = ip[4];
                instIndex = pt_asuint(ip[5].x);
                nextInst++; instLeft--;
                if (STATS) cn.instanceVisits++;
                nodeOffset = pt_asuint(ints.x); triOffset = pt_asuint(ints.y); attrOffset = pt_asuint(ints.z);
                const v3 wd = xyz(f4_array(B, 2u * myKind)[2u * (size_t)mySlot + 1u]);           // the direction row of the lane's ray record
                const v4 lo4 = mul44c(w0, w1, w2, w3, v4{O.x, O.y, O.z, 1.0f});
                const v4 ld4 = mul44c(w0, w1, w2, w3, v4{wd.x, wd.y, wd.z, 0.0f});
                const float tKeep = rs.hit.t;
                ray_begin(rs, mk3(lo4.x, lo4.y, lo4.z), mk3(ld4.x, ld4.y, ld4.z), myKind != 0u);
                rs.hit.t = tKeep;
                hitFound = false;
                inBlas = true;
            };
            if (walkT && runWalk) {
                bool finished = false;
                for (uint32_t stepT = 0; stepT < PT_WF_TLAS_STEPS; ++stepT) {
                    // the lanes still walking are the wave's active lanes here; when only a few are left they wait for company (the
                    // lanes whose instance ends in this wave iteration) instead of stepping through the TLAS at 1-2 lanes per instruction
                    if (PT_WF_TLAS_CONT > 1u && stepT > 0u && (uint32_t)__popcll(__ballot(true)) < PT_WF_TLAS_CONT) break;
#ifdef PT_TLAS_DIAG
                    if (PT_TLAS_DIAG == 2) { cn.pixelsRead++; if (lane == (uint32_t)(__ffsll((long long)__ballot(true)) - 1)) cn.pixelsWritten++; }   // lane-steps / wave-steps inside the walk block
#endif
                    if (!PT_WF_TLAS_SPLIT_ENTRY && instLeft > 0u) { enter_instance(); break; }
                    if (needPop) {
                        if (tsp == 0u) { finished = true; break; }
                        nodeIndex = tpop();
                        needPop = false;
                    }
                    // visit TLAS node nodeIndex (tlas.hlsl:246-331)
                    float4 a, b, c, e;
                    if (kCached && nodeIndex < cachedNodes) {
                        const float4* cp = &s_tcache[nodeIndex * 4u];
                        a = cp[0]; b = cp[1]; c = cp[2]; e = cp[3];
                        asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(e.x));        // ds_read_b128 here, global_load below: never one flat load (see tpush)
                    } else {
                        const float4* np = (const float4*)(T + (size_t)nodeIndex * 16u);
                        a = np[0]; b = np[1]; c = np[2]; e = np[3];
                        // all four rows requested before the first use: left alone, the compiler fetches the instance count first, waits, and
                        // requests the boxes (or the leaf's first index) under the branch on it -- two dependent round trips per TLAS step
                        asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(c.w), "+v"(e.x), "+v"(e.w));
                    }
                    if (STATS) cn.tlasNodeVisits++;
                    const uint32_t instanceCount = pt_asuint(c.w);
                    if (instanceCount == 0u) {
                        uint32_t left = pt_asuint(a.w), right = pt_asuint(b.w);
                        const v3 t1a = (mk3(a.x, a.y, a.z) - O) * rD, t2a = (mk3(b.x, b.y, b.z) - O) * rD;
                        const float tmina = pt_max(pt_max(pt_max(pt_min(t1a.x, t2a.x), pt_min(t1a.y, t2a.y)), pt_min(t1a.z, t2a.z)), 0.0f);
                        const float tmaxa = pt_min(pt_min(pt_min(pt_max(t1a.x, t2a.x), pt_max(t1a.y, t2a.y)), pt_max(t1a.z, t2a.z)), rs.hit.t);
                        float dist1 = tmina > tmaxa ? PT_FAR_PLANE : tmina;
                        const v3 t1b = (mk3(c.x, c.y, c.z) - O) * rD, t2b = (mk3(e.x, e.y, e.z) - O) * rD;
                        const float tminb = pt_max(pt_max(pt_max(pt_min(t1b.x, t2b.x), pt_min(t1b.y, t2b.y)), pt_min(t1b.z, t2b.z)), 0.0f);
                        const float tmaxb = pt_min(pt_min(pt_min(pt_max(t1b.x, t2b.x), pt_max(t1b.y, t2b.y)), pt_max(t1b.z, t2b.z)), rs.hit.t);
                        float dist2 = tminb > tmaxb ? PT_FAR_PLANE : tminb;
                        if (dist1 > dist2) {
                            const float h = dist1; dist1 = dist2; dist2 = h;
                            const uint32_t t = left; left = right; right = t;
                        }
                        if (dist1 == PT_FAR_PLANE) needPop = true;
                        else {
                            nodeIndex = left;
                            if (dist2 != PT_FAR_PLANE) tpush(right);
                        }
                    } else {
                        nextInst = pt_asuint(e.w);
                        instLeft = instanceCount;
                        needPop = true;                                    // after the leaf's instances
                        if (PT_WF_TLAS_SPLIT_ENTRY) break;                 // to the entry block below
                    }
                }
                if (finished) {
                    if (myKind == 0u) { if (!(rs.hit.t < PT_FAR_PLANE)) store_miss(B, mySlot); }
                    else store_occlusion(B, myKind, mySlot, false);
                    have = false;
                }
            }
            if (PT_WF_TLAS_SPLIT_ENTRY && have && !inBlas && instLeft > 0u) enter_instance();
#ifdef PT_TLAS_DIAG
            const unsigned long long dT1 = __builtin_readcyclecounter();
            if (PT_TLAS_DIAG == 3 && have && inBlas) cn.pixelsWritten++;                             // lanes in the CWBVH step
#endif
            // ---- the hot step: one CWBVH iteration inside the current instance
            if (have && inBlas) {
                const float tBefore = rs.hit.t;
                bool blasDone = false;
                {
                    // ray_step with instance offsets and the HAS_TLAS triangle test (tlas.hlsl:149-206)
                    if (rs.ng.y > 0x00FFFFFFu) {
                        cwbvh_visit_node<STATS>(S.nodes, nodeOffset, rs.o, rs.invDir, rs.octinv4, rs.hit.t, rs.ng, rs.tg, rs.sp, rs.overflow, st, cn);
                    } else {
                        rs.tg = rs.ng;
                        rs.ng = make_uint2(0u, 0u);
                    }
                    while (rs.tg.y != 0u) {
                        const uint32_t triangleIndex = 31u - (uint32_t)__clz((int)rs.tg.y);
                        const uint32_t triAddr = rs.tg.x + triangleIndex * 3u;
                        hitFound = intersect_triangle_inst(S.tris, triOffset + triAddr, attrOffset, rs.o, rs.d, rs.hit) | hitFound;
                        if (STATS) cn.triTests++;
                        rs.tg.y -= 1u << triangleIndex;
                        if (rs.anyHit && hitFound) { rs.tg.y = 0u; rs.ng.y = 0u; rs.sp = 0u; }
                    }
                    if (rs.ng.y <= 0x00FFFFFFu) {
                        if (rs.sp > 0u) { --rs.sp; rs.ng = stack_pop(st, rs.sp); }
                        else blasDone = true;
                    }
                }
                if (myKind == 0u && rs.hit.t < tBefore) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                if (blasDone) {
                    if (STATS && rs.overflow) cn.overflows++;
                    inBlas = false;
                    if (rs.anyHit) {
                        if (hitFound) { store_occlusion(B, myKind, mySlot, true); have = false; }       // stopNow (tlas.hlsl:196-200)
                    } else if (hitFound) {
                        // tlas.hlsl:208-229: world-space hit position and distance of the instance that improved the hit
                        if (STATS) cn.attrFetches++;
                        const float4* ip = S.instances + (size_t)instIndex * 9;
                        const float4 l0 = ip[0], l1 = ip[1], l2 = ip[2], l3 = ip[3];
                        const v3 lp = rs.o + rs.hit.t * rs.d;
                        const v4 wp = mul44c(l0, l1, l2, l3, v4{lp.x, lp.y, lp.z, 1.0f});
                        const v3 pos = mk3(wp.x, wp.y, wp.z);
                        rs.hit.t = length3(pos - O);
                        ((float*)&f4_array(B, PT_F4_HIT)[mySlot])[0] = rs.hit.t;
                        f4_array(B, PT_F4_HIT2)[mySlot] = make_float4(pos.x, pos.y, pos.z, pt_asfloat(instIndex));
                    }
                }
            }
#ifdef PT_TLAS_DIAG
            if (PT_TLAS_DIAG == 4 && dl0) { const unsigned long long dT2 = __builtin_readcyclecounter(); cn.pixelsRead += (uint32_t)((dT1 - dT0) >> 4); cn.pixelsWritten += (uint32_t)((dT2 - dT1) >> 4); }
#endif
            nIdle = (uint32_t)__popcll(__ballot(!have));
        } while (nIdle < stopAt);
    }
    __builtin_amdgcn_wave_barrier();
    flush_counters<STATS>(cn, B.statRows, (B.numSlots >> 6) + *PT_LDS_WORD(s_gw[wv]), lane);
}

// ------------------------------------------------------------------------------------------
// persistent trace (schedule 1): the refill kernel above, made persistent.  The grid is exactly the number of
// waves the chip holds; a wave that has scanned its 64-slot chunk pulls the NEXT chunk index from a device counter
// while its remaining rays are still in flight, so lanes are refilled continuously and nothing drains until the
// very end of the launch.  Chunk counters are sharded 8 ways (one per XCD under round-robin workgroup placement,
// blockIdx % 8) on separate 128-byte lines; a wave takes from its own shard first and steals from the others when
// it runs dry: ~numSlots/64 returning atomics per launch in total, i.e. one per 192 candidate rays.
// Which wave traces which ray depends on timing; what is computed for a ray does not.
// ------------------------------------------------------------------------------------------
PT_DEV bool next_chunk(uint32_t* heads, uint32_t numChunks, uint32_t shard, uint32_t& chunk)
{
    // lane 0 asks; everybody gets the answer.  Shard sh owns the chunks c with c % 8 == sh.
    uint32_t result = 0xFFFFFFFFu;
    if ((threadIdx.x & 63u) == 0u) {
        for (uint32_t t = 0; t < PT_WF_SHARDS; ++t) {
            const uint32_t sh = (shard + t) & (PT_WF_SHARDS - 1u);
            const uint32_t inShard = (numChunks + PT_WF_SHARDS - 1u - sh) / PT_WF_SHARDS;
            if (inShard == 0u) continue;
            const uint32_t k = atomicAdd(&heads[sh * 32u], 1u);
            if (k < inShard) { result = k * PT_WF_SHARDS + sh; break; }
        }
    }
    result = __shfl(result, 0, 64);
    chunk = result;
    return result != 0xFFFFFFFFu;
}

template <bool STATS>
__global__ __launch_bounds__(64, PT_WF_TRACE_MIN_WAVES) void pt_wf_trace_persist(DScene S, PTWfBuffers B, uint32_t iteration)
{
    __shared__ uint2 s_stack[PT_WF_LDS_STACK][64];
    if (tail_cut(B, iteration)) return;
    __shared__ uint32_t s_xchg[64];
    const uint32_t lane = threadIdx.x;
    const uint32_t numChunks = (B.numSlots + PT_WF_CHUNK - 1u) / PT_WF_CHUNK;
    const uint32_t shard = blockIdx.x & (PT_WF_SHARDS - 1u);
    const uint32_t nItems = 3u * PT_WF_CHUNK;
    pt_lds_u32 xchg = (pt_lds_u32)&s_xchg[0];

    Counters cn = {};
    TravStackT<PT_WF_LDS_STACK, true> st;
    st.lds = PT_LDS_U2(&s_stack[0][lane]);
    st.stride = 64u;
    st.gbase = B.stackSpill;
    __shared__ uint32_t s_gw;
    if (lane == 0u) s_gw = blockIdx.x;
    st.gwave = PT_LDS_WORD(s_gw);                                                 // grid <= numSlots / 64 waves
    RayState rs;
    rs.sp = 0u; rs.anyHit = false; rs.overflow = false;
    bool have = false;
    uint32_t mySlot = 0u, myKind = 0u;
    uint32_t cursor = nItems, slotBase = 0u;        // no chunk yet
    bool more = true;                                // chunks may remain

    while (true) {
        uint32_t nIdle = (uint32_t)__popcll(__ballot(!have));
        // ---- refill from the current chunk, pulling new chunks as needed
        while (nIdle >= PT_WF_REFILL || nIdle == 64u) {
            if (cursor >= nItems) {
                uint32_t chunk;
                if (!more || !next_chunk(B.chunkHeads, numChunks, shard, chunk)) { more = false; break; }
                slotBase = chunk * PT_WF_CHUNK;
                cursor = 0u;
            }
            const unsigned long long idle = __ballot(!have);
            const uint32_t item = cursor + lane;
            const uint32_t kind = item / PT_WF_CHUNK;
            const uint32_t slot = slotBase + (item & (PT_WF_CHUNK - 1u));
            bool valid = item < nItems && slot < B.numSlots;
            if (valid) valid = ray_exists(B.flags[slot], kind);
            const unsigned long long V = __ballot(valid);
            const uint32_t nV = (uint32_t)__popcll(V);
            const uint32_t take = nIdle < nV ? nIdle : nV;
            const uint32_t rankV = rank_below(V);
            uint32_t consumed = 64u;
            if (take < nV) consumed = (uint32_t)__ffsll((long long)__ballot(valid && rankV == take)) - 1u;
            if (valid && rankV < take) xchg[rankV] = item;
            __builtin_amdgcn_wave_barrier();
            const uint32_t rankI = rank_below(idle);
            if (!have && rankI < take) {
                const uint32_t it = xchg[rankI];
                myKind = it / PT_WF_CHUNK;
                mySlot = slotBase + (it & (PT_WF_CHUNK - 1u));
                v3 o, d;
                fetch_ray(B, mySlot, myKind, o, d);
                if (myKind == 0u) cn.closestRays++; else cn.shadowRays++;
                if (ray_begin(rs, o, d, myKind != 0u)) {
                    if (myKind == 0u) store_miss(B, mySlot);
                    else store_occlusion(B, myKind, mySlot, false);
                } else have = true;
            }
            __builtin_amdgcn_wave_barrier();
            cursor += consumed;
            nIdle = (uint32_t)__popcll(__ballot(!have));
        }
        if (nIdle == 64u) break;                                      // no chunk left and nothing in flight
        const uint32_t stopAt = more ? PT_WF_REFILL : 64u;
        do {
            if (have) {
                const float tBefore = rs.hit.t;
                const bool fin = PT_WF_STEP<STATS>(S, rs, st, cn);
                if (myKind == 0u && rs.hit.t < tBefore) f4_array(B, PT_F4_HIT)[mySlot] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                if (fin) {
                    if (myKind == 0u) { if (!(rs.hit.t < PT_FAR_PLANE)) store_miss(B, mySlot); }
                    else store_occlusion(B, myKind, mySlot, rs.hit.t < PT_FAR_PLANE);
                    have = false;
                }
            }
            nIdle = (uint32_t)__popcll(__ballot(!have));
        } while (nIdle < stopAt);
    }
    flush_counters<STATS>(cn, B.statRows, (B.numSlots >> 6) + blockIdx.x, lane);
}

// ------------------------------------------------------------------------------------------
// shade: one lane per slot
// ------------------------------------------------------------------------------------------
#ifndef PT_WF_SHADE_MIN_WAVES
#define PT_WF_SHADE_MIN_WAVES 3
#endif

// The path state of one slot as the flags word f describes it (state != DONE).  Used by the cleanup kernel; the shade kernel
// requests the same words all at once (shade_slot).
PT_DEV void load_path(const PTWfBuffers& B, uint32_t slot, uint32_t f, PathRegs& r)
{
    r.state = fl_state(f);
    r.hasPending = fl_pending(f);
    r.env.valid = fl_env(f);
    r.light.valid = fl_light(f);
    r.green = (f >> 6) & 1u;
    r.sampleIdx = (f >> 7) & 0xFFFu;
    r.depth = f >> 19;
    r.rng = B.rng[slot];
    float4 q;
    q = B.ray[0][2u * slot]; r.ro = xyz(q); r.scatterPdf = q.w;
    q = B.ray[0][2u * slot + 1u]; r.rd = xyz(q); r.maxRoughness = q.w;
    r.radiance = xyz(B.rad[slot]);
    r.throughput = xyz(B.thr[slot]);
    r.color = xyz(B.color[slot]);
    r.env.dir = mk3(0.0f); r.light.dir = mk3(0.0f); r.neeOrigin = mk3(0.0f);
    r.env.contribution = mk3(0.0f); r.light.contribution = mk3(0.0f); r.pendThroughput = mk3(0.0f);
    if (r.hasPending) {
        r.env.contribution = xyz(B.envC[slot]);
        r.light.contribution = xyz(B.lightC[slot]);
        r.pendThroughput = xyz(B.pthr[slot]);
    }
}

// Everything the shade step does for one slot (flags word f already read, state != DONE).
// Every word of the slot's state is requested up front in ONE batch, whether the flags say it is meaningful or not (the pending
// NEE terms, the two occlusion bytes, the hit record): requested where they are used they formed a chain of four dependent
// round trips (state -> pending terms -> occlusion bytes -> hit record, ~870 cycles each at the shade kernel's 4 waves/SIMD)
// in front of the attribute / material / texture chain.  A value the flags do not cover is read and ignored.
// `slot` indexes the path-state arrays; `pixelSlot` is the pixel the state belongs to (the same number in schedules 1-3, where
// every pixel has its own slot; a context's current pixel in schedule 4, which also wants the finished pixel's sample sum in
// B.pixsum[pixelSlot]: PIXSUM).
template <bool STATS, bool PIXSUM = false>
PT_DEV bool shade_slot(const DScene& S, const PTFrameParams& P, const PTTileMap& tm, const PTWfBuffers& B, uint32_t slot, uint32_t pixelSlot,
                       uint32_t f, Counters& cn)
{
    uint32_t px, py, pass;
    pt_slot_to_pixel(tm, pixel_slot_of(B, pixelSlot, pass), px, py);
    uint32_t rng = B.rng[slot];
    float4 qro = B.ray[0][2u * slot], qrd = B.ray[0][2u * slot + 1u], qrad = B.rad[slot], qthr = B.thr[slot], qcol = B.color[slot];
    float4 qenvC = B.envC[slot], qlightC = B.lightC[slot], qpthr = B.pthr[slot], qhit = B.hit[slot];
    uint32_t o0 = B.occl[slot], o1 = B.occl[(size_t)B.numSlots + slot];
    float4 qhit2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (S.hasTlas) qhit2 = B.hit2[slot];
    // pinned: left alone, the compiler sinks each load into the branch that uses it again
    asm volatile("" : "+v"(rng), "+v"(qro.x), "+v"(qrd.x), "+v"(qrad.x), "+v"(qthr.x), "+v"(qcol.x), "+v"(qenvC.x), "+v"(qlightC.x), "+v"(qpthr.x),
                      "+v"(qhit.x), "+v"(o0), "+v"(o1), "+v"(qhit2.x));
    PathRegs r;
    r.state = fl_state(f);
    r.hasPending = fl_pending(f);
    r.env.valid = fl_env(f);
    r.light.valid = fl_light(f);
    r.green = (f >> 6) & 1u;
    r.sampleIdx = (f >> 7) & 0xFFFu;
    r.depth = f >> 19;
    r.rng = rng;
    r.ro = xyz(qro); r.scatterPdf = qro.w;
    r.rd = xyz(qrd); r.maxRoughness = qrd.w;
    r.radiance = xyz(qrad);
    r.throughput = xyz(qthr);
    r.color = xyz(qcol);
    r.env.dir = mk3(0.0f); r.light.dir = mk3(0.0f); r.neeOrigin = mk3(0.0f);
    r.env.contribution = r.hasPending ? xyz(qenvC) : mk3(0.0f);
    r.light.contribution = r.hasPending ? xyz(qlightC) : mk3(0.0f);
    r.pendThroughput = r.hasPending ? xyz(qpthr) : mk3(0.0f);
    const bool occEnv = r.hasPending && o0 != 0u, occLight = r.hasPending && o1 != 0u;
    HitRecord ch;
    ch.h.t = PT_FAR_PLANE; ch.h.u = 0.0f; ch.h.v = 0.0f; ch.h.triIndex = 0u;
    ch.pos = mk3(0.0f); ch.inst = 0u;
    if (r.state == PS_TRACE) {
        ch.h.t = qhit.x; ch.h.u = qhit.y; ch.h.v = qhit.z; ch.h.triIndex = pt_asuint(qhit.w);
        if (S.hasTlas) { ch.pos = xyz(qhit2); ch.inst = pt_asuint(qhit2.w); }
    }
    // the NEE rays of the bounce go to their arrays as soon as they are final (NeeSink): 15 registers free while the BSDF is sampled
    struct Sink {
        const PTWfBuffers& B; uint32_t slot;
        PT_DEV void operator()(PathRegs& q) const {
            uint32_t s2 = slot;
            asm volatile("" : "+v"(s2));
            B.ray[1][2u * s2] = f4(q.neeOrigin, 0.0f);
            B.ray[1][2u * s2 + 1u] = f4(q.env.dir, 0.0f);
            B.ray[2][2u * s2] = f4(q.neeOrigin, 0.0f);
            B.ray[2][2u * s2 + 1u] = f4(q.light.dir, 0.0f);
            B.envC[s2] = f4(q.env.contribution, 0.0f);
            B.lightC[s2] = f4(q.light.contribution, 0.0f);
            B.pthr[s2] = f4(q.pendThroughput, 0.0f);
        }
    };
    path_step<STATS, false, Sink>(S, P, r, ch, occEnv, occLight, px, py, py * P.OutputWidth + px, nullptr, nullptr, cn, Sink{B, slot});
    // the store addresses are formed HERE from a slot the compiler cannot connect with the one the loads used: otherwise the
    // thirteen 64-bit load addresses stay in registers across the whole step to be reused by these stores
    uint32_t storeSlot = slot;
    asm volatile("" : "+v"(storeSlot));
    store_path(B, storeSlot, r, false);                             // the NEE arrays were written by the sink
    if (PIXSUM && r.state == PS_DONE) {
        uint32_t ps = pixelSlot;
        asm volatile("" : "+v"(ps));
        B.pixsum[ps] = f4(r.color, 0.0f);
    }
    return r.state != PS_DONE;
}

// one lane per slot, in screen order.  (Walking a compacted or class-sorted list of live slots instead was built twice and is
// slower -- round 2: shade VALU -41 %, throughput -20 % -- because path state is then read and written through gathers;
// DESIGN.md 5.1.)
#ifndef PT_WF_SHADE_BLOCK
#define PT_WF_SHADE_BLOCK 64u          // one wave per workgroup: a finished wave frees its 128 VGPRs at once (256 -> 64: +4 %)
#endif
template <bool STATS>
__global__ __launch_bounds__(PT_WF_SHADE_BLOCK, PT_WF_SHADE_MIN_WAVES) void pt_wf_shade(DScene S, PTFrameParams P, PTTileMap tm, PTWfBuffers B,
                                                                         uint32_t iteration)
{
    const uint32_t vb = blockIdx.x;
    const uint32_t slot = vb * PT_WF_SHADE_BLOCK + threadIdx.x;
    if (vb == 0u && threadIdx.x < PT_WF_SHARDS) B.chunkHeads[threadIdx.x * 32u] = 0u;   // for the next trace launch (schedule 3)
    if (tail_cut(B, iteration)) return;                               // (aliveFlags[iteration] stays 0: the later launches return too)
    const uint32_t f = B.flags[slot];
    Counters cn = {};
    bool alive = false;
#ifdef PT_SHADE_DIAG
    // diagnostics build: wave-level executions of the shade body and lanes active in them (reported in the TLAS counters)
    if (__any(fl_state(f) != PS_DONE) && (threadIdx.x & 63u) == 0u) cn.tlasNodeVisits++;
    if (fl_state(f) != PS_DONE) cn.instanceVisits++;
#endif
    if (fl_state(f) != PS_DONE) alive = shade_slot<STATS>(S, P, tm, B, slot, slot, f, cn);
    {
        const uint32_t nAlive = (uint32_t)__popcll(__ballot(alive));
        if (B.tailCut != 0u && nAlive && (threadIdx.x & 63u) == 0u)
            atomicAdd(&B.aliveFlags[iteration * PT_WF_ALIVE_SHARDS + (blockIdx.x & (PT_WF_ALIVE_SHARDS - 1u))], nAlive);   // one atomic per wave that still has paths
    }
    flush_counters<STATS>(cn, B.statRows, vb * (PT_WF_SHADE_BLOCK / 64u) + (threadIdx.x >> 6), threadIdx.x & 63u);
}

// ------------------------------------------------------------------------------------------
// Schedule 4: the FUSED PERSISTENT wavefront.  One launch per pass; no launch boundary anywhere inside it.
//
// Paths of different pixels never exchange anything (PathTracer.compute:60), so nothing in a pass needs a grid-wide barrier: the
// 52 x 3 launches of schedule 1 only exist because a launch is the unit in which that schedule alternates between the lean trace
// code and the register-hungry shading code.  Here a persistent wave does the alternating itself, over path CONTEXTS it owns:
//
//     a wave owns PT_WF_FUSED_GROUPS x 64 contexts (path state in HBM, indexed by context, not by pixel);
//     loop {  refill:  every context whose pixel has finished takes the next pixel of the frame from a device counter
//                      (one atomic per wave and round; the camera ray of its first sample is written);
//             trace:   the refill scheduler of pt_wf_trace_refill over the contexts' 3 x 64 x GROUPS candidate rays -- 64 resumable
//                      traversals in flight, idle lanes re-filled by ballot / rank compaction -- until the candidates are used up
//                      and <= PT_WF_SUSPEND rays are left; those are parked as records in LDS and resumed in the next round;
//             shade:   path_step() for every context whose rays have all returned; a pixel that finished its last sample
//                      leaves its sample sum in the per-pixel array the resolve kernel reads.  }
//
// Every context is busy until the frame runs out of pixels: there is no per-launch ramp-up and drain, no tail launch, no cleanup
// kernel, and ONE pass in flight fills the machine -- a host that synchronises after every pass (the reference presents every
// pass, PathTracer.cs:251-272) gets the pipelined throughput, with one path-state set of ~1 M contexts instead of twelve sets
// of one slot per pixel.  A wave only ever touches the state of its own contexts, so ordering is program order within the
// wave (workgroup-scope fences between phases).  Which context renders which pixel depends on timing; what is computed for a
// pixel does not: same device functions, same per-path order as every other schedule -> frames and counters are bit-identical.
// ------------------------------------------------------------------------------------------
#ifndef PT_WF_FUSED_WAVES
#define PT_WF_FUSED_WAVES 4         // waves per SIMD the fused kernel is compiled for (128 VGPRs)
#endif

#define PT_WF_NO_PIXEL 0xFFFFFFFFu

// add a phase's counters to the wave's LDS totals (one wave per workgroup; 16 words, PTStats order)
template <bool STATS>
PT_DEV void counters_to_lds(const Counters& cn, pt_lds_u32 tot, uint32_t lane)
{
    uint32_t vals[PT_NUM_COUNTERS];
    counters_to_array(cn, vals);
#pragma unroll
    for (int i = 0; i < PT_NUM_COUNTERS; ++i) {
        if (!STATS && (i >= 3 && i <= 9)) continue;
        if (!STATS && i >= 12) continue;
        if (i == 12) {
            const uint32_t m = wave_max_u32(vals[i]);
            if (lane == 0u && m > tot[i]) tot[i] = m;
        } else {
            const uint32_t v = wave_sum_u32(vals[i]);
            if (lane == 0u && v) tot[i] += v;
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(64, PT_WF_FUSED_WAVES) void pt_wf_fused(DScene S, PTFrameParams P, PTBatch batch, PTTileMap tm, PTWfBuffers B)
{
    constexpr uint32_t K = PT_WF_FUSED_GROUPS;
    constexpr uint32_t kSusp = PT_WF_SUSPEND ? PT_WF_SUSPEND : 1u;
    static_assert((K & (K - 1u)) == 0u && K >= 1u && K <= 16u, "context groups per wave: a power of two");
    __shared__ uint2 s_stack[PT_WF_LDS_STACK][64];
    __shared__ uint32_t s_xchg[64];
    __shared__ uint32_t s_pix[K][64];           // the pixel (slot of the frame's enumeration, pt_slot_to_pixel) a context renders, or PT_WF_NO_PIXEL
    __shared__ uint32_t s_pend[2][K][2];        // [buffer][group][low / high word]: contexts that have a ray parked in s_susp
    __shared__ uint4 s_susp[kSusp][PT_WF_SUSP_ROWS];
    __shared__ uint32_t s_cnt[PT_NUM_COUNTERS];
    __shared__ uint32_t s_gw;
    const uint32_t lane = threadIdx.x;
    const uint32_t ctxBase = blockIdx.x * (K * 64u);                 // context = ctxBase + group * 64 + lane
    pt_lds_u32 xchg = (pt_lds_u32)&s_xchg[0];
    pt_lds_u32 pix = (pt_lds_u32)&s_pix[0][0];
    pt_lds_u32 cnt = (pt_lds_u32)&s_cnt[0];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        s_pix[k][lane] = PT_WF_NO_PIXEL;
        B.flags[ctxBase + k * 64u + lane] = PS_DONE;                 // a context without a pixel has no rays (the trace scan reads the flag words)
    }
    if (lane < PT_NUM_COUNTERS) s_cnt[lane] = 0u;
    if (lane < 4u * K) (&s_pend[0][0][0])[lane] = 0u;
    if (lane == 0u) s_gw = blockIdx.x;
    __builtin_amdgcn_wave_barrier();
    uint32_t nSusp = 0u, pb = 0u;
    bool more = true;
#ifdef PT_FUSED_DIAG
    // diagnostics build: wave time per phase (s_memtime ticks), rounds, reported through the shading counters of the FULL-stats variant:
    // attrFetches = refill ticks / 1024, materialFetches = trace ticks / 1024, lightFetches = shade ticks / 1024, texDescFetches = rounds
    unsigned long long tRefill = 0ull, tTrace = 0ull, tShade = 0ull, tMark = __builtin_readcyclecounter();
    uint32_t rounds = 0u, dIter = 0u, dScan = 0u, dTri = 0u;
    unsigned long long tScan = 0ull;
#define PT_DIAG_INC(x) do { x++; } while (0)
#define PT_DIAG_LAP(acc) do { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - tMark; tMark = now_; } while (0)
#else
#define PT_DIAG_LAP(acc) do { } while (0)
#define PT_DIAG_INC(x) do { } while (0)
#endif

    while (true) {
        // ---- (0) refill: a context without a pixel takes the next pixel of the frame and starts its first sample
        bool any = false;
#pragma unroll 1
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t ctx = ctxBase + k * 64u + lane;
            uint32_t mine = pix[k * 64u + lane];
            const unsigned long long E = __ballot(mine == PT_WF_NO_PIXEL);
            if (more && E != 0ull) {                                 // wave-uniform
                const uint32_t n = (uint32_t)__popcll(E);
                uint32_t base = 0u;
                if (lane == 0u) base = atomicAdd(&B.chunkHeads[0], n);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base + n >= B.numSlots) more = false;
                Counters cn = {};
                if (mine == PT_WF_NO_PIXEL) {
                    const uint32_t slot = base + rank_below(E);
                    uint32_t px, py, pass;
                    if (slot < B.numSlots && pt_slot_to_pixel(tm, pixel_slot_of(B, slot, pass), px, py)) {      // (slots of partially covered edge tiles have no pixel)
                        PathRegs r;
                        uint32_t seedRoot, currentSample;
                        pt_batch_pick(batch, pass, seedRoot, currentSample);
                        path_init(P, seedRoot, currentSample, px, py, py * P.OutputWidth + px, r, cn);
                        store_path(B, ctx, r, false);
                        mine = slot;
                        pix[k * 64u + lane] = slot;
                    }
                }
                counters_to_lds<false>(cn, cnt, lane);
            }
            any = any || __ballot(mine != PT_WF_NO_PIXEL) != 0ull;
        }
        if (!any) { if (!more) break; else continue; }              // nothing to do: out of pixels -> done; else an edge tile gave none, pull again
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        PT_DIAG_LAP(tRefill);

        // ---- (1) trace: resume what was parked, then scan the contexts' flag words kind-major and keep 64 traversals in flight
        {
            Counters cn = {};
            TravStackT<PT_WF_LDS_STACK, true> st;
            st.lds = PT_LDS_U2(&s_stack[0][lane]);
            st.stride = 64u;
            st.gbase = B.stackSpill;
            st.gwave = PT_LDS_WORD(s_gw);
            RayState rs;
            rs.sp = 0u; rs.anyHit = false; rs.overflow = false;
            rs.o = mk3(0.0f); rs.d = mk3(0.0f); rs.invDir = mk3(0.0f); rs.octinv4 = 0u;
            rs.ng = make_uint2(0u, 0u); rs.tg = make_uint2(0u, 0u);
            rs.hit.t = PT_FAR_PLANE; rs.hit.u = 0.0f; rs.hit.v = 0.0f; rs.hit.triIndex = 0u;
            bool have = false;
            uint32_t myRef = 0u;                                     // kind << 30 | context index within the wave (group * 64 + lane)
            const uint32_t po = pb, pn = pb ^ 1u;                    // pend masks: previous round (read), this round (written)
            if (lane < 2u * K) (&s_pend[pn][0][0])[lane] = 0u;
            // parked rays first: lane i takes record i (all lanes are idle here)
            if (PT_WF_SUSPEND > 0u && lane < nSusp) {
                const uint4 a = s_susp[lane][0], b = s_susp[lane][1];
                myRef = a.x;
                const uint32_t kind = myRef >> 30;
                const uint32_t ctx = ctxBase + (myRef & 0x3FFFFFFFu);
                v3 o, d;
                fetch_ray(B, ctx, kind, o, d);
                ray_begin(rs, o, d, kind != 0u);                     // same invDir / octinv4 as when the ray started
                rs.sp = a.y & 0xFFu;
                rs.overflow = (a.y & 0x100u) != 0u;
                rs.ng = make_uint2(a.z, a.w);
                rs.tg = make_uint2(b.x, b.y);
                rs.hit.t = pt_asfloat(b.z);
                if (kind == 0u && rs.hit.t < PT_FAR_PLANE) {         // the best hit so far was left in the hit array
                    const float4 h = f4_array(B, PT_F4_HIT)[ctx];
                    rs.hit.u = h.y; rs.hit.v = h.z; rs.hit.triIndex = pt_asuint(h.w);
                }
#pragma unroll
                for (uint32_t e = 0; e < PT_WF_SUSP_STACK_ROWS; ++e) {
                    const uint4 er = s_susp[lane][2u + e];
                    st.lds[(2u * e) * 64u] = pt_u2{er.x, er.y};
                    if (2u * e + 1u < PT_WF_LDS_STACK) st.lds[(2u * e + 1u) * 64u] = pt_u2{er.z, er.w};
                }
                have = true;
            }
            nSusp = 0u;
            const uint32_t nItems = 3u * 64u * K;
            uint32_t cursor = 0u;
            bool startedNew = false;

            auto finish = [&]() {                                    // the lane's ray is complete: write its result
                const uint32_t kind = myRef >> 30;
                const uint32_t ctx = ctxBase + (myRef & 0x3FFFFFFFu);
                if (kind == 0u) {
                    if (!(rs.hit.t < PT_FAR_PLANE)) store_miss(B, ctx);
                    else f4_array(B, PT_F4_HIT)[ctx] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                } else store_occlusion(B, kind, ctx, rs.hit.t < PT_FAR_PLANE);
                have = false;
            };

            while (true) {
                uint32_t nIdle = (uint32_t)__popcll(__ballot(!have));
                while (cursor < nItems && (nIdle >= PT_WF_REFILL || nIdle == 64u)) {
                    PT_DIAG_INC(dScan);
                    const unsigned long long idle = __ballot(!have);
                    const uint32_t rankI = rank_below(idle);
                    const uint32_t item = cursor + lane;
                    const uint32_t kind = item / (64u * K);
                    const uint32_t local = item & (64u * K - 1u);    // group * 64 + lane of the context
                    bool valid = item < nItems;
                    if (valid) {
                        const uint32_t pw = s_pend[po][local >> 6][(local >> 5) & 1u];
                        // (a context with a parked ray got all its rays last round; one without a pixel is DONE)
                        valid = !((pw >> (local & 31u)) & 1u) && ray_exists(B.flags[ctxBase + local], kind);
                    }
                    const unsigned long long V = __ballot(valid);
                    const uint32_t nV = (uint32_t)__popcll(V);
                    const uint32_t take = nIdle < nV ? nIdle : nV;
                    const uint32_t rankV = rank_below(V);
                    uint32_t consumed = 64u;
                    if (take < nV) consumed = (uint32_t)__ffsll((long long)__ballot(valid && rankV == take)) - 1u;   // first ray NOT taken
                    if (valid && rankV < take) xchg[rankV] = (kind << 30) | local;
                    __builtin_amdgcn_wave_barrier();
                    if (!have && rankI < take) {
                        myRef = xchg[rankI];
                        const uint32_t myKind = myRef >> 30;
                        const uint32_t ctx = ctxBase + (myRef & 0x3FFFFFFFu);
                        v3 o, d;
                        fetch_ray(B, ctx, myKind, o, d);
                        if (myKind == 0u) cn.closestRays++; else cn.shadowRays++;
                        if (ray_begin(rs, o, d, myKind != 0u)) {
                            if (myKind == 0u) store_miss(B, ctx);             // NaN ray: certain miss
                            else store_occlusion(B, myKind, ctx, false);
                        } else have = true;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (take > 0u) startedNew = true;
                    cursor += consumed;
                    nIdle = (uint32_t)__popcll(__ballot(!have));
                }
                if (nIdle == 64u) break;
                const bool exhausted = cursor >= nItems;
                // parking is allowed only in a round that started new rays: a round that merely resumed must finish them
                const bool mayPark = PT_WF_SUSPEND > 0u && exhausted && startedNew;
                const uint32_t stopAt = !exhausted ? PT_WF_REFILL : (mayPark ? 64u - PT_WF_SUSPEND : 64u);
                while (nIdle < stopAt) {
                    PT_DIAG_INC(dIter);
                    const bool wantTri = have && rs.tg.y != 0u;
                    const uint32_t nT = (uint32_t)__popcll(__ballot(wantTri));
                    const uint32_t nN = (uint32_t)__popcll(__ballot(have && !wantTri));
                    bool fin = false;
                    if ((nT >= PT_WF_TRI_PARK || nN == 0u) && wantTri) { PT_DIAG_INC(dTri); fin = ray_tri_one<STATS>(S, rs, cn); }
                    if (have && !fin && rs.tg.y == 0u) fin = ray_node_one<STATS>(S, rs, st, cn);
                    if (fin) finish();
                    nIdle = (uint32_t)__popcll(__ballot(!have));
                }
                if (mayPark && nIdle < 64u) {
                    // a ray whose stack reaches into the HBM slab stays (a record only holds the LDS entries): finish those first
                    while (__ballot(have && rs.sp > PT_WF_LDS_STACK) != 0ull) {
                        if (have) {
                            const bool fin = rs.tg.y != 0u ? ray_tri_one<STATS>(S, rs, cn) : ray_node_one<STATS>(S, rs, st, cn);
                            if (fin) finish();
                        }
                    }
                    const unsigned long long act = __ballot(have);
                    if (have) {
                        const uint32_t kind = myRef >> 30, local = myRef & 0x3FFFFFFFu;
                        if (kind == 0u && rs.hit.t < PT_FAR_PLANE)
                            f4_array(B, PT_F4_HIT)[ctxBase + local] = make_float4(rs.hit.t, rs.hit.u, rs.hit.v, pt_asfloat(rs.hit.triIndex));
                        const uint32_t rec = rank_below(act);
                        s_susp[rec][0] = make_uint4(myRef, rs.sp | (rs.overflow ? 0x100u : 0u), rs.ng.x, rs.ng.y);
                        s_susp[rec][1] = make_uint4(rs.tg.x, rs.tg.y, pt_asuint(rs.hit.t), 0u);
#pragma unroll
                        for (uint32_t e = 0; e < PT_WF_SUSP_STACK_ROWS; ++e) {
                            const pt_u2 x = st.lds[(2u * e) * 64u];
                            const pt_u2 y = (2u * e + 1u < PT_WF_LDS_STACK) ? (pt_u2)st.lds[(2u * e + 1u) * 64u] : pt_u2{0u, 0u};
                            s_susp[rec][2u + e] = make_uint4(x.x, x.y, y.x, y.y);
                        }
                        atomicOr(&s_pend[pn][local >> 6][(local >> 5) & 1u], 1u << (local & 31u));
                        have = false;
                    }
                    nSusp = (uint32_t)__popcll(act);
                    break;
                }
            }
            pb = pn;
            counters_to_lds<STATS>(cn, cnt, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        PT_DIAG_LAP(tTrace);

        // ---- (2) shade: every context whose rays have all returned
#pragma unroll 1
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t ctx = ctxBase + k * 64u + lane;
            const uint32_t mine = pix[k * 64u + lane];
            const uint32_t pw = s_pend[pb][k][lane >> 5];
            const bool parked = (pw >> (lane & 31u)) & 1u;
            Counters cn = {};
            if (mine != PT_WF_NO_PIXEL && !parked) {
                const uint32_t f = B.flags[ctx];
                if (!shade_slot<STATS, true>(S, P, tm, B, ctx, mine, f, cn)) pix[k * 64u + lane] = PT_WF_NO_PIXEL;    // pixel finished: its sum is in B.pixsum
            }
            counters_to_lds<STATS>(cn, cnt, lane);                   // every lane of the wave takes part in the reductions
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        PT_DIAG_LAP(tShade);
#ifdef PT_FUSED_DIAG
        rounds++;
#endif
    }
#ifdef PT_FUSED_DIAG
    if (lane == 0u) { cnt[5] = (uint32_t)(tRefill >> 10); cnt[6] = (uint32_t)(tTrace >> 10); cnt[7] = (uint32_t)(tShade >> 10); cnt[9] = rounds;
                      cnt[8] = dIter; cnt[14] = dScan; cnt[15] = dTri; }
    __builtin_amdgcn_wave_barrier();
#endif
    // the wave's totals -> its counter row (a wave owns row blockIdx.x for the whole launch)
    __builtin_amdgcn_wave_barrier();
    if (lane < PT_NUM_COUNTERS) {
        unsigned long long* p = B.statRows + (size_t)blockIdx.x * 16u + lane;
        const unsigned long long v = cnt[lane];
        if (lane == 12u) { if (v > *p) *p = v; } else if (v) *p += v;
    }
}

// ------------------------------------------------------------------------------------------
// shade, regrouped through LDS (round 3).  path_step() is three stages: (1) add the pending NEE of the last bounce, (2) shade the
// closest hit that came back -- attributes, lights, material + textures, two NEE evaluations, BSDF sampling, roulette: ~90 % of the
// kernel's instructions -- and (3) end-of-sample bookkeeping / next camera ray.  In slot order a wave runs stage 2 with whatever
// lanes happen to need it: 75 % of the slots are alive, about 60 % of those have a surface hit, and the wave still issues every
// instruction of every branch any lane takes (38 % VALU lane utilisation, profiles/r03_pmc.json).  Sorting the slot LIST fixed
// that and lost to the state gathers it caused (round 2: shade VALU -41 %, throughput -20 %).
//
// Here loads and stores stay coalesced in slot order (the HOME lane of a slot runs stages 1 and 3 and does all global state
// traffic), and only stage 2 is regrouped: inside a 256-slot workgroup the slots that need it are ranked -- surface hits first,
// then misses / light candidates, stable in slot order -- their stage-2 inputs (20 words) go to LDS record [rank], lane i of the
// workgroup shades record i, writes the 31 words of results back in place, and the home lanes pick them up.  Waves of the workgroup
// are therefore full surface-hit waves, at most one mixed wave, and waves with nothing to shade, which skip stage 2 altogether.
// The values a path sees are those of path_step() in the same order: frames and counters are bit-identical.
// LDS: 35 fields x 256 x 4 B = 35 KB per workgroup, SoA (one ds_*_b32 per field and lane, conflict-free) -> 4 workgroups per CU.
//
// MEASURED (round 3, Sponza-class 1080p / 8 spp; tools/experiments/sorted_pmc.sh): SQ_INSTS_VALU of the shade launches 4.46e9 ->
// 3.35e9 per pass (-25 %), lanes per VALU instruction 38 % -> 54 %, bit-identical frames and counters (the parity suite runs it) --
// and the launches take 9.8 instead of 7.7 ms per pass, the pipelined bench 4,390 instead of 5,030 Mrays/s.  The shade step is a
// chain of dependent fetches (state -> attributes -> material -> texels -> lights) that needs many waves in flight to hide; after
// the regrouping a workgroup's surface hits sit in ~2 of its 4 waves and the other two WAIT at the barriers holding their
// registers and the workgroup's LDS: per CU, 6-7 waves do stage 2 where 16 one-wave workgroups did before.  Fewer instructions,
// fewer waves to overlap their latency -- the second effect wins.  Kept as a variant (PT_WF_SHADE_SORTED=1 in the environment
// selects it); what would cash the instruction saving in is a regrouping that does not park waves: stage 2 as its own launch
// over compacted records.
// ------------------------------------------------------------------------------------------
#define PT_SH_FIELDS 31u
#define PT_SH_HOME 4u
template <bool STATS>
__global__ __launch_bounds__(256, PT_WF_SHADE_MIN_WAVES) void pt_wf_shade_sorted(DScene S, PTFrameParams P, PTTileMap tm, PTWfBuffers B, uint32_t iteration)
{
    __shared__ uint32_t s_rec[PT_SH_FIELDS][256];      // stage-2 records, field-major
    __shared__ uint32_t s_home[PT_SH_HOME][256];       // what a home lane parks across stage 2: throughput before the bounce (or radiance / rng of a slot that skips stage 2)
    __shared__ uint32_t s_cnt[2][4];                   // per wave: slots of class 0 (surface hit), class 1 (miss)
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t slot = blockIdx.x * 256u + tid;
    if (blockIdx.x == 0u && tid < PT_WF_SHARDS) B.chunkHeads[tid * 32u] = 0u;   // for the next trace launch (schedule 3)
    if (tail_cut(B, iteration)) return;
    Counters cn = {};
    const uint32_t f = B.flags[slot];
    const bool live = fl_state(f) != PS_DONE;
    auto put = [&](uint32_t field, uint32_t idx, uint32_t v) { s_rec[field][idx] = v; };
    auto putf = [&](uint32_t field, uint32_t idx, float v) { s_rec[field][idx] = pt_asuint(v); };
    auto get = [&](uint32_t field, uint32_t idx) -> uint32_t { return s_rec[field][idx]; };
    auto getf = [&](uint32_t field, uint32_t idx) -> float { return pt_asfloat(s_rec[field][idx]); };

    // ---- stage 1 at home: the slot's state (one batch of coalesced loads), pending NEE applied
    uint32_t key = 2u;                                              // 0: surface hit to shade, 1: miss (sky / analytic light), 2: no stage 2
    {
        uint32_t rng = B.rng[slot];
        float4 qro = B.ray[0][2u * slot], qrd = B.ray[0][2u * slot + 1u], qrad = B.rad[slot], qthr = B.thr[slot];
        float4 qenvC = B.envC[slot], qlightC = B.lightC[slot], qpthr = B.pthr[slot], qhit = B.hit[slot];
        uint32_t o0 = B.occl[slot], o1 = B.occl[(size_t)B.numSlots + slot];
        float4 qhit2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (S.hasTlas) qhit2 = B.hit2[slot];
        asm volatile("" : "+v"(rng), "+v"(qro.x), "+v"(qrd.x), "+v"(qrad.x), "+v"(qthr.x), "+v"(qenvC.x), "+v"(qlightC.x), "+v"(qpthr.x), "+v"(qhit.x), "+v"(o0), "+v"(o1), "+v"(qhit2.x));
        if (live) {
            PathRegs r;
            r.state = fl_state(f);
            r.hasPending = fl_pending(f);
            r.env.valid = fl_env(f);
            r.light.valid = fl_light(f);
            r.green = (f >> 6) & 1u;
            r.radiance = xyz(qrad);
            r.env.contribution = r.hasPending ? xyz(qenvC) : mk3(0.0f);
            r.light.contribution = r.hasPending ? xyz(qlightC) : mk3(0.0f);
            r.pendThroughput = r.hasPending ? xyz(qpthr) : mk3(0.0f);
            const bool occEnv = r.hasPending && o0 != 0u, occLight = r.hasPending && o1 != 0u;
            path_apply_pending(r, occEnv, occLight);
            if (r.state == PS_TRACE) key = qhit.x < PT_FAR_PLANE ? 0u : 1u;
            qrad = f4(r.radiance, 0.0f);
        }
        // ---- rank the slots that need stage 2 (stable partition of the workgroup's 256 slots: class 0, then class 1)
        const unsigned long long m0 = __ballot(key == 0u), m1 = __ballot(key == 1u);
        if (lane == 0u) { s_cnt[0][wave] = (uint32_t)__popcll(m0); s_cnt[1][wave] = (uint32_t)__popcll(m1); }
        __syncthreads();
        uint32_t base0 = 0u, base1 = 0u, total0 = 0u, total1 = 0u;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) {
            const uint32_t c0 = s_cnt[0][w], c1 = s_cnt[1][w];
            if (w < wave) { base0 += c0; base1 += c1; }
            total0 += c0; total1 += c1;
        }
        const uint32_t dest = key == 0u ? base0 + rank_below(m0) : (key == 1u ? total0 + base1 + rank_below(m1) : 0xFFFFu);
        key |= dest << 2;                                            // the lane's own ticket, kept across stage 2 in ONE register
        key |= (total0 + total1) << 20;
        if (dest != 0xFFFFu) {
            putf(0, dest, qro.x); putf(1, dest, qro.y); putf(2, dest, qro.z);
            putf(3, dest, qrd.x); putf(4, dest, qrd.y); putf(5, dest, qrd.z);
            put(6, dest, rng); put(7, dest, f >> 19);               // depth
            putf(8, dest, qro.w); putf(9, dest, qrd.w);             // scatterPdf, maxRoughness
            putf(10, dest, qthr.x); putf(11, dest, qthr.y); putf(12, dest, qthr.z);
            putf(13, dest, qrad.x); putf(14, dest, qrad.y); putf(15, dest, qrad.z);
            putf(16, dest, qhit.x); putf(17, dest, qhit.y); putf(18, dest, qhit.z); putf(19, dest, qhit.w);
            if (S.hasTlas) { putf(20, dest, qhit2.x); putf(21, dest, qhit2.y); putf(22, dest, qhit2.z); putf(23, dest, qhit2.w); }
            s_home[0][tid] = pt_asuint(qthr.x); s_home[1][tid] = pt_asuint(qthr.y); s_home[2][tid] = pt_asuint(qthr.z);    // the throughput the NEE of this bounce applies to
        } else {
            s_home[0][tid] = pt_asuint(qrad.x); s_home[1][tid] = pt_asuint(qrad.y); s_home[2][tid] = pt_asuint(qrad.z);
            s_home[3][tid] = rng;
        }
    }
    __syncthreads();

    // ---- stage 2: lane i shades record i (waves beyond the last record skip it)
    const uint32_t nWork = key >> 20;
    if (tid < nWork) {
        PathRegs w;
        w.ro = mk3(getf(0, tid), getf(1, tid), getf(2, tid));
        w.rd = mk3(getf(3, tid), getf(4, tid), getf(5, tid));
        w.rng = get(6, tid); w.depth = get(7, tid);
        w.scatterPdf = getf(8, tid); w.maxRoughness = getf(9, tid);
        w.throughput = mk3(getf(10, tid), getf(11, tid), getf(12, tid));
        w.radiance = mk3(getf(13, tid), getf(14, tid), getf(15, tid));
        HitRecord ch;
        ch.h.t = getf(16, tid); ch.h.u = getf(17, tid); ch.h.v = getf(18, tid); ch.h.triIndex = get(19, tid);
        ch.pos = mk3(0.0f); ch.inst = 0u;
        if (S.hasTlas) { ch.pos = mk3(getf(20, tid), getf(21, tid), getf(22, tid)); ch.inst = get(23, tid); }
        w.state = PS_TRACE; w.hasPending = false; w.green = false; w.sampleIdx = 0u;
        w.color = mk3(0.0f);
        w.env.valid = 0u; w.light.valid = 0u;
        w.env.dir = mk3(0.0f); w.light.dir = mk3(0.0f); w.neeOrigin = mk3(0.0f);
        w.env.contribution = mk3(0.0f); w.light.contribution = mk3(0.0f); w.pendThroughput = mk3(0.0f);
        // the NEE rays of the bounce go to the record as soon as they are final: their 15 registers are free while the BSDF is sampled
        struct Sink {
            uint32_t (*rec)[256]; uint32_t i;
            PT_DEV void operator()(PathRegs& q) const {
                rec[16][i] = pt_asuint(q.neeOrigin.x); rec[17][i] = pt_asuint(q.neeOrigin.y); rec[18][i] = pt_asuint(q.neeOrigin.z);
                rec[19][i] = pt_asuint(q.env.dir.x); rec[20][i] = pt_asuint(q.env.dir.y); rec[21][i] = pt_asuint(q.env.dir.z);
                rec[22][i] = pt_asuint(q.light.dir.x); rec[23][i] = pt_asuint(q.light.dir.y); rec[24][i] = pt_asuint(q.light.dir.z);
                rec[25][i] = pt_asuint(q.env.contribution.x); rec[26][i] = pt_asuint(q.env.contribution.y); rec[27][i] = pt_asuint(q.env.contribution.z);
                rec[28][i] = pt_asuint(q.light.contribution.x); rec[29][i] = pt_asuint(q.light.contribution.y); rec[30][i] = pt_asuint(q.light.contribution.z);
            }
        };
        path_shade_hit<STATS, Sink>(S, P, w, ch, cn, Sink{s_rec, tid});
        putf(0, tid, w.ro.x); putf(1, tid, w.ro.y); putf(2, tid, w.ro.z);
        putf(3, tid, w.rd.x); putf(4, tid, w.rd.y); putf(5, tid, w.rd.z);
        put(6, tid, w.rng);
        put(7, tid, (w.state & 3u) | ((w.hasPending ? 1u : 0u) << 2) | ((w.env.valid & 3u) << 3) | ((w.light.valid & 1u) << 5) | ((w.green ? 1u : 0u) << 6) | ((w.depth & 0x1FFFu) << 19));
        putf(8, tid, w.scatterPdf); putf(9, tid, w.maxRoughness);
        putf(10, tid, w.throughput.x); putf(11, tid, w.throughput.y); putf(12, tid, w.throughput.z);
        putf(13, tid, w.radiance.x); putf(14, tid, w.radiance.y); putf(15, tid, w.radiance.z);
    }
    __syncthreads();

    // ---- stage 3 at home: results of stage 2 (if any), end of the sample, coalesced stores
    bool alive = false;
    if (live) {
        uint32_t px, py, pass;
        pt_slot_to_pixel(tm, pixel_slot_of(B, slot, pass), px, py);
        const uint32_t dest = (key >> 2) & 0xFFFFu;
        PathRegs r;
        r.sampleIdx = (f >> 7) & 0xFFFu;
        r.color = xyz(B.color[slot]);
        r.env.dir = mk3(0.0f); r.light.dir = mk3(0.0f); r.neeOrigin = mk3(0.0f);
        r.env.contribution = mk3(0.0f); r.light.contribution = mk3(0.0f); r.pendThroughput = mk3(0.0f);
        r.env.valid = 0u; r.light.valid = 0u;
        bool writeNee = false;
        if (dest != 0xFFFFu) {
            const uint32_t m = get(7, dest);
            r.state = m & 3u; r.hasPending = (m >> 2) & 1u; r.env.valid = (m >> 3) & 3u; r.light.valid = (m >> 5) & 1u; r.green = (m >> 6) & 1u;
            r.depth = m >> 19;
            r.ro = mk3(getf(0, dest), getf(1, dest), getf(2, dest));
            r.rd = mk3(getf(3, dest), getf(4, dest), getf(5, dest));
            r.rng = get(6, dest);
            r.scatterPdf = getf(8, dest); r.maxRoughness = getf(9, dest);
            r.throughput = mk3(getf(10, dest), getf(11, dest), getf(12, dest));
            r.radiance = mk3(getf(13, dest), getf(14, dest), getf(15, dest));
            if (r.hasPending) {
                writeNee = true;
                r.neeOrigin = mk3(getf(16, dest), getf(17, dest), getf(18, dest));
                r.env.dir = mk3(getf(19, dest), getf(20, dest), getf(21, dest));
                r.light.dir = mk3(getf(22, dest), getf(23, dest), getf(24, dest));
                r.env.contribution = mk3(getf(25, dest), getf(26, dest), getf(27, dest));
                r.light.contribution = mk3(getf(28, dest), getf(29, dest), getf(30, dest));
                r.pendThroughput = mk3(pt_asfloat(s_home[0][tid]), pt_asfloat(s_home[1][tid]), pt_asfloat(s_home[2][tid]));
            }
        } else {
            // no closest hit to shade: the path was waiting for its last NEE (state ENDING)
            r.state = fl_state(f); r.hasPending = false; r.green = false;
            r.depth = f >> 19;
            r.radiance = mk3(pt_asfloat(s_home[0][tid]), pt_asfloat(s_home[1][tid]), pt_asfloat(s_home[2][tid]));
            r.rng = s_home[3][tid];
            const float4 qro = B.ray[0][2u * slot], qrd = B.ray[0][2u * slot + 1u], qthr = B.thr[slot];
            r.ro = xyz(qro); r.scatterPdf = qro.w;
            r.rd = xyz(qrd); r.maxRoughness = qrd.w;
            r.throughput = xyz(qthr);
        }
        path_end_sample<false>(P, r, px, py, py * P.OutputWidth + px, nullptr, nullptr, cn);
        store_path(B, slot, r, writeNee);
        alive = r.state != PS_DONE;
    }
    {
        const uint32_t nAlive = (uint32_t)__popcll(__ballot(alive));
        if (B.tailCut != 0u && nAlive && lane == 0u)
            atomicAdd(&B.aliveFlags[iteration * PT_WF_ALIVE_SHARDS + (blockIdx.x & (PT_WF_ALIVE_SHARDS - 1u))], nAlive);
    }
    flush_counters<STATS>(cn, B.statRows, blockIdx.x * 4u + wave, lane);
}

// ------------------------------------------------------------------------------------------
// shade, split into three launches (round 3, PTWfBuffers.shadeSorted == 2): the regrouping of pt_wf_shade_sorted without its
// parked waves.  pre: every slot's HOME lane adds the pending NEE (stage 1), and a slot with a closest hit to shade gets a RECORD:
// rank within the workgroup by ballot + one atomic per workgroup and class for the base (surface hits fill the record array
// from the front, misses from the back), inputs written as five coalesced float4 planes.  hit: one lane per record, one wave per
// workgroup -- full waves of surface hits, full waves of misses, nothing idle -- runs path_shade_hit() and writes nine planes of
// results.  post: the home lanes pick the results up (rank order = slot order inside a workgroup: near-contiguous reads), finish the
// sample (stage 3) and store the state coalesced.  Which record a slot gets depends on timing; what is computed for it does not.
// ------------------------------------------------------------------------------------------
#define PT_SP_PLANES 10u        // float4 planes of the record array (PTWfBuffers.recPlanes, stride numSlots)
PT_DEV float4* rec_plane(const PTWfBuffers& B, uint32_t k) { return B.recPlanes + (size_t)k * B.numSlots; }

__global__ __launch_bounds__(256) void pt_wf_shade_pre(DScene S, PTWfBuffers B, uint32_t iteration)
{
    __shared__ uint32_t s_cnt[2][4];
    __shared__ uint32_t s_base[2];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t slot = blockIdx.x * 256u + tid;
    if (blockIdx.x == 0u && tid < PT_WF_SHARDS) B.chunkHeads[tid * 32u] = 0u;   // for the next trace launch (schedule 3)
    if (tail_cut(B, iteration)) return;
    const uint32_t f = B.flags[slot];
    const bool live = fl_state(f) != PS_DONE;
    uint32_t key = 2u;
    uint32_t rng = 0u;
    float4 qro = make_float4(0, 0, 0, 0), qrd = qro, qrad = qro, qthr = qro, qhit = qro, qhit2 = qro;
    if (live) {
        const bool pending = fl_pending(f);
        qrad = B.rad[slot];
        qhit = B.hit[slot];
        if (pending) {
            const float4 qenvC = B.envC[slot], qlightC = B.lightC[slot], qpthr = B.pthr[slot];
            const uint32_t o0 = B.occl[slot], o1 = B.occl[(size_t)B.numSlots + slot];
            PathRegs r;
            r.hasPending = true;
            r.env.valid = fl_env(f); r.light.valid = fl_light(f);
            r.green = (f >> 6) & 1u;
            r.radiance = xyz(qrad);
            r.env.contribution = xyz(qenvC); r.light.contribution = xyz(qlightC); r.pendThroughput = xyz(qpthr);
            path_apply_pending(r, o0 != 0u, o1 != 0u);
            qrad = f4(r.radiance, 0.0f);
            B.rad[slot] = qrad;                                       // stage 3 (and a slot that skips stage 2) reads it back
        }
        if (fl_state(f) == PS_TRACE) {
            key = qhit.x < PT_FAR_PLANE ? 0u : 1u;
            rng = B.rng[slot];
            qro = B.ray[0][2u * slot]; qrd = B.ray[0][2u * slot + 1u]; qthr = B.thr[slot];
            if (S.hasTlas) qhit2 = B.hit2[slot];
        }
    }
    const unsigned long long m0 = __ballot(key == 0u), m1 = __ballot(key == 1u);
    if (lane == 0u) { s_cnt[0][wave] = (uint32_t)__popcll(m0); s_cnt[1][wave] = (uint32_t)__popcll(m1); }
    __syncthreads();
    uint32_t base0 = 0u, base1 = 0u, total0 = 0u, total1 = 0u;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) {
        const uint32_t c0 = s_cnt[0][w], c1 = s_cnt[1][w];
        if (w < wave) { base0 += c0; base1 += c1; }
        total0 += c0; total1 += c1;
    }
    if (tid == 0u) {
        s_base[0] = total0 ? atomicAdd(&B.recCount[0], total0) : 0u;
        s_base[1] = total1 ? atomicAdd(&B.recCount[1], total1) : 0u;
    }
    __syncthreads();
    uint32_t dest = 0xFFFFFFFFu;
    if (key == 0u) dest = s_base[0] + base0 + rank_below(m0);
    else if (key == 1u) dest = B.numSlots - 1u - (s_base[1] + base1 + rank_below(m1));      // misses fill the array from the back
    if (live) B.recDest[slot] = dest;
    if (dest != 0xFFFFFFFFu) {
        rec_plane(B, 0)[dest] = qro;                                  // origin, scatterPdf
        rec_plane(B, 1)[dest] = qrd;                                  // direction, maxRoughness
        rec_plane(B, 2)[dest] = make_float4(qthr.x, qthr.y, qthr.z, pt_asfloat(rng));
        rec_plane(B, 3)[dest] = make_float4(qrad.x, qrad.y, qrad.z, pt_asfloat(f >> 19));     // radiance, depth
        rec_plane(B, 4)[dest] = qhit;
        if (S.hasTlas) rec_plane(B, 5)[dest] = qhit2;
        B.pthr[slot] = qthr;                                          // the throughput the NEE of this bounce applies to (if it produces one)
    }
}

template <bool STATS>
__global__ __launch_bounds__(64, PT_WF_SHADE_MIN_WAVES) void pt_wf_shade_hit(DScene S, PTFrameParams P, PTWfBuffers B, uint32_t iteration)
{
    if (tail_cut(B, iteration)) return;
    const uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    const uint32_t c0 = B.recCount[0], c1 = B.recCount[1];
    const bool mine = idx < c0 || idx >= B.numSlots - c1;
    Counters cn = {};
    if (mine) {
        const float4 p0 = rec_plane(B, 0)[idx], p1 = rec_plane(B, 1)[idx], p2 = rec_plane(B, 2)[idx], p3 = rec_plane(B, 3)[idx], p4 = rec_plane(B, 4)[idx];
        PathRegs w;
        w.ro = xyz(p0); w.scatterPdf = p0.w;
        w.rd = xyz(p1); w.maxRoughness = p1.w;
        w.throughput = xyz(p2); w.rng = pt_asuint(p2.w);
        w.radiance = xyz(p3); w.depth = pt_asuint(p3.w);
        HitRecord ch;
        ch.h.t = p4.x; ch.h.u = p4.y; ch.h.v = p4.z; ch.h.triIndex = pt_asuint(p4.w);
        ch.pos = mk3(0.0f); ch.inst = 0u;
        if (S.hasTlas) { const float4 p5 = rec_plane(B, 5)[idx]; ch.pos = xyz(p5); ch.inst = pt_asuint(p5.w); }
        w.state = PS_TRACE; w.hasPending = false; w.green = false; w.sampleIdx = 0u;
        w.color = mk3(0.0f);
        w.env.valid = 0u; w.light.valid = 0u;
        w.env.dir = mk3(0.0f); w.light.dir = mk3(0.0f); w.neeOrigin = mk3(0.0f);
        w.env.contribution = mk3(0.0f); w.light.contribution = mk3(0.0f); w.pendThroughput = mk3(0.0f);
        struct Sink {                                                // the NEE rays of the bounce leave for their planes as soon as they are final
            const PTWfBuffers& B; uint32_t i;
            PT_DEV void operator()(PathRegs& q) const {
                uint32_t j = i;
                asm volatile("" : "+v"(j));
                rec_plane(B, 5)[j] = f4(q.neeOrigin, 0.0f);
                rec_plane(B, 6)[j] = f4(q.env.dir, 0.0f);
                rec_plane(B, 7)[j] = f4(q.light.dir, 0.0f);
                rec_plane(B, 8)[j] = f4(q.env.contribution, 0.0f);
                rec_plane(B, 9)[j] = f4(q.light.contribution, 0.0f);
            }
        };
        path_shade_hit<STATS, Sink>(S, P, w, ch, cn, Sink{B, idx});
        uint32_t j = idx;
        asm volatile("" : "+v"(j));
        rec_plane(B, 0)[j] = f4(w.ro, w.scatterPdf);
        rec_plane(B, 1)[j] = f4(w.rd, w.maxRoughness);
        rec_plane(B, 2)[j] = make_float4(w.throughput.x, w.throughput.y, w.throughput.z, pt_asfloat(w.rng));
        rec_plane(B, 3)[j] = make_float4(w.radiance.x, w.radiance.y, w.radiance.z,
                                         pt_asfloat((w.state & 3u) | ((w.hasPending ? 1u : 0u) << 2) | ((w.env.valid & 3u) << 3) | ((w.light.valid & 1u) << 5) |
                                                    ((w.green ? 1u : 0u) << 6) | ((w.depth & 0x1FFFu) << 19)));
    }
    flush_counters<STATS>(cn, B.statRows, blockIdx.x, threadIdx.x);
}

template <bool STATS>
__global__ __launch_bounds__(256) void pt_wf_shade_post(PTFrameParams P, PTTileMap tm, PTWfBuffers B, uint32_t iteration)
{
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t slot = blockIdx.x * 256u + tid;
    if (blockIdx.x == 0u && tid < 2u) B.recCount[tid] = 0u;           // for the next iteration's pre launch (this launch does not read them)
    if (tail_cut(B, iteration)) return;
    Counters cn = {};
    const uint32_t f = B.flags[slot];
    bool alive = false;
    if (fl_state(f) != PS_DONE) {
        uint32_t px, py, pass;
        pt_slot_to_pixel(tm, pixel_slot_of(B, slot, pass), px, py);
        const uint32_t dest = B.recDest[slot];
        PathRegs r;
        r.sampleIdx = (f >> 7) & 0xFFFu;
        r.color = xyz(B.color[slot]);
        r.env.dir = mk3(0.0f); r.light.dir = mk3(0.0f); r.neeOrigin = mk3(0.0f);
        r.env.contribution = mk3(0.0f); r.light.contribution = mk3(0.0f); r.pendThroughput = mk3(0.0f);
        r.env.valid = 0u; r.light.valid = 0u;
        bool writeNee = false;
        if (dest != 0xFFFFFFFFu) {
            const float4 q0 = rec_plane(B, 0)[dest], q1 = rec_plane(B, 1)[dest], q2 = rec_plane(B, 2)[dest], q3 = rec_plane(B, 3)[dest];
            const uint32_t m = pt_asuint(q3.w);
            r.state = m & 3u; r.hasPending = (m >> 2) & 1u; r.env.valid = (m >> 3) & 3u; r.light.valid = (m >> 5) & 1u; r.green = (m >> 6) & 1u;
            r.depth = m >> 19;
            r.ro = xyz(q0); r.scatterPdf = q0.w;
            r.rd = xyz(q1); r.maxRoughness = q1.w;
            r.throughput = xyz(q2); r.rng = pt_asuint(q2.w);
            r.radiance = xyz(q3);
            if (r.hasPending) {
                writeNee = true;
                r.neeOrigin = xyz(rec_plane(B, 5)[dest]);
                r.env.dir = xyz(rec_plane(B, 6)[dest]);
                r.light.dir = xyz(rec_plane(B, 7)[dest]);
                r.env.contribution = xyz(rec_plane(B, 8)[dest]);
                r.light.contribution = xyz(rec_plane(B, 9)[dest]);
                r.pendThroughput = xyz(B.pthr[slot]);
            }
        } else {
            // no closest hit to shade: the path was waiting for its last NEE (state ENDING, radiance updated by the pre launch)
            r.state = fl_state(f); r.hasPending = false; r.green = false;
            r.depth = f >> 19;
            r.radiance = xyz(B.rad[slot]);
            r.rng = B.rng[slot];
            const float4 qro = B.ray[0][2u * slot], qrd = B.ray[0][2u * slot + 1u], qthr = B.thr[slot];
            r.ro = xyz(qro); r.scatterPdf = qro.w;
            r.rd = xyz(qrd); r.maxRoughness = qrd.w;
            r.throughput = xyz(qthr);
        }
        path_end_sample<false>(P, r, px, py, py * P.OutputWidth + px, nullptr, nullptr, cn);
        store_path(B, slot, r, writeNee);
        alive = r.state != PS_DONE;
    }
    {
        const uint32_t nAlive = (uint32_t)__popcll(__ballot(alive));
        if (B.tailCut != 0u && nAlive && lane == 0u)
            atomicAdd(&B.aliveFlags[iteration * PT_WF_ALIVE_SHARDS + (blockIdx.x & (PT_WF_ALIVE_SHARDS - 1u))], nAlive);
    }
    flush_counters<STATS>(cn, B.statRows, blockIdx.x * 4u + wave, lane);
}

// cleanup: pixels still alive after the fixed number of iterations are run to completion here, one lane per slot with
// the megakernel's loop (trace <= 3 rays, path_step, repeat).  Normally a handful of lanes; correctness for any path length.
template <bool STATS, bool TLAS>
__global__ __launch_bounds__(256, 2) void pt_wf_cleanup(DScene S, PTFrameParams P, PTTileMap tm, PTWfBuffers B)
{
    __shared__ uint2 s_stack[PT_LDS_STACK][256];
    Counters cn = {};
    // a few resident workgroups stride over the slot blocks: almost every block is finished already, and scanning 256 flag
    // words is cheaper than scheduling a 24-KB-LDS workgroup for them
    for (uint32_t blk = blockIdx.x; blk < (B.numSlots >> 8); blk += gridDim.x) {
    const uint32_t slot = blk * 256u + threadIdx.x;
    const uint32_t f = B.flags[slot];
    if (__any(fl_state(f) != PS_DONE)) {
        if (fl_state(f) != PS_DONE) {
            uint32_t px, py, pass;
            pt_slot_to_pixel(tm, pixel_slot_of(B, slot, pass), px, py);
            PathRegs r;
            load_path(B, slot, f, r);
            if (r.hasPending) {
                r.neeOrigin = xyz(B.ray[1][2u * slot]);
                r.env.dir = xyz(B.ray[1][2u * slot + 1u]);
                r.light.dir = xyz(B.ray[2][2u * slot + 1u]);
            }
            TravStack st;
            st.lds = PT_LDS_U2(&s_stack[0][threadIdx.x]);
            st.stride = 256u;
            while (r.state != PS_DONE) {
                HitRecord ch;
                ch.h.t = PT_FAR_PLANE; ch.h.u = 0.0f; ch.h.v = 0.0f; ch.h.triIndex = 0u;
                ch.pos = mk3(0.0f); ch.inst = 0u;
                bool occEnv = false, occLight = false;
                if (r.hasPending && r.env.valid != 0u) {
                    HitRecord h = ch;
                    if (TLAS) occEnv = traverse_tlas<STATS>(S, r.neeOrigin, r.env.dir, true, h, st, cn);
                    else { traverse_cwbvh<STATS>(S, r.neeOrigin, r.env.dir, true, h.h, st, cn); occEnv = h.h.t < PT_FAR_PLANE; }
                    cn.shadowRays++;
                }
                if (r.hasPending && r.light.valid != 0u) {
                    HitRecord h = ch;
                    if (TLAS) occLight = traverse_tlas<STATS>(S, r.neeOrigin, r.light.dir, true, h, st, cn);
                    else { traverse_cwbvh<STATS>(S, r.neeOrigin, r.light.dir, true, h.h, st, cn); occLight = h.h.t < PT_FAR_PLANE; }
                    cn.shadowRays++;
                }
                if (r.state == PS_TRACE) {
                    if (TLAS) traverse_tlas<STATS>(S, r.ro, r.rd, false, ch, st, cn);
                    else traverse_cwbvh<STATS>(S, r.ro, r.rd, false, ch.h, st, cn);
                    cn.closestRays++;
                }
                path_step<STATS, false>(S, P, r, ch, occEnv, occLight, px, py, py * P.OutputWidth + px, nullptr, nullptr, cn);
            }
            B.flags[slot] = pack_flags(r);
            B.color[slot] = f4(r.color, 0.0f);
        }
    }
    }
    flush_counters<STATS>(cn, B.statRows, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63u);
}

// resolve: the pixel write of PathTracer.compute:89-98, applied to every pixel's sample sum -- for a batch, once per pass and in
// pass order, each pass reading what the previous one would have written (the intermediate frames are never stored; the running
// mean is the same chain of fp32 operations)
__global__ __launch_bounds__(256) void pt_wf_resolve(PTFrameParams P, PTBatch batch, PTTileMap tm, PTWfBuffers B, const float4* __restrict__ sums,
                                                     const float4* __restrict__ accumulated, float4* __restrict__ output)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t px, py;
    Counters cn = {};
    if (slot < B.slotsPerPass && pt_slot_to_pixel(tm, slot, px, py)) {
        const uint32_t pixelIndex = py * P.OutputWidth + px;
        const uint32_t numSamples = P.SamplesPerPass > 1 ? (uint32_t)P.SamplesPerPass : 1u;
        const float fSamples = (float)numSamples;
        v3 acc = mk3(0.0f);
        for (uint32_t j = 0; j < batch.count; ++j) {
            uint32_t seedRoot, currentSample;
            pt_batch_pick(batch, j, seedRoot, currentSample);
            const v3 color = xyz(sums[(size_t)j * B.slotsPerPass + slot]);      // per-pixel sample sum: B.color (schedules 1-3) or B.pixsum (schedule 4)
            if (currentSample > 0u) {
                if (j == 0u) {
                    const float4 a = accumulated[pixelIndex];
                    acc = mk3(a.x, a.y, a.z);
                }
                cn.pixelsRead++;
                const float cs = (float)currentSample;
                acc = (color + acc * cs) / (cs + fSamples);
            } else {
                acc = color / fSamples;
            }
            cn.pixelsWritten++;
        }
        output[pixelIndex] = make_float4(acc.x, acc.y, acc.z, 1.0f);
    }
    flush_counters<false>(cn, B.statRows, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63u);
}

// fold the per-wave rows into the context's 16 counters (PTStats order) and clear them
__global__ __launch_bounds__(256) void pt_wf_fold_rows(unsigned long long* rows, uint32_t numRows, unsigned long long* gstats)
{
    unsigned long long acc[PT_NUM_COUNTERS] = {};
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < numRows; r += gridDim.x * 256u) {
        unsigned long long* p = rows + (size_t)r * 16u;
#pragma unroll
        for (int i = 0; i < PT_NUM_COUNTERS; ++i) {
            unsigned long long v = p[i];
            if (i == 12) acc[i] = v > acc[i] ? v : acc[i]; else acc[i] += v;
            p[i] = 0ull;
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int i = 0; i < PT_NUM_COUNTERS; ++i) {
        unsigned long long v = acc[i];
        for (int off = 32; off > 0; off >>= 1) {
            unsigned long long o = __shfl_xor(v, off, 64);
            if (i == 12) v = o > v ? o : v; else v += o;
        }
        if (lane == 0 && v) { if (i == 12) atomicMax(&gstats[i], v); else atomicAdd(&gstats[i], v); }
    }
}

} // namespace


// One pass = a fixed sequence of launches on `stream`, no host synchronisation (see the file header).
// `orderAfter` (may be null) is the event of the previous pass's resolve: this pass's resolve reads that pass's output as
// AccumulatedOutput and, with ping-pong frames, overwrites the frame that resolve was still reading.
hipError_t pt_launch_wavefront(const DScene& S, const PTFrameParams& P, const PTBatch& batch, const float4* accumulated, float4* output,
                               const PTTileMap& tm, const PTWfBuffers& B, unsigned long long* gstats, bool fullStats,
                               hipStream_t stream, hipEvent_t orderAfter, bool zeroOutputFirst,
                               uint32_t* launchesOut, int traceVariant, uint32_t iterationsOverride)
{
    const uint32_t nb = B.numSlots >> 8, nbPass = B.slotsPerPass >> 8;
    uint32_t launches = 0;
    hipError_t e;
    const bool tlas = S.hasTlas != 0u;
    if (tlas && traceVariant == 4) traceVariant = 2;      // schedule 4 with HAS_TLAS: the two-level walk runs through schedule 1's refill kernel
    const bool tlasRefill = tlas && traceVariant == 2;    // schedule 1: two-level traversal through the refill scheduler
    if (tlas) traceVariant = 1;          // schedules 2, 3: the one-ray-per-lane trace kernel
    if (traceVariant == 4 && !tlas) {
        // schedule 4: one persistent launch renders the whole pass (pt_wf_fused); then the ordered pixel write and the counter fold
        if ((e = hipMemsetAsync(B.chunkHeads, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
        const uint32_t maxWaves = B.numSlots / (64u * PT_WF_FUSED_GROUPS);           // contexts never outnumber the frame's slots (array sizes)
        uint32_t waves = B.residentWaves / 8u * (uint32_t)PT_WF_FUSED_WAVES;          // CUs x 4 SIMDs x waves per SIMD
        if (waves > maxWaves) waves = maxWaves;
        if (waves == 0u) waves = 1u;
        if (fullStats) hipLaunchKernelGGL(pt_wf_fused<true>, dim3(waves), dim3(64), 0, stream, S, P, batch, tm, B);
        else hipLaunchKernelGGL(pt_wf_fused<false>, dim3(waves), dim3(64), 0, stream, S, P, batch, tm, B);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (orderAfter && (e = hipStreamWaitEvent(stream, orderAfter, 0)) != hipSuccess) return e;
        if (zeroOutputFirst &&
            (e = hipMemsetAsync(output, 0, (size_t)P.OutputWidth * P.OutputHeight * sizeof(float4), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(pt_wf_resolve, dim3(nbPass), dim3(256), 0, stream, P, batch, tm, B, (const float4*)B.pixsum, accumulated, output);
        hipLaunchKernelGGL(pt_wf_fold_rows, dim3(256), dim3(256), 0, stream, B.statRows, B.numStatRows, gstats);
        if (launchesOut) *launchesOut = 3u;
        return hipGetLastError();
    }
    if ((e = hipMemsetAsync(B.aliveFlags, 0, sizeof(uint32_t) * B.maxIterations * PT_WF_ALIVE_SHARDS, stream)) != hipSuccess) return e;
    if (B.shadeSorted == 2u && (e = hipMemsetAsync(B.recCount, 0, 2 * sizeof(uint32_t), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(pt_wf_init, dim3(nb), dim3(256), 0, stream, P, batch, tm, B);
    launches++;
    const uint32_t spp = P.SamplesPerPass > 1 ? (uint32_t)P.SamplesPerPass : 1u;
    const uint32_t bounces = P.MaxRayBounces > 1u ? P.MaxRayBounces : 1u;
    // a sample needs at most (bounces + 1) closest-hit iterations + 1 to apply its last NEE; alpha-skips beyond that go to cleanup
    uint32_t iterations = iterationsOverride ? iterationsOverride : spp * (bounces + 2u) + 4u;
    if (iterations > B.maxIterations) iterations = B.maxIterations;
    const uint32_t refillBlocks = (B.numSlots + PT_WF_RANGE - 1u) / PT_WF_RANGE;                 // one wave per workgroup
    for (uint32_t it = 0; it < iterations; ++it) {
        if (traceVariant == 0) {
            const uint32_t numChunks = (B.numSlots + PT_WF_CHUNK - 1u) / PT_WF_CHUNK;
            uint32_t waves = B.residentWaves;
            if (waves > numChunks) waves = numChunks;
            if (fullStats) hipLaunchKernelGGL(pt_wf_trace_persist<true>, dim3(waves), dim3(64), 0, stream, S, B, it);
            else hipLaunchKernelGGL(pt_wf_trace_persist<false>, dim3(waves), dim3(64), 0, stream, S, B, it);
        } else if (traceVariant == 2) {
            const bool wide = PT_WF_RANGE >= 128u && pt_wf_wide_ranges(B.numSlots, B.residentWaves);         // pt_launch.h
            const uint32_t blocks = wide ? (B.numSlots + 127u) / 128u : (B.numSlots + 63u) / 64u;
            if (wide) {
                if (fullStats) hipLaunchKernelGGL((pt_wf_trace_refill<true, false, 128u>), dim3(blocks), dim3(64), 0, stream, S, B, it);
                else hipLaunchKernelGGL((pt_wf_trace_refill<false, false, 128u>), dim3(blocks), dim3(64), 0, stream, S, B, it);
            } else {
                if (fullStats) hipLaunchKernelGGL((pt_wf_trace_refill<true, false, 64u>), dim3(blocks), dim3(64), 0, stream, S, B, it);
                else hipLaunchKernelGGL((pt_wf_trace_refill<false, false, 64u>), dim3(blocks), dim3(64), 0, stream, S, B, it);
            }
            if (PT_WF_SUSPEND > 0u) {
                const uint32_t tailBlocks = (blocks + PT_WF_TAIL_GROUP - 1u) / PT_WF_TAIL_GROUP;
                if (wide) {
                    if (fullStats) hipLaunchKernelGGL((pt_wf_trace_refill<true, true, 128u>), dim3(tailBlocks), dim3(64), 0, stream, S, B, it);
                    else hipLaunchKernelGGL((pt_wf_trace_refill<false, true, 128u>), dim3(tailBlocks), dim3(64), 0, stream, S, B, it);
                } else {
                    if (fullStats) hipLaunchKernelGGL((pt_wf_trace_refill<true, true, 64u>), dim3(tailBlocks), dim3(64), 0, stream, S, B, it);
                    else hipLaunchKernelGGL((pt_wf_trace_refill<false, true, 64u>), dim3(tailBlocks), dim3(64), 0, stream, S, B, it);
                }
                launches++;
            }
        } else if (tlas && tlasRefill) {
            if (PT_WF_TLAS_WG_WAVES > 1u && S.tlasNodeCount <= 0xFFFFu) {     // 16-bit TLAS stack entries
                const uint32_t wgs = (refillBlocks + PT_WF_TLAS_WG_WAVES - 1u) / PT_WF_TLAS_WG_WAVES;
                if (fullStats) hipLaunchKernelGGL((pt_wf_trace_refill_tlas<true, PT_WF_TLAS_WG_WAVES>), dim3(wgs), dim3(64u * PT_WF_TLAS_WG_WAVES), 0, stream, S, B, it);
                else hipLaunchKernelGGL((pt_wf_trace_refill_tlas<false, PT_WF_TLAS_WG_WAVES>), dim3(wgs), dim3(64u * PT_WF_TLAS_WG_WAVES), 0, stream, S, B, it);
            } else {
                if (fullStats) hipLaunchKernelGGL((pt_wf_trace_refill_tlas<true, 1u>), dim3(refillBlocks), dim3(64), 0, stream, S, B, it);
                else hipLaunchKernelGGL((pt_wf_trace_refill_tlas<false, 1u>), dim3(refillBlocks), dim3(64), 0, stream, S, B, it);
            }
        } else if (tlas) {
            if (fullStats) hipLaunchKernelGGL((pt_wf_trace<true, true>), dim3(nb * 3u), dim3(256), 0, stream, S, B, it);
            else hipLaunchKernelGGL((pt_wf_trace<false, true>), dim3(nb * 3u), dim3(256), 0, stream, S, B, it);
        } else {
            if (fullStats) hipLaunchKernelGGL((pt_wf_trace<true, false>), dim3(nb * 3u), dim3(256), 0, stream, S, B, it);
            else hipLaunchKernelGGL((pt_wf_trace<false, false>), dim3(nb * 3u), dim3(256), 0, stream, S, B, it);
        }
        if (B.shadeSorted == 2u) {
            hipLaunchKernelGGL(pt_wf_shade_pre, dim3(nb), dim3(256), 0, stream, S, B, it);
            if (fullStats) {
                hipLaunchKernelGGL(pt_wf_shade_hit<true>, dim3(B.numSlots / 64u), dim3(64), 0, stream, S, P, B, it);
                hipLaunchKernelGGL(pt_wf_shade_post<true>, dim3(nb), dim3(256), 0, stream, P, tm, B, it);
            } else {
                hipLaunchKernelGGL(pt_wf_shade_hit<false>, dim3(B.numSlots / 64u), dim3(64), 0, stream, S, P, B, it);
                hipLaunchKernelGGL(pt_wf_shade_post<false>, dim3(nb), dim3(256), 0, stream, P, tm, B, it);
            }
            launches += 2;
        } else if (B.shadeSorted) {
            if (fullStats) hipLaunchKernelGGL(pt_wf_shade_sorted<true>, dim3(nb), dim3(256), 0, stream, S, P, tm, B, it);
            else hipLaunchKernelGGL(pt_wf_shade_sorted<false>, dim3(nb), dim3(256), 0, stream, S, P, tm, B, it);
        } else {
            if (fullStats) hipLaunchKernelGGL(pt_wf_shade<true>, dim3(B.numSlots / PT_WF_SHADE_BLOCK), dim3(PT_WF_SHADE_BLOCK), 0, stream, S, P, tm, B, it);
            else hipLaunchKernelGGL(pt_wf_shade<false>, dim3(B.numSlots / PT_WF_SHADE_BLOCK), dim3(PT_WF_SHADE_BLOCK), 0, stream, S, P, tm, B, it);
        }
        launches += 2;
    }
    const uint32_t cleanupBlocks = nb < 1024u ? nb : 1024u;          // 256 CUs x 4 workgroups; each strides over the slot blocks
    if (tlas) {
        if (fullStats) hipLaunchKernelGGL((pt_wf_cleanup<true, true>), dim3(cleanupBlocks), dim3(256), 0, stream, S, P, tm, B);
        else hipLaunchKernelGGL((pt_wf_cleanup<false, true>), dim3(cleanupBlocks), dim3(256), 0, stream, S, P, tm, B);
    } else {
        if (fullStats) hipLaunchKernelGGL((pt_wf_cleanup<true, false>), dim3(cleanupBlocks), dim3(256), 0, stream, S, P, tm, B);
        else hipLaunchKernelGGL((pt_wf_cleanup<false, false>), dim3(cleanupBlocks), dim3(256), 0, stream, S, P, tm, B);
    }
    if (orderAfter && (e = hipStreamWaitEvent(stream, orderAfter, 0)) != hipSuccess) return e;
    if (zeroOutputFirst &&
        (e = hipMemsetAsync(output, 0, (size_t)P.OutputWidth * P.OutputHeight * sizeof(float4), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(pt_wf_resolve, dim3(nbPass), dim3(256), 0, stream, P, batch, tm, B, (const float4*)B.color, accumulated, output);
    hipLaunchKernelGGL(pt_wf_fold_rows, dim3(256), dim3(256), 0, stream, B.statRows, B.numStatRows, gstats);
    launches += 3;
    if (launchesOut) *launchesOut = launches;
    return hipGetLastError();
}
