#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s of the path-tracing hot path at 1920x1080, 8 spp (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1 under a launcher (WORLD_SIZE set, the driver's `python -m torch.distributed.run ... bench.py --gpus N`): this process is
one rank.  N > 1 WITHOUT a launcher (`python bench.py --gpus 8`): bench.py starts `torch.distributed.run --nproc-per-node N`
on itself as a CHILD process, before anything here has touched HIP, and relays the child's JSON line and exit code.

A "step" is one progressive pass (one DispatchCompute of the reference, PathTracer.cs:251) over the whole
1920x1080 frame at 8 samples per pixel: step k uses CurrentSample = 8k, RngSeedRoot_k = 0x12345678 + k*0x9E3779B9
and the previous step's frame as AccumulatedOutput (SURVEY.md 8d, config C5's pass schedule).  Scene buffers and
frames are resident in HBM before the timed region.  One ray = one RayIntersectBvh call (closest-hit or NEE
shadow ray); the count comes from the kernels' own always-on counters, which the parity tests pin to the oracle's.

N > 1: one process per GPU; the frame is sharded as interleaved 16x16-pixel blocks (PTSetTileOwnership), every
rank renders only its blocks, packs them densely (PTPackOwnedTiles) and ONE RCCL gather per step moves the packed
tiles -- 1/N of the frame per rank -- to rank 0, which scatters them into the assembled frame (PTUnpackTiles).  No
float is added anywhere, so the frame is bit-identical for every N.  The total work is fixed as N grows ->
"scaling": "strong".

Rank 0 prints ONE JSON line.  `roofline` prices the render kernel against the 8 TB/s HBM peak with ALGORITHMIC
bytes (SURVEY.md 8d formula over the kernel's own node/triangle/attribute/material/texel counters) divided by the
kernel's mean duration measured with HIP events on the stream it is launched on.  `cpu_baseline` times the
reference's tinybvh CPU walkers (oracle/_ref, compiled from /root/reference) on a bounded sample of the very rays
of this workload, on the host cores of this box.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

# consecutive passes overlap on several streams; the HIP runtime maps streams onto this many hardware queues (default 4) and
# streams that share a queue serialise.  The host's choice, made before the first HIP call (the plugin never touches the
# environment; it sizes its default number of passes in flight to this variable, PTSetPassesInFlight).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from unity_webgpu_pathtracer_amd import distributed as dist_helpers  # noqa: E402
from unity_webgpu_pathtracer_amd import scenes  # noqa: E402
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
SEED0, SEED_STEP = 0x12345678, 0x9E3779B9

WORKLOADS = {
    # BASELINE.json configs[3]: the configuration the >= 1 Gray/s target is quoted on
    "sponza": dict(scene="sponza", kw=dict(tex_size=1024, detail=1.0), desc="Sponza-class atrium (~250k tris, 24 textured Disney materials, 2 rect lights + sky)"),
    "bunny": dict(scene="bunny", kw={}, desc="bunny-class displaced icosphere (69,620 tris) in the Cornell box"),
    "cornell": dict(scene="cornell", kw={}, desc="Cornell box (8 tris, 1 rectangle light)"),
    # the HAS_TLAS variant (util/tlas.hlsl) -- 200 instances of three small meshes over a 2-wide TLAS
    "instanced": dict(scene="instanced", kw=dict(count=200, detail=48), desc="HAS_TLAS: 200 instances (924,888 instanced triangles) of three meshes, rotations and non-uniform scales"),
}


def seed_of(step):
    return (SEED0 + step * SEED_STEP) & 0xFFFFFFFF


def cpu_baseline(scene, width, height, spp, pt, budget_rays=1_500_000):
    """Reference tinybvh traversal (BVH::Intersect / IsOccluded, scalar build as the plugin ships it) over the
    exact rays the oracle traces for a bounded sample of this workload's first pass."""
    from oracle import pyoracle     # checker: allowed here (cpu_baseline leg only)
    ref = pyoracle.load_ref_tinybvh()
    if ref is None:
        return None
    b = pyoracle.SceneBuffers(scene, pt._bvhScene.bvh_nodes, pt._bvhScene.bvh_tris)
    p = scenes.frame_params(scene, width, height, spp=spp, current_sample=0, seed=seed_of(0))
    # sample: a 3x3 grid of 48x24-pixel windows spread over the frame
    rays = []
    t0 = time.time()
    ww, wh = min(48, width), min(24, height)
    for gy in range(3):
        for gx in range(3):
            x0 = int((gx + 0.5) * width / 3) - ww // 2
            y0 = int((gy + 0.5) * height / 3) - wh // 2
            _, _, dump = pyoracle.render(b, p, window=(x0, y0, x0 + ww, y0 + wh), ray_dump_capacity=budget_rays // 9)
            rays.append(dump)
    rays = np.concatenate(rays)
    gen_s = time.time() - t0
    n = rays.shape[0]
    flat = np.zeros((n, 8), dtype=np.float32)
    flat[:, 0:3] = rays["origin"]
    flat[:, 3:6] = rays["direction"]
    flat[:, 6] = rays["tmax"]
    flat[:, 7] = rays["kind"]
    verts = np.ascontiguousarray(scene.vertices, dtype=np.float32)
    build_ms = C.c_double()
    h = ref.ref_bvh_build(verts.ctypes.data_as(C.c_void_p), scene.tri_count, C.byref(build_ms))
    cores = max(1, min(os.cpu_count() or 1, 64))
    ref.ref_bvh_trace(h, flat.ctypes.data_as(C.c_void_p), n, cores, None, None)         # warm
    total, reps = 0.0, 0
    while total < 3.0 and reps < 50:
        total += ref.ref_bvh_trace(h, flat.ctypes.data_as(C.c_void_p), n, cores, None, None)
        reps += 1
    one = ref.ref_bvh_trace(h, flat.ctypes.data_as(C.c_void_p), min(n, 200_000), 1, None, None)
    ref.ref_bvh_destroy(h)
    out = {
        "value": round(n * reps / total / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "reference",
        "sample": (f"{n} rays (closest-hit + NEE shadow) traced by the oracle for nine {ww}x{wh}-pixel windows of pass 0 of this "
                   f"workload, x{reps} repeats; tinybvh BVH::Intersect/IsOccluded, scalar binned-SAH build {build_ms.value:.0f} ms; "
                   f"single thread {min(n, 200_000) / one / 1e6:.3f} Mrays/s; oracle ray generation {gen_s:.1f} s"),
    }
    # second leg: the reference's OWN CWBVH walker, BVH8_CWBVH::Intersect (tiny_bvh.h:7006-7114; shadow rays go
    # through its FALLBACK_SHADOW_QUERY = a full Intersect), over the IDENTICAL ray list
    cw = pyoracle.load_ref_cwbvh()
    if cw is not None:
        bms = C.c_double()
        h2 = cw.ref_cwbvh_build(verts.ctypes.data_as(C.c_void_p), scene.tri_count, C.byref(bms))
        cw.ref_cwbvh_trace(h2, flat.ctypes.data_as(C.c_void_p), n, cores, None)          # warm
        tot2, reps2 = 0.0, 0
        while tot2 < 3.0 and reps2 < 50:
            tot2 += cw.ref_cwbvh_trace(h2, flat.ctypes.data_as(C.c_void_p), n, cores, None)
            reps2 += 1
        one2 = cw.ref_cwbvh_trace(h2, flat.ctypes.data_as(C.c_void_p), min(n, 200_000), 1, None)
        cw.ref_cwbvh_destroy(h2)
        out["cwbvh_walker"] = {"value": round(n * reps2 / tot2 / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "reference",
                               "single_thread": round(min(n, 200_000) / one2 / 1e6, 3),
                               "sample": f"the same {n} rays x{reps2}; tinybvh BVH8_CWBVH::Intersect (scalar code in the AVX-enabled TU, "
                                         f"scalar build pipeline {bms.value:.0f} ms, -ffp-contract=off)"}
    # context: the full CPU oracle render of BASELINE.json configs[0] (Cornell box 256x256, 1 spp), all threads
    try:
        c1 = scenes.make_scene("cornell")
        from unity_webgpu_pathtracer_amd import plugin as _plugin
        cn, ct = _plugin.build_cwbvh(c1.vertices)
        cb = pyoracle.SceneBuffers(c1, cn, ct)
        cp = scenes.frame_params(c1, 256, 256, spp=1, current_sample=0, seed=SEED0)
        t1 = time.time()
        _, cst = pyoracle.render(cb, cp)
        dt = time.time() - t1
        out["oracle_c1"] = {"seconds": round(dt, 4), "mrays_per_s": round(cst.rays / dt / 1e6, 3),
                            "what": "CPU oracle (port of the whole shader, all host threads), Cornell box 256x256, 1 spp"}
    except Exception as e:
        out["oracle_c1"] = {"error": str(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="sponza", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--schedule", type=int, default=-1, help="-1 auto, 0 megakernel, 1 wavefront (refill trace + shade launches), 2/3 variants, 4 fused persistent wavefront")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (default); gloo only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: run the N>1 code path (process group, per-frame reduce) even with one rank")
    ap.add_argument("--wf-iterations", type=int, default=0, help="wavefront trace+shade iterations per pass before the cleanup kernel (0 = automatic; any value gives the same frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alg-replay", action="store_true", help="profiling runs: skip the untimed full-counter replay (roofline.achieved is then null)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads reported under 'other_workloads'")
    ap.add_argument("--passes-in-flight", type=int, default=0, help="PTSetPassesInFlight: 0 = library default for GPU_MAX_HW_QUEUES, 1 = passes back to back")
    ap.add_argument("--latency-sub-frames", type=int, default=0, help="PTSetSubFrames for the latency leg (0 = the library's passes in flight, at most 6)")
    ap.add_argument("--sub-frames", type=int, default=1, help="PTSetSubFrames for the timed (pipelined) run")
    ap.add_argument("--latency-steps", type=int, default=6, help="passes of the latency leg (host synchronises after every pass); 0 = skip")
    ap.add_argument("--build-device", action="store_true", help="also render over the tree PTBuildBVHDevice builds on the GPU (reported under 'device_built_tree')")
    ap.add_argument("--batch", type=int, default=0, help="passes per launch sequence (PTRenderPassBatchTo, 1..8): the steps are issued in groups of this many -- same frames, "
                                                         "larger launches.  0 (default) = 1 on one GPU, min(N, 8) on N GPUs, where one rank's share of ONE pass is too small "
                                                         "a launch to fill an MI355X (1/8 of a 1080p pass: 2.92 ms alone, 2.28 ms per pass in batches of eight)")
    ap.add_argument("--master-port", type=int, default=29671, help="rendezvous port when bench.py launches its own ranks")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the ranks ourselves, as a child, BEFORE this process touches HIP
    # (a process that has initialised the GPU must never exec another program; a child process is fine)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_dist:
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        text = child.stdout.decode(errors="replace")
        lines = [l for l in text.splitlines() if l.startswith("{")]
        sys.stdout.write(lines[-1] + "\n" if lines else text)
        sys.stdout.flush()
        sys.exit(child.returncode)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W, H, SPP = args.width, args.height, args.spp
    if args.batch <= 0:
        args.batch = 1 if world == 1 else min(world, 8)
    args.batch = min(args.batch, 8)

    def run_workload(name, steps, warmup, full, build_device=None):
        wl = WORKLOADS[name]
        scene = scenes.make_scene(wl["scene"], **wl["kw"])
        pt = PathTracer(scene, device=local_rank, width=W, height=H, samplesPerPass=SPP, maxRayBounces=args.bounces,
                        rank=rank, world_size=world, build_device=build_device)
        pt.set_schedule(args.schedule)
        if args.passes_in_flight:
            pt.set_passes_in_flight(args.passes_in_flight)
        if args.sub_frames > 1:
            pt.set_sub_frames(args.sub_frames)
        if args.wf_iterations:
            pt.set_wavefront_iterations(args.wf_iterations)
        frames = [torch.zeros((H, W, 4), dtype=torch.float32, device=device) for _ in range(2)]
        # N > 1: the rank's two frames keep its own tiles (they are the next pass's AccumulatedOutput); the frame is
        # assembled on rank 0 in a third buffer from every rank's packed tiles
        assembled = torch.zeros((H, W, 4), dtype=torch.float32, device=device) if (distributed and rank == 0) else None
        scratch = {}
        ext = torch.cuda.ExternalStream(pt.stream(), device=device)
        torch.cuda.synchronize(device)

        def step(k, cur, count=1):
            """passes k .. k + count - 1 into frames[cur] (accumulated frame: frames[1 - cur])"""
            ps = [scenes.frame_params(scene, W, H, spp=SPP, current_sample=(k + j) * SPP, seed=seed_of(k + j), max_bounces=args.bounces) for j in range(count)]
            p = ps[-1]
            if count == 1:
                pt.render_pass_to(p, frames[cur].data_ptr(), frames[1 - cur].data_ptr() if k > 0 else 0)
            else:
                pt.render_batch_to(ps, frames[cur].data_ptr(), frames[1 - cur].data_ptr() if k > 0 else 0)
            if distributed:
                with torch.cuda.stream(ext):                                  # pack -> RCCL gather over xGMI -> unpack, once per frame
                    dist_helpers.gather_frame(frames[cur], rank, world, dst=0, assembled=assembled, pt=pt, params=p, scratch=scratch)

        def sync_all():
            pt.synchronize()
            torch.cuda.synchronize(device)
            if distributed:
                dist.barrier()
                torch.cuda.synchronize(device)

        def run_steps(k0, k1, cur):
            """steps k0 .. k1 - 1 in groups of --batch passes (the last group may be shorter)"""
            k = k0
            while k < k1:
                n = min(max(1, args.batch), k1 - k)
                step(k, cur, n)
                cur = 1 - cur
                k += n
            return cur

        # untimed priming, not steps: every state set the timed region will use is allocated and touched (one full batch each),
        # and the GPU has left its idle clocks (the first run on a fresh box used to read up to 15 % low); pass 0 of the warm-up
        # overwrites the frame this leaves
        ps0 = [scenes.frame_params(scene, W, H, spp=SPP, current_sample=0, seed=seed_of(1000 + j), max_bounces=args.bounces) for j in range(max(1, args.batch))]
        for _ in range(max(12, pt.passes_in_flight())):
            if len(ps0) > 1:
                pt.render_batch_to(ps0, frames[0].data_ptr(), 0)
            else:
                pt.render_pass_to(ps0[0], frames[0].data_ptr(), 0)
        sync_all()
        cur = run_steps(0, warmup, 0)
        sync_all()
        pt.reset_stats()
        pt.set_profiling(True)
        pt.reset_timings()
        sync_all()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(ext)              # HIP events on the context's own stream, which every pass joins when it completes
        cur = run_steps(warmup, warmup + steps, cur)
        ev1.record(ext)
        sync_all()
        elapsed = time.perf_counter() - t0
        region_ms = ev0.elapsed_time(ev1)
        st = pt.stats()
        tm = pt.timings()
        pt.set_profiling(False)
        final = assembled if distributed else frames[1 - cur]
        crc = zlib.crc32(final.cpu().numpy().tobytes()) if (full and rank == 0) else 0   # identical for every N: owned tiles sum with zeros
        rays = torch.tensor([st.rays, st.closestHitRays, st.shadowRays, st.paths], dtype=torch.float64, device=device)
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        total_rays = float(rays[0].item())
        res = {
            "mrays_per_s": total_rays / elapsed / 1e6,
            "ms_per_step": elapsed / steps * 1e3,
            "rays_per_step": total_rays / steps,
            "closest_hit_rays": float(rays[1].item()) / steps, "shadow_rays": float(rays[2].item()) / steps,
            "msamples_per_s": float(rays[3].item()) / elapsed / 1e6,
            "kernel_ms": region_ms / steps,                        # device time per pass over the timed region (passes may overlap)
            "pass_span_ms": tm.kernelMsTotal / max(1, tm.passes),  # first-launch-to-resolve span of one pass
            "tris": scene.tri_count, "frame_crc32": crc,
        }
        if full and rank == 0 and not args.no_alg_replay:
            # algorithmic bytes: replay the same timed steps once, untimed, with the full-counter kernel variant
            pt.set_stats_level(1)
            pt.reset_stats()
            c2 = cur
            for k in range(warmup, warmup + steps):
                p = scenes.frame_params(scene, W, H, spp=SPP, current_sample=k * SPP, seed=seed_of(k), max_bounces=args.bounces)
                pt.render_pass_to(p, frames[c2].data_ptr(), frames[1 - c2].data_ptr() if k > 0 else 0)
                c2 = 1 - c2
            pt.synchronize()
            fs = pt.stats()
            pt.set_stats_level(0)
            res["alg_bytes_per_launch"] = fs.algorithmic_bytes() / steps
            res["counters_per_launch"] = {k: v / steps for k, v in fs.as_dict().items() if k not in ("maxStackDepth", "stackOverflows")}
            res["max_stack_depth"] = int(fs.maxStackDepth)
        if distributed and full:
            dist.barrier()
        if full and rank == 0 and not distributed and args.latency_steps > 0:
            # latency leg: what a host gets that consumes every frame before it asks for the next (the reference presents each
            # pass: OnRenderImage -> DispatchCompute -> Blit, PathTracer.cs:251-272): one pass at a time, host sync after each
            pt.set_stats_level(0)
            pt.synchronize()
            pt.reset_stats()
            sub = args.latency_sub_frames or min(4, pt.passes_in_flight())
            pt.set_sub_frames(max(1, sub))
            c2, t_lat = cur, []
            for k in range(warmup + steps, warmup + steps + args.latency_steps):
                p = scenes.frame_params(scene, W, H, spp=SPP, current_sample=k * SPP, seed=seed_of(k), max_bounces=args.bounces)
                t1 = time.perf_counter()
                pt.render_pass_to(p, frames[c2].data_ptr(), frames[1 - c2].data_ptr() if k > 0 else 0)
                pt.synchronize()
                t_lat.append(time.perf_counter() - t1)
                c2 = 1 - c2
            ls = pt.stats()
            pt.set_sub_frames(max(1, args.sub_frames))
            lat = sorted(t_lat)[len(t_lat) // 2]
            res["latency_mode"] = {"ms_per_pass": round(lat * 1e3, 4), "mrays_per_s": round(ls.rays / len(t_lat) / lat / 1e6, 3),
                                   "passes": len(t_lat), "sub_frames": max(1, sub), "ms_per_pass_all": [round(t * 1e3, 3) for t in t_lat],
                                   "what": "median wall time of one pass when the host synchronises after every pass (enqueue + "
                                           "device + hipStreamSynchronize): one pass in flight, as under the reference's "
                                           "OnRenderImage -> Blit (PathTracer.cs:251-272); `value` keeps several passes in flight"}
        res["schedule"] = pt.schedule()
        res["passes_in_flight"] = pt.passes_in_flight()
        res["_pt"], res["_scene"] = pt, scene
        return res

    main_res = run_workload(args.workload, args.steps, args.warmup, full=True)
    out = None
    if rank == 0:
        kernel_s = main_res["kernel_ms"] / 1e3
        main_res.setdefault("alg_bytes_per_launch", 0.0)
        main_res.setdefault("counters_per_launch", {})
        main_res.setdefault("max_stack_depth", 0)
        achieved = main_res["alg_bytes_per_launch"] / kernel_s / 1e9 if kernel_s > 0 else 0.0
        traffic, traffic_source = None, None
        tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tj):
            try:
                rec = json.load(open(tj))
                if rec.get("workload") == args.workload and rec.get("n_gpus", 1) == world:
                    traffic = rec.get("bytes_per_launch")
                    # NOT measured in this run: counters need rocprofv3; the committed summary of the same command is quoted
                    traffic_source = "committed profile, not this run: " + rec.get("source", "profiles/hbm_traffic.json")
            except Exception:
                traffic = None
        valu = None
        vj = os.path.join(ROOT, "profiles", "valu_insts.json")
        if os.path.exists(vj):
            try:
                rec = json.load(open(vj))
                if rec.get("workload") == args.workload and rec.get("n_gpus", 1) == world and kernel_s > 0:
                    # second roofline: wave-level VALU instructions per pass (SQ_INSTS_VALU, committed profile) against the
                    # MEASURED wave64 issue peak (tools/microbench/valu_issue.hip) over the kernel time measured NOW
                    valu = {"wave_insts_per_launch": rec["valu_wave_insts_per_launch"], "issue_peak_per_s": rec["issue_peak_wave_insts_per_s"],
                            "frac": round(rec["valu_wave_insts_per_launch"] / kernel_s / rec["issue_peak_wave_insts_per_s"], 4),
                            "valu_source": "committed profile, not this run: " + rec.get("source", "profiles/valu_insts.json"),
                            "peak_source": rec.get("issue_peak_source", "")}
            except Exception:
                valu = None
        # the committed rocprofv3 summary of THIS command (profiles/collect.sh): span-based time per pass with the passes pipelined
        # as here, and per-kernel durations from a second trace with one pass in flight
        profile = None
        try:
            import glob
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))[-1]
            rec = json.load(open(pj))
            if rec.get("workload") == args.workload and "trace" in rec:
                ser = rec.get("serialised", {})
                profile = {"source": os.path.relpath(pj, ROOT) + " (profiles/collect.sh; not this run)",
                           "pipelined_ms_per_pass": round(rec["trace"].get("steady_ms_per_pass", 0.0), 4),
                           "pipelined_mean_kernels_in_flight": round(rec["trace"].get("mean_kernels_in_flight", 0.0), 2),
                           "serialised_ms_per_pass": round(ser.get("trace", {}).get("steady_ms_per_pass", 0.0), 4),
                           "serialised_kernel_ms_per_pass": {k: round(v["total_per_pass"], 4) for k, v in ser.get("kernel_ms", {}).items() if v["total_per_pass"] >= 0.01},
                           "lane_utilisation": {k: round(v, 3) for k, v in rec.get("pmc", {}).get("lane_utilisation", {}).items() if "trace" in k or "shade" in k}}
        except Exception:
            profile = None
        out = {
            "metric": "Mrays/s at 1920x1080, 8 spp", "value": round(main_res["mrays_per_s"], 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(main_res["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]['desc']}, {W}x{H}, {SPP} spp/pass, depth {args.bounces}, "
                                   f"progressive passes (CurrentSample = 8k)",
                       "tris": main_res["tris"], "rays_per_step": round(main_res["rays_per_step"]),
                       "msamples_per_s": round(main_res["msamples_per_s"], 3), "schedule": main_res["schedule"],
                       "passes_in_flight": main_res["passes_in_flight"], "passes_per_launch_sequence": max(1, args.batch),
                       "frame_crc32": main_res["frame_crc32"],
                       "sharding": "interleaved 16x16 blocks, one RCCL gather of owned tiles per frame" if distributed else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": {0: "pt_megakernel", 1: "pt_wf_trace_refill (main + tail) + pt_wf_shade (whole pass)", 2: "pt_wf_trace + pt_wf_shade (whole pass)",
                                    3: "pt_wf_trace_persist + pt_wf_shade (whole pass)", 4: "pt_wf_fused (one persistent launch per pass)"}[main_res["schedule"]],
                         "kernel_ms": round(main_res["kernel_ms"], 4), "pass_span_ms": round(main_res["pass_span_ms"], 4),
                         "algorithmic_bytes_per_launch": round(main_res["alg_bytes_per_launch"]),
                         "counters_per_launch": {k: round(v) for k, v in main_res["counters_per_launch"].items()},
                         "max_stack_depth": main_res["max_stack_depth"],
                         "valu_issue": valu,
                         "binding_roof": ({"name": "valu_issue", "frac": valu["frac"],
                                           "why": "the scene (148 MB) is Infinity-Cache resident, so the 8 TB/s HBM roof above is nominal (SURVEY.md 8d defines it); "
                                                  "what the pass saturates first is VALU issue under divergence: wave instructions of a pass / measured issue peak"}
                                          if valu else None),
                         "profile": profile,
                         "note": "rank-0 kernels; achieved = algorithmic bytes of a pass / its device time (HIP events on the context stream, passes overlap); "
                                 "the scene is Infinity-Cache resident, so HBM is the nominal roof only: the pass is bound by VALU issue at 40-55 % lane "
                                 "utilisation (valu_issue.frac), with the texture-data path ~70 % busy (DESIGN.md 5.4)"},
        }
    if out is not None and "latency_mode" in main_res:
        out["latency_mode"] = main_res["latency_mode"]
    if args.build_device and not distributed and out is not None:
        # the same workload over the tree the MI355X builds itself (PTBuildBVHDevice: LBVH -> 8-wide collapse -> CWBVH encode),
        # against the reference-identical binned-SAH tree of the main run: throughput and traversal work per ray
        main_res["_pt"].close()
        main_res["_pt"] = None
        r = run_workload(args.workload, max(4, args.steps // 2), max(2, args.warmup), full=True, build_device=local_rank)
        c_dev, c_sah = r.get("counters_per_launch", {}), main_res.get("counters_per_launch", {})
        rays_dev, rays_sah = max(1.0, r["rays_per_step"]), max(1.0, main_res["rays_per_step"])
        out["device_built_tree"] = {
            "mrays_per_s": round(r["mrays_per_s"], 3), "ms_per_step": round(r["ms_per_step"], 4),
            "vs_sah_tree": round(r["mrays_per_s"] / main_res["mrays_per_s"], 4),
            "node_visits_per_ray": round(c_dev.get("nodeVisits", 0) / rays_dev, 3), "tri_tests_per_ray": round(c_dev.get("triTests", 0) / rays_dev, 3),
            "sah_node_visits_per_ray": round(c_sah.get("nodeVisits", 0) / rays_sah, 3), "sah_tri_tests_per_ray": round(c_sah.get("triTests", 0) / rays_sah, 3),
            "nodes": int(r["_pt"]._bvhScene.bvh_nodes.nbytes // 80), "build_ms": r["_pt"]._bvhScene.build_ms,
            "what": "PTBuildBVHDevice tree (hit-equal, not byte-equal, to BuildBVH's): frames differ only where ties are broken differently"}
        r["_pt"].close()
    if not args.no_extra and not distributed:
        extra = {}
        for name in WORKLOADS:
            if name == args.workload:
                continue
            r = run_workload(name, max(4, args.steps), max(2, args.warmup), full=False)
            r["_pt"].close()
            extra[name] = {"mrays_per_s": round(r["mrays_per_s"], 3), "ms_per_step": round(r["ms_per_step"], 4), "tris": r["tris"],
                           "schedule": r["schedule"]}
        if out is not None:
            out["other_workloads"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        wl = WORKLOADS[args.workload]
        scene = main_res.get("_scene")
        pt = main_res.get("_pt")
        if pt is None:
            pt = PathTracer(scene, device=local_rank, width=W, height=H, samplesPerPass=SPP)
        try:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, SPP, pt)
        except Exception as e:       # the checker being unavailable must not hide the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "reference", "sample": f"unavailable: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
