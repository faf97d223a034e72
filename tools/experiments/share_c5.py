"""BASELINE.json configs[4] as one rank of eight sees it, on ONE GPU: Sponza-class, 3840x2160, 64 spp = 8 passes of 8 spp issued as
ONE launch sequence (PTRenderPassBatchTo), rank 0's tiles only, no exchange: ms per 64-spp frame and the aggregate rate eight
such GPUs would give before the gather."""
import sys, time, os; sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=1024)
W, H, SPP = 3840, 2160, 8
def seed_of(k): return (0x12345678 + k*0x9E3779B9) & 0xFFFFFFFF
for world, batch in ((8, 8), (8, 1), (1, 1)):
    pt = PathTracer(s, width=W, height=H, samplesPerPass=SPP, rank=0, world_size=world, schedule=1)
    if world == 1: pt.set_passes_in_flight(6)
    frames = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    def run(k0, n, cur):
        k = k0
        while k < k0 + n:
            ps = [scenes.frame_params(s, W, H, spp=SPP, current_sample=SPP*(k+j), seed=seed_of(k+j)) for j in range(batch)]
            pt.render_batch_to(ps, frames[cur].data_ptr(), frames[1-cur].data_ptr() if k > 0 else 0)
            cur = 1 - cur; k += batch
        pt.synchronize()
        return cur
    cur = run(0, 16 if world == 1 else 48, 0)
    pt.reset_stats()
    K = 16 if world == 1 else 64
    t0 = time.perf_counter(); cur = run(1000, K, cur); wall = (time.perf_counter() - t0) * 1e3 / K
    rays = pt.stats().rays / K
    print(f"4K, world {world}, {batch} passes per sequence: {wall:.3f} ms per 8-spp pass on one rank = {wall*8:.1f} ms per 64-spp frame -> {rays/wall/1e3*world:.0f} Mrays/s aggregate", flush=True)
    pt.close(); del frames
