"""One rank's share of a sharded 1080p / 8 spp frame on ONE GPU (rank 0 of `world`, no exchange), passes issued in batches
(PTRenderPassBatchTo): ms per pass and the aggregate rate `world` such GPUs would give before the gather.
usage: python tools/experiments/share_batch.py [tex_size]"""
import sys, time, os; sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
tex = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s = scenes.make_scene("sponza", tex_size=tex)
W, H, SPP = 1920, 1080, 8
def seed_of(k): return (0x12345678 + k*0x9E3779B9) & 0xFFFFFFFF
K = 48
for world in (1, 2, 4, 8):
    for batch in (1, 2, 4, 8):
        if world == 1 and batch > 2: continue
        pt = PathTracer(s, width=W, height=H, samplesPerPass=SPP, rank=0, world_size=world, schedule=1)
        frames = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        def run(k0, n, cur):
            k = k0
            while k < k0 + n:
                ps = [scenes.frame_params(s, W, H, spp=SPP, current_sample=SPP*(k+j), seed=seed_of(k+j)) for j in range(batch)]
                pt.render_batch_to(ps, frames[cur].data_ptr(), frames[1-cur].data_ptr() if k > 0 else 0)
                cur = 1 - cur; k += batch
            pt.synchronize()
            return cur
        cur = run(0, 16 * batch if batch < 4 else 32, 0)
        pt.reset_stats()
        t0 = time.perf_counter(); cur = run(1000, K, cur); wall = (time.perf_counter() - t0) * 1e3 / K
        rays = pt.stats().rays / K
        print(f"world {world} batch {batch}: {wall:.3f} ms/pass on one rank -> {rays/wall/1e3*world:.0f} Mrays/s aggregate", flush=True)
        pt.close()
        del frames
