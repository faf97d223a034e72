#!/usr/bin/env python3
"""Times the kernels of the §8(f) rows built after the hot path (presentation blit, mesh processing, texture copy, environment
map) at production sizes, for `rocprofv3 --kernel-trace --stats`.  Prints one JSON line with host-side timings; the per-kernel
durations come from the profiler (profiles/r01_next_rows_stats.csv)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from unity_webgpu_pathtracer_amd import abi, ingest, plugin, scenes  # noqa: E402
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer  # noqa: E402


def main():
    import torch
    out = {}
    # presentation blit, 4K frame resident in HBM
    W, H = 3840, 2160
    pt = PathTracer(scenes.cornell_box(), width=W, height=H)
    src = torch.rand((H, W, 4), dtype=torch.float32, device="cuda") * 4.0
    dst = torch.zeros_like(src)
    torch.cuda.synchronize()
    q = pt.present_params()
    for mode in (4, 1):
        q.Mode = mode
        for _ in range(3):
            pt.present_to(q, src.data_ptr(), dst.data_ptr())
        pt.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pt.present_to(q, src.data_ptr(), dst.data_ptr())
        pt.synchronize()
        dt = (time.perf_counter() - t0) / 20
        out[f"present_4k_mode{mode}"] = {"ms": round(dt * 1e3, 4), "GB/s": round(W * H * 32 / dt / 1e9, 1)}
    # ingestion: a 250k-triangle indexed mesh with all attributes, and 24 1024x1024 textures
    rng = np.random.RandomState(1)
    nv, nt = 150000, 250000
    m = ingest.Mesh(rng.normal(0, 1, (nv, 3)).astype(np.float32), rng.normal(0, 1, (nv, 3)).astype(np.float32),
                    rng.normal(0, 1, (nv, 3)).astype(np.float32), rng.uniform(0, 1, (nv, 2)).astype(np.float32),
                    rng.randint(0, nv, nt * 3).astype(np.uint32), np.diag([2.0, 1.0, 0.5, 1.0]), 0)
    t0 = time.perf_counter()
    pos, attrs = ingest.process_meshes(pt.ctx, [m])
    out["process_meshes_250k"] = {"ms_incl_pcie": round((time.perf_counter() - t0) * 1e3, 2), "tris": nt}
    imgs = [(rng.uniform(0, 1, (1024, 1024, 4)).astype(np.float32), True) for _ in range(4)]
    t0 = time.perf_counter()
    td = ingest.copy_texture_data(pt.ctx, imgs)
    out["copy_texture_4x1024"] = {"ms_incl_pcie": round((time.perf_counter() - t0) * 1e3, 2), "uints": int(td.size)}
    pt.close()
    # environment-map scene, 1080p / 8 spp passes
    s = scenes.material_zoo(env_map=(2048, 1024))
    pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8)
    for k in range(2):
        pt.OnRenderImage(1000 + k)
    pt.synchronize()
    pt.reset_stats()
    t0 = time.perf_counter()
    for k in range(4):
        pt.OnRenderImage(2000 + k)
    pt.synchronize()
    dt = (time.perf_counter() - t0) / 4
    out["zoo_env_1080p_8spp"] = {"ms_per_pass": round(dt * 1e3, 3), "Mrays/s": round(pt.stats().rays / 4 / dt / 1e6, 1)}
    pt.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
