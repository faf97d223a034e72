import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import pyoracle as oracle
from unity_webgpu_pathtracer_amd import scenes, plugin, abi
from test_bvh_builder_gpu import _rays
s = scenes.make_scene("sponza", tex_size=4, detail=0.15)
ref_nodes, ref_tris = plugin.build_cwbvh(s.vertices)
for rep in range(3):
    dev_nodes, dev_tris = plugin.build_cwbvh(s.vertices, device=0)
    rays = _rays(s, 20000, seed=11)
    t_ref, p_ref, st_ref = oracle.trace(oracle.SceneBuffers(s, ref_nodes, ref_tris), rays)
    t_dev, p_dev, st_dev = oracle.trace(oracle.SceneBuffers(s, dev_nodes, dev_tris), rays)
    bad = np.nonzero(t_ref != t_dev)[0]
    print("rep", rep, "differ", len(bad), "dev missed (t larger):", int((t_dev[bad] > t_ref[bad]).sum()), "dev closer:", int((t_dev[bad] < t_ref[bad]).sum()), "mean steps ref/dev", st_ref.mean(), st_dev.mean(), "max", st_ref.max(), st_dev.max())
    for k in bad[:5]: print("   ray", k, t_ref[k], t_dev[k], p_ref[k], p_dev[k])
n = np.frombuffer(dev_nodes.tobytes(), abi.CWBVH_NODE)
print("nodes", len(n), "tris", s.tri_count)
