"""bench.py end to end on the GPU box: the JSON contract at N=1, and the N>1 code path rehearsed with two ranks that
share cuda:0 (gloo stands in for RCCL, which refuses two ranks on one device).  Tile sharding must not change a bit of
the assembled frame: frame_crc32 is the same for N=1 and N=2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--workload", "cornell", "--width", "320", "--height", "200", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]


def _last_json(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_contract_and_two_rank_rehearsal():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL,
                                             env=env, timeout=600, stderr=subprocess.STDOUT))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in one
    assert one["n_gpus"] == 1 and one["steps"] == 3 and one["value"] > 0 and one["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(one["roofline"])
    assert one["roofline"]["achieved"] > 0

    two = _last_json(subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device"] + SMALL,
        env=env, timeout=900, stderr=subprocess.STDOUT))
    assert two["n_gpus"] == 2 and two["value"] > 0
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]          # strong scaling: same frame, same rays
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"] != 0


@pytest.mark.gpu
def test_bench_rccl_code_path_with_one_rank():
    """The N>1 path with the real backend (nccl = RCCL): process group on the device, per-frame reduce on the context's
    stream, barriers.  One rank is all a one-GPU box allows RCCL; the frame must equal the plain N=1 frame."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL,
                                               env=env, timeout=600, stderr=subprocess.STDOUT))
    forced = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist"] + SMALL,
                                                env=env, timeout=600, stderr=subprocess.STDOUT))
    assert forced["config"]["frame_crc32"] == plain["config"]["frame_crc32"] != 0
    assert "RCCL" in forced["config"]["sharding"] and forced["value"] > 0
