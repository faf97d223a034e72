import sys, os, subprocess, json, glob
# run bench main workload only for each variant .so in tools/experiments/variants
for so in sorted(glob.glob("tools/experiments/variants/*.so")):
    env = dict(os.environ, PT_PLUGIN=os.path.abspath(so))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--no-extra", "--no-cpu-baseline", "--no-alg-replay"] + sys.argv[1:], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1]); print(os.path.basename(so), "Mrays/s", j["value"], "ms", j["ms_per_step"], flush=True)
    else:
        print(os.path.basename(so), "FAILED", out.stderr[-500:], flush=True)
