#!/bin/bash
# usage (GPU box): bash tools/experiments/sizes.sh -- Mrays/s by frame size and samples per pass (what a bigger launch / a longer pass is worth)
python3 bench.py --no-extra --no-cpu-baseline --no-alg-replay > /dev/null 2>&1
for cfg in "1920 1080 8 24" "1920 1080 64 4" "3840 2160 8 8" "3840 2160 64 2" "960 540 8 40" "1920 1080 16 12"; do
  set -- $cfg
  python3 bench.py --width $1 --height $2 --spp $3 --steps $4 --warmup 2 --no-extra --no-cpu-baseline --no-alg-replay > gpurun_out/size_tmp.json 2>/dev/null
  python3 -c "import json; j=json.load(open('gpurun_out/size_tmp.json')); print('$1x$2 spp $3:', j['value'], 'Mrays/s', j['ms_per_step'], 'ms/pass')"
done
