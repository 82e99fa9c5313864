# Runs on the GPU box: per-stage launch times against the batch size: slope = steady state, intercept = what a launch loses at its ends
set -o pipefail
for b in 768 1024 1536 2048 3072 4096 8192; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --batch $b --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages']
print('B=$b', 'step', d['ms_per_step'], ' '.join('%s %.4f' % (k, (v['ms'] if isinstance(v, dict) else v)) for k, v in s.items()))" || exit 1
done
