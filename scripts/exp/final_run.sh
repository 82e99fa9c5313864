set -o pipefail
TAG=$1
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/gputest_$TAG.log 2>&1 || { tail -20 gpurun_out/gputest_$TAG.log; exit 1; }
tail -2 gpurun_out/gputest_$TAG.log
bash scripts/profile_round.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_$TAG.log; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline'], d['config'].get('secondary',{}).get('values', d['config'].get('secondary')))
"
