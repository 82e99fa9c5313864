#!/bin/bash
# GPU box: forward blend GEMM with / without the XCD-aware tile map: time (bench events) and L2-side fetch traffic (one PMC pass each).
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/xcd
mkdir -p $OUT
for X in 0 1; do
  export SAGA_GEMM_XCD=$X
  timeout -k 10 200 python3 $PWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$X.json 2> $OUT/bench_$X.err || exit 1
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_$X -o p --output-format csv -- python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$X.json 2> $OUT/pmc_$X.err || exit 1
done
python3 - <<PY
import csv, glob, json
for X in (0, 1):
    d = json.loads(open('$OUT/bench_%d.json' % X).read().strip().splitlines()[-1])
    f = glob.glob('$OUT/pmc_%d/**/*counter_collection.csv' % X, recursive=True)[0]
    for name, key in (('void gemm_nt_f32_kernel<0, 16, 4, 2', 'blend_gemm_fwd'), ('void gemm_nt_f32_kernel<1, 16, 5, 2', 'blend_gemm_bwd')):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith(name) and r['Counter_Name'] == 'FETCH_SIZE':
                tot += float(r['Counter_Value']); n += 1
        print('xcd_map', X, 'ms/step', d['ms_per_step'], key, 'ms', d['config']['kernels'][key]['ms_per_step'],
              'FETCH_SIZE per launch (KB units)', tot / max(n, 1), 'launches', n)
PY
