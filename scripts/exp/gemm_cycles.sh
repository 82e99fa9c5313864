#!/bin/bash
# Cycle counts of the blend GEMM launches under library variants: scripts/exp/gemm_cycles.sh "" build/variants/libsaga_x.so ...
# (one rocprofv3 --pmc pass per variant; GRBM_GUI_ACTIVE / 8 XCDs = shader cycles of the launch, with the dispatch timestamps -> clock)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gemm_cycles; mkdir -p $OUT
for L in "$@"; do
  N=$(basename "${L:-default}" .so)
  if [ -n "$L" ]; then export SAGA_HIP_LIB=$PWD/$L; else unset SAGA_HIP_LIB; fi
  rm -rf $OUT/$N
  timeout -s KILL 180 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/$N -o p --output-format csv -- python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$N.json 2> $OUT/$N.err || { echo "$N failed"; tail -3 $OUT/$N.err; exit 1; }
  python3 - "$OUT/$N" "$N" <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'gemm_nt_f32_kernel' not in k: continue
    key = k.split('(')[0][-40:] + ' g=' + r.get('Grid_Size', '')
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    if 'Start_Timestamp' in r and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        acc[key]['ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, c in sorted(acc.items()):
    g = sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE']) / 8
    if g < 1e6: continue
    ns = sum(c['ns']) / len(c['ns']) if c['ns'] else 0
    print('%-14s %-52s cycles %.0f  us %.1f  GHz %.3f  wait_inst %.3f wait_any %.3f' % (name, k, g, ns / 1e3, g / ns if ns else 0,
          sum(c['SQ_WAIT_INST_ANY']) / sum(c['SQ_WAVE_CYCLES']), sum(c['SQ_WAIT_ANY']) / sum(c['SQ_WAVE_CYCLES'])))
PY
done
