#!/usr/bin/env python3
"""Aggregate rocprofv3 output of scripts/profile_round.sh: per kernel, mean duration and mean counter values per launch
(library kernels only; torch's element-wise kernels are summed into one line)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
csv.field_size_limit(1 << 30)

def short(name):
    n = name.split('(')[0]
    return n.replace('void ', '').strip()

OURS = ('gemm_nt', 'skin_', 'pose_', 'joints_', 'grasp_loss', 'nn_scan', 'vertex_normals', 'cont6d', 'rodrigues', 'fit_', 'adam', 'param')
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not k.startswith(OURS):
            k = '(other: torch / rocBLAS kernels)'
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('== kernel trace: launches, mean us, total ms')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('%-60s %6d %10.1f %10.2f' % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not k.startswith(OURS):
            continue
        cnt[k][r['Counter_Name']].append(float(r['Counter_Value']))
print('== PMC: mean per launch')
for k in sorted(cnt):
    print(k)
    for c, v in sorted(cnt[k].items()):
        print('    %-32s %18.1f   (n=%d)' % (c, sum(v) / len(v), len(v)))
