#!/usr/bin/env python3
"""Copy a profile summary from gpurun_out/ into profiles/ (tracked) and stamp it with the git head it was taken at.
usage: scripts/commit_profile.py gpurun_out/prof_TAG roundN_TAG   -> profiles/roundN_TAG_pmc_summary.json, _kernel_stats.csv, _summary.txt"""
import glob
import json
import os
import shutil
import subprocess
import sys

src, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
head = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=root, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(['git', 'status', '--porcelain', '--', 'saga_amd/csrc'], cwd=root, capture_output=True, text=True).stdout.strip())
d = json.load(open(os.path.join(src, 'summary.json')))
d.setdefault('_meta', {})['git_head'] = head + ('+uncommitted csrc changes' if dirty else '')
json.dump(d, open(os.path.join(root, 'profiles', name + '_pmc_summary.json'), 'w'), indent=1, sort_keys=True)
shutil.copy(os.path.join(src, 'summary.txt'), os.path.join(root, 'profiles', name + '_kernel_trace_and_pmc_summary.txt'))
stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(root, 'profiles', name + '_kernel_stats.csv'))
b = os.path.join(src, 'bench_trace.json')
if os.path.exists(b):
    shutil.copy(b, os.path.join(root, 'profiles', name + '_bench_under_rocprof.json'))
if '--current' in sys.argv:      # the summary bench.py reads for roofline.traffic
    json.dump(d, open(os.path.join(root, 'profiles', 'current_pmc_summary.json'), 'w'), indent=1, sort_keys=True)
print('profiles/%s_* written at %s' % (name, d['_meta']['git_head']))
