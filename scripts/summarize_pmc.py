#!/usr/bin/env python3
"""Aggregate rocprofv3 output of scripts/profile_round.sh: per kernel (name + grid size), mean duration and mean counter
values per launch.  Writes <dir>/summary.txt (human) and <dir>/summary.json (read by bench.py for `roofline.traffic`).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md 'HBM': FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of
1024 B by rocprofv3; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so reads are doubled; WRITE_SIZE is exact
for streaming stores.  Separate --pmc passes (one per counter group)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

root = sys.argv[1]
csv.field_size_limit(1 << 30)
OURS = ('gemm_nt', 'blend_fwd_xt', 'blend_bwd_xt', 'skin_', 'pose_', 'joints_', 'grasp_loss', 'nn_scan', 'nn_fast', 'vertex_normals', 'cont6d', 'rodrigues',
        'params_', 'loss_sum', 'adam_', 'best_update', 'hand_ring', 'p2p_', 'pick_rows', 'nn_grad', 'da_sum')


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp'])):      # launch order
        k = short(r['Kernel_Name'])
        if k.startswith(OURS):
            k = '%s grid=%s' % (k, int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']))
        else:
            k = '(other: torch / rocBLAS kernels)'
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not k.startswith(OURS):
            continue
        cnt['%s grid=%s' % (k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))

out = {}
# mean over EVERY launch of the command (set-up, warm-up and timed steps: the first launches run at ramping clocks) and the mean
# over the last 10 launches in launch order (= the timed region of `--steps 10`: what bench.py's HIP events measure)
lines = ['== kernel trace: launches, mean us (all), mean us (last 10 = timed steps), total ms']
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    last = v[-10:]
    lines.append('%-64s %6d %10.1f %10.1f %10.2f' % (k, len(v), sum(v) / len(v), sum(last) / len(last), sum(v) / 1e3))
    out[k] = {'launches': len(v), 'mean_us': sum(v) / len(v), 'mean_us_last10': sum(last) / len(last)}
lines.append('== PMC: mean per launch (+ derived: HBM bytes per the guide, MFMA / VALU utilisation)')
for k in sorted(cnt):
    c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
    d = out.setdefault(k, {})
    d['counters'] = c
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        d['hbm_read_bytes'] = 2.0 * c['FETCH_SIZE'] * 1024.0       # gfx950: 128-B requests tallied at 64 B
        d['hbm_write_bytes'] = c['WRITE_SIZE'] * 1024.0
        d['hbm_bytes'] = d['hbm_read_bytes'] + d['hbm_write_bytes']
    if c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_conflict_ratio'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']   # conflict cycles per active LDS cycle
    if c.get('GRBM_GUI_ACTIVE'):
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0                            # summed over the 8 XCDs
        d['gpu_cycles'] = cyc
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            d['mfma_util'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc      # 1024 SIMDs
        if 'SQ_ACTIVE_INST_VALU' in c:
            d['valu_issue_util'] = c['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / cyc   # quad-cycles per SIMD
    lines.append(k)
    for n, v in sorted(c.items()):
        lines.append('    %-32s %18.1f' % (n, v))
    for n in ('hbm_bytes', 'mfma_util', 'valu_issue_util', 'lds_conflict_ratio'):
        if n in d:
            lines.append('    -> %-29s %18.4g' % (n, d[n]))
# what the numbers describe: the kernel sources (sha256[:16] per file), the workload of the profiled command and the GEMM
# environment switches -- bench.py emits `roofline.traffic` only when these match the run it reports (git_head is added when the
# summary is copied into profiles/ in the build container: the GPU box has no .git)
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'saga_amd', 'csrc')
meta = {'src_sha16': {os.path.basename(p): hashlib.sha256(open(p, 'rb').read()).hexdigest()[:16]
                      for p in sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')))},
        'env': {k: os.environ.get(k, '') for k in ('SAGA_GEMM_LEPI', 'SAGA_GEMM_XCD', 'SAGA_GEMM_XT', 'SAGA_GEMM_XT_BWD')},
        'kernels': sorted(k for k in out if k.startswith(OURS))}
try:
    line = [l for l in open(os.path.join(root, 'bench_trace.json')) if l.startswith('{')][-1]
    cfg = json.loads(line)['config']
    meta.update(B=cfg['candidates_per_gpu'], n_obj=cfg['object_points'], V=10475 if 'V=10475' in cfg['workload'] else None,
                command='bench.py --steps 10 --warmup 3 --no-cpu-baseline (kernel trace); --steps 3 --warmup 1 (PMC passes)')
except (OSError, IndexError, KeyError, ValueError):
    pass
out['_meta'] = meta
open(os.path.join(root, 'summary.txt'), 'w').write('\n'.join(lines) + '\n')
json.dump(out, open(os.path.join(root, 'summary.json'), 'w'), indent=1, sort_keys=True)
print('\n'.join(lines))
