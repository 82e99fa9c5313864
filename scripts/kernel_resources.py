#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in the shipped library, from the code objects' metadata notes.

    python scripts/kernel_resources.py [--lib saga_amd/lib/libsaga_hip.so] [--all] [--fail-on-spill]

Prints one row per kernel that spills (or every kernel with --all): vgpr / agpr counts, vgpr_spill_count, sgpr_spill_count,
private_segment_fixed_size (scratch bytes per lane) and LDS bytes.  A kernel that starts spilling shows up here before it shows up
as unexplained HBM write traffic in a PMC pass.  Runs on the CPU (llvm-objdump / llvm-readelf from ROCm).
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_notes(lib):
    """-> list of dicts (one per kernel) parsed from the amdhsa metadata of every gfx950 code object bundled in lib."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)                 # llvm-objdump extracts next to its input
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if 'amdgcn' not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r'\s*(- )?\.(\w+):\s+(.*)$', line)
                if not m:
                    continue
                dash, key, val = m.groups()
                if dash and key in ('agpr_count', 'args'):
                    # a new kernel record starts with the first key of its map
                    cur = {}
                    out.append(cur)
                if cur is None:
                    continue
                if key in ('name', 'symbol'):
                    cur[key] = val.strip()
                elif key in ('agpr_count', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
                             'private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size'):
                    cur[key] = int(val)
    return [k for k in out if 'name' in k and 'vgpr_count' in k]


def demangle(names):
    try:
        r = subprocess.run([os.path.join(LLVM, 'llvm-cxxfilt')], input='\n'.join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=os.path.join(ROOT, 'saga_amd', 'lib', 'libsaga_hip.so'))
    ap.add_argument('--all', action='store_true', help='every kernel, not only the ones that spill')
    ap.add_argument('--fail-on-spill', action='store_true', help='exit 1 if any kernel has vgpr_spill_count > 0')
    a = ap.parse_args()
    ks = kernel_notes(a.lib)
    names = demangle([k['name'] for k in ks])
    spilled = 0
    print(f'{"vgpr":>5} {"agpr":>5} {"vspill":>6} {"sspill":>6} {"scratchB":>8} {"ldsB":>7}  kernel')
    for k, n in sorted(zip(ks, names), key=lambda t: -t[0].get('vgpr_spill_count', 0)):
        sp = k.get('vgpr_spill_count', 0)
        spilled += sp > 0
        if not a.all and sp == 0 and k.get('private_segment_fixed_size', 0) == 0:
            continue
        n = re.sub(r'\(.*$', '', n)
        print(f'{k["vgpr_count"]:5d} {k.get("agpr_count", 0):5d} {sp:6d} {k.get("sgpr_spill_count", 0):6d} '
              f'{k.get("private_segment_fixed_size", 0):8d} {k.get("group_segment_fixed_size", 0):7d}  {n}')
    print(f'{len(ks)} kernels, {spilled} with VGPR spills')
    if a.fail_on_spill and spilled:
        sys.exit(1)


if __name__ == '__main__':
    main()
