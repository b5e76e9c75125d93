#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of libbeta_cores.so, read from the code object's metadata notes.

  python tools/kernel_resources.py [path/to/libbeta_cores.so] [name-filter]

Uses llvm-readelf --notes on the gfx950 code object extracted with clang-offload-bundler; prints one line per kernel:
vgpr, agpr, sgpr, spilled vgprs / sgprs, scratch bytes, static LDS bytes."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'beta_cores_amd', 'libbeta_cores.so')
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    notes = ''
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
        blob = open(fat, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]      # one bundle per translation unit
        for i, a in enumerate(starts):
            part = os.path.join(tmp, 'b%d.bin' % i)
            open(part, 'wb').write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, 'dev%d.co' % i)
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + part,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], stderr=subprocess.DEVNULL)
            notes += subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], text=True)
    rows = []
    for blk in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
        blk = '.agpr_count:' + blk
        get = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
        name = get('name')
        try:
            name = subprocess.check_output([os.path.join(LLVM, 'llvm-cxxfilt'), name], text=True).strip()
        except Exception:
            pass
        if flt and flt not in name:
            continue
        rows.append((name, get('vgpr_count'), get('agpr_count'), get('sgpr_count'), get('vgpr_spill_count'),
                     get('sgpr_spill_count'), get('private_segment_fixed_size'), get('group_segment_fixed_size')))
    print('%-110s %5s %5s %5s %6s %6s %8s %7s' % ('kernel', 'vgpr', 'agpr', 'sgpr', 'vspill', 'sspill', 'scratch', 'lds'))
    for r in sorted(rows):
        print('%-110s %5s %5s %5s %6s %6s %8s %7s' % ((r[0][:110],) + r[1:]))


if __name__ == '__main__':
    main()
