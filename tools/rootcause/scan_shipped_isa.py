#!/usr/bin/env python3
"""Disassemble the device code of the SHIPPED libraries (inv3d_amd/libeg3d_hip*.so) and fail if any packed-fp32 instruction uses a low-lane operand swizzle
(`v_pk_{add,mul,fma}_f32 ... op_sel:[...]`): on gfx950 `v_pk_add_f32 D, A, B op_sel:[0,1]` was measured to return src0.lo + 0 in lanes 48-63, sporadically
(tools/rootcause/slp_isa_patch.py).  `op_sel_hi` forms (the HIGH lane reading a low half: broadcasts) are what the compiler emits for
explicit two-element vectors; they measured clean and are allowed.   python tools/rootcause/scan_shipped_isa.py [lib.so ...]  -> exit status 1 on a hit"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PAT = re.compile(r'\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[')


def scan(lib):
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, '--offloading', local], cwd=td, check=True, capture_output=True)
        hits, npk, nobj = [], 0, 0
        for co in sorted(glob.glob(local + '.*amdgcn*')):
            nobj += 1
            dis = subprocess.run([OBJDUMP, '-d', co], check=True, capture_output=True, text=True).stdout
            kern = '?'
            for l in dis.split('\n'):
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', l)
                if m:
                    kern = m.group(1)
                if 'v_pk_' in l and '_f32' in l:
                    npk += 1
                    if PAT.search(l):
                        hits.append((kern, l.split('//')[0].strip()))
        return nobj, npk, hits


if __name__ == '__main__':
    libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, '3dgan-inversion_amd', 'inv3d_amd', 'libeg3d_hip.so')) + glob.glob(os.path.join(ROOT, '3dgan-inversion_amd', 'inv3d_amd', 'libeg3d_hip_det.so')))
    bad = 0
    for lib in libs:
        nobj, npk, hits = scan(lib)
        print('%s: %d code objects, %d packed-fp32 instructions, %d with a low-lane op_sel' % (os.path.basename(lib), nobj, npk, len(hits)))
        for k, l in hits[:10]:
            print('   ', k[:80], '|', l)
        bad += len(hits)
    sys.exit(1 if bad else 0)
