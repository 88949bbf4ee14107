#!/usr/bin/env python3
"""Disassemble the device code of the SHIPPED libraries (inv3d_amd/libeg3d_hip*.so) and fail if
(1) any kernel reaches an s_barrier with an LDS MEMORY operation of its own still in flight (issued, not covered by an lgkmcnt wait: the write-after-read /
    read-after-write window of DESIGN.md section 5.9a -- checked here on every kernel of the binary that ships, not only on the LDS-DMA family), or
(2) any packed-fp32 instruction uses a low-lane operand swizzle
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
LDS_MEM = re.compile(r'^ds_(read|write|load|store|add|sub|min|max|and|or|xor|inc|dec|cmpst|wrxchg|append|consume|gws)')          # not ds_bpermute / ds_permute / ds_swizzle / ds_nop


def scan(lib):
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, '--offloading', local], cwd=td, check=True, capture_output=True)
        hits, npk, nobj, nbar, inflight = [], 0, 0, 0, []
        for co in sorted(glob.glob(local + '.*amdgcn*')):
            nobj += 1
            dis = subprocess.run([OBJDUMP, '-d', co], check=True, capture_output=True, text=True).stdout
            kern, q = '?', []
            for l in dis.split('\n'):
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', l)
                if m:
                    kern, q = m.group(1), []
                    continue
                ins = l.split('//')[0].strip()
                if not ins:
                    continue
                op = ins.split()[0]
                if op.startswith('ds_') or op.startswith('s_load') or op.startswith('s_buffer_load'):           # lgkmcnt model (LDS returns in order; a counted wait with
                    q.append(ins)                                                                                # scalar loads in flight only ever UNDER-counts what has returned)
                elif op == 's_waitcnt':
                    w = re.search(r'lgkmcnt\((\d+)\)', ins)
                    if w:
                        n = int(w.group(1)); q = q[len(q) - n:] if n else []
                elif op == 's_barrier':
                    nbar += 1
                    pend = [x for x in q if LDS_MEM.match(x)]
                    if pend:
                        inflight.append((kern, len(pend), pend[0]))
                if 'v_pk_' in ins and '_f32' in ins:
                    npk += 1
                    if PAT.search(ins):
                        hits.append((kern, ins))
        return nobj, npk, hits, nbar, inflight


if __name__ == '__main__':
    libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, '3dgan-inversion_amd', 'inv3d_amd', 'libeg3d_hip.so')) + glob.glob(os.path.join(ROOT, '3dgan-inversion_amd', 'inv3d_amd', 'libeg3d_hip_det.so')))
    bad = 0
    for lib in libs:
        nobj, npk, hits, nbar, inflight = scan(lib)
        print('%s: %d code objects, %d packed-fp32 instructions, %d with a low-lane op_sel; %d s_barrier sites, %d with an LDS memory operation in flight' % (
            os.path.basename(lib), nobj, npk, len(hits), nbar, len(inflight)))
        for k, l in hits[:10]:
            print('   ', k[:80], '|', l)
        for k, n, l in inflight[:10]:
            print('   ', k[:80], '|', n, 'in flight, first:', l)
        bad += len(hits) + len(inflight)
    sys.exit(1 if bad else 0)
