#!/usr/bin/env python3
"""Instruction-level bisect of the SLP-vectoriser fault of conv_v2's forward epilogue.

  build  (no GPU):  compile csrc/conv_v2.hip WITH the SLP vectoriser to device assembly, write patched copies of the assembly, assemble each into a code
                    object tools/rootcause/hsaco/conv_v2_<variant>.hsaco
  run    (GPU):     load each code object through the HIP module API, launch conv_v2_kernel<9,true,false,4,false,1> on a full-size layer with the fused forward
                    epilogue, compare with the plain-store launch of the product library + the epilogue in torch; count wrong elements
Variants:  slp (unpatched) | nop_opsel (s_nop 7 in front of every `v_pk_add_f32 ... op_sel:[0,1]`) | scalar_opsel (that instruction replaced by two v_add_f32) |
           nop_pk (s_nop 1 in front of every v_pk_*_f32) | noslp (the -fno-slp-vectorize assembly, control)
"""
import ctypes as C
import math
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'rootcause', 'hsaco')
KERNEL = '_ZN12_GLOBAL__N_114conv_v2_kernelILi9ELb1ELb0ELi4ELb0ELi1EEEv19eg3d_conv_v2_paramsi'
LLVM = '/opt/rocm/lib/llvm/bin'
PK_OPSEL = re.compile(r'^\s*v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\]\s*$')


def patch(lines, variant):
    out = []
    n = 0
    for l in lines:
        m = PK_OPSEL.match(l)
        if variant == 'nop_opsel' and m:
            out.append('\ts_nop 7'); n += 1
        if variant == 'scalar_opsel' and m:
            a, b, c, d, e, f = (int(x) for x in m.groups())
            assert b == a + 1 and d == c + 1 and f == e + 1 and a != f
            out.append('\tv_add_f32_e32 v%d, v%d, v%d' % (a, c, f))        # lo = src0.lo + src1.HI
            out.append('\tv_add_f32_e32 v%d, v%d, v%d' % (b, d, f))        # hi = src0.hi + src1.HI
            n += 1
            continue
        if variant == 'nop_pk' and re.match(r'^\s*v_pk_\w+_f32 ', l):
            out.append('\ts_nop 1'); n += 1
        out.append(l)
    return out, n


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, '3dgan-inversion_amd', 'csrc', 'conv_v2.hip')
    base = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '--cuda-device-only', '-S', src]
    for name, extra in (('slp', []), ('noslp', ['-fno-slp-vectorize'])):
        subprocess.run(base + extra + ['-o', os.path.join(OUT, 'conv_v2_%s.s' % name)], check=True, stderr=subprocess.DEVNULL)
    slp = open(os.path.join(OUT, 'conv_v2_slp.s')).read().split('\n')
    for v in ('nop_opsel', 'scalar_opsel', 'nop_pk'):
        lines, n = patch(slp, v)
        open(os.path.join(OUT, 'conv_v2_%s.s' % v), 'w').write('\n'.join(lines))
        print(v, 'patched sites:', n)
    for v in ('slp', 'noslp', 'nop_opsel', 'scalar_opsel', 'nop_pk'):
        s, o, h = (os.path.join(OUT, 'conv_v2_%s.%s' % (v, e)) for e in ('s', 'o', 'hsaco'))
        subprocess.run([LLVM + '/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', s, '-o', o], check=True)
        subprocess.run([LLVM + '/ld.lld', '-shared', o, '-o', h], check=True)
        os.remove(o)
    print('code objects in', OUT)


def run():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, '3dgan-inversion_amd'))
    import torch
    from inv3d_amd import hipops as H, _lib as L
    hip = C.CDLL('libamdhip64.so')
    DEV = 'cuda'
    ci, h, co = 128, 512, 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, ci, h, h, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(1, ci, generator=g)).to(DEV)
    d = (0.5 + torch.rand(1, co, generator=g)).to(DEV)
    noise, strength = torch.randn(h, h, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    aimg = H.split_activation(x, H.absmax(x), in_scale=s)
    wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
    cls = H.classes_corr(h, h, 3, 3, 1)
    z = H.empty_cl(1, co, h, h, DEV)
    H.conv_v2(aimg, wimg, z, cls, epi=L.EPI_STORE, patch_rows=8)
    ref = torch.nn.functional.leaky_relu(z * d[:, :, None, None] + noise * 0.3 + bias[None, :, None, None], 0.2) * 1.4
    ref_nonoise = torch.nn.functional.leaky_relu(z * d[:, :, None, None] + bias[None, :, None, None], 0.2) * 1.4
    torch.cuda.synchronize()
    for v in ('noslp', 'slp', 'nop_opsel', 'scalar_opsel', 'nop_pk'):
        path = os.path.join(OUT, 'conv_v2_%s.hsaco' % v).encode()
        mod, fn = C.c_void_p(), C.c_void_p()
        assert hip.hipModuleLoad(C.byref(mod), path) == 0, 'hipModuleLoad ' + v
        assert hip.hipModuleGetFunction(C.byref(fn), mod, KERNEL.encode()) == 0, 'hipModuleGetFunction'
        lds = 73728
        hip.hipFuncSetAttribute(fn, 8, lds)                                  # hipFuncAttributeMaxDynamicSharedMemorySize
        wrong = miss = 0
        for it in range(20):
            out, am = H.empty_cl(1, co, h, h, DEV), torch.zeros(1, device=DEV)
            p = H._conv_v2_params(aimg, wimg, out, cls, 1, L.EPI_FWD, d, bias, noise, 0, strength, 'lrelu', 0.2, 1.4, -1.0, None, None, None, am)
            p.products, p.ksplit, p.patch_rows = 3, 1, 8
            size = C.sizeof(p)
            buf = (C.c_char * (size + 8))()
            C.memmove(buf, C.byref(p), size)
            C.memmove(C.addressof(buf) + size, C.byref(C.c_int32(0)), 4)     # cls_base
            nbytes = C.c_size_t(size + 4)
            extra = (C.c_void_p * 5)(1, C.addressof(buf), 2, C.addressof(nbytes), 3)
            tiles = (h // 8) * (h // 32) * (co // 128)
            rc = hip.hipModuleLaunchKernel(fn, tiles, 1, 1, 256, 1, 1, lds, C.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra)
            assert rc == 0, 'launch rc %d' % rc
            torch.cuda.synchronize()
            bad = (out - ref).abs() > 1e-4
            wrong += int(bad.sum())
            miss += int((bad & ((out - ref_nonoise).abs() < 1e-5)).sum())
        print('%-13s wrong elements in 20 launches: %8d   (of those = the epilogue without its noise term: %d)' % (v, wrong, miss), flush=True)
        hip.hipModuleUnload(mod)


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
