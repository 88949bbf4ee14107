#!/usr/bin/env python3
"""Protocol trace of a conv_v2-style kernel from hipcc -S output (round-6 root-cause work).

For one kernel symbol: the instruction stream reduced to the events that make up the LDS-DMA / barrier protocol --
s_barrier, s_waitcnt (decoded counters), buffer_load ... lds (LDS-DMA issue, with the m0 / offset it writes to), ds_read / ds_write,
v_mfma (run-length), s_nop, branch labels -- so that two builds of the same source can be diffed event by event.

  python tools/rootcause/isa_protocol.py FILE.s --kernel 'ILi9ELb1ELb0ELi4ELb0ELi1E' [--full] [--checks]
--checks prints, per basic block between two s_barriers, the facts the protocol relies on:
  * every ds_read between barrier k and barrier k+1 is followed by an lgkmcnt wait before barrier k+1 (no LDS read in flight across a barrier)
  * the number of LDS-DMA issues between consecutive counted vmcnt waits (the immediate must equal the issues allowed to stay in flight)
"""
import re, sys, argparse

def kernel_body(lines, key):
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and key in l and re.match(r'^_Z\w+:', l):
            start = i
        elif start is not None and l.strip().startswith('.end_amdhsa_kernel'):
            return lines[start:i]
        elif start is not None and l.strip().startswith('s_endpgm') and False:
            pass
    if start is None:
        raise SystemExit('kernel %r not found' % key)
    return lines[start:]

def events(body):
    ev = []
    for l in body:
        s = l.split(';')[0].strip()
        if not s or s.startswith('.'):
            continue
        if s.endswith(':'):
            ev.append(('label', s[:-1])); continue
        op = s.split()[0]
        if op == 's_barrier': ev.append(('barrier', ''))
        elif op == 's_waitcnt': ev.append(('wait', s[len(op):].strip()))
        elif op.startswith('buffer_load') and ' lds' in s: ev.append(('dma', s))
        elif op.startswith('buffer_load') or op.startswith('global_load') or op.startswith('flat_load'): ev.append(('vload', op))
        elif op.startswith('buffer_store') or op.startswith('global_store') or op.startswith('global_atomic') or op.startswith('flat_store') or op.startswith('flat_atomic'): ev.append(('vstore', op))
        elif op.startswith('ds_read') or op.startswith('ds_load'): ev.append(('dsr', op))
        elif op.startswith('ds_write') or op.startswith('ds_store') or op.startswith('ds_add') or op.startswith('ds_max'): ev.append(('dsw', op))
        elif op.startswith('ds_'): ev.append(('ds', op))
        elif op.startswith('v_mfma'): ev.append(('mfma', op))
        elif op.startswith('s_load') or op.startswith('s_buffer_load'): ev.append(('sload', op))
        elif op in ('s_nop', 's_sleep', 's_setprio'): ev.append((op, s[len(op):].strip()))
        elif op.startswith('s_cbranch') or op == 's_branch': ev.append(('br', s))
        elif op == 's_endpgm': ev.append(('end', ''))
        elif op.startswith('s_mov_b32') and s.split()[1].rstrip(',') == 'm0': ev.append(('m0', s))
    return ev

def compress(ev):
    out, i = [], 0
    while i < len(ev):
        k = ev[i][0]
        if k in ('mfma', 'dsr', 'dsw', 'dma', 'vload', 'vstore', 'm0'):
            j = i
            while j < len(ev) and ev[j][0] == k: j += 1
            out.append('%s x%d' % (k, j - i)); i = j
        else:
            out.append(('%s %s' % ev[i]).strip()); i += 1
    return out

def checks(ev):
    # segments between barriers in layout order (straight-line main loop is fully unrolled, so layout order == execution order inside a block)
    seg, k = [], 0
    cur = []
    for e in ev:
        if e[0] == 'barrier':
            seg.append(cur); cur = []
        else:
            cur.append(e)
    seg.append(cur)
    bad_reads = 0
    for si, s in enumerate(seg[:-1]):
        # LDS reads issued in this segment with no lgkmcnt(0)-class wait after the last one, before the next barrier
        last_read = max([i for i, e in enumerate(s) if e[0] == 'dsr'], default=-1)
        if last_read >= 0:
            waited = any(e[0] == 'wait' and 'lgkmcnt' in e[1] for e in s[last_read + 1:])
            used = any(e[0] == 'mfma' for e in s[last_read + 1:])
            if not waited:
                bad_reads += 1
                print('  segment %d: %d ds_read, last one NOT followed by an lgkmcnt wait before the next s_barrier (mfma after it: %s)' % (si, sum(e[0] == 'dsr' for e in s), used))
    print('segments between barriers: %d; segments whose last ds_read is not waited for before the next barrier: %d' % (len(seg) - 1, bad_reads))
    # DMA issues per segment and the vmcnt immediates
    print('per-segment [dma issues | vmcnt waits | mfma | ds_read]:')
    for si, s in enumerate(seg):
        dm = sum(e[0] == 'dma' for e in s); mf = sum(e[0] == 'mfma' for e in s); dr = sum(e[0] == 'dsr' for e in s)
        vw = [re.search(r'vmcnt\((\d+)\)', e[1]).group(1) for e in s if e[0] == 'wait' and 'vmcnt' in e[1]]
        lab = [e[1] for e in s if e[0] == 'label']
        print('  %3d: dma %2d  vmcnt %-12s mfma %3d  dsr %3d  %s' % (si, dm, ','.join(vw), mf, dr, ' '.join(lab[:4])))

def inflight_at_barriers(body):
    """LDS reads still in flight (issued, not yet covered by an lgkmcnt wait) at every s_barrier, in layout order."""
    q, res, nb = [], [], 0
    for l in body:
        s = l.split(';')[0].strip()
        if not s or s.startswith('.') or s.endswith(':'): continue
        op = s.split()[0]
        if op.startswith('ds_') or op.startswith('s_load') or op.startswith('s_buffer_load'):
            q.append(s)
        elif op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', s)
            if m:
                n = int(m.group(1)); q = q[len(q) - n:] if n else []
        elif op == 's_barrier':
            res.append((nb, [x for x in q if x.startswith('ds_read')])); nb += 1
    return res


def all_kernels(lines):
    """(name, body) for every kernel of the file: from its label to its .end_amdhsa_kernel directive."""
    out, start, name = [], None, None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m and start is None:
            start, name = i, m.group(1)
        if start is not None and l.strip().startswith('.end_amdhsa_kernel'):
            out.append((name, lines[start:i])); start = None
    return out


def demangle_v2(name):
    m = re.search(r'conv_v2_kernelILi(\d+)ELb([01])ELb([01])ELi(\d+)ELb([01])ELi(\d+)E', name)
    return tuple(int(x) for x in m.groups()) if m else None


def v2_model(ntaps, rpw):
    """The constexpr schedule of csrc/conv_v2.hip (v2_sched) restated: per step (tap, last) -> (vmcnt immediate at its boundary, LDS-DMA issues after it)."""
    nparts = ((2 * rpw + 2) * 34 + 63) // 64
    apt = (nparts + ntaps - 1) // ntaps
    na_taps = nparts // apt
    n_a = lambda tap, last: apt if (not last and tap < na_taps) else 0
    b_static = lambda tap, last: (not last) or tap + 2 < ntaps
    n_b = lambda tap, last: 2 if b_static(tap, last) else 0

    def allow(tap, last):
        if ntaps == 1:
            return 0
        ptap, plast = (tap - 1, last) if tap >= 1 else (ntaps - 1, False)
        return n_b(ptap, plast) + (0 if tap == 0 else n_a(ptap, plast))
    steps = [(allow(t, last), (n_a(t, last) + n_b(t, last)) if ntaps > 1 else None) for last in (False, True) for t in range(ntaps)]
    return nparts, steps


def verify(path, quiet=False):
    """Every kernel of the file that issues LDS-DMA: no LDS read may be in flight at any s_barrier.  conv_v2_kernel instantiations: the (vmcnt immediate,
    LDS-DMA issues of the step) pairs of the main loop must be the multiset the constexpr schedule gives.  Returns the list of violations."""
    lines = open(path).read().split('\n')
    bad, seen = [], 0
    for name, body in all_kernels(lines):
        if not any('buffer_load' in l and ' lds' in l for l in body):
            continue
        seen += 1
        fl = inflight_at_barriers(body)
        worst = max([len(q) for _, q in fl], default=0)
        if worst:
            bad.append('%s: %d LDS read(s) in flight at an s_barrier (barrier #%d)' % (name, worst, [nb for nb, q in fl if len(q) == worst][0]))
        key = demangle_v2(name)
        note = ''
        if key:
            ntaps, full, atomic, rpw, rgb, kh = key
            nparts, steps = v2_model(ntaps, rpw)
            ev = events(body)
            # boundaries = the asm statements `s_waitcnt vmcnt(N) lgkmcnt(0)` directly followed by s_barrier; the segment after each one up to the next barrier
            segs, cur, imm = [], None, None
            for i, e in enumerate(ev):
                if e[0] == 'barrier':
                    if cur is not None: segs.append((imm, cur))
                    prev = ev[i - 1] if i else ('', '')
                    m = re.search(r'vmcnt\((\d+)\)', prev[1]) if prev[0] == 'wait' else None
                    imm, cur = (int(m.group(1)) if m and 'lgkmcnt(0)' in prev[1] else None), 0
                elif e[0] == 'dma' and cur is not None:
                    cur += 1
            if cur is not None: segs.append((imm, cur))
            # (layout order is not execution order at the loop's back edge -- the compiler rotates the loop -- so the check is per kernel, not per segment:
            #  the multiset of vmcnt immediates of the counted boundaries, and the total number of LDS-DMA instructions = prologue + every step once)
            imms = sorted(a for a, d in segs if a is not None)
            ndma = sum(1 for e in ev if e[0] == 'dma')
            if ntaps > 1:
                want_imms = sorted([a for a, d in steps] + [0])                      # + the boundary in front of the epilogue
                want_dma = nparts + 4 + sum(d for a, d in steps)
                nz = lambda v: [x for x in v if x]                                     # (a fenced __syncthreads also prints as `s_waitcnt vmcnt(0) lgkmcnt(0)` + s_barrier: zeros may be more)
                if nz(imms) != nz(want_imms) or len(imms) < len(want_imms):
                    bad.append('%s: vmcnt immediates %s != schedule %s' % (name, imms, want_imms))
                elif not (want_dma <= ndma <= want_dma + 2):        # (+ the prologue's `S > 1 ? B(step 1) : filler` alternative: two more instructions, or one after tail merging)
                    bad.append('%s: %d LDS-DMA instructions != schedule %d' % (name, ndma, want_dma))
                else:
                    note = ' schedule ok (%d boundaries, %d LDS-DMA)' % (len(imms), ndma)
        if not quiet:
            print('%-110s barriers %3d  max LDS reads in flight at a barrier %d%s' % (name[:110], len(fl), worst, note))
    if not quiet:
        print('%s: %d LDS-DMA kernels, %d violation(s)' % (path, seen, len(bad)))
    return bad


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('file', nargs='+'); ap.add_argument('--kernel'); ap.add_argument('--full', action='store_true'); ap.add_argument('--checks', action='store_true')
    ap.add_argument('--verify', action='store_true')
    a = ap.parse_args()
    if a.verify:
        bad = [b for f in a.file for b in verify(f)]
        for b in bad: print('VIOLATION', b)
        sys.exit(1 if bad else 0)
    body = kernel_body(open(a.file[0]).read().split('\n'), a.kernel)
    ev = events(body)
    if a.checks: checks(ev)
    else:
        for l in (['%s %s' % e for e in ev] if a.full else compress(ev)): print(l)
