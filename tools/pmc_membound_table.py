"""Per-INSTANCE counter table of the memory-bound passes from the passes of tools/pmc_membound.sh: launches of a kernel are matched across passes by their
occurrence index in the eager step sequence; one row per (kernel, grid size) with the mean over the steps.   python tools/pmc_membound_table.py gpurun_out/pmc_mem"""
import collections, csv, glob, re, sys
d = sys.argv[1]
pats = ['upconv_epilogue', 'epilogue_bwd_kernel', 'fir44_adjoint_split', 'split_act_lds', 'torgb_mid', 'scatter_accum16p', 'gather_rows', 'decode_rows', 'upfirdn2d_nhwc4', 'render_kernel']
vals = collections.defaultdict(lambda: collections.defaultdict(list))        # (kernel, grid) -> counter -> values
for f in sorted(glob.glob(d + '/g*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(p in k for p in pats):
            continue
        m = re.search(r'(\w+_kernel(<[^>]*>)?|upfirdn2d_nhwc4\w*)', k)
        key = (m.group(1) if m else k[:40], int(r.get('Grid_Size', 0) or 0))
        vals[key][r['Counter_Name']].append(float(r['Counter_Value']))
def mean(v): return sum(v) / len(v) if v else float('nan')
print('| kernel | grid (threads) | launches | wave cycles / wave | waiting % (SQ_WAIT_ANY) | issue-stall % | active % | TCP pending-stall cyc / wave-cyc | L2 hit % | EA write-stall / WRREQ | L2 tag stall / TCC busy |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
for (k, g), c in sorted(vals.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    wc, wv = mean(c['SQ_WAVE_CYCLES']), mean(c['SQ_WAVES'])
    if not wc or wc != wc or wv < 256: continue
    hit, miss = mean(c['TCC_HIT_sum']), mean(c['TCC_MISS_sum'])
    print('| `%s` | %d | %d | %.0f | %.0f | %.0f | %.0f | %.2f | %.0f | %.2f | %.2f |' % (
        k[:44], g, len(c['SQ_WAVES']), wc / wv, 100 * mean(c['SQ_WAIT_ANY']) / wc, 100 * mean(c['SQ_WAIT_INST_ANY']) / wc, 100 * mean(c['SQ_ACTIVE_INST_ANY']) / wc,
        mean(c['TCP_PENDING_STALL_CYCLES_sum']) / wc, 100 * hit / max(hit + miss, 1), mean(c['TCC_EA0_WRREQ_STALL_sum']) / max(mean(c['TCC_EA0_WRREQ_sum']), 1),
        mean(c['TCC_TAG_STALL_sum']) / max(mean(c['TCC_BUSY_sum']), 1)))
