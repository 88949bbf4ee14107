"""The LDS-DMA step boundary (csrc/common.h `step_sync`) checked on the COMPILED code of every build (no GPU needed: hipcc cross-compiles).

Root cause of the two schedule-dependent faults of the pre-split convolution family (rounds 2 and 5): with a raw
`__builtin_amdgcn_s_barrier()` the scheduler left LDS reads of the previous step in flight ACROSS the barrier, and another wave's LDS-DMA could overwrite
the ring slot they address.  `make protocol-check` disassembles every translation unit that issues `buffer_load ... lds` and fails when
  * any LDS read is still in flight (issued, not covered by an lgkmcnt wait) at any s_barrier of such a kernel, or
  * conv_v2_kernel's counted vmcnt immediates / number of LDS-DMA instructions differ from its constexpr issue schedule (v2_sched, restated in the tool).
No reference counterpart: the reference gets its convolutions from cuDNN (torch_utils/ops/conv2d_gradfix.py:37-45)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'rootcause'))
import isa_protocol as P      # noqa: E402

GOOD = '''
_Z4goodv:
	buffer_load_dwordx4 v1, s[0:3], 0 offen lds
	ds_read_b128 v[4:7], v2
	;;#ASMSTART
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_barrier
	;;#ASMEND
	v_mfma_f32_32x32x16_f16 v[8:23], v[4:7], v[4:7], v[8:23]
	buffer_load_dwordx4 v1, s[0:3], 0 offen lds
	s_endpgm
	.end_amdhsa_kernel
'''
BAD = '''
_Z3badv:
	buffer_load_dwordx4 v1, s[0:3], 0 offen lds
	ds_read_b128 v[4:7], v2
	ds_read_b128 v[24:27], v2 offset:49152
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_barrier
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_32x32x16_f16 v[8:23], v[4:7], v[24:27], v[8:23]
	buffer_load_dwordx4 v1, s[0:3], 0 offen lds
	s_endpgm
	.end_amdhsa_kernel
'''


def test_checker_flags_lds_reads_in_flight_across_a_barrier(tmp_path):
    g, b = tmp_path / 'good.s', tmp_path / 'bad.s'
    g.write_text(GOOD); b.write_text(BAD)
    assert P.verify(str(g), quiet=True) == []
    bad = P.verify(str(b), quiet=True)
    assert len(bad) == 1 and '2 LDS read(s) in flight' in bad[0], bad


def test_schedule_model_matches_the_hand_counted_immediates():
    # conv_v2_kernel<9, ..., RPW = 4>: 6 A parts per wave and chunk, one per step during taps 0..5; B(step + 2) two operations per step
    nparts, steps = P.v2_model(9, 4)
    assert nparts == 6
    assert [a for a, d in steps[:9]] == [2, 3, 3, 3, 3, 3, 3, 2, 2] and [d for a, d in steps[:9]] == [3, 3, 3, 3, 3, 3, 2, 2, 2]
    assert [a for a, d in steps[9:]] == [2, 2, 2, 2, 2, 2, 2, 2, 0] and [d for a, d in steps[9:]] == [2, 2, 2, 2, 2, 2, 2, 0, 0]
    nparts, steps = P.v2_model(2, 4)            # two-tap classes: 3 A parts in both steps -- tap 0 must not leave the previous step's A parts in flight
    assert [a for a, d in steps[:2]] == [2, 5]


def test_every_lds_dma_kernel_of_the_library_keeps_the_step_boundary():
    r = subprocess.run(['make', '-C', os.path.join(ROOT, '3dgan-inversion_amd'), '-j8', 'protocol-check'], capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert 'VIOLATION' not in r.stdout
    lines = [l for l in r.stdout.splitlines() if 'LDS-DMA kernels' in l]
    assert sum(int(l.split(':')[1].split()[0]) for l in lines) >= 30, tail        # conv_v2 x15, s2adj x4, up2 x2, v3 x8, wgrad_v2 x6, scatter_accum2


def test_shipped_libraries_have_no_lds_operation_in_flight_at_a_barrier_and_no_low_lane_op_sel():
    """The instruction form behind the round-2 'SLP miscompile' (`v_pk_add_f32 D, A, B op_sel:[0,1]` returns src0.lo + 0 in lanes 48-63 on
    gfx950, sporadically) must not occur in the device code of the libraries that ship -- whatever flag or compiler release produces it next."""
    libs = [os.path.join(ROOT, '3dgan-inversion_amd', 'inv3d_amd', n) for n in ('libeg3d_hip.so', 'libeg3d_hip_det.so')]
    libs = [l for l in libs if os.path.exists(l)]
    assert libs, 'build the libraries first (__graft_entry__.build())'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rootcause', 'scan_shipped_isa.py')] + libs, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert r.stdout.count(' 0 with a low-lane op_sel') == len(libs), r.stdout
    # ... and no kernel of the shipped binaries -- all of them, not only the LDS-DMA family -- reaches an s_barrier with an LDS memory operation of its own in flight
    # (the round-5 library: 261 of 1444 barrier sites)
    assert r.stdout.count(' 0 with an LDS memory operation in flight') == len(libs), r.stdout
