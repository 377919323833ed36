"""VALU utilisation and wave-time split of the dominant kernels from a pmc_sq.sh counter summary:
VALU utilisation = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x 32 SIMDs per shader engine -- BUSY_CYCLES is summed over the 32
engines); wave split = SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (quad-cycles, disjoint).
usage: python tools/valu_util.py profiles/r5_pmc_sq_counters.txt"""
import re, sys
txt = open(sys.argv[1]).read()
want = ('sweep_iso_kernel<false>', 'compositen_kernel<0, 4, true, unsigned int, 3, false>', 'fragment_bwd_kernel<0, 3, 2, unsigned int, true, true>',
        'binA_kernel<true>', 'binB_kernel<false>', 'sweep_iso_kernel<true>', 'compositen_kernel<0, 4, true, unsigned int, 3, true>',
        'fragment_bwd_kernel<0, 3, 2, unsigned int, false, false>', 'rays_fwd_kernel', 'sweep_iso_kernel', 'trace_fwd_kernel<1, false>',
        # round 6's template parameter lists (GEN an int; DIAG):
        'compositen_kernel<0, 4, true, unsigned int, 3, 0>', 'compositen_kernel<0, 4, true, unsigned int, 3, 1>',
        'compositen_kernel<0, 4, true, unsigned int, 3, 2>', 'fragment_bwd_kernel<0, 3, 2, unsigned int, true, true, false>',
        'fragment_bwd_kernel<0, 3, 2, unsigned int, false, false, false>', 'fragment_bwd_kernel<0, 3, 2, unsigned int, false, true, true>',
        'binB_kernel<true>', 'binA_kernel<false>')
for b in re.split(r'\n(?=voge::)', txt):
    name = b.split('  vgpr')[0].replace('voge::', '')
    if name not in want:
        continue
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r'(SQ_\w+)\s+(\d+)', b)}
    simd_cycles = c['SQ_BUSY_CYCLES'] * 32
    print(f"{name:60s} VALU {c['SQ_INSTS_VALU'] / 1e6:6.2f} M  SALU {c['SQ_INSTS_SALU'] / 1e6:5.2f} M  LDS {c['SQ_INSTS_LDS'] / 1e6:5.2f} M  "
          f"busy {c['SQ_BUSY_CYCLES'] / 32 / 2.4e3:6.1f} us at 2.4 GHz  VALU utilisation {4 * c['SQ_ACTIVE_INST_VALU'] / simd_cycles:4.2f}  "
          f"wave cycles: issuing {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:4.2f}  waiting (s_waitcnt) {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:4.2f}  "
          f"issue-stalled {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:4.2f}  LDS bank-conflict cycles / LDS cycles "
          f"{c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):4.2f}")
