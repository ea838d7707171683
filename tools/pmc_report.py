#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>_*/ (tools/pmc_profile.sh) for the fast stepper kernel."""
import csv, glob, sys
tag = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else None     # substring of the kernel name (default: the L = 128 fast steppers)
res = {}
for d in sorted(glob.glob('gpurun_out/pmc_%s_*/*/*_counter_collection.csv' % tag)):
    for r in csv.DictReader(open(d)):
        name = r['Kernel_Name']
        if (match in name) if match else ('stepper_kernel<128, false' in name or 'stepper_pair_kernel' in name):
            res[r['Counter_Name']] = res.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
for k, v in sorted(res.items()):
    print("   %-28s %.4g" % (k, v))
wc = res['SQ_WAVE_CYCLES']; gui = res['GRBM_GUI_ACTIVE'] / 8
print('kernel cycles %.4g ; avg resident waves/SIMD %.2f' % (gui, wc * 4 / gui / 1024))
print('VALU active / SIMD-cycle %.3f   LDS busy / CU-cycle %.3f' % (res['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / gui, res['SQ_LDS_IDX_ACTIVE'] / 256 / gui))
print('wave-time shares: valu %.3f lds %.3f wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f' % tuple(res[k] / wc for k in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS')))
print('cycles per VALU inst (active) %.2f ; SIMD-cycles per VALU inst %.2f ; VALU/LDS insts per wave %.4g / %.4g' % (4 * res['SQ_ACTIVE_INST_VALU'] / res['SQ_INSTS_VALU'], 1024 * gui / res['SQ_INSTS_VALU'], res['SQ_INSTS_VALU'] / res['SQ_WAVES'], res['SQ_INSTS_LDS'] / res['SQ_WAVES']))
