#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>_*/ (tools/pmc_profile.sh) for one stepper kernel.
    python tools/pmc_report.py <tag> [kernel-name substring] [--steps T --iters I]
With the number of time steps and of inner iterations the profiled launch ran (bench line: config.T,
inner_iterations_per_pass) the instruction counts are also given per time step and per inner iteration."""
import csv, glob, sys
argv = sys.argv[1:]
opts = {argv[i]: argv[i + 1] for i in range(len(argv) - 1) if argv[i].startswith("--")}
args = [a for i, a in enumerate(argv) if not a.startswith("--") and not (i > 0 and argv[i - 1].startswith("--"))]
tag = args[0]
match = args[1] if len(args) > 1 else None     # substring of the kernel name (default: the L = 128 fast steppers)
res = {}
names = set()
for d in sorted(glob.glob('gpurun_out/pmc_%s_*/*/*_counter_collection.csv' % tag)):
    for r in csv.DictReader(open(d)):
        name = r['Kernel_Name']
        if (match in name) if match else ('stepper_kernel<128, false' in name or 'stepper_pair_kernel' in name):
            res[r['Counter_Name']] = res.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
            names.add(name.split('(')[0])
print("kernel(s): %s" % "; ".join(sorted(names)))
for k, v in sorted(res.items()):
    print("   %-28s %.4g" % (k, v))
wc = res['SQ_WAVE_CYCLES']; gui = res['GRBM_GUI_ACTIVE'] / 8
print('kernel cycles %.4g ; avg resident waves/SIMD %.2f' % (gui, wc * 4 / gui / 1024))
print('VALU active / SIMD-cycle %.3f   LDS busy / CU-cycle %s' % (res['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / gui, ('%.3f' % (res['SQ_LDS_IDX_ACTIVE'] / 256 / gui)) if 'SQ_LDS_IDX_ACTIVE' in res else 'n/a (pass 2 not collected)'))
print('wave-time shares: valu %.3f lds %.3f scalar %.3f wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f' % tuple(res.get(k, float('nan')) / wc for k in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS')))
if 'SQ_INSTS_VALU' not in res:
    sys.exit(0)          # PMC_SETS="1 4": occupancy and busy shares only
print('cycles per VALU inst (active) %.2f ; SIMD-cycles per VALU inst %.2f ; VALU/LDS/SALU insts per wave %.4g / %.4g / %.4g' % (4 * res['SQ_ACTIVE_INST_VALU'] / res['SQ_INSTS_VALU'], 1024 * gui / res['SQ_INSTS_VALU'], res['SQ_INSTS_VALU'] / res['SQ_WAVES'], res['SQ_INSTS_LDS'] / res['SQ_WAVES'], res['SQ_INSTS_SALU'] / res['SQ_WAVES']))
if '--steps' in opts:
    T = float(opts['--steps']); waves = res['SQ_WAVES']
    print('per wave and time step (T = %d): VALU %.1f, LDS %.1f, SALU %.1f, v_rcp_f64 %.1f instructions' % (
        T, res['SQ_INSTS_VALU'] / waves / T, res['SQ_INSTS_LDS'] / waves / T, res['SQ_INSTS_SALU'] / waves / T, res['SQ_INSTS_VALU_TRANS_F64'] / waves / T))
    if '--iters' in opts:
        it = float(opts['--iters'])       # inner iterations of the launch, summed over its SYSTEMS
        print('inner iterations per system and step %.3f' % (it / float(opts.get('--systems', waves)) / T))
