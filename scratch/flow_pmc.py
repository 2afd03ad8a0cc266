"""SQ counters of the flow-embedding kernels alone (scratch/flow_probe.py under rocprofv3 --pmc, one pass per counter group;
run on the GPU box from the repo root). Prints per kernel the per-launch averages and a few ratios."""
import glob, os, subprocess, sys, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'r06_flow_pmc')
GROUPS = [
    ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'GRBM_GUI_ACTIVE'],
    ['SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM'],
    ['SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_INST_CYCLES_VMEM'],
]
os.makedirs(OUT, exist_ok=True)
env = dict(os.environ)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for gi, grp in enumerate(GROUPS):
    d = os.path.join(OUT, 'raw%d' % gi)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + grp + ['--output-format', 'csv', '-d', d, '--', sys.executable, os.path.join(ROOT, 'scratch', 'flow_probe.py')]
    with open(os.path.join(OUT, 'pass%d.log' % gi), 'w') as fh:
        rc = subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, cwd=ROOT, env=env).returncode
    print('pass', gi, 'rc', rc, flush=True)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        seen = set()
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                kn = row['Kernel_Name']
                if 'flow' not in kn or 'kernel' not in kn:
                    continue
                key = (kn.split('(')[0][-40:], row['Grid_Size'])
                acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
                if gi == 0 and row['Dispatch_Id'] not in seen and row.get('End_Timestamp'):
                    seen.add(row['Dispatch_Id'])
                    acc[key]['_dur_ns'].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
with open(os.path.join(OUT, 'summary.txt'), 'w') as out:
    for key, ctrs in sorted(acc.items()):
        lines = ['== %s grid %s' % key]
        avg = {c: sum(v) / len(v) for c, v in ctrs.items()}
        for c in sorted(avg):
            lines.append('   %-28s %16.0f  (%d launches)' % (c, avg[c], len(ctrs[c])))
        if 'SQ_BUSY_CYCLES' in avg and 'GRBM_GUI_ACTIVE' in avg:
            g = avg['GRBM_GUI_ACTIVE']
            simd_cycles = g * 256 * 4
            for c, scale, what in (('SQ_ACTIVE_INST_VALU', 4, 'VALU issue busy (x4 cycles per wave64 op? raw/SIMD-cycles)'), ('SQ_VALU_MFMA_BUSY_CYCLES', 1, 'MFMA busy'),
                                   ('SQ_ACTIVE_INST_LDS', 1, 'LDS inst active'), ('SQ_ACTIVE_INST_VMEM', 1, 'VMEM inst active'), ('SQ_WAIT_INST_ANY', 1, 'wait inst any (per wave)'),
                                   ('SQ_WAIT_ANY', 1, 'wait any (per wave)'), ('SQ_WAVE_CYCLES', 1, 'wave cycles')):
                if c in avg:
                    lines.append('   ratio %-24s / (GUI_ACTIVE x 1024 SIMDs) = %.3f   [%s]' % (c, avg[c] / simd_cycles, what))
        print('\n'.join(lines)); out.write('\n'.join(lines) + '\n')
