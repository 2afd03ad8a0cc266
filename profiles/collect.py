#!/usr/bin/env python3
"""Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box, from the repo root):

    python3 profiles/collect.py --tag r02 [--configs c2,c4,c5] [--skip-pmc]

Per config: (1) `rocprofv3 --kernel-trace --stats` over `python3 bench.py --config X ...` -> the kernel-stats CSV;
(2) two SEPARATE `--pmc` passes (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md
'rocprofv3 PMC slots'; never combined with a trace domain other than --kernel-trace) -> HBM bytes per launch.
A calibration pass (a float4 copy of known size through torch) measures how FETCH_SIZE reads on this box for
wide streaming loads (the guide: exactly 1/2 of the bytes on gfx950); `bytes_per_launch` = FETCH_SIZE x that
factor + WRITE_SIZE, both raw figures are kept beside it. Everything is written under gpurun_out/<tag>/ (the
only directory gpurun copies back); the summaries to commit are then copied into profiles/ by hand or with
--install when run in the build container on the merged gpurun_out/.

The JSON is stamped with the git commit (if known) and a hash of the kernel sources; bench.py attaches the figures
to its `roofline.traffic` only when that hash matches the kernels it is running.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (the f32-matrix-path kernels run once, in the range-checked first forward: they get span names of their own so that their
# counters are not averaged into the split-f16 kernels' -- a default build times the split-f16 kernels)
KERNEL_TO_SPAN = [('fps_', 'fps_clouds'), ('sa_msg_kernel', 'sa_msg_fused'), ('knn_rows_kernel', 'knn_rows'),
                  ('flow16_kernel', 'flow_embedding'), ('flow32_kernel', 'flow_embedding'), ('flow_kernel', 'flow_embedding_f32'),
                  ('head16', 'head_conv_fused'), ('head_fused_kernel', 'head_conv_fused_f32'),
                  ('linear_kernel', 'linear_pair'), ('fc_kernel', 'fc')]
# enough steps for several grouped launches of every kind in steady state (c2: 10 batches per launch, c5: 20), so that
# the per-kernel averages are those of the contended run bench.py's `avg_us` reports
BENCH_ARGS = {'c2': ['--steps', '60', '--warmup', '20'], 'c4': ['--steps', '6', '--warmup', '2'],
              'c5': ['--steps', '80', '--warmup', '40']}
CALIB = os.path.join(ROOT, 'profiles', 'calib_copy.py')
# the measurement lines beside the c2 headline: bench arguments per mode
MODES = {
    'driver': ['--steps', '20', '--warmup', '5'],                           # the driver's own arguments
    'default': [],                                                          # 200 steps
    'strict': ['--strict', '--steps', '200', '--warmup', '20'],            # one batch of 8 pairs per launch, no cross-batch fusion
    'serial': ['--no-overlap', '--steps', '100', '--warmup', '10'],        # one batch at a time, one stream
    'h2d': ['--h2d'],                                                       # batch copied from pinned host memory every step
    'h2d_driver': ['--h2d', '--steps', '20', '--warmup', '5'],
    'ring': ['--clouds', 'ring'],                                           # LiDAR-density clouds: nsample caps reached
    'ring_strict': ['--clouds', 'ring', '--strict', '--steps', '200', '--warmup', '20'],
    'ring_c5': ['--config', 'c5', '--clouds', 'ring'],
    'c4': ['--config', 'c4'],
    'c5': ['--config', 'c5'],
    'latency': ['--latency', '--steps', '50', '--warmup', '10'],            # B = 1, one pair per predict call
    'latency_ring': ['--latency', '--clouds', 'ring', '--steps', '50', '--warmup', '10'],
}


def span_of(kernel_name: str, calib: bool = False):
    if calib:
        return 'calibration' if 'vectorized_elementwise_kernel' in kernel_name else None
    for key, span in KERNEL_TO_SPAN:
        if key in kernel_name:
            return span
    return None


def run(cmd, log):
    with open(log, 'w') as fh:
        print('+', ' '.join(cmd), flush=True)
        rc = subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, cwd=ROOT).returncode
    if rc != 0:
        raise SystemExit('{} failed with {} (see {})'.format(cmd[0], rc, log))


def counter_rows(directory: str):
    """(kernel name, counter name, value) from every *counter_collection.csv below `directory`; every dispatch also yields
    the pseudo-counter '_dur_ns' (End_Timestamp - Start_Timestamp) once."""
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        seen = set()
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                yield row['Kernel_Name'], row['Counter_Name'], float(row['Counter_Value'])
                if row['Dispatch_Id'] not in seen and row.get('End_Timestamp'):
                    seen.add(row['Dispatch_Id'])
                    yield row['Kernel_Name'], '_dur_ns', float(row['End_Timestamp']) - float(row['Start_Timestamp'])


def per_span_average(directory: str, counter: str, calib: bool = False):
    acc = {}
    for kname, cname, value in counter_rows(directory):
        span = span_of(kname, calib)
        if cname != counter or span is None:
            continue
        tot, cnt = acc.get(span, (0.0, 0))
        acc[span] = (tot + value, cnt + 1)
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def git_commit():
    try:
        return subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=ROOT, capture_output=True, text=True,
                              check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return os.environ.get('DCLR_COMMIT', 'unknown (no .git on the GPU box)')


def run_modes(args, out, py):
    """The bench lines beside the headline (MODES), each with the rocprofv3 kernel-stats summary of the same command. Run
    AFTER the PMC passes: profiles/pmc_traffic.json then holds this tree's figures and the lines carry `traffic` / `mfma_busy`."""
    for mode in [m for m in args.modes.split(',') if m]:
        extra = MODES[mode]
        line = os.path.join(out, '{}_bench_{}.json'.format(args.tag, mode))
        # (the driver's own command carries the `secondary` block; every other line is measured alone)
        run([py, 'bench.py'] + extra + ([] if mode == 'driver' else ['--no-secondary']), line)
        with open(line) as fh:                                 # keep the JSON line only (stderr chatter goes to the log)
            txt = fh.read()
        rows = [l for l in txt.splitlines() if l.startswith('{')]
        with open(line, 'w') as fh:
            fh.write((rows[-1] if rows else txt) + '\n')
        if rows:
            doc_ = json.loads(rows[-1])
            print(mode, 'value', round(doc_['value'], 1), doc_['unit'], 'ms/step', round(doc_['ms_per_step'], 4),
                  'pose', doc_.get('pose_delta_vs_oracle'), flush=True)
        d = os.path.join(out, 'raw_stats_' + mode)
        run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--', py, 'bench.py'] + extra +
            ['--no-cpu-baseline', '--no-launch-timer', '--no-secondary'], os.path.join(out, '{}_{}_bench_under_rocprof.log'.format(args.tag, mode)))
        for path in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
            shutil.copy(path, os.path.join(out, '{}_{}_kernel_stats.csv'.format(args.tag, mode)))
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--configs', default='c2,c4,c5')
    ap.add_argument('--skip-pmc', action='store_true')
    ap.add_argument('--skip-stats', action='store_true')
    ap.add_argument('--install', action='store_true', help='copy the summaries from gpurun_out/<tag>/ into profiles/')
    ap.add_argument('--modes', default='', help='comma list of the lines beside the headline (MODES below): per mode the '
                                                'bench line (JSON) and the rocprofv3 kernel-stats summary of the same command')
    args = ap.parse_args()
    out = os.path.join(ROOT, 'gpurun_out', args.tag)
    if args.install:
        for path in glob.glob(os.path.join(out, '*')):
            if os.path.isfile(path):
                shutil.copy(path, os.path.join(ROOT, 'profiles', os.path.basename(path)))
        traffic = os.path.join(out, '{}_pmc_traffic.json'.format(args.tag))
        if os.path.exists(traffic):
            shutil.copy(traffic, os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))
        return
    os.makedirs(out, exist_ok=True)
    py = sys.executable
    if args.configs == 'none':
        run_modes(args, out, py)
        return
    from bench import kernel_source_hash
    doc = {'_how': __doc__.split('\n\n')[1].replace('\n', ' '), 'commit': git_commit(),
           'kernel_source_hash': kernel_source_hash(), 'configs': {}}
    fetch_factor = None
    if not args.skip_pmc:
        calib = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(out, 'raw_calib_' + counter)
            run(['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '--', py, CALIB],
                os.path.join(out, 'calib_{}.log'.format(counter)))
            calib[counter] = per_span_average(d, counter, calib=True).get('calibration', (0.0, 0))
        known = 256 * 1024 * 1024                          # bytes read and bytes written by each calibration copy
        fetch_factor = known / (calib['FETCH_SIZE'][0] * 1024) if calib['FETCH_SIZE'][0] else None
        doc['calibration'] = {'known_bytes_read': known, 'known_bytes_written': known,
                              'FETCH_SIZE_KB': calib['FETCH_SIZE'][0], 'WRITE_SIZE_KB': calib['WRITE_SIZE'][0],
                              'launches': calib['FETCH_SIZE'][1],
                              'fetch_factor': fetch_factor,
                              'write_factor': known / (calib['WRITE_SIZE'][0] * 1024) if calib['WRITE_SIZE'][0] else None,
                              'note': 'float4 streaming copy of 256 MiB (> Infinity Cache); MI355X_MICROARCH.md expects '
                                      'fetch_factor 2.0 and write_factor 1.0 for 16 B/lane streams'}
        print('calibration:', json.dumps(doc['calibration']), flush=True)
    for cfg in [c for c in args.configs.split(',') if c]:
        bench = [py, 'bench.py', '--config', cfg, '--no-cpu-baseline', '--no-secondary'] + BENCH_ARGS[cfg]
        if not args.skip_stats:
            d = os.path.join(out, 'raw_stats_' + cfg)
            run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--'] + bench + ['--no-launch-timer'],
                os.path.join(out, '{}_{}_bench_under_rocprof.log'.format(args.tag, cfg)))
            for path in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
                shutil.copy(path, os.path.join(out, '{}_{}_kernel_stats.csv'.format(args.tag, cfg)))
            # the launches one after another, nothing else on the chip: backs bench.py's `alone_us` / `frac_alone`
            d = os.path.join(out, 'raw_alone_' + cfg)
            run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--', py, 'bench.py', '--config', cfg,
                 '--alone-only', '6'], os.path.join(out, '{}_{}_alone_under_rocprof.log'.format(args.tag, cfg)))
            for path in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
                shutil.copy(path, os.path.join(out, '{}_{}_alone_kernel_stats.csv'.format(args.tag, cfg)))
        if args.skip_pmc:
            continue
        # the launch sizes of this configuration's spans (`fps_clouds[160x16384]` ...): bench.py attaches a figure only to
        # the span it was collected on
        names_log = os.path.join(out, '{}_{}_bench_line.json'.format(args.tag, cfg))
        run(bench, names_log)
        with open(names_log) as fh:
            rows_ = [l for l in fh.read().splitlines() if l.startswith('{')]
        line = json.loads(rows_[-1]) if rows_ else {}
        with open(names_log, 'w') as fh:
            fh.write((rows_[-1] if rows_ else '') + '\n')
        full_name = {}
        for name in line.get('kernels_us', {}):
            base = name.split('[')[0]
            full_name.setdefault(base, name)
            if len(name) > len(full_name[base]) or name > full_name[base]:      # several sizes (a partial first group): the larger
                full_name[base] = name
        spans = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(out, 'raw_pmc_{}_{}'.format(cfg, counter))
            run(['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '--'] + bench + ['--no-launch-timer'],
                os.path.join(out, 'pmc_{}_{}.log'.format(cfg, counter)))
            for span, (avg, cnt) in per_span_average(d, counter).items():
                spans.setdefault(span, {})[counter + '_KB'] = avg
                spans[span]['launches_' + counter] = cnt
        # matrix-pipe busy cycles of the launches running ALONE (the same launch sizes one after another): a pass of its
        # own, program directly after `--`, no trace domain beside --kernel-trace
        d = os.path.join(out, 'raw_pmc_{}_mfma'.format(cfg))
        run(['rocprofv3', '--kernel-trace', '--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', '--output-format', 'csv', '-d', d,
             '--', py, 'bench.py', '--config', cfg, '--alone-only', '6'], os.path.join(out, 'pmc_{}_mfma.log'.format(cfg)))
        busy, active = per_span_average(d, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_span_average(d, 'GRBM_GUI_ACTIVE')
        dur = per_span_average(d, '_dur_ns')
        for span in busy:
            if span in active and active[span][0] > 0 and busy[span][0] > 0:
                rec = spans.setdefault(span, {})
                rec['SQ_VALU_MFMA_BUSY_CYCLES'], rec['GRBM_GUI_ACTIVE'] = busy[span][0], active[span][0]
                # busy cycles summed over the 1024 SIMDs / (1024 x kernel cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
                # A fraction of the kernel's CYCLES at the clock the profiled run had (stored beside it: the chip clocks
                # lower under the counters than in the timed runs; a kernel bound by the fabric keeps its duration).
                rec['mfma_busy'] = busy[span][0] / (1024.0 * active[span][0] / 8.0)
                if span in dur and dur[span][0] > 0:
                    rec['mfma_pass_us'] = dur[span][0] * 1e-3
                    rec['mfma_pass_clock_ghz'] = active[span][0] / 8.0 / dur[span][0]
        for span, rec in spans.items():
            f, w = rec.get('FETCH_SIZE_KB', 0.0) * 1024, rec.get('WRITE_SIZE_KB', 0.0) * 1024
            rec['bytes_per_launch_raw'] = f + w
            rec['bytes_per_launch'] = f * (fetch_factor or 2.0) + w
            rec['span'] = full_name.get(span)
        doc['configs'][cfg] = spans
        print(cfg, json.dumps(spans), flush=True)
    if not args.skip_pmc:
        path = os.path.join(out, '{}_pmc_traffic.json'.format(args.tag))
        with open(path, 'w') as fh:
            json.dump(doc, fh, indent=1)
        print('wrote', path)
        shutil.copy(path, os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))      # the lines below attach these figures
    run_modes(args, out, py)
    for d in glob.glob(os.path.join(out, 'raw_*')):        # keep gpurun_out small: the summaries are what travels
        shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
