#!/usr/bin/env python
"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_*) into the files committed under profiles/:
       python tools/profile_digest.py r03
   rNN_bench_kernel_stats.csv / rNN_pc_kernel_stats.csv / rNN_train_kernel_stats.csv   rocprofv3 --stats tables (copied)
   rNN_*_line_under_rocprof.json                                                       the JSON line the profiled command printed
   rNN_counters.json / rNN_counters.txt      per kernel of the bench step: launches per step, average duration (trace), the PMC
                                             passes (SQ / FETCH_SIZE / WRITE_SIZE / L2), HBM bytes calibrated on the 256 MiB copy of
                                             the same pass, MFMA-pipe busy time and its share of the launch -- what bench.py quotes
All passes profile the same command (the bench step in the shipped schedule), so one table covers every kernel of the step."""
import csv, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
SIMDS, CLOCK_GHZ = 1024, 2.4
COPY_BYTES = 268435456.0          # the calibration copy: 64 Mi floats read + 64 Mi floats written


def short(name):
    """kernel name as bench.py's plan names it: no 'void ', no argument list, template args reduced to the leading integers"""
    k = name.replace('void ', '').split('(')[0].strip()
    m = re.match(r'([\w:]+)<(.*)>$', k)
    if m:
        ints = [a.strip() for a in m.group(2).split(',')]
        lead = []
        for a_ in ints:
            if re.match(r'^-?\d+$', a_):
                lead.append(a_)
            else:
                break
        base = m.group(1)
        if base == 'wino3x3_c128_tn_kernel' or base == 'wino3x3_c128_stack_kernel':
            return '{}<{}>'.format(base, lead[0])
        if base == 'wino4_3x3_kernel' and len(ints) >= 4:        # <WT, RES, CIN, COUT, SHUF>: one name per layer type
            return '{}<{}, {}>'.format(base, ints[2], ints[3])
        if base.startswith(('wino3x3_c128', 'wino4_3x3_c128')):
            return base
        return base + ('<{}>'.format(', '.join(ints)) if ints else '')
    return k


CAL = {}          # counter -> value of the LARGEST copy dispatch of its pass (the 256 MiB calibration copy; small copies abound)


def per_kernel(path):
    acc, calls = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    per_dispatch = defaultdict(lambda: defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k].add(r['Dispatch_Id'])
            if 'copyBuffer' in k:
                per_dispatch[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for c, d in per_dispatch.items():
        CAL[c] = max(d.values())
    return {k: {c: v / len(calls[k]) for c, v in acc[k].items()} for k in acc}, {k: len(v) for k, v in calls.items()}


def find(d, suffix, prefix=None):
    """the file `<pass tag>_<suffix>` of pass directory gpurun_out/<d> (tools/profile.sh names every output after its pass:
    `-o <tag>`).  EXACT name -- round 3 took the first file that ended in the suffix and committed a round-1 leftover
    (`b_kernel_stats.csv`) under a round-3 name; if several copies exist (rocprofv3 nests per-host directories) the newest wins."""
    dd = os.path.join(G, d)
    if not os.path.isdir(dd):
        return None
    want = (prefix if prefix is not None else d.replace('prof_', '', 1)) + '_' + suffix
    hits = [os.path.join(root, fn) for root, _, files in os.walk(dd) for fn in files if fn == want]
    return max(hits, key=os.path.getmtime) if hits else None


def library_symbols():
    """every kernel base name the shipped library holds (its code objects carry the mangled names as plain bytes)"""
    so = os.path.join(ROOT, 'imgcomp_cvpr_amd', 'libimgcomp_hip.so')
    return open(so, 'rb').read() if os.path.exists(so) else None


def check_stats_describe_head(path, blob):
    """refuse a stats table whose dominant kernels are not symbols of the library as built now (a stale profile)"""
    if blob is None:
        return
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r.get('TotalDurationNs') or r.get('Percentage') or 0))
    ours = [r for r in rows if not re.search(r'at::|rocblas|Cijk_|hipblas|copyBuffer|fillBuffer|elementwise|reduce_kernel|vectorized', r['Name'])]
    for r in ours[:4]:
        base = re.split(r'[<(]', r['Name'].replace('void ', ''))[0].strip().split('::')[-1]
        if base.encode() not in blob:
            sys.exit('STALE PROFILE: {} lists kernel `{}` (top of the table) which is not in imgcomp_cvpr_amd/libimgcomp_hip.so as built now; '
                     're-run tools/profile_round.sh'.format(os.path.relpath(path, ROOT), base))


def bench_line(d):
    log = os.path.join(G, d, 'stdout.log')
    if not os.path.exists(log):
        return None
    m = re.findall(r'^\{"metric".*\}$', open(log).read(), flags=re.M)
    return json.loads(m[-1]) if m else None


BLOB = library_symbols()
for t in ('bench', 'alone', 'bench1', 'pc', 'train'):
    src = find('prof_' + t, 'kernel_stats.csv')
    if src:
        check_stats_describe_head(src, BLOB)
        shutil.copy(src, os.path.join(P, '{}_{}_kernel_stats.csv'.format(tag, t)))
    line = bench_line('prof_' + t)
    if line:
        json.dump(line, open(os.path.join(P, '{}_{}_line_under_rocprof.json'.format(tag, t)), 'w'))

# ---- per-kernel durations of the bench step from the kernel trace ----
trace = find('prof_bench', 'kernel_trace.csv')
dur, cnt = defaultdict(float), defaultdict(int)
wgs, all_k = {}, []
if trace:
    with open(trace) as f:
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            b, e = float(r['Start_Timestamp']), float(r['End_Timestamp'])
            dur[k] += (e - b) / 1e3
            cnt[k] += 1
            wgs[k] = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(
                int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']), 1)
            all_k.append((b, e, k))
# the same kernels one image at a time (tools/profile_round.sh `alone`): a launch's duration with nothing beside it
dur1, cnt1 = defaultdict(float), defaultdict(int)
trace1 = find('prof_alone', 'kernel_trace.csv')
if trace1:
    with open(trace1) as f:
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            dur1[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
            cnt1[k] += 1
line = bench_line('prof_bench') or {}
n_if = int(line.get('images_in_flight') or 1)
steps = int(line.get('steps', 20)) + int(line.get('warmup', 5)) + (n_if if n_if > 1 else 0)      # + one set-up pass per pipeline
cfg = line.get('config', {})
# launches of different images overlap (bench.py --in_flight): concurrency = sum of the kernel durations inside the span of the
# steps / that span.  A kernel's share of the chip's time per launch is avg_us / concurrency.
# (steady state only: the window from the middle 3x3 launch of the run to the last one -- the first steps load code objects)
k3s = sorted(t for t in all_k if t[2].startswith(('wino3x3', 'wino4_3x3_kernel<128, 128>', 'conv3x3_c128')))
span_us = in_span = union_us = 0.0
steps_in_window = 0.0
k3_in_window_us = k3_in_window_n = 0
if len(k3s) >= 4:
    w0, w1 = k3s[len(k3s) // 2][0], max(t[1] for t in k3s)
    span_us = (w1 - w0) / 1e3
    inside = [(max(b, w0), min(e, w1), k) for b, e, k in all_k if e > w0 and b < w1 and 'copyBuffer' not in k]
    in_span = sum((e - b) / 1e3 for b, e, k in inside)
    # time with at least one kernel running (union of the intervals): span - union = the window's idle time
    cur_b = cur_e = None
    for b, e, k in sorted(inside):
        if cur_e is None or b > cur_e:
            if cur_e is not None:
                union_us += (cur_e - cur_b) / 1e3
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union_us += (cur_e - cur_b) / 1e3
    per_step_3x3 = len(k3s) / float(steps)                      # 3x3 launches per step (64 for the cvpr networks)
    in_w = [t for t in k3s if t[0] >= w0]
    k3_in_window_n = len(in_w)
    k3_in_window_us = sum((e - b) / 1e3 for b, e, k in in_w)
    steps_in_window = k3_in_window_n / per_step_3x3
concurrency = in_span / span_us if span_us > 0 else 1.0
out = {'source': 'profiles/{}_counters.txt: rocprofv3 --kernel-trace and --pmc passes over `python bench.py --steps 20 --warmup 5 --no_extras '
                 '--calib_copy` (tools/profile_round.sh), one counter group per pass'.format(tag),
       'input_shape': [cfg.get('batch_per_gpu'), 3, cfg.get('height'), cfg.get('width')], 'branch_sharing': line.get('branch_sharing'),
       'steps_profiled': steps, 'value_under_rocprof': line.get('value'), 'images_in_flight': line.get('images_in_flight'),
       # the WINDOW every figure below refers to: from the middle 3x3 launch of the trace to the end of the last one (steady state;
       # the first steps load code objects).  Round 4's file put the window's sums next to the count of ALL steps; now each sum
       # has the number of steps it covers beside it, and the identities hold:
       #   concurrency = sum_of_kernel_durations_us / span_of_the_window_us
       #   sum_of_kernel_durations_us / steps_in_window = kernel time per step (3x3 part: launches_per_step x avg_us_in_flight)
       'window': {'steps_in_window': round(steps_in_window, 2), 'span_of_the_window_us': round(span_us, 1),
                  'sum_of_kernel_durations_us': round(in_span, 1), 'union_busy_us': round(union_us, 1),
                  'kernel_us_per_step': round(in_span / steps_in_window, 1) if steps_in_window else None,
                  'wall_us_per_step': round(span_us / steps_in_window, 1) if steps_in_window else None,
                  'launches_3x3_in_window': k3_in_window_n,
                  'avg_us_3x3_in_window': round(k3_in_window_us / k3_in_window_n, 2) if k3_in_window_n else None},
       'span_of_the_steps_us': round(span_us, 1), 'sum_of_kernel_durations_us': round(in_span, 1), 'concurrency': round(concurrency, 3),
       'kernels': {}}
groups = {}
for grp in ('sq', 'fetch', 'write', 'l2'):
    path = find('prof_b_' + grp, 'counter_collection.csv')
    if path:
        groups[grp] = per_kernel(path)
lines = ['rocprofv3 passes over the bench step (python bench.py --steps 20 --warmup 5 --no_extras --calib_copy), per-launch averages per kernel.',
         'Images in flight: {}; launches of different images overlap: sum of kernel durations / span of the steps = {:.3f} (concurrency);'.format(
             line.get('images_in_flight'), concurrency),
         'avg_us is a launch\'s duration one image at a time (pass `alone`: the same kernels, nothing beside them); in flight = its duration',
         'in the shipped schedule under the tracer (first work-group start to last work-group end), / concurrency = its share of the chip\'s time.',
         'FETCH_SIZE / WRITE_SIZE are in KiB; the 256 MiB device copy at the end of the same pass calibrates them (gfx950: FETCH_SIZE tallies',
         '128-byte fabric reads at 64 B -> x2, MI355X_MICROARCH.md "HBM"; WRITE_SIZE exact).  mfma_busy_us = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs',
         '/ 2.4 GHz.  avg_us = kernel-trace duration.', '']
names = sorted(dur, key=lambda k: -dur[k])
cal = dict(CAL)
f_fetch = COPY_BYTES / (cal['FETCH_SIZE'] * 1024.0) if cal.get('FETCH_SIZE') else 2.0
f_write = COPY_BYTES / (cal['WRITE_SIZE'] * 1024.0) if cal.get('WRITE_SIZE') else 1.0
f_req = COPY_BYTES / cal['TCP_TCC_READ_REQ_sum'] if cal.get('TCP_TCC_READ_REQ_sum') else 128.0
out['calibration'] = {'fetch_factor': round(f_fetch, 3), 'write_factor': round(f_write, 3), 'bytes_per_tcp_tcc_read_req': round(f_req, 1)}
for k in names:
    if 'copyBuffer' in k or cnt[k] < steps // 2:
        continue
    ent = {'launches_per_step': round(cnt[k] / float(steps), 2),
           'avg_us_rocprof': round(dur1[k] / cnt1[k], 2) if cnt1.get(k) else round(dur[k] / cnt[k], 2),     # alone (one image at a time)
           'avg_us_in_flight': round(dur[k] / cnt[k], 2),
           'share_of_kernel_time': round(dur[k] / sum(dur.values()), 4), 'work_groups': wgs.get(k),
           'avg_us_over_concurrency': round(dur[k] / cnt[k] / concurrency, 2)}
    c = {}
    for grp, (vals, _) in groups.items():
        c.update(vals.get(k, {}))
    if c:
        ent['counters'] = {n: round(v, 1) for n, v in sorted(c.items())}
        if 'FETCH_SIZE' in c:
            ent['hbm_read_bytes_per_launch'] = int(c['FETCH_SIZE'] * 1024.0 * f_fetch)
        if 'WRITE_SIZE' in c:
            ent['hbm_write_bytes_per_launch'] = int(c['WRITE_SIZE'] * 1024.0 * f_write)
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            ent['hbm_bytes_per_launch'] = ent['hbm_read_bytes_per_launch'] + ent['hbm_write_bytes_per_launch']
            ent['hbm_gb_per_s'] = round(ent['hbm_bytes_per_launch'] / ent['avg_us_rocprof'] / 1e3, 0)
        if 'TCP_TCC_READ_REQ_sum' in c:
            ent['l2_to_l1_bytes_per_launch'] = int(c['TCP_TCC_READ_REQ_sum'] * f_req)
        if c.get('TCC_HIT_sum') is not None and c.get('TCC_MISS_sum') is not None and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            ent['l2_hit_rate'] = round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 3)
        if c.get('SQ_INSTS_MFMA'):
            ent['mfma_per_launch'] = int(c['SQ_INSTS_MFMA'])
            # SQ_INSTS_VALU counts the MFMAs too (they are VALU-class): vector instructions BESIDE the matrix pipe = the difference
            ent['valu_per_mfma'] = round((c.get('SQ_INSTS_VALU', 0) - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'], 2)
            ent['valu_incl_mfma_per_mfma'] = round(c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_MFMA'], 2)
        if c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            ent['mfma_busy_us'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / SIMDS / CLOCK_GHZ / 1e3, 2)
            ent['mfma_busy_share'] = round(ent['mfma_busy_us'] / ent['avg_us_rocprof'], 3)
            if wgs.get(k) and wgs[k] < 256:
                # one work-group per CU on fewer than all CUs (the whole-K 3x3 form on a Kodak map: 192): the busy cycles are
                # counted chip-wide, the launch only occupies work_groups / 256 of the chip
                ent['mfma_busy_share_of_occupied_cus'] = round(ent['mfma_busy_share'] * 256.0 / wgs[k], 3)
    out['kernels'][k] = ent
    lines.append('== {}   ({} per step, {} work-groups, avg {} us alone, {} us in flight, / concurrency {} us, {:.1f} % of the step\'s kernel time)'.format(
        k, ent['launches_per_step'], ent['work_groups'], ent['avg_us_rocprof'], ent['avg_us_in_flight'], ent['avg_us_over_concurrency'],
        100 * ent['share_of_kernel_time']))
    for n, v in sorted(ent.get('counters', {}).items()):
        lines.append('   {:28s} {:16.1f}'.format(n, v) + ('      [256 MiB copy: {:.1f}]'.format(cal[n]) if n in cal else ''))
    for n in ('hbm_read_bytes_per_launch', 'hbm_write_bytes_per_launch', 'hbm_gb_per_s', 'l2_to_l1_bytes_per_launch', 'l2_hit_rate', 'valu_per_mfma',
              'mfma_busy_us', 'mfma_busy_share', 'mfma_busy_share_of_occupied_cus'):
        if n in ent:
            lines.append('   -> {:30s} {}'.format(n, ent[n]))
    lines.append('')
k3 = [k for k in out['kernels'] if k.startswith(('wino3x3_c128', 'wino4_3x3_kernel<128, 128>', 'conv3x3_c128'))]
out['plan_3x3'] = ' + '.join(sorted(k3, key=lambda k: -out['kernels'][k]['launches_per_step'])) if k3 else None
open(os.path.join(P, tag + '_counters.txt'), 'w').write('\n'.join(lines) + '\n')
json.dump(out, open(os.path.join(P, tag + '_counters.json'), 'w'), indent=1)
print('\n'.join(lines[:400]))
