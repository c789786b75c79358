#!/usr/bin/env python
"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_*) into the files committed under profiles/:
   python tools/profile_digest.py r02"""
import csv, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')


def per_kernel(path):
    acc, calls = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k].add(r['Dispatch_Id'])
    return {k: {c: v / len(calls[k]) for c, v in acc[k].items()} for k in acc}, {k: len(v) for k, v in calls.items()}


def find(d, suffix, prefix=''):
    for fn in sorted(os.listdir(os.path.join(G, d))):
        if fn.endswith(suffix) and fn.startswith(prefix):
            return os.path.join(G, d, fn)
    return None


for t in ('bench', 'pc', 'train'):
    src = find('prof_' + t, 'kernel_stats.csv', t + '_')
    if src:
        shutil.copy(src, os.path.join(P, '{}_{}_kernel_stats.csv'.format(tag, t)))
    log = os.path.join(G, 'prof_' + t, 'stdout.log')
    if os.path.exists(log):
        m = re.findall(r'^\{"metric".*\}$', open(log).read(), flags=re.M)
        if m:
            open(os.path.join(P, '{}_{}_line_under_rocprof.json'.format(tag, t)), 'w').write(m[-1] + '\n')

traffic = {'shape': [1, 128, 128, 192], 'source': 'profiles/{}_wino_pmc.txt: rocprofv3 --pmc passes over tools/run_layer.py (tools/profile_round.sh), one counter group per pass'.format(tag),
           'kernels': {}}
lines = ['rocprofv3 --pmc passes, one 3x3 128->128 layer on the Kodak residual-stack shape (1,128,128,192), 20 back-to-back launches per pass,',
         'per-launch averages.  FETCH_SIZE / WRITE_SIZE are in KiB (TCC_EA0 requests x 64 B / 1024).  Calibration in the same pass: a 256 MiB',
         'device-to-device copy (64 Mi floats read, 64 Mi written).', '']
for form, kname in (('seg3', 'wino3x3_c128_tn_kernel<3, false, true>'), ('wholek', 'wino3x3_c128_shared_kernel<true>')):
    ent = {}
    lines.append('== form {} : kernel {}'.format(form, kname))
    for grp in ('fetch', 'write', 'l2', 'ea', 'sq'):
        path = find('prof_l_{}_{}'.format(form, grp), 'counter_collection.csv')
        if not path:
            continue
        vals, calls = per_kernel(path)
        k = vals.get(kname, {})
        cal = vals.get('__amd_rocclr_copyBuffer', {})
        for c, v in sorted(k.items()):
            lines.append('   {:28s} {:16.1f}'.format(c, v) + ('      [256 MiB copy: {:.1f}]'.format(cal[c]) if c in cal else ''))
            ent[c] = v
            if c in cal:
                ent['calib_copy_' + c] = cal[c]
    if 'FETCH_SIZE' in ent:
        # gfx950: FETCH_SIZE tallies 128-byte fabric reads at 64 B (MI355X_MICROARCH.md, HBM section); the 256 MiB copy of the same
        # pass calibrates it: bytes = counter KiB * 1024 * (268435456 / (calib KiB * 1024))
        f = 268435456.0 / (ent['calib_copy_FETCH_SIZE'] * 1024.0)
        w = 268435456.0 / (ent['calib_copy_WRITE_SIZE'] * 1024.0)
        rd, wr = ent['FETCH_SIZE'] * 1024.0 * f, ent['WRITE_SIZE'] * 1024.0 * w
        l2l1 = ent['TCP_TCC_READ_REQ_sum'] * (268435456.0 / ent['calib_copy_TCP_TCC_READ_REQ_sum'])
        kkey = re.sub(r'<(\d+), \w+, \w+>', r'<\1>', kname).replace('<true>', '')      # bench.py's plan names
        traffic['kernels'][kkey] = {
            'fetch_counter_kib': ent['FETCH_SIZE'], 'write_counter_kib': ent['WRITE_SIZE'], 'fetch_calibration_factor': round(f, 3),
            'write_calibration_factor': round(w, 3), 'hbm_read_bytes_per_launch': int(rd), 'hbm_write_bytes_per_launch': int(wr),
            'hbm_bytes_per_launch': int(rd + wr), 'l2_to_l1_bytes_per_launch': int(l2l1),
            'l2_hit_rate': round(ent['TCC_HIT_sum'] / (ent['TCC_HIT_sum'] + ent['TCC_MISS_sum']), 3),
            'mfma_per_launch': int(ent.get('SQ_INSTS_MFMA', 0)), 'valu_per_mfma': round(ent.get('SQ_INSTS_VALU', 0) / max(ent.get('SQ_INSTS_MFMA', 1), 1), 2),
            'mfma_busy_share_of_wave_cycles': round(ent.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(4.0 * ent.get('SQ_WAVE_CYCLES', 1), 1), 3)}
        lines.append('   -> HBM read {:.1f} MB + written {:.1f} MB per launch (calibrated x{:.2f} / x{:.2f}); L2 -> L1 {:.1f} MB; L2 hit rate {:.2f}'.format(
            rd / 1e6, wr / 1e6, f, w, l2l1 / 1e6, traffic['kernels'][kkey]['l2_hit_rate']))
    lines.append('')
open(os.path.join(P, tag + '_wino_pmc.txt'), 'w').write('\n'.join(lines) + '\n')
json.dump(traffic, open(os.path.join(P, tag + '_conv3x3_traffic.json'), 'w'), indent=1)
print('\n'.join(lines))
