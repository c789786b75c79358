#!/usr/bin/env python
"""In-kernel launch stamps of the F(4x4) kernel on the schedule BENCH measures -- the tracer-free evidence for `roofline.frac`
(rocprofv3 serialises the streams, 160.7 against 257 Mpix/s, so no kernel trace describes the in-flight step).

Needs a library built with -DW4_LAUNCH_STAMPS (every launch of wino4_3x3_kernel then writes one record: first instruction of its first
wave and last store acknowledgement of its last wave on the 100 MHz real-time counter all XCDs share, the waves' shader clocks, MFMAs):
    tools/build_variants.sh conv3x3_wino4.hip ls="-fno-slp-vectorize -DW4_LAUNCH_STAMPS"
    IMGCOMP_HIP_LIB=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_ls.so python tools/w4_inflight_stamps.py [--out profiles/r05_inflight_stamps.json]

Two scenarios, both with the default plan flags of bench.py (4 Kodak-sized images in flight):
  step    bench.InFlight(4): the timed region of bench.py itself, K steps.  Every F(4x4) launch of it -- 64 x <128,128> per image plus
          h2 <256,128> and h12 <128,256> -- gives [start, end]; the UNION of the <128,128> intervals / their number is the time in which
          the chip completes one launch of the dominant kernel under the real concurrency (other kernels share the chip meanwhile).
  stacks  what bench.py's `roofline` times with HIP events: 4 residual stacks in flight, nothing else.  The same run is measured both
          ways, so the events figure is checked against the stamps.
Chip-level MFMA-issue share = MFMAs issued x 32 clocks (8 passes of 4) / (1024 SIMDs x union time x shader clock), the shader clock
under this load from the waves' own two counters."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from imgcomp_cvpr_amd import _lib, weights as W

PEAK = bench.PEAK_F32_MFMA_TFLOPS
CAP_WAVES = 1 << 23          # 4 x u64 each: 256 MB
CAP_LAUNCHES = 1 << 16


def union_us(iv):
    """iv: (n, 2) ticks of 10 ns -> total length of the union of the intervals, in microseconds"""
    iv = iv[np.argsort(iv[:, 0])]
    tot, cs, ce = 0, iv[0, 0], iv[0, 1]
    for s, e in iv[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return (tot + ce - cs) * 0.01


class Stamps(object):
    def __init__(self, dev):
        self.raw = ctypes.CDLL(_lib.LIB_PATH)
        if not hasattr(self.raw, 'ic_wino4_launch_stamps_set'):
            raise SystemExit('this library has no launch stamps: build with -DW4_LAUNCH_STAMPS and point IMGCOMP_HIP_LIB at it')
        self.raw.ic_wino4_launch_stamps_count.restype = ctypes.c_longlong
        self.buf = torch.zeros((CAP_WAVES, 4), dtype=torch.int64, device=dev)
        self.table = (ctypes.c_longlong * (3 * CAP_LAUNCHES))()
        self.arm()

    def arm(self):
        torch.cuda.synchronize()
        self.buf.zero_()
        torch.cuda.synchronize()
        self.raw.ic_wino4_launch_stamps_set(ctypes.c_void_p(self.buf.data_ptr()), ctypes.c_longlong(CAP_WAVES), self.table, ctypes.c_longlong(CAP_LAUNCHES))

    def read(self):
        torch.cuda.synchronize()
        n = int(self.raw.ic_wino4_launch_stamps_count())
        assert 0 < n < CAP_LAUNCHES, n
        tab = np.array(self.table[:3 * n], dtype=np.int64).reshape(n, 3)
        used = int(tab[-1, 0] + tab[-1, 1])
        b = self.buf[:used].cpu().numpy()
        start, end, clocks, ticks, waves, mfma = (np.zeros(n, np.float64) for _ in range(6))
        start, end, waves = start.astype(np.int64), end.astype(np.int64), waves.astype(np.int64)
        for i, (first, nw, kind) in enumerate(tab):
            r = b[first:first + nw]
            assert (r[:, 1] > 0).all(), 'launch {} has waves without a stamp'.format(i)
            start[i], end[i] = r[:, 0].min(), r[:, 1].max()
            clocks[i], ticks[i], waves[i], mfma[i] = r[:, 2].sum(), (r[:, 1] - r[:, 0]).sum(), nw, r[:, 3].sum()
        return {'start': start, 'end': end, 'clocks': clocks, 'ticks': ticks, 'waves': waves, 'mfma': mfma, 'kind': tab[:, 2].copy()}


def digest(r, flop_per_launch_dominant):
    out = {}
    ghz = float(r['clocks'].sum() / r['ticks'].sum() * 0.1)          # shader clocks per 10 ns tick -> GHz
    iv_all = np.stack([r['start'], r['end']], 1)
    u_all = union_us(iv_all)
    out['launches'] = int(len(r['start']))
    out['shader_clock_ghz_under_load'] = round(ghz, 3)
    out['union_busy_us_all_f4_launches'] = round(u_all, 2)
    out['chip_mfma_issue_share'] = round(float(r['mfma'].sum() * 32.0 / (1024.0 * u_all * 1e-6 * ghz * 1e9)), 4)
    out['chip_mfma_issue_share_note'] = 'MFMAs x 32 clocks / (1024 SIMDs x union time x measured clock); against the 2.4 GHz peak multiply by clock / 2.4'
    kinds = {}
    for k in sorted(set(r['kind'].tolist())):
        m = r['kind'] == k
        iv = iv_all[m]
        dur = (iv[:, 1] - iv[:, 0]) * 0.01
        u = union_us(iv)
        ent = {'launches': int(m.sum()), 'waves_per_launch': int(np.median(r['waves'][m])),
               'launch_duration_us': {'mean': round(float(dur.mean()), 2), 'median': round(float(np.median(dur)), 2),
                                      'p10': round(float(np.percentile(dur, 10)), 2), 'p90': round(float(np.percentile(dur, 90)), 2)},
               'union_busy_us': round(u, 2), 'us_per_launch_under_concurrency': round(u / int(m.sum()), 3),
               'concurrency': round(float(dur.sum() / u), 3),
               'wave_busy_clocks_mean': round(float(r['clocks'][m].sum() / r['waves'][m].sum()), 0)}
        kinds['wino4_3x3_kernel<{}, {}>'.format(k // 1000, k % 1000)] = ent
    out['kernels'] = kinds
    dom = kinds.get('wino4_3x3_kernel<128, 128>')
    if dom:
        us = dom['us_per_launch_under_concurrency']
        out['dominant'] = {'kernel': 'wino4_3x3_kernel<128, 128>', 'executed_flop_per_launch': flop_per_launch_dominant,
                           'us_per_launch_under_concurrency': us,
                           'achieved_tflops': round(flop_per_launch_dominant / (us * 1e-6) / 1e12, 2),
                           'frac': round(flop_per_launch_dominant / (us * 1e-6) / 1e12 / PEAK, 4),
                           'frac_alone_equivalent': round(flop_per_launch_dominant / (dom['launch_duration_us']['mean'] * 1e-6) / 1e12 / PEAK, 4)}
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_inflight_stamps.json'))
    p.add_argument('--steps', type=int, default=48)
    p.add_argument('--in_flight', type=int, default=4)
    p.add_argument('--height', type=int, default=512)
    p.add_argument('--width', type=int, default=768)
    a = p.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.lib
    N, H, Wd, n = 1, a.height, a.width, a.in_flight
    st = Stamps(dev)
    first = bench.Pipeline(dev, 'low', 'serial', seed=0).set_input(N, H, Wd)
    sched = bench.InFlight(torch, first, dev, n, 'low', 0)
    flags = sched.pipes[0].ae.plan_flags
    form = int(lib.ic_conv3x3_c128_pick_form(N, H // 4, Wd // 4, flags))
    assert form == 2, 'the plan does not pick F(4x4) here (form {})'.format(form)
    flop = bench.CONV3_FLOP_PER_OUT_PX * N * (H // 4) * (Wd // 4) * 36.0 / 144.0
    res = {'source': 'tools/w4_inflight_stamps.py on a -DW4_LAUNCH_STAMPS build of conv3x3_wino4.hip (in-kernel s_memrealtime / s_memtime, no tracer)',
           'input_shape': [N, 3, H, Wd], 'images_in_flight': n, 'peak_tflops': PEAK}
    # ---- scenario `step`: bench.py's timed region ----
    for _ in range(3 * n):
        sched.step()
    st.arm()
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sched.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d = digest(st.read(), flop)
    d['steps'] = a.steps
    d['ms_per_step_wall'] = round(dt / a.steps * 1e3, 4)
    d['mpix_per_s_wall'] = round(N * H * Wd * a.steps / dt / 1e6, 2)
    d['note'] = 'stamps cost time: compare mpix_per_s_wall with BENCH on the shipped library before quoting absolute figures'
    res['step'] = d
    # ---- scenario `stacks`: what bench.py's roofline times with HIP events ----
    pipe = first
    enc = pipe.ae.encode(pipe.x, False)
    stream = _lib.current_stream(dev)
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        _lib.check(lib.ic_event_create(ctypes.byref(e)))
    gos = [bench.res_stack_runner(torch, lib, _lib, W, pipe.ae, pipe.ae_cfg, pipe, enc, 'enc', flags, stream)[0] for _ in range(n)]
    nl = 32
    bench.timed_concurrent_stacks(torch, lib, _lib, dev, ev, stream, gos, 2, 1)
    st.arm()
    ms = bench.timed_concurrent_stacks(torch, lib, _lib, dev, ev, stream, gos, 6, 0) / nl
    d2 = digest(st.read(), flop)
    d2['hip_events_us_per_launch'] = round(ms * 1e3, 3)
    d2['hip_events_frac'] = round(flop / (ms * 1e-3) / 1e12 / PEAK, 4)
    d2['events_over_stamps'] = round(ms * 1e3 / d2['dominant']['us_per_launch_under_concurrency'], 4)
    res['stacks'] = d2
    for e in ev:
        lib.ic_event_destroy(e)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps({'step': {k: res['step'][k] for k in ('dominant', 'chip_mfma_issue_share', 'shader_clock_ghz_under_load', 'mpix_per_s_wall')},
                      'stacks': {k: res['stacks'][k] for k in ('dominant', 'hip_events_us_per_launch', 'hip_events_frac', 'events_over_stamps')}}))


if __name__ == '__main__':
    main()
