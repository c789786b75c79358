#!/usr/bin/env python
"""Kernel micro-benchmarks on one MI355X (HIP events on the launch stream, variants interleaved inside one process).

  python tools/kbench.py conv3 [--shapes kodak,256,256b8,train,4k,544] [--rounds 5]   every form of the 3x3 layer per shape
  python tools/kbench.py pc                                                          context model, standalone
  python tools/kbench.py edge                                                        h1 / h2 / to_bn / from_bn / h12 / h13
  python tools/kbench.py train [--steps 10]                                          cfg3 training step
Writes one JSON object per line to stdout (and to --out when given).
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from imgcomp_cvpr_amd import _lib  # noqa: E402

lib = _lib.lib
SHAPES = {'kodak': (1, 128, 192), 'kodak_p': (1, 192, 128), '256': (1, 64, 64), '256b2': (2, 64, 64), '256b4': (4, 64, 64),
          '256b8': (8, 64, 64), 'train': (32, 32, 32), '4k': (1, 540, 960), '544': (1, 136, 240), 'kodakb2': (2, 128, 192)}
FORMS = [('auto', 0), ('leave_idle', _lib.CONV3_LEAVE_IDLE_CUS), ('wholek', _lib.CONV3_WINO_WHOLEK), ('ksplit', _lib.CONV3_WINO_KSPLIT),
         ('t16', _lib.CONV3_WINO_T16), ('seg1', _lib.CONV3_WINO_SEG1), ('seg2', _lib.CONV3_WINO_SEG2), ('seg3', _lib.CONV3_WINO_SEG3), ('pair', _lib.CONV3_WINO_PAIR),
         ('seg3_pk', _lib.CONV3_WINO_SEG3 | _lib.CONV3_PACKED_TRANSFORM), ('seg2_pk', _lib.CONV3_WINO_SEG2 | _lib.CONV3_PACKED_TRANSFORM),
         ('seg3_noxcd', _lib.CONV3_WINO_SEG3 | _lib.CONV3_NO_XCD_RUNS), ('direct', _lib.CONV3_DIRECT),
         ('w4', _lib.CONV3_WINO4 | _lib.CONV3_WINO4_WG4), ('w4_wg8', _lib.CONV3_WINO4 | _lib.CONV3_WINO4_WG8)]


class Timer(object):
    def __init__(self, dev):
        self.dev = dev
        self.st = _lib.current_stream(dev)
        self.ev = [ctypes.c_void_p(), ctypes.c_void_p()]
        for e in self.ev:
            _lib.check(lib.ic_event_create(ctypes.byref(e)))

    def __call__(self, fn, reps):
        torch.cuda.synchronize(self.dev)
        lib.ic_event_record(self.ev[0], self.st)
        for _ in range(reps):
            fn()
        lib.ic_event_record(self.ev[1], self.st)
        ms = ctypes.c_float()
        _lib.check(lib.ic_event_elapsed_ms(self.ev[0], self.ev[1], ctypes.byref(ms)))
        return ms.value / reps * 1e3       # us


def emit(obj, out):
    line = json.dumps(obj)
    print(line, flush=True)
    if out:
        with open(out, 'a') as f:
            f.write(line + '\n')


def cmd_conv3(a, dev):
    t = Timer(dev)
    for name in a.shapes.split(','):
        N, H, W = SHAPES[name]
        # 8 different filters cycled (the real step cycles 64: a single filter would sit L2-hot in one place)
        nf = 8
        wps = []
        for i in range(nf):
            w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
            wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
            _lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, t.st))
            wps.append(wp)
        x = [torch.randn((N, 128, H, W), device=dev) for _ in range(2)]
        r = torch.randn((N, 128, H, W), device=dev)
        sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
        flop = 2.0 * 9 * 128 * 128 * N * H * W
        groups = N * -(-H // 4) * -(-W // 32)

        def launch(flags):
            cnt = [0]

            def go():
                i = cnt[0]
                cnt[0] += 1
                _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x[i & 1]), _lib.ptr(wps[i % nf]), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None,
                                                        _lib.ptr(x[(i + 1) & 1]), N, H, W, 1, flags, t.st))
            return go
        forms = [(n_, f) for n_, f in FORMS if not (name == '4k' and n_ in ('ksplit', 'direct', 'seg1', 't16'))
                 and not (n_ in ('t16', 'pair') and not lib.ic_build_has_tuning_forms())]
        res = {n_: [] for n_, _ in forms}
        for n_, f in forms:              # warm-up (clock ramp, code objects)
            t(launch(f), 10)
        for rnd in range(a.rounds):
            for n_, f in forms:
                res[n_].append(t(launch(f), a.reps))
            x[0].normal_()
        pl = (ctypes.c_longlong * 5)()
        lib.ic_wino3x3_c128_plan(N, H, W, 0, pl)
        out = {'bench': 'conv3', 'shape': name, 'N': N, 'H': H, 'W': W, 'tile_groups': groups, 'auto_plan': list(pl),
               'us_median': {}, 'us_min': {}, 'executed_frac': {}}
        for n_, _ in forms:
            v = sorted(res[n_])
            med = v[len(v) // 2]
            out['us_median'][n_] = round(med, 2)
            out['us_min'][n_] = round(v[0], 2)
            ex = flop * (1.0 if n_ == 'direct' else (0.25 if n_.startswith('w4') else 16.0 / 36.0))
            out['executed_frac'][n_] = round(ex / med / 1e6 / 157.3, 3)
        emit(out, a.out)


def cmd_pc(a, dev):
    from imgcomp_cvpr_amd import probclass, config_parser as cp, weights as W
    t = Timer(dev)
    for ae_name, shp in (('low', (1, 32, 64, 96)), ('low', (1, 32, 32, 32)), ('low', (8, 32, 32, 32)), ('hi', (1, 64, 270, 480))):
        ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', ae_name))
        pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
        wts = W.synthetic_weights(ae_cfg, pc_cfg)
        pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
        centers = torch.as_tensor(wts['autoencoder/encoder/centers']).to(dev)
        sym = torch.randint(0, 6, shp, device=dev)
        q = centers[sym].contiguous()
        fn = lambda: pc.bitcost(q, sym, False, pad_value=float(centers[0]))
        t(fn, 5)
        us = sorted(t(fn, 20) for _ in range(5))[2]
        n = sym.numel()
        emit({'bench': 'pc', 'shape': list(shp), 'us': round(us, 2), 'live_tap_frac_of_mfma_peak': round(36912.0 * n / us / 1e6 / 157.3, 4),
              'dense_frac': round(47520.0 * n / us / 1e6 / 157.3, 4)}, a.out)


def cmd_edge(a, dev):
    from imgcomp_cvpr_amd import autoencoder, config_parser as cp, weights as W
    t = Timer(dev)
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
    N, H, Wd = 1, 512, 768
    one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    p = lambda s: ae._plan[s]
    x = torch.rand((N, 3, H, Wd), device=dev) * 255
    half = torch.randn((N, 64, H // 2, Wd // 2), device=dev)
    quar = torch.randn((N, 128, H // 4, Wd // 4), device=dev)
    bott = torch.randn((N, 33, H // 8, Wd // 8), device=dev)
    q = torch.randn((N, 32, H // 8, Wd // 8), device=dev)
    xo = torch.empty_like(x)
    E, D = W.ENC, W.DEC
    cases = {
        'h1': (lambda: lib.ic_conv2d_bn_act_f32(_lib.ptr(x), _lib.ptr(p(E + '/h1')[0]), _lib.ptr(p(E + '/h1')[1]), _lib.ptr(p(E + '/h1')[2]), None, None,
                                                _lib.ptr(half), N, 3, H, Wd, 64, 5, 5, 2, 1, None, None, t.st), 2.0 * 75 * 64 * N * H * Wd / 4),
        'h2': (lambda: lib.ic_conv2d_mfma_bn_act_f32(_lib.ptr(half), _lib.ptr(p(E + '/h2')[0]), _lib.ptr(p(E + '/h2')[1]), _lib.ptr(p(E + '/h2')[2]),
                                                     _lib.ptr(quar), N, 64, H // 2, Wd // 2, 128, 5, 5, 2, 0, 1, t.st), 2.0 * 25 * 64 * 128 * N * H * Wd / 16),
        'to_bn': (lambda: lib.ic_conv2d_mfma_bn_act_f32(_lib.ptr(quar), _lib.ptr(p(E + '/to_bn')[0]), _lib.ptr(p(E + '/to_bn')[1]), _lib.ptr(p(E + '/to_bn')[2]),
                                                        _lib.ptr(bott), N, 128, H // 4, Wd // 4, 33, 5, 5, 2, 0, 0, t.st), 2.0 * 25 * 128 * 33 * N * H * Wd / 64),
        'from_bn': (lambda: lib.ic_deconv2d_bn_act_f32(_lib.ptr(q), _lib.ptr(p(D + '/from_bn')[0]), _lib.ptr(p(D + '/from_bn')[1]), _lib.ptr(p(D + '/from_bn')[2]),
                                                       _lib.ptr(quar), N, 32, H // 8, Wd // 8, 128, 3, 3, 1, None, None, 0, t.st), 2.0 * 9 * 32 * 128 * N * H * Wd / 64),
        'h12': (lambda: lib.ic_conv2d_mfma_bn_act_f32(_lib.ptr(quar), _lib.ptr(p(D + '/h12')[0]), _lib.ptr(p(D + '/h12')[1]), _lib.ptr(p(D + '/h12')[2]),
                                                      _lib.ptr(half), N, 128, H // 4, Wd // 4, 64, 5, 5, 2, 1, 1, t.st), 2.0 * 25 * 128 * 64 * N * H * Wd / 16),
        'h13': (lambda: lib.ic_deconv2d_bn_act_f32(_lib.ptr(half), _lib.ptr(p(D + '/h13')[0]), _lib.ptr(p(D + '/h13')[1]), _lib.ptr(p(D + '/h13')[2]),
                                                   _lib.ptr(xo), N, 64, H // 2, Wd // 2, 3, 5, 5, 0, None, None, 0, t.st), 2.0 * 25 * 64 * 3 * N * H * Wd / 4),
    }
    for name, (fn, flop) in cases.items():
        rc = fn()
        assert rc == 0, (name, rc)
        t(fn, 20)
        us = sorted(t(fn, 40) for _ in range(5))[2]
        emit({'bench': 'edge', 'layer': name, 'us': round(us, 2), 'tflops': round(flop / us / 1e6, 1), 'frac': round(flop / us / 1e6 / 157.3, 3)}, a.out)


def cmd_train(a, dev):
    import time
    from imgcomp_cvpr_amd import config_parser as cp, weights as W, training
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    tr = training.Trainer(ae_cfg, pc_cfg, wts, dev, num_itr_per_epoch=1000)
    x = torch.as_tensor(W.synthetic_image((32, 3, 128, 128), 'natural', seed=0)).float().to(dev)
    for _ in range(3):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.step(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    emit({'bench': 'train', 'ms_per_step': round(ms, 3), 'img_per_s': round(32e3 / ms, 1), 'last': {k: round(float(v), 5) for k, v in out.items()}}, a.out)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('what', choices=['conv3', 'pc', 'edge', 'train'])
    p.add_argument('--shapes', default='kodak,256,256b8,train,544,kodakb2,4k')
    p.add_argument('--rounds', type=int, default=5)
    p.add_argument('--reps', type=int, default=40)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--out', default=None)
    a = p.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    {'conv3': cmd_conv3, 'pc': cmd_pc, 'edge': cmd_edge, 'train': cmd_train}[a.what](a, dev)


if __name__ == '__main__':
    main()
