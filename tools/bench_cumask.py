#!/usr/bin/env python
"""Experiment: decoder and context model on disjoint CU sets (hipExtStreamCreateWithCUMask).

The decoder's Winograd launches run one work-group per CU (512 registers per lane) and, for a Kodak map, only use 192 of the
256 CUs; a context-model work-group that lands on one of "their" CUs makes the next Winograd work-group wait for the whole
SIMD.  Giving each branch its own CUs removes that interference.

    python tools/bench_cumask.py [--dec_cus 192] [--steps 30]
"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask failed: %d' % rc)
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dec_cus', type=int, default=192)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--height', type=int, default=512)
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--layout', default='interleaved', choices=['interleaved', 'blocked'])
    ap.add_argument('--only', default='', help='substring of the variant names to run')
    ap.add_argument('--null_stream', type=int, default=0, help='1: leave the main branch on the legacy default stream')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W
    hip = ctypes.CDLL('libamdhip64.so')
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
    pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
    x = torch.as_tensor(W.synthetic_image((1, 3, a.height, a.width), 'natural', seed=0)).float().to(dev)
    pad_value = float(wts['autoencoder/encoder/centers'][0])
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    nd = a.dec_cus
    if a.layout == 'interleaved':
        dec_bits, pc_bits = list(range(nd)), list(range(nd, ncu))
    else:   # per-XCD blocks of 32: the first nd/8 CUs of every block
        per = nd // 8
        dec_bits = [32 * x8 + i for x8 in range(8) for i in range(per)]
        pc_bits = [32 * x8 + i for x8 in range(8) for i in range(per, 32)]
    s_dec, s_pc = masked_stream(hip, dec_bits), masked_stream(hip, pc_bits)
    s_side = torch.cuda.Stream(device=dev)

    def step_plain():
        cur = torch.cuda.current_stream(dev)
        enc = ae.encode(x, is_training=False)
        s_side.wait_stream(cur)
        with torch.cuda.stream(s_side):
            bc = pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=pad_value)
            bpp = bits.bitcost_to_bpp(bc, x)
        out = ae.decode(enc.qhard, is_training=False)
        cur.wait_stream(s_side)
        return bpp, out

    def step_masked():
        cur = torch.cuda.current_stream(dev)
        enc = ae.encode(x, is_training=False)
        s_dec.wait_stream(cur); s_pc.wait_stream(cur)
        with torch.cuda.stream(s_pc):
            bc = pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=pad_value)
            bpp = bits.bitcost_to_bpp(bc, x)
        with torch.cuda.stream(s_dec):
            out = ae.decode(enc.qhard, is_training=False)
        cur.wait_stream(s_pc); cur.wait_stream(s_dec)
        return bpp, out

    def step_pc_masked():
        cur = torch.cuda.current_stream(dev)
        enc = ae.encode(x, is_training=False)
        s_pc.wait_stream(cur)
        with torch.cuda.stream(s_pc):
            bc = pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=pad_value)
            bpp = bits.bitcost_to_bpp(bc, x)
        out = ae.decode(enc.qhard, is_training=False)
        cur.wait_stream(s_pc)
        return bpp, out

    def step_serial():
        enc = ae.encode(x, is_training=False)
        bc = pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=pad_value)
        bpp = bits.bitcost_to_bpp(bc, x)
        out = ae.decode(enc.qhard, is_training=False)
        return bpp, out

    def only(stream, what):
        def f():
            cur = torch.cuda.current_stream(dev)
            enc = ae.encode(x, is_training=False)
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                r = ae.decode(enc.qhard, is_training=False) if what == 'dec' else pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=pad_value)
            cur.wait_stream(stream)
            return r
        return f

    def timeit(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, r

    if not a.null_stream:
        # CU-masked streams are created blocking: they order themselves against the legacy default stream on every launch
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ref = step_serial()
    torch.cuda.synchronize()
    for name, f in (('serial', step_serial), ('side stream (bench.py)', step_plain), ('disjoint CU sets', step_masked), ('pc on its CUs, decoder unmasked', step_pc_masked),
                    ('encode + decode on default', only(torch.cuda.current_stream(dev), 'dec')),
                    ('encode + decode on %d CUs' % nd, only(s_dec, 'dec')),
                    ('encode + pc on default', only(torch.cuda.current_stream(dev), 'pc')),
                    ('encode + pc on %d CUs' % (ncu - nd), only(s_pc, 'pc'))):
        if a.only and a.only not in name:
            continue
        ms, r = timeit(f)
        extra = ''
        if isinstance(r, tuple):
            extra = '  bpp %.5f  same output %s' % (float(r[0]), bool(torch.equal(r[1], ref[1])))
        print('%-32s %.3f ms%s' % (name, ms, extra))


if __name__ == '__main__':
    main()
