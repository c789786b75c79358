#!/usr/bin/env python
"""Real-bpp (BASELINE config 4) timing: parallel tables + host arithmetic encoder, then the sequential decode on the
device (ic_pc_decode_f32) and, for small volumes, through the reference-style host loop."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from imgcomp_cvpr_amd import autoencoder, probclass, config_parser as cp, weights as W, bit_counter


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--height', type=int, default=512)
    p.add_argument('--width', type=int, default=768)
    p.add_argument('--host_loop', action='store_true', help='also time the one-context-per-round-trip decoder')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
    pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
    x = torch.as_tensor(W.synthetic_image((1, 3, a.height, a.width), 'natural', seed=0)).float().to(dev)
    sym = ae.encode(x, False).symbols[0].cpu().numpy()
    pred = probclass.PredictionNetwork(pc, pc_cfg, ae.get_centers_variable())
    padded = pred.pad_symbols_volume(sym)
    fd, path = tempfile.mkstemp()
    t = time.time(); nbits, first, theory = bit_counter._encode(fd, padded, sym, pred); t_enc = time.time() - t
    data = open(path, 'rb').read()
    pred.decode_stream(data[:64], (1, 2, 2), first)                # warm-up
    t = time.time(); out = pred.decode_stream(data, sym.shape, first); t_dec = time.time() - t
    ok = bool(np.array_equal(out, sym))
    from imgcomp_cvpr_amd import _lib
    prev = _lib.lib.ic_pc_decode_set_mode(1)
    t = time.time(); out2 = pred.decode_stream(data, sym.shape, first); t_dec2 = time.time() - t
    _lib.lib.ic_pc_decode_set_mode(prev)
    res = {'symbols': int(sym.size), 'bits': int(nbits), 'bpp_real': nbits / (a.height * a.width), 'bits_theory': theory,
           'encode_s': round(t_enc, 3), 'device_decode_s': round(t_dec, 3), 'device_decode_us_per_symbol': round(t_dec / sym.size * 1e6, 2),
           'round_trip_ok': ok,
           'launch_per_layer_decode_s': round(t_dec2, 3), 'launch_per_layer_ok': bool(np.array_equal(out2, sym))}
    if a.host_loop:
        t = time.time()
        ref = pred.undo_pad_symbols_volume(bit_counter._decode(path, padded.shape, pred.input_ctx_shape, first, pred.get_freqs))
        res['host_loop_decode_s'] = round(time.time() - t, 3)
        res['host_loop_ok'] = bool(np.array_equal(ref, sym))
    os.remove(path)
    print(res)


if __name__ == '__main__':
    main()
