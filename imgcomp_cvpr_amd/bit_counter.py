"""Real bit count of a symbol volume: arithmetic-code it to a file, decode it back, compare
(mirror of the reference's code/bit_counter.py; BASELINE config 4).

Differences from the reference, all on the speed side, none in the stream:
  * encode: the frequency tables of ALL contexts come from ONE parallel pass of the context model
    (PredictionNetwork.get_all) instead of two sess.run per symbol (bit_counter.py:125-126);
  * decode stays sequential by nature (a table depends on the symbols decoded so far); by default the whole loop
    runs on the device without host round trips (PredictionNetwork.decode_stream -> ic_pc_decode_f32); with
    device_decode=False it asks the prediction network one context at a time from Python, exactly like
    bit_counter.py:137-164 (both are tested to return the same symbols).
Raster order C -> H -> W, the first symbol is not coded (only its -log2 p enters the theoretical cost),
and the three run-time checks of the reference are kept (bit_counter.py:51,56,68).
"""
import itertools
import os
import tempfile

import numpy as np

from . import arithmetic_coding as ac
from . import probclass


def encode_decode_to_file_ctx(syms, prediction_net, syms_format='HWC', verbose=False, device_decode=True):
    """:return: number of bits needed to encode all symbols in `syms` (HWC / CHW, or a batch of them)."""
    _print = print if verbose else (lambda *a, **k: None)
    if len(syms.shape) == 4:
        return int(np.sum([encode_decode_to_file_ctx(syms[b, ...], prediction_net, syms_format, verbose, device_decode)
                           for b in range(syms.shape[0])]))
    assert len(syms.shape) == 3, 'Expected HWC or CHW'
    assert syms_format in ('HWC', 'CHW')
    if syms_format == 'HWC':
        syms = np.transpose(syms, (2, 0, 1))
    syms = np.asarray(syms)
    ctx_shape = prediction_net.input_ctx_shape

    fd, fout_p = tempfile.mkstemp()
    try:
        _print('Encoding symbols of shape {} ({} symbols) with context shape {}...'.format(
            syms.shape, int(np.prod(syms.shape)), ctx_shape))
        syms_padded = prediction_net.pad_symbols_volume(syms)
        virtual_num_bits, first_sym, theoretical_bit_cost = _encode(fd, syms_padded, syms, prediction_net)
        assert abs(virtual_num_bits - theoretical_bit_cost) < 50, 'Virtual: {} -- Theoretical: {}'.format(
            virtual_num_bits, theoretical_bit_cost)
        actual_num_bits = os.path.getsize(fout_p) * 8
        assert actual_num_bits == virtual_num_bits, '{} != {}'.format(actual_num_bits, virtual_num_bits)

        _print('Decoding symbols to shape {}, first_sym={}...'.format(syms_padded.shape, first_sym))
        if device_decode and hasattr(prediction_net, 'decode_stream'):
            with open(fout_p, 'rb') as fin:
                syms_dec = prediction_net.decode_stream(fin.read(), syms.shape, first_sym)
        else:
            syms_dec_padded = _decode(fout_p, syms_padded.shape, ctx_shape, first_sym, prediction_net.get_freqs)
            syms_dec = prediction_net.undo_pad_symbols_volume(syms_dec_padded)
        np.testing.assert_array_equal(syms, syms_dec)
        _print('Decoded symbols match input!')
    finally:
        os.remove(fout_p)
    return actual_num_bits


def _new_sym_idxs_itr(syms_shape, ctx_size):
    """indices (d, h, w) of the coded symbols inside the padded volume, raster order (bit_counter.py:94-100)."""
    D, H, W = syms_shape
    pad = ctx_size // 2
    return itertools.product(range(pad, D), range(pad, H - pad), range(pad, W - pad))


def _encode(fd, syms_padded, syms, prediction_net):
    """all tables at once, then the host-side coder over symbols 1..n-1 in raster order."""
    pr, freqs = prediction_net.get_all(syms_padded)                  # (n, L) each, raster C,H,W
    flat = syms.reshape(-1).astype(np.int64)
    assert pr.shape[0] == flat.shape[0], (pr.shape, flat.shape)
    first_sym = int(flat[0])
    cost = -np.log2(pr[np.arange(flat.shape[0]), flat].astype(np.float64))
    theoretical_bit_cost = float(cost.sum())
    with open(fd, 'wb') as fout:
        num_bits = ac.encode_sequence(flat[1:], freqs[1:], fout)
    return num_bits, first_sym, theoretical_bit_cost


def _decode(fout_p, symbols_shape_padded, ctx_shape, first_sym, get_freqs):
    with open(fout_p, 'rb') as fin:
        dec = ac.ArithmeticDecoder(ac.BitInputStream(fin))
        symbols_decoded = np.zeros(symbols_shape_padded, dtype=np.int32)
        ctx_size = probclass.context_size_from_context_shape(ctx_shape)
        idxs = _new_sym_idxs_itr(symbols_shape_padded, ctx_size)
        blocks = probclass._iter_block_idices(symbols_shape_padded, ctx_shape)
        next(blocks)                                                # the first context is not coded
        symbols_decoded[next(idxs)] = first_sym
        for (cs, hs, ws), idx in zip(blocks, idxs):
            freqs = ac.SimpleFrequencyTable(get_freqs(symbols_decoded[cs, hs, ws]))
            symbols_decoded[idx] = dec.read(freqs)
        return symbols_decoded
