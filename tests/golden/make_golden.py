#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by RUNNING the reference's own Python code
(/root/reference/code) in this container.  The reference cannot travel to the GPU box, so only the
data it produced is committed (inputs + expected outputs); this script is the recipe.

What can be run: the NumPy / pure-Python parts of the reference (tensorflow and fjcommon are stubbed
with MagicMock -- they are not installed and the stubbed symbols are never reached by the calls below):
  * ms_ssim_np.MultiScaleSSIM                      (code/ms_ssim_np.py:51-110)   -> msssim.npz
  * arithmetic_coding encoder/decoder + freq tables (code/arithmetic_coding.py)  -> arithcoding.npz
  * probclass NumPy helpers: masks, pad_for_probclass3d, undo_pad, iter_over_blocks, num_blocks,
    context shape arithmetic                       (code/probclass.py:18-57,145-176,268-292,341-387)
                                                                                 -> probclass_np.npz
What cannot: anything that executes a TF op (conv, BN, softmax, conv3d) -- see oracle/oracle.py header.

usage: python tests/golden/make_golden.py        (rewrites the .npz files next to this script)
"""
import io
import os
import sys
from unittest import mock

import numpy as np

REF = '/root/reference/code'
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    for m in ('tensorflow', 'tensorflow.contrib', 'tensorflow.contrib.slim', 'fjcommon'):
        sys.modules[m] = mock.MagicMock()
    sys.path.insert(0, REF)
    import ms_ssim_np           # noqa: E402
    import arithmetic_coding    # noqa: E402
    import probclass            # noqa: E402
    return ms_ssim_np, arithmetic_coding, probclass


def make_msssim(ms_ssim_np):
    """inputs are regenerated from seeds by tests/util.py:msssim_case (legacy RandomState streams are
    frozen by numpy); the fixture keeps their CRC32 and the reference's outputs."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.util import msssim_case, MSSSIM_CASES
    import zlib
    cases = {}
    for name in MSSSIM_CASES:
        img1, img2 = msssim_case(name)
        cases['crc_' + name] = np.int64(zlib.crc32(img1.tobytes() + img2.tobytes()))
        cases['msssim_' + name] = np.float64(ms_ssim_np.MultiScaleSSIM(img1, img2, max_val=255))
        ssim, cs = ms_ssim_np._SSIMForMultiScale(img1, img2, max_val=255)
        cases['ssim_' + name] = np.float64(ssim)
        cases['cs_' + name] = np.float64(cs)
    np.savez_compressed(os.path.join(HERE, 'msssim.npz'), **cases)
    print('msssim:', {k: float(v) for k, v in cases.items() if k.startswith('msssim')})


def make_arithcoding(ac):
    rs = np.random.RandomState(1)
    n, L = 2000, 6
    syms = rs.randint(0, L, n)
    # per-symbol frequency tables as the reference builds them: int(p * 1e9), floored at 1
    logits = rs.normal(0, 1.5, (n, L))
    p = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    freqs = np.maximum((p * 1e9).astype(np.int64), 1)
    buf = io.BytesIO()
    bitout = ac.BitOutputStream(buf)
    enc = ac.ArithmeticEncoder(bitout)
    for s, f in zip(syms, freqs):
        enc.write(ac.SimpleFrequencyTable(f.tolist()), int(s))
    enc.finish()
    bitout.close() if hasattr(bitout, 'close') else None
    data = np.frombuffer(buf.getvalue() if not buf.closed else b'', dtype=np.uint8)
    if data.size == 0:      # BitOutputStream.close() closed the BytesIO: redo with a non-closing wrapper
        class _Keep(io.BytesIO):
            def close(self):
                pass
        buf = _Keep()
        bitout = ac.BitOutputStream(buf)
        enc = ac.ArithmeticEncoder(bitout)
        for s, f in zip(syms, freqs):
            enc.write(ac.SimpleFrequencyTable(f.tolist()), int(s))
        enc.finish()
        bitout.close()
        data = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    # decode with the reference decoder to prove the stream is self-consistent
    dec = ac.ArithmeticDecoder(ac.BitInputStream(io.BytesIO(data.tobytes())))
    out = [dec.read(ac.SimpleFrequencyTable(f.tolist())) for f in freqs]
    assert out == syms.tolist()
    np.savez_compressed(os.path.join(HERE, 'arithcoding.npz'), symbols=syms.astype(np.int64), freqs=freqs,
                        stream=data)
    print('arithcoding: {} symbols -> {} bytes'.format(n, data.size))


def make_probclass(pc):
    out = {}
    cfg = mock.MagicMock()
    cfg.kernel_size = 3
    with mock.patch.object(pc._Network3D, '_make_tf_conv3d_mask', staticmethod(lambda m: m)):
        net = pc._ResShallow(cfg, num_centers=6)
        out['first_mask'] = np.asarray(net.create_first_mask())
        out['other_mask'] = np.asarray(net.create_other_mask())
    out['context_size'] = np.int64(pc._ResShallow.get_context_size(cfg))
    out['context_shape'] = np.asarray(pc._ResShallow.get_context_shape(cfg), np.int64)
    rs = np.random.RandomState(2)
    vol = rs.randint(0, 6, (3, 4, 5)).astype(np.int64)
    out['vol'] = vol
    out['vol_padded'] = pc.pad_for_probclass3d(vol, 9)                       # pad value 0, CHW branch
    out['vol_padded_v7'] = pc.pad_for_probclass3d(vol[None].astype(np.float32), 9, pad_value=7.5)
    out['vol_unpadded'] = pc.undo_pad_for_probclass3d(out['vol_padded'], 9)
    blocks = list(pc.iter_over_blocks(out['vol_padded'], (5, 9, 9)))
    out['num_blocks'] = np.int64(pc.num_blocks(out['vol_padded'].shape, (5, 9, 9)))
    out['blocks'] = np.stack(blocks)
    np.savez_compressed(os.path.join(HERE, 'probclass_np.npz'), **out)
    print('probclass: context', out['context_shape'], 'blocks', out['blocks'].shape)


if __name__ == '__main__':
    ms_ssim_np, ac, pc = _import_reference()
    make_msssim(ms_ssim_np)
    make_arithcoding(ac)
    make_probclass(pc)
