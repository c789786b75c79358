"""-m "not gpu": pins the CPU oracle -- against loop-level definitions of the TF ops it restates, against
the fixtures produced by running the reference's own NumPy code (tests/golden/make_golden.py), and through
the invariants the reference asserts at run time."""
import os

import numpy as np
import pytest
import torch

from oracle import naive_np as NP
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('H,W,k,stride', [(8, 10, 5, 2), (6, 6, 3, 1), (7, 9, 5, 2), (5, 4, 3, 2)])
def test_conv2d_same_matches_definition(H, W, k, stride):
    rs = np.random.RandomState(H * 10 + W)
    x = rs.normal(size=(2, 3, H, W))
    w = rs.normal(size=(k, k, 3, 4))
    got = O.conv2d_same(torch.as_tensor(x), w, stride).numpy()
    np.testing.assert_allclose(got, NP.conv2d_same(x, w, stride), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('h,w,k', [(4, 5, 5), (3, 3, 3), (6, 4, 5)])
def test_conv2d_transpose_same_matches_definition(h, w, k):
    rs = np.random.RandomState(h * 10 + w + k)
    x = rs.normal(size=(2, 4, h, w))
    wt = rs.normal(size=(k, k, 3, 4))               # [kh, kw, cout, cin]
    got = O.conv2d_transpose_same(torch.as_tensor(x), wt, 2).numpy()
    assert got.shape == (2, 3, 2 * h, 2 * w)
    np.testing.assert_allclose(got, NP.conv2d_transpose_same(x, wt, 2), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('k', [3, 5])
def test_transposed_conv_is_adjoint_of_same_conv(k):
    """<conv(y), x> == <y, deconv(x)> pins the transposed-conv crop."""
    rs = np.random.RandomState(k)
    y = torch.as_tensor(rs.normal(size=(1, 3, 12, 16)))       # big side
    x = torch.as_tensor(rs.normal(size=(1, 5, 6, 8)))         # small side
    w_fwd = rs.normal(size=(k, k, 3, 5))                      # conv filter [kh,kw,cin=3,cout=5]
    lhs = float((O.conv2d_same(y, w_fwd, 2) * x).sum())
    # the transposed op takes the same array read as [kh,kw,cout=3,cin=5]
    rhs = float((y * O.conv2d_transpose_same(x, w_fwd, 2)).sum())
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))
    # and PyTorch's default (padding=k//2, output_padding=1) is NOT that operator
    wt = torch.as_tensor(w_fwd).permute(3, 2, 0, 1)
    other = torch.nn.functional.conv_transpose2d(x, wt, stride=2, padding=k // 2, output_padding=1)
    assert float((other - O.conv2d_transpose_same(x, w_fwd, 2)).abs().max()) > 1e-3


def test_conv3d_matches_definition():
    rs = np.random.RandomState(3)
    x = rs.normal(size=(1, 2, 4, 6, 7))
    w = rs.normal(size=(2, 3, 3, 2, 3))
    b = rs.normal(size=(3,))
    wd = {'s/weights': w.astype(np.float64), 's/biases': b}
    got = O._conv3d(torch.as_tensor(x), wd, 's', np.ones((2, 3, 3)), relu=False).numpy()
    ref = NP.conv3d_valid(x, w) + b.reshape(1, 3, 1, 1, 1)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_masks_and_pad_match_reference_fixture():
    g = np.load(os.path.join(GOLD, 'probclass_np.npz'))
    first, other = O.pc_masks(3)
    np.testing.assert_array_equal(first, g['first_mask'][..., 0, 0])
    np.testing.assert_array_equal(other, g['other_mask'][..., 0, 0])
    assert int(first.sum()) == 13 and int(other.sum()) == 14
    vol = torch.as_tensor(g['vol'][None].astype(np.float32))
    np.testing.assert_array_equal(O.pad_for_probclass3d(vol, 9, 7.5).numpy(), g['vol_padded_v7'])
    np.testing.assert_array_equal(O.pad_for_probclass3d(vol, 9, 0).numpy()[0], g['vol_padded'].astype(np.float32))


def test_quantizer_definition():
    rs = np.random.RandomState(0)
    z = rs.normal(0, 1.5, (1, 4, 5, 6)).astype(np.float32)
    c = np.linspace(-2, 2, 6).astype(np.float32)
    qs, qh, sym = O.quantize(torch.as_tensor(z), c, 1.0)
    d = (z[..., None] - c) ** 2
    np.testing.assert_array_equal(sym.numpy(), np.argmin(d, -1))
    np.testing.assert_array_equal(qh.numpy(), c[np.argmin(d, -1)])
    p = np.exp(-d.astype(np.float64)); p /= p.sum(-1, keepdims=True)
    np.testing.assert_allclose(qs.numpy(), (p * c).sum(-1), rtol=1e-5, atol=1e-6)
    # masked-out positions (z == 0) quantise to the centre nearest 0, ties to the lower index
    _, qh0, s0 = O.quantize(torch.zeros(1, 1, 1, 1), c, 1.0)
    assert int(s0) == 2 and float(qh0) == pytest.approx(-0.4)


def test_pc_logits_causal_and_shape(syn_weights):
    """logits at raster index <= i do not depend on symbol i (probclass.py:150-176)."""
    rs = np.random.RandomState(4)
    c = syn_weights['autoencoder/encoder/centers']
    C, h, w = 6, 5, 7
    sym = rs.randint(0, 6, (1, C, h, w))
    q = torch.as_tensor(c[sym]).double()
    _, base = O.bitcost(q, torch.as_tensor(sym), syn_weights, float(c[0]))
    assert base.shape == (1, C, h, w, 6)
    assert float(base.min()) >= 0.0                       # final conv3d keeps its ReLU
    flat = base.reshape(-1, 6)
    for idx in (0, 37, C * h * w - 2):
        s2 = sym.copy().reshape(-1); s2[idx] = (s2[idx] + 2) % 6
        _, out = O.bitcost(torch.as_tensor(c[s2.reshape(sym.shape)]).double(), torch.as_tensor(sym), syn_weights, float(c[0]))
        out = out.reshape(-1, 6)
        assert torch.equal(out[:idx + 1], flat[:idx + 1])
        assert not torch.equal(out[idx + 1:], flat[idx + 1:])


def test_blockwise_logits_equal_full_volume(syn_weights):
    """a (5,9,9) context evaluated on its own gives the logits of the full-volume evaluation
    (what PredictionNetwork relies on, probclass.py:425-444) -- val.py:174's invariant in the oracle."""
    rs = np.random.RandomState(5)
    c = syn_weights['autoencoder/encoder/centers']
    sym = rs.randint(0, 6, (1, 4, 5, 6))
    q = torch.as_tensor(c[sym]).double()
    _, full = O.bitcost(q, torch.as_tensor(sym), syn_weights, float(c[0]))
    qp = O.pad_for_probclass3d(q, 9, float(c[0]))
    for (ci, y, x) in ((0, 0, 0), (3, 4, 5), (2, 1, 3)):
        block = qp[:, ci:ci + 5, y:y + 9, x:x + 9]
        one = O.pc_logits(block, syn_weights)
        assert one.shape == (1, 1, 1, 1, 6)
        assert float((one[0, 0, 0, 0] - full[0, ci, y, x]).abs().max()) < 1e-12


def test_val_wiring_shapes_and_bpp(configs, syn_weights):
    from imgcomp_cvpr_amd import weights as W
    ae_cfg, _ = configs
    x = W.synthetic_image((1, 3, 32, 48), 'natural', 0)
    r = O.validate_forward(x, syn_weights, ae_cfg.as_dict(), torch.float32)
    enc = r['enc']
    assert enc.symbols.shape == (1, 32, 4, 6) and enc.symbols.dtype == torch.int64
    assert r['x_out'].shape == (1, 3, 32, 48) and r['x_out_uint8'].dtype == torch.uint8
    assert float(r['x_out'].min()) >= 0 and float(r['x_out'].max()) <= 255
    assert r['bpp'] == pytest.approx(float(r['bits'].sum()) / (32 * 48))
    # uint8 conversion truncates (val.py:91)
    assert torch.equal(r['x_out_uint8'], torch.floor(r['x_out']).to(torch.uint8))


def test_fp32_oracle_close_to_fp64_shadow(configs, syn_weights):
    from imgcomp_cvpr_amd import weights as W
    ae_cfg, _ = configs
    x = W.synthetic_image((1, 3, 32, 32), 'natural', 2)
    a = O.encode(torch.as_tensor(x).float(), syn_weights, ae_cfg.as_dict())
    b = O.encode(torch.as_tensor(x).double(), syn_weights, ae_cfg.as_dict())
    assert float((a.z.double() - b.z).abs().max()) < 1e-4 * max(1.0, float(b.z.abs().max()))


def test_oracle_matches_frozen_end_to_end_fixture():
    """regression pin: today's oracle reproduces the end-to-end numbers frozen in tests/golden/oracle_e2e.npz
    (make_oracle_e2e.py) -- a change of the oracle cannot silently move the target of the GPU parity tests."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_oracle_e2e', os.path.join(GOLD, 'make_oracle_e2e.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.compute()
    g = np.load(os.path.join(GOLD, 'oracle_e2e.npz'))
    assert np.array_equal(got['symbols'], g['symbols'])
    for k in ('z', 'heatmap', 'bitcost', 'x_out', 'bpp'):
        assert np.max(np.abs(got[k] - g[k])) <= 1e-9 * max(1.0, float(np.abs(g[k]).max())), k
