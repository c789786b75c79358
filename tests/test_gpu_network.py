"""-m gpu: the whole hot path through the plugin surface (autoencoder / probclass / bits) against the
oracle, with the val.py wiring (reference code/val.py:81-94)."""
import os
import numpy as np
import pytest
import torch

from tests.util import assert_close, dev, rel_err, record_flips, RTOL, NET_RTOL, HEATMAP_RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nets(cuda, configs, syn_weights):
    from imgcomp_cvpr_amd import autoencoder, probclass
    ae_cfg, pc_cfg = configs
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg)
    pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers)
    ae.load_weights(syn_weights, cuda)
    pc.load_weights(syn_weights, cuda)
    return ae, pc


@pytest.fixture(params=['direct3x3', 'winograd3x3', 'wino_seg3', 'wino_seg2', 'wino4'])
def algo(request, nets):
    """force one form of the 3x3 layers for the whole network (the default picks per launch from the shape): a per-object
    plan flag that every encode / decode call of THIS autoencoder passes down -- the library has no process-wide switch.
    'wino4' also puts h2 / h12 on the F(4x4) kernel (over phases) where the map's width allows."""
    from imgcomp_cvpr_amd import _lib
    ae, _ = nets
    ae.plan_flags = {'direct3x3': _lib.CONV3_DIRECT, 'winograd3x3': _lib.CONV3_WINO | _lib.CONV3_NO_WINO4, 'wino_seg3': _lib.CONV3_WINO_SEG3,
                     'wino_seg2': _lib.CONV3_WINO_SEG2, 'wino4': _lib.CONV3_WINO4 | _lib.CONV5_WINO4}[request.param]
    yield request.param
    ae.plan_flags = 0


def _boundary_margin(z64, centers):
    """distance of every z to the nearest quantiser decision midpoint (float64)."""
    c = np.sort(np.asarray(centers, np.float64))
    mids = (c[1:] + c[:-1]) / 2
    return np.min(np.abs(z64[..., None] - mids), axis=-1)


@pytest.mark.parametrize('shape,kind', [((1, 3, 64, 64), 'natural'), ((2, 3, 40, 72), 'noise')])
def test_encode_matches_oracle(cuda, configs, syn_weights, nets, shape, kind, algo):
    from imgcomp_cvpr_amd import weights as W
    from oracle import oracle as O
    ae, pc = nets
    ae_cfg, _ = configs
    x = W.synthetic_image(shape, kind, seed=1)
    enc = ae.encode(dev(x, cuda), is_training=False)
    torch.cuda.synchronize()
    ref = O.encode(torch.as_tensor(x).double(), syn_weights, ae_cfg.as_dict())
    assert enc.symbols.dtype == torch.int64 and enc.symbols.shape == ref.symbols.shape
    assert_close(enc.heatmap, ref.heatmap, 'heatmap', HEATMAP_RTOL)
    assert_close(enc.z, ref.z, 'z', NET_RTOL)
    # symbols: bit-exact wherever z is not within the fp32 error band of a decision midpoint
    centers = syn_weights['autoencoder/encoder/centers']
    zerr = float((enc.z.double().cpu() - ref.z).abs().max())
    margin = _boundary_margin(ref.z.numpy(), centers)
    flips = (enc.symbols.cpu() != ref.symbols).numpy()
    assert not np.any(flips & (margin > 2 * zerr + 1e-12)), 'symbol flips away from decision boundaries'
    assert record_flips('encode {} {} {}'.format(shape, kind, algo), flips) < 2e-3, 'too many boundary flips: {}'.format(flips.mean())
    same = ~flips
    assert torch.equal(enc.qhard.cpu()[torch.as_tensor(same)], ref.qhard.float()[torch.as_tensor(same)])
    # given the kernel's own z, symbols/qhard are exactly the oracle's
    _, qh, sym = O.quantize(enc.z.cpu(), centers, 1.0)
    assert torch.equal(sym, enc.symbols.cpu()) and torch.equal(qh, enc.qhard.cpu())
    assert torch.equal(enc.qbar.cpu(), (ae._last_qsoft + (enc.qhard - ae._last_qsoft)).cpu())


@pytest.mark.parametrize('shape', [(1, 3, 32, 96), (3, 3, 48, 32), (2, 3, 96, 160)])
def test_f4_everywhere_small_and_narrow_maps(cuda, configs, syn_weights, nets, shape):
    """F(4x4) forced for the 3x3 layers AND for h2 / h12 (phase form) on maps that take the 2 x 8-tile segments (8 x 24, 12 x 8) and
    on one that takes 1 x 16 (24 x 40): encoder and decoder against the oracle."""
    from imgcomp_cvpr_amd import weights as W, _lib
    from oracle import oracle as O
    ae, _ = nets
    ae_cfg, _ = configs
    flags = _lib.CONV3_WINO4 | _lib.CONV5_WINO4
    x = W.synthetic_image(shape, 'natural', seed=9)
    enc = ae.encode(dev(x, cuda), is_training=False, plan_flags=flags)
    torch.cuda.synchronize()
    ref = O.encode(torch.as_tensor(x).double(), syn_weights, ae_cfg.as_dict())
    assert_close(enc.z, ref.z, 'F(4x4) everywhere, z {}'.format(shape), NET_RTOL)
    assert_close(enc.heatmap, ref.heatmap, 'F(4x4) everywhere, heatmap {}'.format(shape), HEATMAP_RTOL)
    xo = ae.decode(dev(ref.qhard.float().numpy(), cuda), is_training=False, plan_flags=flags)
    assert_close(xo, O.decode(ref.qhard, syn_weights, ae_cfg.as_dict()), 'F(4x4) everywhere, x_out {}'.format(shape), NET_RTOL)
    assert not torch.equal(enc.z, ae.encode(dev(x, cuda), is_training=False, plan_flags=_lib.CONV5_NO_WINO4 | _lib.CONV3_NO_WINO4).z)


@pytest.mark.parametrize('heatmap', [True, False])
def test_encode_without_quantizer(cuda, configs, syn_weights, heatmap):
    """_Network(config, quantize=False) (autoencoder.py:34-36, :127-129): the encoder output is the (masked) bottleneck itself --
    qbar = z, no qhard, no symbols -- and z is the quantising network's z bit for bit."""
    from imgcomp_cvpr_amd import autoencoder, config_parser as cp, weights as W
    from oracle import oracle as O
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))       # (a private copy: the fixture's is shared)
    ae_cfg.heatmap = heatmap
    wts = W.synthetic_weights(ae_cfg, configs[1])
    x = W.synthetic_image((2, 3, 48, 64), 'natural', seed=5)
    cls = autoencoder.get_network_cls(ae_cfg)
    plain = cls(ae_cfg, quantize=False).load_weights(wts, cuda).encode(dev(x, cuda), is_training=False)
    quant = cls(ae_cfg).load_weights(wts, cuda).encode(dev(x, cuda), is_training=False)
    torch.cuda.synchronize()
    assert plain.qhard is None and plain.symbols is None and plain.qbar is plain.z
    assert torch.equal(plain.z, quant.z) and (plain.heatmap is None) == (not heatmap)
    ref = O.encode(torch.as_tensor(x).double(), wts, ae_cfg.as_dict())
    assert_close(plain.qbar, ref.z, 'encoder output without quantiser (heatmap {})'.format(heatmap), NET_RTOL)


@pytest.mark.parametrize('shape', [(1, 32, 8, 8), (2, 32, 5, 9)])
def test_decode_matches_oracle(cuda, configs, syn_weights, nets, shape, algo):
    from oracle import oracle as O
    ae, _ = nets
    ae_cfg, _ = configs
    centers = syn_weights['autoencoder/encoder/centers']
    sym = np.random.RandomState(5).randint(0, 6, shape)
    q = centers[sym]
    x_out = ae.decode(dev(q, cuda), is_training=False)
    torch.cuda.synchronize()
    ref = O.decode(torch.as_tensor(q).double(), syn_weights, ae_cfg.as_dict())
    assert float(x_out.min()) >= 0 and float(x_out.max()) <= 255
    assert_close(x_out, ref, 'decoded RGB', NET_RTOL)
    # uint8 truncation (val.py:91) can differ only where the float64 value sits on an integer boundary
    u_hip = x_out.to(torch.uint8).cpu()
    u_ref = ref.to(torch.uint8)
    frac = (ref - torch.floor(ref))
    near_int = (torch.minimum(frac, 1 - frac) < 255 * RTOL)
    assert not bool(((u_hip != u_ref) & ~near_int).any())


def test_val_wiring(cuda, configs, syn_weights, nets, algo):
    """encode -> decode(qhard) ; bitcost(qbar, symbols, pad=centers[0]) -> bpp (val.py:85-89)."""
    from imgcomp_cvpr_amd import bits, weights as W
    from oracle import oracle as O
    ae, pc = nets
    ae_cfg, _ = configs
    x = W.synthetic_image((1, 3, 64, 96), 'natural', seed=3)
    xd = dev(x, cuda)
    enc = ae.encode(xd, False)
    x_out = ae.decode(enc.qhard, False)
    bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pc.auto_pad_value(ae))
    bpp = float(bits.bitcost_to_bpp(bc, xd))
    # oracle, fed with the device's symbols so that the comparison is not chaotic in the (rare) flips
    sym = enc.symbols.cpu()
    centers = syn_weights['autoencoder/encoder/centers']
    q = torch.as_tensor(centers)[sym].double()
    assert torch.equal(q.float(), enc.qhard.cpu())
    rb, _ = O.bitcost(q, sym, syn_weights, float(centers[0]))
    assert_close(bc, rb, 'bit cost', NET_RTOL)
    assert abs(bpp - O.bitcost_to_bpp(rb, torch.as_tensor(x))) < 1e-4
    ref_out = O.decode(q, syn_weights, ae_cfg.as_dict())
    assert_close(x_out, ref_out, 'x_out', NET_RTOL)
    # val.py:174 invariant: bitcost from symbols alone == bitcost from the encoder's qbar
    # (qbar = qsoft + (qhard - qsoft) equals qhard only up to fp32 rounding, so: close, not identical)
    bc2 = pc.bitcost(dev(q.float().numpy(), cuda), enc.symbols, False, pad_value=float(centers[0]))
    assert rel_err(bc2, bc) < 1e-5
    assert abs(float(bits.bitcost_to_bpp(bc2, xd)) - bpp) < 1e-3


def test_full_size_properties(cuda, configs, syn_weights, nets, algo):
    """BASELINE configs[1] shape (Kodak 512x768, batch 1): size-independent properties --
    determinism, batch-independence (image n of a batch == the image alone), tile-variant independence."""
    from imgcomp_cvpr_amd import weights as W, _lib
    ae, pc = nets
    x = dev(W.synthetic_image((1, 3, 512, 768), 'natural', seed=0), cuda)
    e1 = ae.encode(x, False)
    z1, s1 = e1.z.clone(), e1.symbols.clone()
    e2 = ae.encode(x, False)
    assert torch.equal(z1, e2.z) and torch.equal(s1, e2.symbols), 'encode is not deterministic'
    if algo == 'direct3x3':
        e3 = ae.encode(x, False, plan_flags=_lib.conv3_direct_variant(0))
        assert torch.equal(z1, e3.z), 'result depends on the MFMA tile variant'
    elif algo == 'winograd3x3':
        # every Winograd decomposition performs the same operations per output: bit-identical
        forms = [_lib.CONV3_WINO_WHOLEK, _lib.CONV3_WINO_SEG1, _lib.CONV3_WINO_SEG2, _lib.CONV3_WINO_SEG3,
                 _lib.CONV3_WINO_SEG3 | _lib.CONV3_PACKED_TRANSFORM]
        if _lib.lib.ic_build_has_tuning_forms():                      # make TUNING=1: the forms the plan never picks
            forms += [_lib.CONV3_WINO_T16, _lib.CONV3_WINO_PAIR]
        for form in forms:
            ae.plan_flags = form
            e3 = ae.encode(x, False)
            assert torch.equal(z1, e3.z), 'Winograd form {:#x} differs'.format(form)
        ae.plan_flags = _lib.CONV3_WINO | _lib.CONV3_NO_WINO4
    xo = ae.decode(e1.qhard, False)
    assert xo.shape == x.shape and float(xo.min()) >= 0 and float(xo.max()) <= 255
    bc = pc.bitcost(e1.qbar, e1.symbols, False, pad_value=pc.auto_pad_value(ae))
    assert bool(torch.isfinite(bc).all()) and float(bc.min()) >= 0
    xb = torch.cat([x[:, :, :256, :384], x[:, :, 256:, 384:]], 0).contiguous()
    eb = ae.encode(xb, False)
    ea = ae.encode(xb[1:].contiguous(), False)
    assert torch.equal(eb.z[1:], ea.z), 'batched encode differs from single-image encode'


def test_real_bpp_round_trip(cuda, configs, syn_weights, nets):
    """BASELINE config 4 in the small: parallel logits -> frequency tables -> arithmetic coder -> file ->
    sequential decode; the run-time checks of bit_counter.py:51,56,68 and val.py:174 must hold."""
    from imgcomp_cvpr_amd import probclass, bpp_helpers, bit_counter
    ae, pc = nets
    _, pc_cfg = configs
    rs = np.random.RandomState(12)
    sym = rs.randint(0, 6, (1, 5, 4, 6)).astype(np.int64)          # 120 symbols
    pred = probclass.PredictionNetwork(pc, pc_cfg, ae.get_centers_variable())
    checker = probclass.ProbclassNetworkTesting(pc, ae)
    # the parallel tables equal the per-context tables bit for bit (what makes the decoder stay in sync)
    padded = pred.pad_symbols_volume(sym[0])
    pr_all, f_all = pred.get_all(padded)
    assert f_all.shape == (120, 6) and f_all.min() >= 1 and f_all.sum(1).max() <= 2 ** 30 + 2
    for i, blk in enumerate(probclass.iter_over_blocks(padded, pred.input_ctx_shape)):
        if i % 17 == 0:
            assert np.array_equal(pred.get_freqs(blk), f_all[i])
            assert np.array_equal(pred.get_pr(blk), pr_all[i])
    bpp_real, bpp_theory = bpp_helpers.BppFetcher(pred, checker).get_bpp(sym, num_pixels=32 * 48)
    bits_theory = checker.get_total_bit_cost(sym)
    assert abs(bpp_theory * 32 * 48 - bits_theory) < 1e-6
    assert abs(bpp_real * 32 * 48 - bits_theory) < 50 + 8          # coder overhead bound + byte padding
    # HWC input format and a batch give the same count
    n1 = bit_counter.encode_decode_to_file_ctx(sym[0], pred, syms_format='CHW')
    n2 = bit_counter.encode_decode_to_file_ctx(np.transpose(sym[0], (1, 2, 0)), pred, syms_format='HWC')
    assert n1 == n2 == int(round(bpp_real * 32 * 48))
    # the reference's host loop (one context per round trip) and the on-device decoder agree
    n3 = bit_counter.encode_decode_to_file_ctx(sym[0], pred, syms_format='CHW', device_decode=False)
    assert n3 == n1


def test_device_decoder_stream(cuda, configs, syn_weights, nets, tmp_path):
    """row N3: ic_pc_decode_f32 decodes what the host-side encoder wrote -- symbols of a real encoder output (skewed
    statistics, long runs), a truncated stream decodes without faulting (zeros past the end, as the reference's
    bit reader), a single-symbol volume needs no bits."""
    from imgcomp_cvpr_amd import probclass, bit_counter, weights as W
    import tempfile
    ae, pc = nets
    _, pc_cfg = configs
    pred = probclass.PredictionNetwork(pc, pc_cfg, ae.get_centers_variable())
    x = dev(W.synthetic_image((1, 3, 64, 96), 'natural', seed=3), cuda)
    sym = ae.encode(x, False).symbols[0].cpu().numpy()                      # (32, 8, 12): 3072 symbols
    padded = pred.pad_symbols_volume(sym)
    fd, path = tempfile.mkstemp(dir=str(tmp_path))
    nbits, first, _ = bit_counter._encode(fd, padded, sym, pred)
    data = open(path, 'rb').read()
    assert len(data) * 8 == nbits
    out = pred.decode_stream(data, sym.shape, first)                        # k = 24: one persistent work-group, activation caches
    assert out.dtype == np.int64 and np.array_equal(out, sym)
    from imgcomp_cvpr_amd import _lib
    out1 = pred.decode_stream(data, sym.shape, first, flags=_lib.PC_DECODE_RECOMPUTE)      # the same without caches (round-1 kernel)
    assert np.array_equal(out1, sym)
    out2 = pred.decode_stream(data, sym.shape, first, flags=_lib.PC_DECODE_PER_LAYER)      # the launch-per-layer loop (any k)
    assert np.array_equal(out2, sym)
    ref = pred.undo_pad_symbols_volume(bit_counter._decode(path, padded.shape, pred.input_ctx_shape, first, pred.get_freqs))
    assert np.array_equal(ref, sym)
    cut = pred.decode_stream(data[:len(data) // 2], sym.shape, first)
    assert cut.shape == sym.shape and cut.min() >= 0 and cut.max() < 6 and not np.array_equal(cut, sym)
    n_same = int(np.argmax(cut.reshape(-1) != sym.reshape(-1)))
    assert n_same > sym.size // 4                                        # everything before the cut is still right
    one = pred.decode_stream(b'', (1, 1, 1), 4)
    assert one.tolist() == [[[4]]]


@pytest.mark.parametrize('L,shape', [(6, (1, 3, 40, 56)), (8, (1, 3, 24, 72)), (3, (1, 3, 8, 8))])
def test_device_decoder_cached_shapes(cuda, L, shape, tmp_path):
    """the activation-cache decoder on volumes with odd extents (5 x 7, 3 x 9, 1 x 1 planes: every tap of the first rows and
    columns is a halo voxel) and with other numbers of centres than the specialised L = 6 (generic lane loops): the three
    device decoders and the reference-style host loop return the encoder's symbols."""
    import tempfile
    from imgcomp_cvpr_amd import autoencoder, probclass, bit_counter, _lib, config_parser as cp, weights as W
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    ae_cfg.num_centers = L
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, cuda)
    pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=L).load_weights(wts, cuda)
    pred = probclass.PredictionNetwork(pc, pc_cfg, ae.get_centers_variable())
    x = dev(W.synthetic_image(shape, 'natural', seed=L), cuda)
    sym = ae.encode(x, False).symbols[0].cpu().numpy()
    assert sym.max() < L
    padded = pred.pad_symbols_volume(sym)
    fd, path = tempfile.mkstemp(dir=str(tmp_path))
    nbits, first, _ = bit_counter._encode(fd, padded, sym, pred)
    data = open(path, 'rb').read()
    for flags in (0, _lib.PC_DECODE_RECOMPUTE, _lib.PC_DECODE_PER_LAYER):
        out = pred.decode_stream(data, sym.shape, first, flags=flags)
        assert np.array_equal(out, sym), 'decoder flags {} lost sync (L = {}, volume {})'.format(flags, L, sym.shape)


def test_blockwise_logits_bit_identical_to_full_volume(cuda, configs, syn_weights, nets):
    """the incremental decoder's contract: a (5,9,9) context evaluated alone gives the SAME fp32 logits as the
    all-position pass (fixed K order per output; SURVEY.md section 7 hard parts)."""
    from oracle import oracle as O
    ae, pc = nets
    centers = syn_weights['autoencoder/encoder/centers']
    rs = np.random.RandomState(13)
    sym = rs.randint(0, 6, (1, 7, 6, 9))
    q = dev(centers[sym], cuda)
    full = pc.logits_unpadded(q, float(centers[0]))
    qp = O.pad_for_probclass3d(torch.as_tensor(centers[sym]), 9, float(centers[0])).to(cuda)
    for (c, y, x) in ((0, 0, 0), (6, 5, 8), (3, 2, 4), (1, 5, 0)):
        blk = qp[:, c:c + 5, y:y + 9, x:x + 9].contiguous()
        one = pc.logits(blk, False)
        assert one.shape == (1, 1, 1, 1, 6)
        assert torch.equal(one[0, 0, 0, 0], full[0, c, y, x])


def test_val_entry_point(cuda, tmp_path, configs, syn_weights):
    """python -m imgcomp_cvpr_amd.val LOG_DIR_ROOT JOB_IDS IMAGES on synthetic PNGs of ragged size."""
    from PIL import Image
    from imgcomp_cvpr_amd import val, weights as W, bpp_helpers
    imgs = tmp_path / 'kodakish'
    imgs.mkdir()
    for i, (h, w) in enumerate(((181, 203), (192, 176))):
        im = W.synthetic_image((1, 3, h, w), 'natural', seed=i)[0].transpose(1, 2, 0)
        Image.fromarray(im).save(str(imgs / 'img{:02d}.png'.format(i)))
    root = tmp_path / 'logs'
    (root / '0515_1103 ae_configs@cvpr@low pc_configs@cvpr@res_shallow').mkdir(parents=True)
    val.main([str(root), '0515_1103', str(imgs), '--weights', 'synthetic', '--save_ours'])
    out = root / '0515_1103 kodakish'
    rows = (out / 'measures.csv').read_text().strip().split('\n')
    assert rows[0] == 'img_name,bpp,ms-ssim,psnr' and len(rows) == 3
    for r in rows[1:]:
        name, bpp, msssim, psnr = r.split(',')
        assert 0 < float(bpp) < 10 and 0 < float(msssim) <= 1 and float(psnr) > 0
    saved = np.asarray(Image.open(str(out / 'imgs' / 'img00.png')))
    assert saved.shape == (184, 208, 3)          # padded to the subsampling factor
    # several images in flight (one Fetcher + stream each) give exactly the rows of one image at a time, in order
    for i, (h, w) in enumerate(((64, 96), (120, 72), (88, 88), (64, 96), (56, 200)), start=2):
        im = W.synthetic_image((1, 3, h, w), 'natural', seed=i)[0].transpose(1, 2, 0)
        Image.fromarray(im).save(str(imgs / 'img{:02d}.png'.format(i)))
    rows_by_mode = {}
    for n in (1, 3):
        val.main([str(root), '0515_1103', str(imgs), '--weights', 'synthetic', '--reset', '--in_flight', str(n)])
        rows_by_mode[n] = (out / 'measures.csv').read_text()
    assert rows_by_mode[1] == rows_by_mode[3] and len(rows_by_mode[1].strip().split('\n')) == 8
    # --host_metrics: MS-SSIM / PSNR in numpy on the host as the reference computes them (val.py:96-108); same rows as the
    # device metrics to the printed precision of a float32-valued metric
    val.main([str(root), '0515_1103', str(imgs), '--weights', 'synthetic', '--reset', '--host_metrics'])
    host_rows = (out / 'measures.csv').read_text().strip().split('\n')
    dev_rows = rows_by_mode[1].strip().split('\n')
    assert len(host_rows) == len(dev_rows) == 8
    for hr, dr in zip(host_rows[1:], dev_rows[1:]):
        hn, hb, hm, hp = hr.split(',')
        dn, db, dm, dp = dr.split(',')
        assert hn == dn and hb == db
        assert abs(float(hm) - float(dm)) < 1e-5 and abs(float(hp) - float(dp)) < 1e-3
    # --real_bpp leg on a small image: arithmetic-coded size vs theoretical vs loss (val.py:163-174)
    ae_cfg, pc_cfg = configs
    f = val.Fetcher(ae_cfg, pc_cfg, syn_weights, cuda)
    img = W.synthetic_image((1, 3, 32, 48), 'natural', seed=5)[0]
    x = torch.as_tensor(img)[None].to(cuda).float()
    enc = f.ae.encode(x, False)
    bc = f.pc.bitcost(enc.qbar, enc.symbols, False, pad_value=f.pc.auto_pad_value(f.ae))
    from imgcomp_cvpr_amd import bits
    bpp_loss = float(bits.bitcost_to_bpp(bc, x))
    bpp_real, bpp_theory = f.real_bpp(enc.symbols.cpu().numpy(), bpp_helpers.num_pixels_in_image(img))
    assert abs(bpp_theory - bpp_loss) < 1e-3
    assert abs(bpp_real - bpp_theory) * 32 * 48 < 58


def test_high_rate_config_matches_oracle(cuda):
    """BASELINE config 5 in the small: ae_configs/cvpr/hi (64 bottleneck channels, to_bn emits 65) with
    pc_configs/cvpr/res_shallow_64 (64 feature maps: the k = 64 instantiations of the context-model kernels) through
    the whole path against the oracle."""
    from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W
    from oracle import oracle as O
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'hi'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow_64'))
    assert ae_cfg.num_chan_bn == 64 and pc_cfg.arch_param__k == 64
    wts = W.synthetic_weights(ae_cfg, pc_cfg, seed=77)
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, cuda)
    pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, cuda)
    x = W.synthetic_image((1, 3, 48, 80), 'natural', seed=9)
    xd = dev(x, cuda)
    enc = ae.encode(xd, False)
    assert tuple(enc.symbols.shape) == (1, 64, 6, 10)
    ref = O.encode(torch.as_tensor(x).double(), wts, ae_cfg.as_dict())
    assert_close(enc.z, ref.z, 'z (hi)', NET_RTOL)
    assert_close(enc.heatmap, ref.heatmap, 'heatmap (hi)', HEATMAP_RTOL)
    assert record_flips('hi + res_shallow_64', (enc.symbols.cpu() != ref.symbols).numpy()) < 2e-3
    x_out = ae.decode(enc.qhard, False)
    bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pc.auto_pad_value(ae))
    sym = enc.symbols.cpu()
    centers = wts['autoencoder/encoder/centers']
    q = torch.as_tensor(centers)[sym].double()
    rb, _ = O.bitcost(q, sym, wts, float(centers[0]))
    assert_close(bc, rb, 'bit cost (hi, k = 64)', NET_RTOL)
    assert_close(x_out, O.decode(q, wts, ae_cfg.as_dict()), 'x_out (hi)', NET_RTOL)
    assert abs(float(bits.bitcost_to_bpp(bc, xd)) - O.bitcost_to_bpp(rb, torch.as_tensor(x))) < 1e-4


def test_device_metrics_match_numpy(cuda):
    """val.py's MS-SSIM / PSNR: the float64 device versions against the numpy ones that are pinned to the reference's
    implementation (tests/golden/msssim.npz), sizes with and without the shrunken-window scales."""
    from imgcomp_cvpr_amd import metrics, weights as W
    for shape, seed in (((1, 3, 256, 320), 1), ((1, 3, 181, 177), 2), ((2, 3, 96, 200), 3)):
        a = W.synthetic_image(shape, 'natural', seed)
        b = np.clip(a.astype(np.float64) + np.random.RandomState(seed).normal(0, 7, a.shape), 0, 255).astype(np.uint8)
        ad, bd = torch.as_tensor(a).to(cuda), torch.as_tensor(b).to(cuda)
        assert abs(metrics.msssim_nchw_uint8_device(ad, bd) - float(metrics.msssim_nchw_uint8(a, b))) < 1e-6
        assert abs(metrics.psnr_uint8_device(ad, bd) - float(metrics.psnr_uint8(a, b))) < 1e-4
        # the library's kernels (csrc/val_metrics.hip) against the tap-by-tap torch twin on the CPU: same operations in the same order
        vals, mse = metrics.val_metrics_device(ad, bd)
        twin = metrics.msssim_scale_values_device(torch.as_tensor(a), torch.as_tensor(b))
        assert float((vals.cpu() - twin).abs().max()) < 1e-12, (vals.cpu(), twin)
        assert float(mse) == float(metrics.mse_uint8_device(torch.as_tensor(a), torch.as_tensor(b)))
    assert metrics.psnr_uint8_device(ad, ad) == float('inf') and metrics.msssim_nchw_uint8_device(ad, ad) == 1.0
    # per image of a batch in one call (val.py's batched step): bit-identical to a call per image
    a = W.synthetic_image((5, 3, 128, 136), 'natural', 9)
    b = np.clip(a.astype(np.float64) + np.random.RandomState(4).normal(0, 5, a.shape), 0, 255).astype(np.uint8)
    ad, bd = torch.as_tensor(a).to(cuda), torch.as_tensor(b).to(cuda)
    m6 = metrics.val_metrics_device_per_image(ad, bd)
    assert tuple(m6.shape) == (5, 6)
    for i in range(5):
        vals, mse = metrics.val_metrics_device(ad[i:i + 1], bd[i:i + 1])
        assert torch.equal(m6[i, :5], vals) and torch.equal(m6[i, 5], mse), i
    assert not torch.equal(m6[0], m6[1])


def test_against_frozen_oracle_fixture(cuda, configs, syn_weights, nets):
    """the HIP path against the committed end-to-end fixture (tests/golden/oracle_e2e.npz: float64 oracle outputs frozen
    by make_oracle_e2e.py), independent of the oracle code that is imported at test time."""
    import os
    from imgcomp_cvpr_amd import bits, weights as W
    ae, pc = nets
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'oracle_e2e.npz'))
    x = W.synthetic_image((1, 3, 64, 96), 'natural', seed=3)
    xd = dev(x, cuda)
    enc = ae.encode(xd, False)
    assert_close(enc.z, torch.as_tensor(g['z']), 'z vs fixture', NET_RTOL)
    assert_close(enc.heatmap, torch.as_tensor(g['heatmap']), 'heatmap vs fixture', HEATMAP_RTOL)
    flips = enc.symbols.cpu().numpy() != g['symbols']
    assert record_flips('frozen fixture', flips) < 2e-3
    if not flips.any():                                   # same symbols -> the rest is comparable element-wise
        bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pc.auto_pad_value(ae))
        assert_close(bc, torch.as_tensor(g['bitcost']), 'bit cost vs fixture', NET_RTOL)
        assert abs(float(bits.bitcost_to_bpp(bc, xd)) - float(g['bpp'])) < 1e-4
        assert_close(ae.decode(enc.qhard, False), torch.as_tensor(g['x_out']), 'x_out vs fixture', NET_RTOL)


def test_branch_streams_cu_range(cuda, configs, syn_weights, nets):
    """The context model on its own CUs next to the decoder (streams.py) gives the same numbers as the serial schedule; the
    CU-range stream really is a separate, usable stream; shapes that fill the chip fall back to the plain side stream."""
    import ctypes
    from imgcomp_cvpr_amd import bits, streams, weights as W, _lib
    ae, pc = nets
    auto = streams.BranchStreams(cuda)                  # default: no second stream, the context model runs ahead of the decoder
    assert auto.share == 'serial' and auto.context_model_stream(1, 512, 768) is auto.main and auto.decode_flags(auto.main) == 0
    assert auto.context_model_stream(1, 256, 256) is auto.main
    auto.close()
    bs = streams.BranchStreams(cuda, share='cu_range')
    try:
        # Kodak: 192 whole-K work-groups -> on a 256-CU MI355X a quarter of the chip is idle; 4K: the decoder fills every round
        assert int(_lib.lib.ic_wino3x3_c128_workgroups(1, 128, 192, 0)) >= 256   # alone, the layer is spread over the whole chip
        wgs = int(_lib.lib.ic_wino3x3_c128_workgroups(1, 128, 192, _lib.CONV3_LEAVE_IDLE_CUS))   # next to a CU-range stream it stays whole-K
        assert wgs == 192
        expect = min((bs.n_cus - wgs) // 8 * 8, bs.n_cus // 2) if wgs < bs.n_cus else 0
        assert bs.idle_cus(1, 512, 768) == expect
        assert bs.idle_cus(1, 2160, 3840) == 0
        assert bs.context_model_stream(1, 2160, 3840) is bs._plain
        side = bs.context_model_stream(1, 512, 768)
        if expect >= bs.MIN_CUS:
            assert side is not bs._plain and side.cuda_stream != bs.main.cuda_stream
        assert bs.context_model_stream(1, 512, 768) is side                    # cached
        x = dev(W.synthetic_image((1, 3, 128, 192), 'natural', seed=5), cuda)
        pad = pc.auto_pad_value(ae)
        enc = ae.encode(x, False)
        ref_out = ae.decode(enc.qhard, False, plan_flags=bs.decode_flags(side))      # the serial schedule, same launch plan
        ref_bpp = float(bits.bitcost_to_bpp(pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad), x))
        torch.cuda.synchronize()
        outer = torch.cuda.current_stream(cuda)
        bs.main.wait_stream(outer)
        with torch.cuda.stream(bs.main):
            for _ in range(3):
                enc = ae.encode(x, False)
                side.wait_stream(bs.main)
                with torch.cuda.stream(side):
                    bpp = bits.bitcost_to_bpp(pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad), x)
                out = ae.decode(enc.qhard, False, plan_flags=bs.decode_flags(side))
                bs.main.wait_stream(side)
        outer.wait_stream(bs.main)
        torch.cuda.synchronize()
        assert torch.equal(out, ref_out)
        assert float(bpp) == ref_bpp
        # out-of-range requests are refused, not clamped
        h = ctypes.c_void_p()
        assert _lib.lib.ic_stream_create_cu_range(max(bs.n_cus - 8, 0), 16, ctypes.byref(h)) != 0
    finally:
        bs.close()


def test_images_in_flight_plan_hint(cuda, nets, configs, syn_weights):
    """IC_CONV3_IN_FLIGHT(n): with n >= 2 independent calls in flight the 3x3 plan takes whole-K jobs for launches of >= 128 tile
    groups (a Kodak map: 192 work-groups), small maps keep their one-launch plan; the outputs do not depend on it
    (same operations per output in every Winograd form), neither do they when several images really are in flight on streams."""
    import ctypes
    from imgcomp_cvpr_amd import autoencoder, weights as W, _lib
    lib = _lib.lib
    pl = (ctypes.c_longlong * 5)()
    _lib.check(lib.ic_wino3x3_c128_plan(1, 128, 192, _lib.CONV3_IN_FLIGHT(4), pl))
    assert list(pl) == [192, 0, 0, 0, 0] and int(lib.ic_wino3x3_c128_workgroups(1, 128, 192, _lib.CONV3_IN_FLIGHT(4))) == 192
    _lib.check(lib.ic_wino3x3_c128_plan(1, 128, 192, _lib.CONV3_IN_FLIGHT(1), pl))
    assert list(pl)[:3] == [0, 192, 3]                              # one call at a time: NB = 3 segment jobs on every CU
    _lib.check(lib.ic_wino3x3_c128_plan(1, 64, 64, _lib.CONV3_IN_FLIGHT(8), pl))
    assert list(pl)[:3] == [0, 32, 1]                               # small maps keep their short NB = 1 jobs
    ae_cfg, _ = configs
    ae, _pc = nets
    n = 3
    xs = [dev(W.synthetic_image((1, 3, 128, 192), 'natural', seed=20 + i), cuda) for i in range(n)]
    ref = []
    for x in xs:
        e = ae.encode(x, False)
        ref.append((e.symbols.clone(), e.z.clone(), ae.decode(e.qhard, False).clone()))
    aes = [autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(syn_weights, cuda) for _ in range(n)]
    strs = [torch.cuda.Stream(device=cuda) for _ in range(n)]
    torch.cuda.synchronize()
    outs = [None] * n
    for rep in range(3):
        for i in range(n):
            with torch.cuda.stream(strs[i]):
                aes[i].plan_flags = _lib.CONV3_IN_FLIGHT(n)
                e = aes[i].encode(xs[i], False)
                outs[i] = (e.symbols, e.z, aes[i].decode(e.qhard, False))
    torch.cuda.synchronize()
    for (s0, z0, o0), (s1, z1, o1) in zip(ref, outs):
        assert torch.equal(s0, s1) and torch.equal(z0, z1) and torch.equal(o0, o1)


def test_next_layer_filter_prefetch_changes_no_bit(cuda, configs, syn_weights):
    """A lone F(4x4) launch inside ic_ae_encode_f32 / ic_ae_decode_f32 pulls the NEXT layer's filter fragments into its XCD's L2 with
    LDS-DMA loads into a sink behind the 72 KB B-operand ring (conv3x3_wino4.hip: M0 = an LDS address above 64 KB).  Were the sink to
    land inside the ring -- a part that honours fewer bits of M0 -- B operands would be corrupted.  The prefetch is on for a
    call alone on the chip (flags 0) and off when the caller announces calls in flight (IC_CONV3_IN_FLIGHT): same kernel, same plan
    otherwise, so the whole encoder and decoder must agree bit for bit, at the Kodak shape and on a map with 2 x 8-tile segments."""
    from imgcomp_cvpr_amd import autoencoder, weights as W, _lib
    ae_cfg, _ = configs
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(syn_weights, cuda)
    for (h, w) in ((512, 768), (512, 640)):
        assert int(_lib.lib.ic_conv3x3_c128_pick_form(1, h // 4, w // 4, 0)) == 2 and int(_lib.lib.ic_conv3x3_c128_pick_form(1, h // 4, w // 4, _lib.CONV3_IN_FLIGHT(4))) == 2
        x = dev(W.synthetic_image((1, 3, h, w), 'natural', seed=h + w), cuda)
        e_pf = ae.encode(x, False, plan_flags=0)
        z, sym, qh = e_pf.z.clone(), e_pf.symbols.clone(), e_pf.qhard.clone()
        e_no = ae.encode(x, False, plan_flags=_lib.CONV3_IN_FLIGHT(4))
        assert torch.equal(z, e_no.z) and torch.equal(sym, e_no.symbols), 'the next-layer prefetch changes the encoder output'
        xo = ae.decode(qh, False, plan_flags=0).clone()
        assert torch.equal(xo, ae.decode(qh, False, plan_flags=_lib.CONV3_IN_FLIGHT(4))), 'the next-layer prefetch changes the decoder output'


def test_val_batches_small_same_shape_images(cuda, tmp_path):
    """a directory of small images of one shape: val.py evaluates consecutive same-shape images as one batch of
    up to 8 -- the rows of measures.csv, their order and the saved reconstructions are those of the one-image-per-step loop
    (`--batch_same_shape 1`); values agree to the fp32 agreement of the plans the two batch sizes run (no cross-image term exists
    in the inference path), a shape change closes a batch."""
    from PIL import Image
    from imgcomp_cvpr_amd import val, weights as W
    imgs = tmp_path / 'small'
    imgs.mkdir()
    shapes = [(128, 128)] * 10 + [(104, 136)] + [(128, 128)] * 2
    for i, (h, w) in enumerate(shapes):
        im = W.synthetic_image((1, 3, h, w), 'natural', seed=60 + i)[0].transpose(1, 2, 0)
        Image.fromarray(im).save(str(imgs / 'img{:02d}.png'.format(i)))
    root = tmp_path / 'logs'
    (root / '0515_1103 ae_configs@cvpr@low pc_configs@cvpr@res_shallow').mkdir(parents=True)
    out = root / '0515_1103 small'
    rows, saved = {}, {}
    for mode in ('1', '8'):
        val.main([str(root), '0515_1103', str(imgs), '--weights', 'synthetic', '--reset', '--save_ours', '--batch_same_shape', mode, '--in_flight', '2'])
        rows[mode] = [r.split(',') for r in (out / 'measures.csv').read_text().strip().split('\n')]
        saved[mode] = [np.asarray(Image.open(str(out / 'imgs' / 'img{:02d}.png'.format(i)))).astype(np.int32) for i in range(len(shapes))]
    assert len(rows['1']) == len(rows['8']) == len(shapes) + 1 and rows['1'][0] == rows['8'][0]
    for a_, b_ in zip(rows['1'][1:], rows['8'][1:]):
        assert a_[0] == b_[0]                                                     # same image in the same row
        assert abs(float(a_[1]) - float(b_[1])) < 1e-4 * max(1.0, float(a_[1]))   # bpp
        assert abs(float(a_[2]) - float(b_[2])) < 1e-4 and abs(float(a_[3]) - float(b_[3])) < 1e-2
    for a_, b_ in zip(saved['1'], saved['8']):
        assert a_.shape == b_.shape and np.abs(a_ - b_).max() <= 1 and (a_ != b_).mean() < 1e-3
    # the grouping that produced the batched rows: 8 + 2 | 1 | 2
    got = [len(b) for b in val._same_shape_batches(((i, np.zeros((3, -(-h // 8) * 8, -(-w // 8) * 8), np.uint8)) for i, (h, w) in enumerate(shapes)), 8)]
    assert got == [8, 2, 1, 2]


def test_val_two_ranks_share_the_images(cuda, tmp_path):
    """multi-GPU inference as val.py does it (image-sharded, no data-path collective, one gather of the per-image
    scalars): two ranks under torch.distributed.run -- on this box's one GPU, so gloo and --device cuda:0 -- write the same
    measures.csv as one rank."""
    import subprocess
    import sys
    from PIL import Image
    from imgcomp_cvpr_amd import val, weights as W
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imgs = tmp_path / 'set5'
    imgs.mkdir()
    for i, (h, w) in enumerate(((64, 96), (120, 72), (88, 88), (64, 96), (56, 200))):
        im = W.synthetic_image((1, 3, h, w), 'natural', seed=40 + i)[0].transpose(1, 2, 0)
        Image.fromarray(im).save(str(imgs / 'img{:02d}.png'.format(i)))
    root = tmp_path / 'logs'
    (root / '0515_1103 ae_configs@cvpr@low pc_configs@cvpr@res_shallow').mkdir(parents=True)
    val.main([str(root), '0515_1103', str(imgs), '--weights', 'synthetic', '--reset'])
    out = root / '0515_1103 set5' / 'measures.csv'
    one = out.read_text()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29581', '-m', 'imgcomp_cvpr_amd.val', str(root), '0515_1103', str(imgs), '--weights', 'synthetic',
           '--reset', '--backend', 'gloo', '--device', 'cuda:0', '--in_flight', '2']
    p = subprocess.run(cmd, cwd=root_dir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-3000:]
    assert out.read_text() == one and len(one.strip().split('\n')) == 6
