"""-m "not gpu": host-side pieces -- config DSL, metrics vs the reference's outputs, NumPy helpers of the
plugin modules vs the reference's outputs, the C ABI surface, weights inventory, sharding over gloo."""
import os
import re
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


# ---- config DSL -----------------------------------------------------------------------------------

def test_config_inheritance_and_values(configs):
    ae, pc = configs
    assert ae.arch == 'CVPR' and ae.num_chan_bn == 32 and ae.arch_param_B == 5
    assert ae.H_target == pytest.approx(0.4) and ae.beta == 500 and ae.num_centers == 6
    assert ae.normalization == 'FIXED' and ae.heatmap is True and ae.crop_size == (160, 160)
    assert ae.lr_initial == pytest.approx(8e-5) and ae.batch_size == 30          # batch_size from ../base
    assert pc.arch == 'res_shallow' and pc.kernel_size == 3 and pc.arch_param__k == 24
    assert pc.use_centers_for_padding is True and pc.regularization_factor is None
    with pytest.raises(AttributeError):
        ae.no_such_parameter


def test_config_variants_and_rel_path():
    from imgcomp_cvpr_amd import config_parser as cp
    hi, rel = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'hi'))
    assert hi.num_chan_bn == 64 and hi.H_target == 1.0 and rel == 'ae_configs/cvpr/hi'
    med, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
    assert med.H_target == pytest.approx(1.2)
    k64, rel = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow_64'))
    assert k64.arch_param__k == 64 and rel == 'pc_configs/cvpr/res_shallow_64'


def test_config_constraints_and_errors(tmp_path):
    from imgcomp_cvpr_amd import config_parser as cp
    (tmp_path / 'base').write_text('constrain mode :: A, B\nmode = A\nx = 2*3\n')
    (tmp_path / 'child').write_text('use base\nmode = B\ny = x + 1  # comment\n')
    c, _ = cp.parse(str(tmp_path / 'child'))
    assert c.mode == 'B' and c.y == 7
    (tmp_path / 'bad').write_text('use base\nmode = C\n')
    with pytest.raises(ValueError):
        cp.parse(str(tmp_path / 'bad'))
    (tmp_path / 'loop').write_text('use loop\n')
    with pytest.raises(ValueError):
        cp.parse(str(tmp_path / 'loop'))
    with pytest.raises(FileNotFoundError):
        cp.parse(str(tmp_path / 'missing'))


# ---- metrics vs the reference's own numbers ------------------------------------------------------------

def test_msssim_matches_reference_outputs():
    from imgcomp_cvpr_amd import metrics
    from tests.util import msssim_case, MSSSIM_CASES
    g = np.load(os.path.join(GOLD, 'msssim.npz'))
    for name in MSSSIM_CASES:
        img1, img2 = msssim_case(name)
        assert zlib.crc32(img1.tobytes() + img2.tobytes()) == int(g['crc_' + name]), 'fixture inputs drifted'
        ssim, cs = metrics._ssim_and_cs(img1, img2)
        assert ssim == pytest.approx(float(g['ssim_' + name]), abs=1e-10)
        assert cs == pytest.approx(float(g['cs_' + name]), abs=1e-10)
        assert metrics.multiscale_ssim(img1, img2) == pytest.approx(float(g['msssim_' + name]), abs=1e-10)
    a, b = msssim_case('a')
    assert metrics.multiscale_ssim(a, a) == pytest.approx(1.0)
    assert float(metrics.msssim_nchw_uint8(a.transpose(0, 3, 1, 2), b.transpose(0, 3, 1, 2))) == pytest.approx(
        float(g['msssim_a']), abs=1e-6)


def test_psnr():
    from imgcomp_cvpr_amd import metrics
    a = np.full((1, 3, 8, 8), 100, np.uint8)
    b = a.copy(); b[0, 0, 0, 0] = 110
    mse = 100.0 / a.size
    assert float(metrics.psnr_uint8(a, b)) == pytest.approx(10 * np.log10(255 ** 2 / mse), rel=1e-6)


# ---- arithmetic coder vs a stream produced by the reference's coder ---------------------------------------------

class _KeepOpen(__import__('io').BytesIO):
    def close(self):
        pass


def test_arithmetic_coder_reproduces_reference_stream():
    import io
    from imgcomp_cvpr_amd import arithmetic_coding as ac
    g = np.load(os.path.join(GOLD, 'arithcoding.npz'))
    buf = _KeepOpen()
    out = ac.CountingBitOutputStream(ac.BitOutputStream(buf))
    enc = ac.ArithmeticEncoder(out)
    for s, f in zip(g['symbols'], g['freqs']):
        enc.write(ac.SimpleFrequencyTable(f), int(s))
    enc.finish()
    out.close()
    assert buf.getvalue() == g['stream'].tobytes(), 'bit stream differs from the reference coder'
    assert out.num_bits == 8 * len(g['stream'])
    buf2 = _KeepOpen()
    assert ac.encode_sequence(g['symbols'], g['freqs'], buf2) == 8 * len(g['stream'])
    assert buf2.getvalue() == g['stream'].tobytes()
    dec = ac.ArithmeticDecoder(ac.BitInputStream(io.BytesIO(g['stream'].tobytes())))
    assert [dec.read(ac.SimpleFrequencyTable(f)) for f in g['freqs']] == g['symbols'].tolist()


def test_arithmetic_coder_edge_cases():
    import io
    from imgcomp_cvpr_amd import arithmetic_coding as ac
    rs = np.random.RandomState(3)
    # skewed tables at the reference's resolution (int(p * 1e9), floor 1), incl. near-deterministic rows
    p = rs.dirichlet([0.05] * 6, size=500)
    freqs = np.maximum((p * 1e9).astype(np.int64), 1)
    assert freqs.sum(1).max() <= ac.MAX_TOTAL                    # 1e9 + 6 <= 2^30 + 2 (probclass.py:474-475)
    syms = np.array([rs.choice(6, p=r) for r in p])
    buf = _KeepOpen()
    ac.encode_sequence(syms, freqs, buf)
    dec = ac.ArithmeticDecoder(ac.BitInputStream(io.BytesIO(buf.getvalue())))
    assert [dec.read(ac.SimpleFrequencyTable(f)) for f in freqs] == syms.tolist()
    with pytest.raises(ValueError):
        ac.ArithmeticEncoder(ac.BitOutputStream(_KeepOpen())).write(ac.SimpleFrequencyTable([2 ** 30, 2 ** 30]), 0)
    with pytest.raises(ValueError):
        ac.ArithmeticEncoder(ac.BitOutputStream(_KeepOpen())).write(ac.SimpleFrequencyTable([0, 5]), 0)
    with pytest.raises(ValueError):
        ac.SimpleFrequencyTable([])
    with pytest.raises(ValueError):
        ac.BitOutputStream(_KeepOpen()).write(2)


# ---- NumPy helpers of the plugin module vs the reference's outputs --------------------------------------

def test_probclass_numpy_helpers_match_reference():
    from imgcomp_cvpr_amd import probclass, config_parser as cp
    g = np.load(os.path.join(GOLD, 'probclass_np.npz'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    cls = probclass.get_network_cls(pc_cfg)
    assert cls.get_context_size(pc_cfg) == int(g['context_size'])
    assert tuple(cls.get_context_shape(pc_cfg)) == tuple(g['context_shape'])
    net = cls(pc_cfg, num_centers=6)
    np.testing.assert_array_equal(net.create_first_mask(), g['first_mask'])
    np.testing.assert_array_equal(net.create_other_mask(), g['other_mask'])
    np.testing.assert_array_equal(probclass.pad_for_probclass3d(g['vol'], 9), g['vol_padded'])
    np.testing.assert_array_equal(probclass.undo_pad_for_probclass3d(g['vol_padded'], 9), g['vol'])
    blocks = np.stack(list(probclass.iter_over_blocks(g['vol_padded'], (5, 9, 9))))
    np.testing.assert_array_equal(blocks, g['blocks'])
    assert probclass.num_blocks(g['vol_padded'].shape, (5, 9, 9)) == int(g['num_blocks'])
    with pytest.raises(KeyError):
        probclass.get_network_cls(type('C', (), {'arch': 'nope'}))


# ---- C ABI ----------------------------------------------------------------------------------------------------

def _header_prototypes():
    text = open(os.path.join(ROOT, 'include', 'imgcomp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(ic_[a-z0-9_]+)\s*\(', text))


def test_conv3x3_plan_queries_are_host_logic():
    """which kernel a 3x3 launch runs is decided on the host (ic_conv3x3_c128_pick_form and friends: no device call): the rules of
    DESIGN section 3 -- F(4x4) where two of its work-groups per CU are resident and the map fills its segments, in the better of the
    two segment shapes; h2 / h12 in phase form where the width allows."""
    from imgcomp_cvpr_amd import _lib as L
    pf, wg = L.lib.ic_conv3x3_c128_pick_form, L.lib.ic_wino4_3x3_c128_workgroups
    assert wg(1, 128, 192) == 192 and wg(8, 128, 192) == 1536 and wg(1, 540, 960) == 4050
    assert wg(32, 32, 32) == 256 and wg(1, 32, 32) == 8            # 8 x 8 tiles: 2 x 8-tile segments (1 x 16 would need 16)
    assert wg(1, 128, 190) == 0 and pf(1, 128, 190, 0) == 1        # W % 4 != 0: F(2x2)
    assert pf(1, 128, 192, 0) == 2 and pf(1, 128, 192, L.CONV3_IN_FLIGHT(4)) == 2 and pf(8, 128, 192, 0) == 2      # a Kodak map: 192 work-groups
    assert pf(1, 128, 128, 0) == 1 and pf(1, 128, 128, L.CONV3_IN_FLIGHT(4)) == 2 and pf(1, 64, 64, L.CONV3_IN_FLIGHT(4)) == 1  # 128 / 32 work-groups
    assert pf(8, 128, 192, L.CONV3_NO_WINO4) == 1 and pf(8, 128, 192, L.CONV3_DIRECT) == 0 and pf(1, 4096, 2048, 0) == 0
    assert pf(30, 40, 40, 0) == 2 and pf(200, 12, 12, 0) == 1      # 62 % / 28 % of the segments' tiles exist
    assert L.lib.ic_wino4_conv5s2_supported(1, 128, 192) == 1 and L.lib.ic_wino4_conv5s2_supported(1, 128, 190) == 0
    assert L.lib.ic_wino4_conv5s2_workgroups(1, 128, 192, 0) == 192 and L.lib.ic_wino4_conv5s2_workgroups(1, 128, 192, 1) == 384
    assert L.lib.ic_conv5s2_both_packed_floats(0) == L.lib.ic_conv2d_mfma_packed_floats(5, 5, 64, 128, 2, 0) + 36 * 256 * 128
    assert L.lib.ic_conv3x3_c128_both_packed_floats() == 9 * 128 * 128 + 2 * 16 * 128 * 128 + 36 * 128 * 128


def test_val_decodes_images_ahead_in_order(tmp_path):
    """val.py's loader threads hand the images to the loop in the order of the file list, whatever order they finish in, and
    one thread (or one image) is the plain loop."""
    from PIL import Image
    from imgcomp_cvpr_amd import val
    paths = []
    for i in range(13):
        a = np.full((8 + i, 16, 3), i, np.uint8)
        p = str(tmp_path / 'img{:02d}.png'.format(i))
        Image.fromarray(a).save(p)
        paths.append(p)
    idx = [12, 0, 5, 7, 1, 2, 3, 11, 4]
    for threads in (1, 3, 8):
        got = list(val._decoded_ahead(paths, idx, 8, threads))
        assert [g[0] for g in got] == idx
        for i, img in got:
            assert img.shape[0] == 3 and img.shape[1] % 8 == 0 and int(img[0, img.shape[1] // 2, 8]) == i
    assert list(val._decoded_ahead(paths, [], 8, 4)) == []


def test_val_batches_consecutive_same_shape_images_in_order():
    """val.py evaluates consecutive small images of one shape as one batch: the batches' members, read in order, are the image list
    in order; a shape change or the per-shape limit closes a batch; Kodak-sized images stay alone."""
    from imgcomp_cvpr_amd import val
    assert val.batch_size_for_shape(256, 256) == 8 and val.batch_size_for_shape(512, 768) == 1 and val.batch_size_for_shape(384, 512) == 2
    assert val.batch_size_for_shape(64, 64) == 8 and val.batch_size_for_shape(64, 64, limit=3) == 3 and val.batch_size_for_shape(2160, 3840) == 1
    shapes = [(256, 256)] * 11 + [(512, 768)] * 3 + [(256, 256)] * 2 + [(256, 264)] + [(256, 256)]
    decoded = [(7 * i % 1000, np.zeros((3,) + sh, np.uint8)) for i, sh in enumerate(shapes)]
    got = list(val._same_shape_batches(iter(decoded), 8))
    assert [len(b) for b in got] == [8, 3, 1, 1, 1, 2, 1, 1]
    assert [i for b in got for i, _ in b] == [i for i, _ in decoded]
    assert all(len({m[1].shape for m in b}) == 1 for b in got)
    assert [len(b) for b in val._same_shape_batches(iter(decoded), 1)] == [1] * len(shapes)
    assert list(val._same_shape_batches(iter([]), 8)) == []


def test_val_loader_threads_default_follows_the_host():
    from imgcomp_cvpr_amd import val
    import os
    n = os.cpu_count() or 1
    assert val.default_loader_threads(1) == max(1, min(16, n))
    assert val.default_loader_threads(8) == max(1, min(16, n // 8))
    assert val.default_loader_threads(10 ** 6) == 1


def test_only_the_entry_points_ask_for_hardware_queues():
    """the images in flight (val.py --in_flight, bench.py) have one stream each and the HIP runtime maps all streams onto
    GPU_MAX_HW_QUEUES hardware queues, 4 by default.  Importing the package leaves the environment alone (a training rank keeps the
    runtime's default); the entry points that keep images in flight ask for more unless the caller has decided."""
    import subprocess
    import sys
    code = 'import os; os.environ.pop("GPU_MAX_HW_QUEUES", None); import imgcomp_cvpr_amd, imgcomp_cvpr_amd.training; print(os.environ.get("GPU_MAX_HW_QUEUES"))'
    assert subprocess.check_output([sys.executable, '-c', code], cwd=ROOT).decode().strip() == 'None'
    code = ('import os; os.environ.pop("GPU_MAX_HW_QUEUES", None); import imgcomp_cvpr_amd as P; a = P.ask_for_hardware_queues(8); '
            'os.environ["GPU_MAX_HW_QUEUES"] = "2"; print(a, P.ask_for_hardware_queues(8))')
    assert subprocess.check_output([sys.executable, '-c', code], cwd=ROOT).decode().strip() == '8 2'
    src = open(os.path.join(ROOT, 'imgcomp_cvpr_amd', 'val.py')).read()
    assert 'ask_for_hardware_queues(8)' in src.split('def main(argv=None):')[1][:300]
    assert "os.environ.setdefault('GPU_MAX_HW_QUEUES'" in open(os.path.join(ROOT, 'bench.py')).read()


def test_training_wino4_mode_is_validated_per_graph(monkeypatch):
    """IMGCOMP_TRAIN_WINO4 is read when a TrainGraph is built (not at import), case- and spelling-tolerant, and a bad value names itself"""
    from imgcomp_cvpr_amd import training
    monkeypatch.delenv('IMGCOMP_TRAIN_WINO4', raising=False)
    assert training.wino4_mode() is True                         # default: both directions
    for v, want in (('0', False), ('', False), ('1', True), ('True', True), (' both ', True), ('2', True), ('FWD', 'fwd'), ('bwd', 'bwd'),
                    (True, True), (False, False)):
        assert training.wino4_mode(v) == want, v
    monkeypatch.setenv('IMGCOMP_TRAIN_WINO4', 'Bwd')
    assert training.wino4_mode() == 'bwd'
    monkeypatch.setenv('IMGCOMP_TRAIN_WINO4', 'sideways')
    with pytest.raises(ValueError, match='IMGCOMP_TRAIN_WINO4'):
        training.wino4_mode()
    assert not hasattr(training.TrainGraph, 'WINO4')              # no class-level (process-wide) setting


def test_abi_header_bindings_and_exports_agree():
    """every ic_* prototype of include/imgcomp_hip.h is bound in _lib.PROTOTYPES and exported by the .so
    (no compute calls here: there is no GPU)."""
    from imgcomp_cvpr_amd import _lib
    names = _header_prototypes()
    assert names, 'no prototypes parsed'
    assert names == set(_lib.PROTOTYPES), (names ^ set(_lib.PROTOTYPES))
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r' T (ic_[a-z0-9_]+)', out))
    assert names <= exported, names - exported
    assert _lib.lib.ic_abi_version() == 2
    # re-entrant, no global mutable state -- the product library exports no setter of any kind (the
    # ic_*_debug_* hooks of profiling builds are compiled out), and the python mirror of the per-call flags matches the header
    assert not [n for n in exported if 'set_' in n or 'debug' in n], [n for n in exported if 'set_' in n or 'debug' in n]
    hdr = open(os.path.join(ROOT, 'include', 'imgcomp_hip.h')).read()
    for name, val in re.findall(r'#define IC_(CONV3_[A-Z0-9_]+|PC_DECODE_PER_LAYER)\s+(0x[0-9a-f]+)', hdr):
        assert getattr(_lib, name) == int(val, 16), name
    assert _lib.conv3_direct_variant(3) == 0x400 and _lib.edge_tiles_per_wg(3) == 3 and _lib.conv3_leave_idle_layers(5) == 0x5000
    assert _lib.lib.ic_strerror(-3) == b'workspace too small'
    assert _lib.lib.ic_conv3x3_c128_packed_floats() == 9 * 128 * 128
    # size queries are pure host arithmetic
    feat = 4 * 24 * (35 * 38 * 38 + 34 * 36 * 36 + 33 * 34 * 34)          # three feature volumes
    packed = 4 * (3 * (3 * 14 * 256) + 3 * 14 * 128)                       # three filters in 32-row MFMA fragment order + the final layer in 16-row fragments
    assert _lib.lib.ic_pc_packed_floats(24, 6) * 4 == packed and _lib.lib.ic_pc_packed_floats(12, 6) == 0
    assert _lib.lib.ic_pc_workspace_bytes(1, 32, 32, 32, 24) == feat + packed
    assert _lib.lib.ic_ae_workspace_bytes(1, 512, 768, 32) > 0 and _lib.lib.ic_ae_workspace_bytes(0, 1, 1, 1) == 0


def test_no_cpu_fallback():
    """CPU tensors are refused loudly by the plugin surface."""
    import torch
    from imgcomp_cvpr_amd import _lib, quantizer, bits
    with pytest.raises(_lib.HipLibraryError):
        quantizer.quantize(torch.zeros(1, 1, 2, 2), torch.zeros(6), 1.0)
    with pytest.raises(_lib.HipLibraryError):
        bits.bitcost_to_bpp(torch.zeros(1, 1, 2, 2), torch.zeros(1, 3, 16, 16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'imgcomp_cvpr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


# ---- weights inventory ------------------------------------------------------------------------------------------

def test_weight_inventory_matches_reference_counts(configs, syn_weights):
    from imgcomp_cvpr_amd import weights as W
    ae, pc = configs
    assert W.num_parameters(syn_weights) == 10039916
    enc = sum(a.size for n, a in syn_weights.items() if n.startswith(W.ENC) and 'moving' not in n)
    dec = sum(a.size for n, a in syn_weights.items() if n.startswith(W.DEC) and 'moving' not in n)
    pcn = sum(a.size for n, a in syn_weights.items() if n.startswith('probclass3d'))
    assert (enc, dec, pcn) == (5042440, 4973638, 23838)
    assert syn_weights[W.DEC + '/from_bn/weights'].shape == (3, 3, 128, 32)   # kh,kw,out,in
    assert syn_weights[W.ENC + '/to_bn/weights'].shape == (5, 5, 128, 33)
    assert syn_weights[W.PC + '/conv3d_conv2_mask/weights'].shape == (2, 3, 3, 24, 6)
    again = W.synthetic_weights(ae, pc)
    assert all(np.array_equal(again[k], syn_weights[k]) for k in syn_weights)


# ---- multi-process sharding (gloo, world_size 2) ----------------------------------------------------------------

_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
from imgcomp_cvpr_amd import sharding
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{port}', rank=int(sys.argv[1]), world_size=2)
rank, world = sharding.rank_and_world()
items = ['img%02d' % i for i in range(7)]
mine = sharding.shard_indices(len(items), rank, world)
local = [(i, (items[i], i * i)) for i in mine]
allr = sharding.gather_in_order(local, len(items))
assert allr == [(items[i], i * i) for i in range(7)], allr
assert sharding.shard(items) == [items[i] for i in mine]
sharding.barrier()                      # (val.py --reset: every rank passes it, whatever rank 0 did before)
dist.destroy_process_group()
print('rank', rank, 'ok', mine)
'''


def test_image_sharding_over_gloo_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT, port=29000 + os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank {} ok'.format(r) in o


_BUCKET_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from imgcomp_cvpr_amd.training import GradBuckets
rank = int(sys.argv[1])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{port}', rank=rank, world_size=2)
flat = {{'pc': torch.full((5,), float(rank + 1)), 'dec': torch.arange(4.) * (rank + 1), 'enc': torch.ones(3) * (10 * rank)}}
b = GradBuckets(flat)
for name in ('pc', 'dec', 'enc'):        # same order on every rank, each launched as soon as "its backward is done"
    b.ready(name)
b.wait()
assert torch.allclose(flat['pc'], torch.full((5,), 1.5)), flat['pc']
assert torch.allclose(flat['dec'], torch.arange(4.) * 1.5), flat['dec']
assert torch.allclose(flat['enc'], torch.ones(3) * 5.0), flat['enc']
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'buckets ok')
'''


def test_gradient_buckets_average_over_gloo_world2(tmp_path):
    """the data-parallel gradient exchange of training.py (three flat buckets, async all-reduce, mean) on CPU."""
    script = tmp_path / 'bworker.py'
    script.write_text(_BUCKET_WORKER.format(root=ROOT, port=31000 + os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank {} buckets ok'.format(r) in o


def test_train_command_line_is_the_references():
    """code/train.py:475-502: every flag of the reference's command line parses, long and short forms."""
    from imgcomp_cvpr_amd import train
    p = train.build_arg_parser()
    f = p.parse_args(['ae', 'pc', '-dtrain', 'a', '-dtest', 'b', '-dcodec', 'c', '-o', 'out', '-ltrain', '7', '-lsave', '8',
                      '-ltest', '-1', '-lmeta', '-lgrads', '-t', '-r', 'dir', '-i', '5', '--restore_continue',
                      '--restore_skip_vars', 'Adam,global_step', '--ckpt_interval', '0.5', '-d', 'text'])
    assert (f.dataset_train, f.dataset_test, f.log_dir_root) == ('a', 'b', 'out')
    assert (f.log_interval_train, f.log_interval_save, f.log_interval_test) == (7, 8, -1)
    assert (f.restore, f.restore_itr, f.restore_continue, f.restore_skip_vars) == ('dir', 5, True, 'Adam,global_step')
    f = p.parse_args(['ae', 'pc', '--from_identity', 'idt', '--log_interval_train', '3', '--log_interval_save', '4',
                      '--log_interval_test', '5', '--dataset_train', 'x', '--dataset_test', 'y'])
    assert (f.from_identity, f.log_interval_train, f.log_interval_save, f.log_interval_test) == ('idt', 3, 4, 5)
    f = p.parse_args(['ae', 'pc', '--log_interval', '9', '--save_interval', '11'])         # round-2 spellings stay as aliases
    assert (f.log_interval_train, f.log_interval_save, f.dataset_train, f.dataset_test) == (9, 11, 'imgnet_train', 'imgnet_test')
    from imgcomp_cvpr_amd import tf_checkpoint as T
    assert T.log_dir_for_restore('/a/b/ckpts') == '/a/b' and T.log_dir_for_restore('/a/b/ckpts/ckpt-5') == '/a/b'


def test_train_host_logic(tmp_path):
    from imgcomp_cvpr_amd import train, training, config_parser as cp
    d1 = train.create_unique_log_dir(['ae_configs/cvpr/low', 'pc_configs/cvpr/res_shallow'], str(tmp_path))
    d2 = train.create_unique_log_dir(['ae_configs/cvpr/low', 'pc_configs/cvpr/res_shallow'], str(tmp_path))
    assert d1 != d2 and os.path.basename(d1).endswith('ae_configs@cvpr@low pc_configs@cvpr@res_shallow')
    from imgcomp_cvpr_amd import val
    assert val.is_log_date(os.path.basename(d1).split(' ')[0])
    ld = train.CropLoader(None, (16, 24), 3, seed=1, synthetic=True)
    b = ld.get_batch()
    assert b.shape == (3, 3, 16, 24) and b.dtype == np.float32 and 0 <= b.min() and b.max() <= 255
    ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    assert training.learning_rate(ae, 0, 10) == pytest.approx(8e-5) and training.learning_rate(ae, 20, 10) == pytest.approx(8e-6)
    b1 = training.GradBuckets({'x': __import__('torch').ones(2)})
    b1.ready('x'); b1.wait()                                  # world size 1: no-op


def test_sharding_single_process():
    from imgcomp_cvpr_amd import sharding
    sharding.barrier()                  # no process group: a no-op
    assert sharding.shard_indices(5, 1, 2) == [1, 3]
    assert sorted(sharding.shard_indices(9, 0, 4) + sharding.shard_indices(9, 1, 4) +
                  sharding.shard_indices(9, 2, 4) + sharding.shard_indices(9, 3, 4)) == list(range(9))
    assert sharding.gather_in_order([(1, 'b'), (0, 'a')], 2) == ['a', 'b']
    with pytest.raises(RuntimeError):
        sharding.gather_in_order([(0, 'a')], 2)
    with pytest.raises(ValueError):
        sharding.shard_indices(3, 2, 2)


# ---- val.py host logic -----------------------------------------------------------------------------------------

def test_val_padding_and_paths(tmp_path):
    from imgcomp_cvpr_amd import val
    im = np.arange(5 * 13 * 3, dtype=np.uint8).reshape(5, 13, 3)
    padded, undo = val.add_padding(im, 8)
    assert padded.shape == (8, 16, 3)
    # centred, the odd pixel goes to the far side (images_iterator.py:45-55): 3 rows -> 1 before / 2 after
    assert np.array_equal(padded[1:6, 1:14], im) and padded[0].sum() == 0 and padded[6:].sum() == 0
    assert np.array_equal(undo(padded), im)
    same, _ = val.add_padding(np.zeros((16, 24, 3), np.uint8), 8)
    assert same.shape == (16, 24, 3)
    rgba, _ = val.add_padding(np.zeros((8, 8, 4), np.uint8), 8)
    assert rgba.shape == (8, 8, 3)
    # job dir name -> config files (logdir_helpers.py:130-151)
    job = tmp_path / '0515_1103 ae_configs@cvpr@low pc_configs@cvpr@res_shallow'
    job.mkdir()
    here = os.path.join(ROOT, 'imgcomp_cvpr_amd')
    ae_p, pc_p = val.config_paths_from_log_dir(str(job), [os.path.join(here, 'ae_configs'), os.path.join(here, 'pc_configs')])
    assert ae_p.endswith('ae_configs/cvpr/low') and pc_p.endswith('pc_configs/cvpr/res_shallow')
    assert val.log_date_from_log_dir(str(job)) == '0515_1103'
    assert list(val.iter_job_dirs(str(tmp_path), '0515')) == [str(job)]
    with pytest.raises(ValueError):
        val.log_date_from_log_dir(str(tmp_path / 'nodate x y'))
    with pytest.raises(ValueError):
        val.get_image_paths(str(tmp_path / 'no_images'))
    w = val.MeasuresWriter(str(tmp_path / 'out'))
    w.append('a.png', {'bpp': 0.5, 'ms-ssim': 0.9, 'psnr': 30.0})
    w.close()
    assert (tmp_path / 'out' / 'measures.csv').read_text() == 'img_name,bpp,ms-ssim,psnr\na.png,0.5,0.9,30.0\n'


def test_tf_checkpoint_bundle_round_trip(tmp_path):
    """N1: the TF-1 tensor-bundle reader/writer -- published CRC-32C check values and table magic, multi-block tables
    with prefix-compressed keys, scalars and several dtypes, the reference's ckpt-directory conventions
    (saver.py:19-43,102-142), corruption is detected."""
    import pickle
    import struct
    from imgcomp_cvpr_amd import tf_checkpoint as T
    assert T.crc32c(b'123456789') == 0xE3069283                      # CRC-32C (Castagnoli) check value
    assert T.crc32c(bytes(32)) == 0x8a9136aa and T.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert T.crc32c(bytes(range(32))) == 0x46dd794e
    rs = np.random.RandomState(0)
    tens = {'autoencoder/encoder/h1/weights': rs.randn(5, 5, 3, 64).astype(np.float32),
            'autoencoder/encoder/h1/weights/Adam': rs.randn(5, 5, 3, 64).astype(np.float32),
            'autoencoder/encoder/centers': rs.randn(6).astype(np.float32),
            'probclass3d/logits/conv3d_conv0_mask/biases': rs.randn(24).astype(np.float32),
            'global_step': np.array(1234, np.int64), 'beta1_power': np.array(0.9, np.float32),
            'some/int32': np.arange(-3, 4, dtype=np.int32), 'some/bool': np.array([True, False])}
    for i in range(300):
        tens['autoencoder/decoder/layer_%03d/BatchNorm/gamma' % i] = rs.randn(7).astype(np.float32)
    ckpt_dir = str(tmp_path / 'logdir' / 'ckpts')
    prefix = ckpt_dir + '/ckpt-1234'
    T.write_bundle(prefix, tens)
    T.write_bundle(ckpt_dir + '/ckpt-200', {k: v for k, v in tens.items() if 'layer' not in k})
    T.write_var_names(ckpt_dir, sorted(tens))
    # a small block size forces several data blocks and restart points through the same reader
    items = [(b'', b'hdr')] + [(('k%05d' % i).encode(), bytes([i % 251]) * (i % 40)) for i in range(2000)]
    T._write_table(str(tmp_path / 't.index'), items, block_size=512)
    assert list(T._read_table(str(tmp_path / 't.index')).items()) == items
    raw = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57       # kTableMagicNumber
    back = T.read_bundle(prefix, verify=True)
    assert list(back) == sorted(tens, key=lambda n: n.encode())
    for k, v in tens.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    assert T.list_variables(prefix)['global_step'] == (np.dtype(np.int64), ())
    # directory conventions
    assert [i for i, _ in T.all_ckpts_with_iterations(ckpt_dir)] == [200, 1234]
    assert T.latest_checkpoint_before_itr(ckpt_dir, -1)[0] == 1234 and T.latest_checkpoint_before_itr(ckpt_dir, 1000)[0] == 200
    with pytest.raises(ValueError):
        T.latest_checkpoint_before_itr(ckpt_dir, 100)
    assert pickle.load(open(ckpt_dir + '/var_names.pkl', 'rb'))[0].endswith(':0')
    assert 'global_step' not in T.read_var_names(ckpt_dir, ['global_step', 'Adam'])
    w = T.load_weights(str(tmp_path / 'logdir'))
    assert 'autoencoder/encoder/h1/weights' in w and not any('Adam' in k or 'global_step' in k for k in w)
    assert len(w) == 303
    assert set(T.load_weights(ckpt_dir, itr=500)) == {k for k in tens if T.is_model_variable(k) and 'layer' not in k}
    # corruption: one flipped byte in an index block / in the tensor data
    bad = bytearray(raw)
    bad[10] ^= 1
    open(prefix + '.index', 'wb').write(bytes(bad))
    with pytest.raises(ValueError):
        T.read_bundle(prefix)
    open(prefix + '.index', 'wb').write(raw)
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[100] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    with pytest.raises(ValueError):
        T.read_bundle(prefix, verify=True)


def test_crop_loader_files_and_synthetic(tmp_path):
    """N4: training input -- crops come out of decoded files through the background shuffle pool, NUM_CROPS_PER_IMG crops
    per decoded image share one flip decision (inputpipeline.py:199-213); the synthetic mode is deterministic."""
    from PIL import Image
    from imgcomp_cvpr_amd import train
    imgs = []
    for i in range(3):
        a = np.random.RandomState(i).randint(0, 255, (80, 112, 3), dtype=np.uint8)
        Image.fromarray(a).save(str(tmp_path / 'im{}.png'.format(i)))
        imgs.append(np.transpose(a, (2, 0, 1)))
    L = train.CropLoader(str(tmp_path / '*.png'), (64, 48), 8, seed=0, capacity=64, min_after_dequeue=24, num_threads=2)
    try:
        for _ in range(4):
            b = L.get_batch()
            assert b.shape == (8, 3, 64, 48) and b.dtype == np.float32
            for c in b:                                  # every crop is a window of one of the files, possibly mirrored
                found = False
                for im in imgs:
                    for cand in (c, c[:, :, ::-1]):
                        y0 = np.where((im[0, :, :1] == cand[0, 0, 0]))[0]
                        win = np.lib.stride_tricks.sliding_window_view(im, (3, 64, 48))[0]
                        found = found or bool((win == cand.astype(np.uint8)).all(axis=(2, 3, 4)).any())
                assert found
    finally:
        L.close()
    with pytest.raises(ValueError):
        train.CropLoader(str(tmp_path / '*.jpg'), (64, 48), 8)
    S1 = train.CropLoader(None, (64, 64), 6, seed=1, synthetic=True)
    S2 = train.CropLoader(None, (64, 64), 6, seed=1, synthetic=True)
    assert np.array_equal(S1.get_batch(), S2.get_batch())
    assert train.NUM_CROPS_PER_IMG == 8


def test_datasets_tfrecord_pickle_glob(tmp_path, monkeypatch):
    """N4: the three dataset forms of inputpipeline.get_dataset -- TFRecord shards ('image/encoded' feature, CRC-checked
    container), a paths pickle relative to its directory, an image glob -- all stream decoded RGB images; the crop loader
    runs on top of a record set."""
    import io
    import pickle
    from PIL import Image
    from imgcomp_cvpr_amd import datasets as D, train
    imgs, encoded = [], []
    for i in range(5):
        a = np.random.RandomState(i).randint(0, 255, (70 + i, 90, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format='PNG')
        imgs.append(a)
        encoded.append(buf.getvalue())
        Image.fromarray(a).save(str(tmp_path / 'im{}.png'.format(i)))
    (tmp_path / 'train').mkdir()
    D.write_tfrecord(str(tmp_path / 'train' / 'a.tfrecord'),
                     [D.make_example({'image/format': b'PNG', 'image/encoded': e}) for e in encoded[:3]])
    D.write_tfrecord(str(tmp_path / 'train' / 'b.tfrecord'),
                     [D.make_example({'image/encoded': e, 'image/class/label': b'7'}) for e in encoded[3:]])
    recs = list(D.iter_tfrecord(str(tmp_path / 'train' / 'a.tfrecord'), verify=True))
    assert len(recs) == 3 and D.example_bytes_feature(recs[1], 'image/encoded') == encoded[1]
    assert D.example_bytes_feature(recs[0], 'image/format') == b'PNG'
    with pytest.raises(KeyError):
        D.example_bytes_feature(recs[0], 'nope')
    raw = bytearray(open(str(tmp_path / 'train' / 'a.tfrecord'), 'rb').read())
    raw[40] ^= 1
    open(str(tmp_path / 'bad.tfrecord'), 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        list(D.iter_tfrecord(str(tmp_path / 'bad.tfrecord'), verify=True))
    monkeypatch.setenv('RECORDS_ROOT', str(tmp_path))
    ds = D.get_dataset('imgnet_train')
    assert ds.num_images == 1281167 and len(ds.files) == 2
    it = ds.stream(np.random.RandomState(0))
    got = [next(it) for _ in range(10)]                               # two epochs
    for g in got:
        assert any(g.shape == a.shape and np.array_equal(g, a) for a in imgs)
    assert len({g.shape[0] for g in got}) == 5
    with pytest.raises(ValueError):
        D.get_dataset('imgnet_test')                                  # no $RECORDS_ROOT/val shards
    with open(str(tmp_path / 'paths.pkl'), 'wb') as f:
        pickle.dump(['im0.png', 'im3.png'], f)
    dp = D.get_dataset(str(tmp_path / 'paths.pkl'))
    assert dp.num_images == 2 and next(dp.stream(np.random.RandomState(1))).shape[1] == 90
    dg = D.get_dataset(str(tmp_path / '*.png'))
    assert dg.num_images == 5
    with pytest.raises(ValueError):
        D.get_dataset(str(tmp_path / '*.bmp'))
    L = train.CropLoader('imgnet_train', (64, 64), 4, seed=0, capacity=32, min_after_dequeue=8, num_threads=2)
    try:
        assert L.get_batch().shape == (4, 3, 64, 64) and L.num_images == 1281167
    finally:
        L.close()


def test_device_metric_twins_on_cpu_tensors():
    """the torch float64 twins of MS-SSIM / PSNR (used on the device by val.py) against the numpy versions."""
    import torch
    from imgcomp_cvpr_amd import metrics, weights as W
    a = W.synthetic_image((1, 3, 192, 224), 'natural', 4)
    b = np.clip(a.astype(np.float64) + np.random.RandomState(0).normal(0, 5, a.shape), 0, 255).astype(np.uint8)
    ta, tb = torch.as_tensor(a), torch.as_tensor(b)
    assert abs(metrics.msssim_nchw_uint8_device(ta, tb) - float(metrics.msssim_nchw_uint8(a, b))) < 1e-7
    assert abs(metrics.psnr_uint8_device(ta, tb) - float(metrics.psnr_uint8(a, b))) < 1e-5


def test_lr_schedule_counts_epochs_like_the_reference():
    """training_helpers.py:22-34,51-60: an epoch is num_images // (batch_size // NUM_CROPS_PER_IMG) iterations -- batch 30
    with 8 crops per decoded image consumes 3 images per step -- and DECAY multiplies the rate by 0.1 every
    lr_schedule_decay_interval epochs, staircase.  ImageNet train (1,281,167 images): 427,055 iterations per epoch, first
    decay after 854,110 iterations (a loader-independent formula: the 10x-too-early decay of round 1 counted batch_size
    images per step)."""
    from imgcomp_cvpr_amd import training, config_parser as cp
    from imgcomp_cvpr_amd.train import NUM_CROPS_PER_IMG
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    assert NUM_CROPS_PER_IMG == 8 and int(ae_cfg.batch_size) == 30
    n = training.get_num_itr_per_epoch(1281167, int(ae_cfg.batch_size), NUM_CROPS_PER_IMG)
    assert n == 1281167 // (30 // 8) == 427055
    assert ae_cfg.lr_schedule == 'DECAY' and ae_cfg.lr_schedule_decay_staircase
    steps = int(n * ae_cfg.lr_schedule_decay_interval)
    assert steps == 854110
    lr0 = float(ae_cfg.lr_initial)
    assert training.learning_rate(ae_cfg, 0, n) == lr0 == training.learning_rate(ae_cfg, steps - 1, n)
    assert abs(training.learning_rate(ae_cfg, steps, n) - lr0 * 0.1) < 1e-12
    assert abs(training.learning_rate(ae_cfg, 2 * steps + 5, n) - lr0 * 0.01) < 1e-12
    # tiny data sets and batches smaller than the crops per image do not divide by zero
    assert training.get_num_itr_per_epoch(5, 4, 8) == 5 and training.get_num_itr_per_epoch(0, 30, 8) == 1


def test_checkpoint_name_filters():
    from imgcomp_cvpr_amd import tf_checkpoint as T
    w = 'autoencoder/encoder/h1/weights'
    assert T.is_model_variable(w) and T.is_model_variable('probclass3d/logits/conv3d_conv0_mask/biases')
    for n in (w + '/Adam_AE', w + '/Adam_AE_1', w + '/Adam', 'global_step', 'Adam_AE/beta1_power', 'beta1_power'):
        assert not T.is_model_variable(n), n
    for n in (w + '/Adam_AE', w + '/Adam_AE_1', 'probclass3d/logits/conv3d_conv0_mask/biases/Adam_PC_1', 'global_step',
              'Adam_PC/beta2_power', 'beta1_power', 'beta2_power', 'beta1_power_1', 'beta2_power_1'):
        assert T.is_training_state(n), n
    assert not T.is_training_state(w)


def test_optimizer_classes_follow_the_config(configs):
    """training_helpers.optimizer_cls (:42-48): ADAM | SGD | MOMENTUM (use_nesterov=True, momentum = optimizer_momentum).  The update
    rules on plain tensors against TF's documented ones; an unknown name raises (round 3 trained with Adam whatever the config said)."""
    import torch
    from imgcomp_cvpr_amd import training
    ae_cfg, pc_cfg = configs
    assert ae_cfg.optimizer == 'ADAM' and pc_cfg.optimizer == 'ADAM' and float(ae_cfg.optimizer_momentum) == 0.9

    class Cfg(object):
        lr_initial = 0.1
        optimizer_momentum = 0.9
    p0, g = torch.tensor([1.0, -2.0, 3.0]), torch.tensor([0.5, -1.0, 0.25])
    c = Cfg()
    c.optimizer = 'SGD'
    p = p0.clone()
    opt = training.create_optimizer(c, [p], [g], flat=None)
    assert isinstance(opt, training.TFGradientDescent) and opt.slot_tensors([p]) == []
    opt.step()
    assert torch.allclose(p, p0 - 0.1 * g)
    opt.step(0.01)
    assert torch.allclose(p, p0 - 0.11 * g)
    c.optimizer = 'MOMENTUM'
    p = p0.clone()
    opt = training.create_optimizer(c, [p], [g], flat=[(p, g)])
    assert isinstance(opt, training.TFMomentum)
    acc, ref = torch.zeros(3), p0.clone()
    for _ in range(3):
        opt.step()
        acc = 0.9 * acc + g                                  # ApplyMomentum, use_nesterov
        ref = ref - 0.1 * g - 0.1 * 0.9 * acc
    assert torch.allclose(p, ref, atol=1e-6)
    (slot,), = [opt.slot_tensors([p])]
    assert torch.allclose(slot[0], acc, atol=1e-6)
    c.optimizer = 'ADAM'
    assert isinstance(training.create_optimizer(c, [p0.clone()], [g], flat=None), training.TFAdam)
    c.optimizer = 'RMSPROP'
    with pytest.raises(ValueError):
        training.create_optimizer(c, [p0.clone()], [g], flat=None)


def test_adam_step_count_from_tf_beta_powers():
    """TF-1 Adam keeps beta ** (t + 1) after t updates (beta1_power starts at beta1, _finish multiplies once per update); a
    step-0 TF checkpoint must restore t = 0, and beta2_power still resolves t where float32 beta1_power has gone to zero."""
    import math
    from imgcomp_cvpr_amd import training

    class Opt(object):
        b1, b2 = 0.9, 0.999
    tr = training.Trainer.__new__(training.Trainer)
    tr.global_step = 777
    for t in (0, 1, 5, 829, 2000, 20000):
        ck = {'beta1_power': np.array(0.9 ** (t + 1), np.float32), 'beta2_power': np.array(0.999 ** (t + 1), np.float32)}
        assert tr._adam_steps_from_checkpoint(ck, 'Adam_AE', Opt) == t, t
    assert tr._adam_steps_from_checkpoint({'beta1_power_1': np.array(0.9 ** 4, np.float32)}, 'Adam_PC', Opt) == 3
    assert tr._adam_steps_from_checkpoint({}, 'Adam_AE', Opt) == 777                      # no beta powers: the global step
    # checkpoints of rounds 1-3 of this repo: `<optimiser>/beta?_power` = beta ** t (not t + 1)
    for t in (0, 1, 12, 3000):
        ck = {'Adam_PC/beta1_power': np.array(0.9 ** t, np.float32), 'Adam_PC/beta2_power': np.array(0.999 ** t, np.float32)}
        assert tr._adam_steps_from_checkpoint(ck, 'Adam_PC', Opt) == t, t


# ---- csrc/isa_audit.py: the build gate for the kernels whose MFMAs are inline asm (profiles/r05_w4_rootcause.md) ----
_AUDIT_HEAD = '_Z1kv:\n'
_AUDIT_TAIL = '.Lfunc_end0:\n'


def _audit(tmp_path, body):
    sys.path.insert(0, os.path.join(ROOT, 'imgcomp_cvpr_amd', 'csrc'))
    import isa_audit
    p = tmp_path / 'k.s'
    p.write_text(_AUDIT_HEAD + body + _AUDIT_TAIL)
    out = []
    for name, lines in isa_audit.kernels(str(p)):
        n, f = isa_audit.audit_kernel(name, lines)
        out += f
    return out


def test_isa_audit_flags_the_two_hazards_and_nothing_else(tmp_path):
    """the exact instruction pairs of round 4's failing build are findings; the shipped patterns (operands from LDS / buffer loads,
    accumulators read behind an s_nop pad, a VALU instruction overwriting SrcA / SrcB right BEHIND the MFMA) are not"""
    mf = '\tv_mfma_f32_16x16x4_f32 v[38:41], v25, v33, v[38:41]\n'
    # (A) spill of the accumulator 0 wait states behind its MFMA; a copy 3 instructions later; an AGPR read
    assert any('(A)' in f for f in _audit(tmp_path, mf + '\tscratch_store_dwordx4 off, v[38:41], off offset:16\n'))
    assert any('(A)' in f for f in _audit(tmp_path, mf + '\ts_add_i32 s1, s2, 4\n\tv_add_f32_e32 v1, v2, v3\n\tv_mov_b64_e32 v[10:11], v[40:41]\n'))
    assert any('(A)' in f for f in _audit(tmp_path, '\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]\n\tv_accvgpr_read_b32 v9, a2\n'))
    assert any('(A)' in f for f in _audit(tmp_path, mf + '\ts_nop 9\n\tv_add_f32_e32 v38, 1.0, v38\n'))          # 10 states: the gate asks for 12
    assert not _audit(tmp_path, mf + '\ts_nop 11\n\tv_add_f32_e32 v38, 1.0, v38\n')
    assert not _audit(tmp_path, mf + '\ts_nop 15\n\ts_nop 15\n\tbuffer_store_dwordx4 v[38:41], v0, s[0:3], 0 offen\n')
    # (B) the allocator's copy feeding SrcC / a VALU result feeding SrcB with fewer than 2 wait states
    assert any('(B)' in f for f in _audit(tmp_path, '\tv_mov_b64_e32 v[10:11], v[44:45]\n\tv_mfma_f32_16x16x4_f32 v[8:11], v41, v21, v[8:11]\n'))
    assert any('(B)' in f for f in _audit(tmp_path, '\tv_fma_f32 v21, v1, v2, v3\n\ts_nop 0\n\tv_mfma_f32_16x16x4_f32 a[8:11], v41, v21, a[8:11]\n'))
    assert not _audit(tmp_path, '\tv_fma_f32 v21, v1, v2, v3\n\ts_nop 1\n\tv_mfma_f32_16x16x4_f32 a[8:11], v41, v21, a[8:11]\n')
    # not hazards: loads into the operands (waited for by s_waitcnt), the accumulate chain, write-after-read of SrcA / SrcB
    assert not _audit(tmp_path, '\tds_read_b128 v[20:23], v84\n\tbuffer_load_dwordx4 v[40:43], v84, s[48:51], s79 offen\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n'
                                '\tv_mfma_f32_16x16x4_f32 a[8:11], v41, v21, a[8:11]\n\tv_mfma_f32_16x16x4_f32 a[8:11], v42, v22, a[8:11]\n'
                                '\tv_fma_f32 v21, -4.0, v1, v2\n\tbuffer_load_dwordx4 v[40:43], v84, s[48:51], s79 offen\n')
    # an MFMA result consumed as the next MFMA's SrcB is a hazard too
    assert any('(A)' in f for f in _audit(tmp_path, mf + '\tv_mfma_f32_16x16x4_f32 a[8:11], v1, v38, a[8:11]\n'))


def test_isa_audit_is_part_of_the_build_and_the_shipped_assembly_is_clean():
    """csrc/Makefile audits every file with inline-asm MFMAs before it builds the object; the assembly the last build audited has no finding"""
    csrc = os.path.join(ROOT, 'imgcomp_cvpr_amd', 'csrc')
    mk = open(os.path.join(csrc, 'Makefile')).read()
    asm_files = sorted(f[:-4] for f in os.listdir(csrc) if f.endswith('.hip') and 'asm volatile("v_mfma' in open(os.path.join(csrc, f)).read())
    assert asm_files == ['conv3x3_wino4', 'conv3x3_wino_stack', 'conv3x3_wino_tn', 'conv3x3_wino_tp']
    for f in asm_files:
        assert 'AUDIT_{} := 1'.format(f) in mk, f
    assert 'isa_audit.py $*.audit.s' in mk and mk.index('isa_audit.py $*.audit.s') < mk.index('-c $< -o $@')
    for f in ('conv3x3_wino4', 'conv3x3_wino_tn'):              # built by build(): the audited assembly is on disk
        s = os.path.join(csrc, f + '.audit.s')
        if not os.path.exists(s):
            pytest.skip('library not built in this checkout')
        import re
        m = re.search(r"AUDIT_ARGS_{} := (.*)".format(f), mk)
        extra = [t.strip("'") for t in m.group(1).split()] if m else []
        r = subprocess.run([sys.executable, os.path.join(csrc, 'isa_audit.py'), s] + extra, stdout=subprocess.PIPE, universal_newlines=True)
        assert r.returncode == 0 and ' 0 finding(s)' in r.stdout, r.stdout[-2000:]
    assert 'isa_audit.py $*.audit.s $(AUDIT_ARGS_$*)' in mk


# ---- tools/pin_reference.py: the one command for the day a TF-written checkpoint arrives (N1) ----
def _hand_assembled_bundle(prefix, tensors, entries_per_block=37):
    """A TF-1 tensor bundle laid out byte by byte from the FORMAT DESCRIPTION (LevelDB table_format.md; tensor_bundle.proto),
    sharing no code with imgcomp_cvpr_amd.tf_checkpoint's writer, and deliberately different from it where the format allows:
    every entry its own restart point (no prefix compression), fixed entries per data block, zero-valued proto fields omitted
    as protobuf serialisers do (shard_id 0, offset 0, little-endian), the VersionDef in the header as BundleWriter writes it."""
    import struct

    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7f
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def crc32c_bitwise(data):                                   # reflected Castagnoli polynomial, bit by bit
        c = 0xFFFFFFFF
        for byte in data:
            c ^= byte
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
        return c ^ 0xFFFFFFFF

    def masked(c):
        return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF

    def crc_of(raw):
        if len(raw) <= 16384:
            return crc32c_bitwise(raw)
        from imgcomp_cvpr_amd import tf_checkpoint as T          # (large tensors: the library's routine, pinned by the published
        return T.crc32c(raw)                                     #  check values in test_tf_checkpoint_bundle_round_trip)

    def block(entries):
        buf, restarts = bytearray(), []
        for k, v in entries:
            restarts.append(len(buf))
            buf += varint(0) + varint(len(k)) + varint(len(v)) + k + v
        for r in restarts:
            buf += struct.pack('<I', r)
        buf += struct.pack('<I', len(restarts))
        return bytes(buf)

    items = [(b'', b'\x08\x01' + b'\x1a\x02\x08\x01')]            # BundleHeaderProto{num_shards: 1, version{producer: 1}}
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(tensors, key=lambda n: n.encode()):
            a = np.ascontiguousarray(tensors[name]) if np.ndim(tensors[name]) else np.asarray(tensors[name])
            raw = a.astype('<' + a.dtype.str[1:]).tobytes()
            f.write(raw)
            dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}[a.dtype]
            shape = b''.join(b'\x12' + varint(len(b'\x08' + varint(d))) + b'\x08' + varint(d) for d in a.shape)
            e = b'\x08' + varint(dt) + b'\x12' + varint(len(shape)) + shape
            if offset:
                e += b'\x20' + varint(offset)
            e += b'\x28' + varint(len(raw)) + b'\x35' + struct.pack('<I', masked(crc_of(raw)))
            items.append((name.encode(), e))
            offset += len(raw)
    with open(prefix + '.index', 'wb') as f:
        def emit(blk):
            off = f.tell()
            f.write(blk + b'\x00' + struct.pack('<I', masked(crc32c_bitwise(blk + b'\x00'))))
            return varint(off) + varint(len(blk))
        index = []
        for i in range(0, len(items), entries_per_block):
            chunk = items[i:i + entries_per_block]
            index.append((chunk[-1][0], emit(block(chunk))))
        meta = emit(block([]))
        idx = emit(block(index))
        f.write((meta + idx).ljust(40, b'\x00') + struct.pack('<Q', 0xdb4775248b80fb57))


def test_pin_reference_inventory_on_written_and_hand_assembled_bundles(tmp_path, configs, syn_weights):
    """tools/pin_reference.py steps 1-3 (no GPU): the log-dir / checkpoint discovery, the Appendix-B inventory against the bundle and
    var_names.pkl, the CRC-verified read -- on a bundle written by tf_checkpoint.write_bundle AND on one assembled by hand from the
    format description (reader and writer cannot share a bug); a missing, mis-shaped or unknown variable is named precisely."""
    import pickle
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pin_reference as P
    from imgcomp_cvpr_amd import tf_checkpoint as T
    state = {'global_step': np.array(7, np.int64), 'beta1_power': np.array(0.9 ** 8, np.float32),
             'autoencoder/encoder/centers/Adam_AE': np.zeros(6, np.float32), 'autoencoder/encoder/centers/Adam_AE_1': np.zeros(6, np.float32)}
    full = dict(syn_weights, **state)
    job = '0515_1103 ae_configs@cvpr@low pc_configs@cvpr@res_shallow'

    def lay_out(root, writer, tensors, names=None):
        ck = os.path.join(str(root), job, 'ckpts')
        os.makedirs(ck, exist_ok=True)
        writer(os.path.join(ck, 'ckpt-7'), tensors)
        with open(os.path.join(ck, 'var_names.pkl'), 'wb') as f:
            pickle.dump([n + ':0' for n in (names if names is not None else sorted(tensors))], f)
        return str(root)

    # 1. this package's writer
    r1 = lay_out(tmp_path / 'a', T.write_bundle, full)
    rep = P.pin(r1, 'unused', inventory_only=True, verbose=False)['inventory']
    assert rep['model_variables'] == len(syn_weights) and rep['training_state_variables'] == 4 and rep['other_variables'] == []
    assert rep['parameters'] == sum(int(np.prod(np.shape(v))) for v in syn_weights.values())
    # 2. laid out by hand
    r2 = lay_out(tmp_path / 'b', _hand_assembled_bundle, full)
    assert P.pin(r2, 'unused', inventory_only=True, verbose=False)['inventory'] == rep
    back = T.read_bundle(os.path.join(r2, job, 'ckpts', 'ckpt-7'), verify=True)
    assert list(back) == sorted(full, key=lambda n: n.encode())
    for k, v in full.items():
        assert back[k].dtype == np.asarray(v).dtype and back[k].shape == np.shape(v) and np.array_equal(back[k], v), k
    assert open(os.path.join(r1, job, 'ckpts', 'ckpt-7.index'), 'rb').read() != open(os.path.join(r2, job, 'ckpts', 'ckpt-7.index'), 'rb').read()
    w = T.load_weights(os.path.join(r2, job))
    assert set(w) == set(syn_weights)
    # 3. precise failures
    gone = {k: v for k, v in full.items() if k != 'autoencoder/encoder/h2/BatchNorm/gamma'}
    with pytest.raises(P.PinError, match=r'missing variable autoencoder/encoder/h2/BatchNorm/gamma \(expected float32 \[128\]\)'):
        P.pin(lay_out(tmp_path / 'c', _hand_assembled_bundle, gone), 'unused', inventory_only=True, verbose=False)
    bent = dict(full)
    bent['autoencoder/decoder/h12/weights'] = np.zeros((5, 5, 128, 64), np.float32)         # in/out swapped
    with pytest.raises(P.PinError, match=r'variable autoencoder/decoder/h12/weights has shape \[5, 5, 128, 64\], expected \[5, 5, 64, 128\]'):
        P.pin(lay_out(tmp_path / 'd', T.write_bundle, bent), 'unused', inventory_only=True, verbose=False)
    extra = dict(full)
    extra['autoencoder/encoder/h3/weights'] = np.zeros((3, 3, 8, 8), np.float32)
    with pytest.raises(P.PinError, match='do not have: autoencoder/encoder/h3/weights'):
        P.pin(lay_out(tmp_path / 'e', T.write_bundle, extra), 'unused', inventory_only=True, verbose=False)
    with pytest.raises(P.PinError, match='var_names.pkl does not list probclass3d/logits/conv3d_conv0_mask/biases'):
        P.pin(lay_out(tmp_path / 'f', T.write_bundle, full, [n for n in sorted(full) if n != 'probclass3d/logits/conv3d_conv0_mask/biases']),
              'unused', inventory_only=True, verbose=False)
    flipped = lay_out(tmp_path / 'g', _hand_assembled_bundle, full)
    dp = os.path.join(flipped, job, 'ckpts', 'ckpt-7.data-00000-of-00001')
    raw = bytearray(open(dp, 'rb').read())
    raw[len(raw) // 2] ^= 4
    open(dp, 'wb').write(bytes(raw))
    with pytest.raises(P.PinError, match='checksum mismatch'):
        P.pin(flipped, 'unused', inventory_only=True, verbose=False)
    with pytest.raises(P.PinError, match='expected exactly one log dir'):
        P.pin(str(tmp_path / 'nowhere'), 'unused', inventory_only=True, verbose=False)


def test_isa_audit_follows_loop_back_edges(tmp_path):
    """a hazard whose two halves sit at the end and at the top of a loop body is found too (the audit follows backward branches)"""
    loop = ('.LBB0_1:\n\tv_add_f32_e32 v38, 1.0, v38\n\ts_nop 15\n\tv_mfma_f32_16x16x4_f32 v[38:41], v25, v33, v[38:41]\n'
            '\ts_cbranch_scc1 .LBB0_1\n')
    f = _audit(tmp_path, loop)
    assert any('(A)' in x and 'back edge' in x for x in f), f
    ok = ('.LBB0_1:\n\ts_nop 11\n\tv_add_f32_e32 v38, 1.0, v38\n\ts_nop 15\n\tv_mfma_f32_16x16x4_f32 v[38:41], v25, v33, v[38:41]\n'
          '\ts_cbranch_scc1 .LBB0_1\n')
    assert not _audit(tmp_path, ok)
    vb = ('.LBB0_2:\n\tv_mfma_f32_16x16x4_f32 a[0:3], v25, v33, a[0:3]\n\ts_nop 15\n\tv_fma_f32 v33, v1, v2, v3\n\ts_cbranch_scc1 .LBB0_2\n')
    assert any('(B)' in x and 'back edge' in x for x in _audit(tmp_path, vb))


def test_isa_audit_follows_taken_forward_branches_and_reads_the_spill_metadata(tmp_path):
    """round 6: the 8-wave F(4x4) form skips its transform slices behind scalar branches.  The skipped instructions are not wait states:
    a hazard that only exists when the branch is TAKEN is found; and a kernel whose metadata reports scratch is a finding of its own
    unless its name is allow-listed (the documented guarantee `nothing was spilled` is now enforced: ADVICE r5)."""
    mf = '\tv_mfma_f32_16x16x4_f32 v[38:41], v25, v33, v[38:41]\n'
    pad12 = ''.join('\tv_add_f32_e32 v{}, 1.0, v{}\n'.format(60 + i, 60 + i) for i in range(12))
    # linear scan: 12 VALU instructions separate the MFMA from the read of its result; taken branch: nothing does
    skip = mf + '\ts_cbranch_vccnz .LBB0_5\n' + pad12 + '.LBB0_5:\n\tv_add_f32_e32 v38, 1.0, v38\n'
    f = _audit(tmp_path, skip)
    assert any('(A)' in x and 'forward branch' in x for x in f), f
    assert not _audit(tmp_path, mf + '\ts_nop 11\n\ts_cbranch_vccnz .LBB0_5\n' + pad12 + '.LBB0_5:\n\tv_add_f32_e32 v38, 1.0, v38\n')
    # (B) across a taken forward branch
    vb = '\tv_fma_f32 v33, v1, v2, v3\n\ts_cbranch_scc1 .LBB0_6\n\ts_nop 3\n.LBB0_6:\n' + mf
    assert any('(B)' in x and 'forward branch' in x for x in _audit(tmp_path, vb))
    # (S) the metadata
    sys.path.insert(0, os.path.join(ROOT, 'imgcomp_cvpr_amd', 'csrc'))
    import isa_audit
    meta = ('amdhsa.kernels:\n  - .args: []\n    .name:           _Z1kv\n    .private_segment_fixed_size: 24\n    .sgpr_count:     10\n'
            '    .vgpr_count:     256\n    .vgpr_spill_count: 5\n')
    p = tmp_path / 'm.s'
    p.write_text(_AUDIT_HEAD + mf + _AUDIT_TAIL + meta)
    assert isa_audit.scratch_of(str(p)) == {'_Z1kv': (24, 5)}
    script = os.path.join(ROOT, 'imgcomp_cvpr_amd', 'csrc', 'isa_audit.py')
    r = subprocess.run([sys.executable, script, str(p)], stdout=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 1 and '(S) 24 bytes of scratch, 5 register(s) spilled' in r.stdout
    r = subprocess.run([sys.executable, script, str(p), '--allow-scratch', '_Z1k'], stdout=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0 and ' 0 finding(s)' in r.stdout


def test_bench_training_flop_count_matches_the_survey_figures():
    """bench.py's executed-FLOP count of a training step (the `roofline` of extra.train / --mode train) from the layer table: in direct
    form it is SURVEY 8(d)'s per-pixel figures x 3 passes (h1 has no data gradient), and the Winograd forms scale the 64 3x3 layers only"""
    import types
    sys.path.insert(0, ROOT)
    import bench
    from imgcomp_cvpr_amd import weights as W
    g = types.SimpleNamespace(C=32, B=5, L=6, k=24, heatmap=True, _w3_f4=(False, False))
    N, H, Wd = 2, 64, 96
    px, sym = N * H * Wd, N * 32 * (H // 8) * (Wd // 8)
    ex, direct = bench.train_step_flops(W, g, N, H, Wd)
    want = 3 * (bench.FLOP_PER_PX_ENC + bench.FLOP_PER_PX_DEC) * px - 2400.0 * px + 3 * bench.FLOP_PER_SYMBOL_PC * sym
    assert abs(direct - want) < 1e-6 * want, (direct, want)
    c3 = 2 * 589824.0 * px                                         # the 64 3x3 layers, one pass, direct form
    rest = direct - 3 * c3 - 3 * bench.FLOP_PER_SYMBOL_PC * sym
    assert abs(ex - (rest + c3 * 3 * 16.0 / 36.0 + 3 * bench.FLOP_PER_SYMBOL_PC_LIVE * sym)) < 1e-6 * ex
    g._w3_f4 = (True, True)
    ex4, _ = bench.train_step_flops(W, g, N, H, Wd)
    assert abs(ex4 - (rest + c3 * (0.25 + 0.25 + 16.0 / 36.0) + 3 * bench.FLOP_PER_SYMBOL_PC_LIVE * sym)) < 1e-6 * ex4


def test_profile_evidence_is_self_consistent():
    """the committed evidence files of the newest round agree with themselves (a digest whose sums and step count described
    different windows): profiles/rNN_counters.json -- concurrency = kernel time / span of the SAME window, kernel time per step = the
    window's sum / the steps it covers, and the 3x3 kernel's launches x duration fit inside it; profiles/rNN_inflight_stamps.json --
    frac = executed FLOPs / (us per launch) / peak for both scenarios, the events figure and the stamps figure of one run related by the
    ratio the file states."""
    import glob
    import json
    cs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_counters.json')))
    c = json.load(open(cs[-1]))
    if 'window' not in c:
        pytest.skip('digest of an older round')
    w = c['window']
    assert abs(c['concurrency'] - w['sum_of_kernel_durations_us'] / w['span_of_the_window_us']) < 2e-3
    assert abs(w['kernel_us_per_step'] - w['sum_of_kernel_durations_us'] / w['steps_in_window']) < 1.0
    assert w['union_busy_us'] <= w['span_of_the_window_us'] + 0.5 and w['sum_of_kernel_durations_us'] >= w['union_busy_us']
    k3 = c['kernels'][c['plan_3x3']]
    assert k3['launches_per_step'] * k3['avg_us_in_flight'] < w['kernel_us_per_step']
    assert abs(w['launches_3x3_in_window'] / w['steps_in_window'] - k3['launches_per_step']) < 0.5
    ss = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_inflight_stamps.json')))
    if not ss:
        pytest.skip('no stamps file')
    s = json.load(open(ss[-1]))
    for scen in ('step', 'stacks'):
        d = s[scen]['dominant']
        frac = d['executed_flop_per_launch'] / (d['us_per_launch_under_concurrency'] * 1e-6) / 1e12 / s['peak_tflops']
        assert abs(frac - d['frac']) < 2e-3, scen
        k = s[scen]['kernels'][d['kernel']]
        assert abs(k['union_busy_us'] / k['launches'] - d['us_per_launch_under_concurrency']) < 0.01
        assert 0 < s[scen]['chip_mfma_issue_share'] < 1 and 1.0 < s[scen]['shader_clock_ghz_under_load'] < 2.5
    st = s['stacks']
    assert abs(st['hip_events_us_per_launch'] / st['dominant']['us_per_launch_under_concurrency'] - st['events_over_stamps']) < 2e-3
    assert abs(st['dominant']['executed_flop_per_launch'] / (st['hip_events_us_per_launch'] * 1e-6) / 1e12 / s['peak_tflops'] - st['hip_events_frac']) < 2e-3
