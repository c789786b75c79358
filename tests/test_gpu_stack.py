"""-m gpu: the residual stack (autoencoder.py:224-234 / :252-262) as ONE persistent launch (csrc/conv3x3_wino_stack.hip) against
the same stack as 6B + 2 per-layer launches: the two must agree BIT FOR BIT (same operations per output), on every shape class
the launcher accepts (NB = 1, 2, 3 segments per job; ragged heights and widths; batches), repeatedly (the hand-off between
work-groups is a race if it is wrong), and the time-out word of the sync area must stay 0."""
import ctypes

import numpy as np
import pytest
import torch

from tests.util import assert_close, dev

pytestmark = pytest.mark.gpu


def _needs_tuning_build():
    """the persistent stack kernel was measured level with per-layer launches and is compiled into tuning builds
    only (make -C imgcomp_cvpr_amd/csrc TUNING=1); in the shipped library IC_CONV3_STACK_KERNEL is ignored"""
    from imgcomp_cvpr_amd import _lib
    if not _lib.lib.ic_build_has_tuning_forms():
        pytest.skip('tuning-build form (make TUNING=1)')


def _stack_inputs(cuda, B, seed):
    from imgcomp_cvpr_amd import _lib
    lib = _lib.lib
    st = _lib.current_stream(cuda)
    g = torch.Generator(device='cpu').manual_seed(seed)
    tens, raw = [], []
    for i in range(6 * B + 2):
        w = (torch.randn((3, 3, 128, 128), generator=g) * 0.03).to(cuda)
        wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=cuda)
        _lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
        sc = (torch.rand(128, generator=g) * 0.6 + 0.5).to(cuda)
        sh = (torch.randn(128, generator=g) * 0.1).to(cuda)
        tens += [wp, sc, sh]
        raw.append((w, sc, sh))
    torch.cuda.synchronize()
    return tens, raw


def _run_stack(cuda, x, tens, B, flags):
    from imgcomp_cvpr_amd import _lib
    lib = _lib.lib
    N, _, H, W = x.shape
    need = lib.ic_ae_res_stack_workspace_bytes(N, H, W)
    ws = torch.zeros(need, dtype=torch.uint8, device=cuda)
    y = torch.empty_like(x)
    tab = _lib.ptr_table(tens)
    _lib.check(lib.ic_ae_res_stack_f32(_lib.ptr(x), tab, B, _lib.ptr(y), N, H, W, _lib.ptr(ws), need, flags, _lib.current_stream(cuda)))
    torch.cuda.synchronize()
    pos = lib.ic_ae_res_stack_sync_pos_bytes(N, H, W)
    timeout_word = int(ws[pos:pos + 4].view(torch.int32).item())
    return y, timeout_word


SHAPES = [(1, 128, 192), (1, 64, 64), (1, 128, 128), (2, 64, 96), (1, 100, 190), (1, 66, 70), (1, 30, 46), (3, 40, 64), (1, 16, 16),
          (4, 32, 32), (1, 6, 2)]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_persistent_stack_is_bit_identical_to_per_layer_launches(cuda, shape):
    _needs_tuning_build()
    from imgcomp_cvpr_amd import _lib
    B = 5
    tens, _ = _stack_inputs(cuda, B, seed=shape[1] * 1000 + shape[2])
    N, H, W = shape
    x = torch.relu(torch.randn((N, 128, H, W), generator=torch.Generator().manual_seed(7))).to(cuda)
    # the persistent kernel runs the F(2x2) NB-segment jobs: its per-layer twin is the F(2x2) plan (since round 5 the automatic plan takes
    # F(4x4) from 160 work-groups on -- a Kodak map -- and F(4x4) rounds differently: 1.3e-3 absolute on this input)
    ref, tw0 = _run_stack(cuda, x, tens, B, _lib.CONV3_NO_WINO4)
    assert tw0 == 0
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    for rep in range(4):
        got, tw = _run_stack(cuda, x, tens, B, _lib.CONV3_STACK_KERNEL | _lib.CONV3_NO_WINO4)
        assert tw == 0, 'a hand-off of the persistent stack kernel timed out at layer {}'.format(tw - 1)
        assert torch.equal(got, ref), 'persistent stack != per-layer launches (run {}): {} of {} values differ, max {}'.format(
            rep, int((got != ref).sum()), ref.numel(), float((got - ref).abs().max()))


def test_persistent_stack_takes_the_shapes_it_should(cuda):
    _needs_tuning_build()
    """the shapes above really run as one launch (plan: NB-segment jobs only, one resident round) -- and shapes that do not
    fit fall back to per-layer launches with the same results"""
    from imgcomp_cvpr_amd import _lib
    lib = _lib.lib
    pl = (ctypes.c_longlong * 5)()
    for (n, h, w), nb in (((1, 128, 192), 3), ((1, 64, 64), 1), ((1, 128, 128), 2)):
        _lib.check(lib.ic_wino3x3_c128_plan(n, h, w, 0, pl))
        assert list(pl)[:3] == [0, n * ((h + 3) // 4) * ((w + 31) // 32), nb]
    B = 1
    tens, _ = _stack_inputs(cuda, B, seed=5)
    x = torch.relu(torch.randn((1, 128, 136, 240), generator=torch.Generator().manual_seed(3))).to(cuda)     # 272 groups: two launches per layer
    a, tw = _run_stack(cuda, x, tens, B, _lib.CONV3_STACK_KERNEL)
    b, _ = _run_stack(cuda, x, tens, B, 0)
    assert tw == 0 and torch.equal(a, b)


def test_persistent_stack_against_the_oracle(cuda):
    """and it is the right function: fp64 restatement of the stack (conv + folded BN [+ ReLU] + skips) on a Kodak-class map"""
    import torch.nn.functional as F
    from imgcomp_cvpr_amd import _lib
    B = 2
    tens, raw = _stack_inputs(cuda, B, seed=11)
    x = torch.relu(torch.randn((1, 128, 32, 48), generator=torch.Generator().manual_seed(1))).to(cuda)
    got, tw = _run_stack(cuda, x, tens, B, _lib.CONV3_STACK_KERNEL)
    assert tw == 0

    def cba(t, i, relu):
        w, sc, sh = raw[i]
        y = F.conv2d(t, w.double().cpu().permute(3, 2, 0, 1), padding=1)
        y = y * sc.double().cpu()[None, :, None, None] + sh.double().cpu()[None, :, None, None]
        return torch.relu(y) if relu else y
    net = res0 = x.double().cpu()
    li = 0
    for b in range(B):
        res_b = net
        for i in range(3):
            t = cba(net, li, True)
            net = cba(t, li + 1, False) + net + (res_b if i == 2 else 0)
            li += 2
    t = cba(net, li, False)
    ref = cba(t, li + 1, False) + net + res0
    assert_close(got, ref, 'persistent residual stack vs fp64', 2e-5)
