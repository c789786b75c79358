"""Which part of the captured distortion graph is not ordered with the work behind its launch?  (round 4, training hazard)

Captures Distortions(...) [+ its gradient] the way training._GraphedDistortion does, then, with the host far ahead of the device,
replays it on fresh inputs and enqueues consumers of the static outputs right behind the launch (same stream).  Wrong results are
counted against an eager evaluation of the same inputs.

  python tools/distortion_graph_probe.py [variant ...]      variants: shipped global fwd_only mse no_thread
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from imgcomp_cvpr_amd import config_parser as cp, training


def build(cfg, shape, dev, mode='thread_local', with_grad=True, inline_backward=False):
    x = torch.zeros(shape, device=dev)
    xo = torch.ones(shape, device=dev)

    def run():
        v = xo.detach().requires_grad_(True)
        d = training.Distortions(cfg, x, v, is_training=True)
        outs = {'loss': d.d_loss_scaled.detach()}
        if with_grad:
            outs['grad'], = torch.autograd.grad(d.d_loss_scaled, v)
        return outs
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    cur.wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        outs = run()
    torch.cuda.synchronize(dev)
    return g, x, xo, outs, run


class _CloneInBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, box):
        ctx.box = box
        return hook * 1.0

    @staticmethod
    def backward(ctx, g):
        ctx.box['got'] = {k: v.clone() for k, v in ctx.box['outs'].items()}      # runs in the autograd engine's device thread
        return g, None


def consume_torch(outs):
    return {k: v.clone() for k, v in outs.items()}


def consume_engine_thread(outs):
    hook = torch.zeros((), device=outs['loss'].device, requires_grad=True)
    box = {'outs': outs}
    _CloneInBackward.apply(hook, box).backward()
    return box['got']


def consume_lib(outs):
    from imgcomp_cvpr_amd import _lib
    from imgcomp_cvpr_amd._lib import lib, check, ptr
    o = {'loss': outs['loss'].clone()}
    if 'grad' in outs:
        gr = outs['grad']
        N, C, H, W = gr.shape
        dev = gr.device
        y = torch.empty_like(gr)
        ones, zeros = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        check(lib.ic_bn_apply_f32(ptr(gr), ptr(ones), ptr(zeros), None, None, ptr(y), N, C, H * W, 0, _lib.current_stream(dev)))
        o['grad'] = y
        o['_keep'] = (ones, zeros)
    return o


CONSUMERS = {'torch': consume_torch, 'engine': consume_engine_thread, 'lib': consume_lib}


def probe(name, cfg, dev, reps=12, consumer='torch', on_side=False, **kw):
    consume = CONSUMERS[consumer]
    shape = (32, 3, 128, 128)
    g, x, xo, outs, run = build(cfg, shape, dev, **kw)
    big = torch.randn(4096, 4096, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    ins, got = [], []
    r_side = [None]
    for r in range(reps):
        a = torch.rand(shape, device=dev, generator=gen) * 255
        b = (a + 20 * torch.randn(shape, device=dev, generator=gen)).clamp(0, 255)
        ins.append((a, b))
    torch.cuda.synchronize(dev)
    for a, b in ins:
        for _ in range(48):
            big @ big                                  # host ahead of the device
        if on_side:
            main = torch.cuda.current_stream(dev)
            if r_side[0] is None:
                r_side[0] = torch.cuda.Stream(device=dev, priority=-1)
            sd = r_side[0]
            sd.wait_stream(main)
            with torch.cuda.stream(sd):
                x.copy_(a)
                xo.copy_(b)
                g.replay()
            main.wait_stream(sd)
        else:
            x.copy_(a)
            xo.copy_(b)
            g.replay()
        got.append(consume(outs))
        float(got[-1]['loss'])                          # the step's read-back: at most one replay in flight
    torch.cuda.synchronize(dev)
    bad = {k: 0 for k in outs}
    first = {}
    for i, ((a, b), o) in enumerate(zip(ins, got)):
        x.copy_(a)
        xo.copy_(b)
        ref = run()
        torch.cuda.synchronize(dev)
        for k in ref:
            w = int(not torch.equal(o[k], ref[k]))
            bad[k] += w
            if w:
                first.setdefault(k, []).append(i)
    print('{:12s} wrong of {}: {}  failing replays: {}'.format(name, reps, bad, first), flush=True)


def main():
    dev = torch.device('cuda', 0)
    cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
    which = sys.argv[1:] or ['shipped', 'global', 'fwd_only', 'mse']
    for w in which:
        if w == 'shipped':
            probe(w, cfg, dev)
        elif w.startswith('reps'):
            probe(w, cfg, dev, reps=int(w[4:]))
        elif w.startswith('side'):
            probe(w, cfg, dev, reps=int(w[4:]), on_side=True)
        elif w in ('engine', 'lib'):
            probe(w, cfg, dev, consumer=w)
        elif w == 'global':
            probe(w, cfg, dev, mode='global')
        elif w == 'fwd_only':
            probe(w, cfg, dev, with_grad=False)
        elif w == 'mse':
            c2 = cfg.copy() if hasattr(cfg, 'copy') else cfg
            c2.distortion_to_minimize = 'mse'
            probe(w, c2, dev)
            c2.distortion_to_minimize = 'ms_ssim'


if __name__ == '__main__':
    main()
