"""Inference driver -- the val.py entry point of the reference (code/val.py) on the MI355X hot path.

    python -m imgcomp_cvpr_amd.val LOG_DIR_ROOT JOB_IDS IMAGES [--save_ours] [--how_many N] [--real_bpp]
                                   [--weights synthetic|FILE.npz|CKPT] [--restore_itr N] [--reset]

Per image (batch 1, as val.py:157-158): encode -> decode(qhard) -> bitcost(qbar, symbols, pad=centers[0]) -> bpp;
output truncated to uint8; MS-SSIM and PSNR in float64 (metrics.py: on the device by default, --host_metrics for the
reference's numpy path -- same numbers, 0.5 s per image slower); `measures.csv`
(`img_name,bpp,ms-ssim,psnr`) in `LOG_DIR_ROOT/{log_date} {dataset}` like val_files.py:62-77; with --real_bpp the
arithmetic-coded size is measured too and |bpp_theory - bpp_loss| < 1e-3 is asserted (val.py:163-174).

Job directories are named as the reference names them, `MMDD_HHMM ae_configs@cvpr@low pc_configs@cvpr@res_shallow`
(logdir_helpers.py:34-56); the configs are recovered from the name (logdir_helpers.py:130-151) and resolved against
$CONFIG_BASE_AE / $CONFIG_BASE_PC or this package's config trees.  Weights: the newest `JOB_DIR/ckpts/ckpt-<itr>`
(or the newest with iteration <= --restore_itr, saver.py:102-127) read straight from the TF-1 tensor bundle
(`.index` + `.data-00000-of-00001`, tf_checkpoint.py), or `--weights` = 'synthetic' | an .npz keyed by variable name |
a checkpoint prefix / ckpts directory / log dir.

Under torch.distributed.run (one process per GPU) the image list is sharded round-robin over the ranks; rank 0
writes the merged measures.
"""
import argparse
import glob
import os
import re
import sys
from collections import defaultdict, namedtuple
from datetime import datetime
from os import path

import numpy as np
import torch

from . import _lib, autoencoder, bits, bpp_helpers, config_parser, metrics, probclass, sharding, streams
from . import weights as _weights

OutputFlags = namedtuple('OutputFlags', ['save_ours', 'ckpt_step', 'real_bpp'])
_LOG_DATE_FORMAT = '%m%d_%H%M'
_MEASURES_FILE_NAME = 'measures.csv'


# ---- images (reference code/images_iterator.py, code/val_images.py) -------------------------------------------

def add_padding(im, pad):
    """HWC uint8 -> zero-padded (centred, extra pixel at the far side) to multiples of `pad`
    (images_iterator.py:39-59).  Returns (padded, undo_fn)."""
    h, w, chan = im.shape
    if chan == 4:
        return add_padding(im[:, :, :3], pad)
    if h % pad == 0 and w % pad == 0:
        return im, lambda x: x
    hp, wp = (-h) % pad, (-w) % pad
    t, l = hp // 2, wp // 2
    padded = np.pad(im, [[t, hp - t], [l, wp - l], [0, 0]], mode='constant')
    return padded, lambda x: x[t:t + h, l:l + w, :]


def get_image_paths(images):
    """directory with PNGs, or a glob -> (sorted paths, dataset name)  (val_images.py:12-46)."""
    if '*' not in images:
        images = path.join(images, '*.png')
    paths = sorted(glob.glob(images))
    if not paths:
        raise ValueError('Not matching any files: {}'.format(images))
    for comp in reversed(images.strip(path.sep).split(path.sep)):
        if '*' not in comp:
            return paths, comp
    raise ValueError('No component without *: {}'.format(images))


def load_image_chw(p, pad):
    """PNG -> padded CHW uint8 (images_iterator.py:28-59).
    (Round 6, measured and not kept.  Decoding into page-locked buffers so that the upload is an asynchronous DMA: an upload of a Kodak
    image from pageable memory on the fetcher's copy stream holds the host for 42 us against 11 us -- nothing to win --, decoder threads
    write page-locked memory slower than pageable (974 against 1109 images/s at 16 threads), and torch's CPU copy_ into a page-locked
    staging batch stalled the step for 17 - 30 ms.  Decoder PROCESSES with shared-memory slots instead of threads: 1700 images/s on their
    own, but 0.5 - 1.5 s to start -- 127 - 206 images/s on a 240-image set against 486 with 8 threads, which already feed the device
    path's 524 images/s.  DESIGN.md section 4.)"""
    from PIL import Image
    im = np.asarray(Image.open(p).convert('RGB'), dtype=np.uint8)
    im, _ = add_padding(im, pad)
    return np.ascontiguousarray(np.transpose(im, (2, 0, 1)))


# ---- log-dir conventions (reference code/logdir_helpers.py) -----------------------------------------------------

def is_log_date(s):
    try:
        datetime.strptime(s, _LOG_DATE_FORMAT)
        return True
    except ValueError:
        return False


def log_date_from_log_dir(log_dir):
    d = path.basename(log_dir.rstrip(path.sep)).split(' ')[0]
    if not is_log_date(d):
        raise ValueError('Invalid log dir: {}'.format(log_dir))
    return d


def config_paths_from_log_dir(log_dir, base_dirs):
    """`{date} {ae cfg with / -> @} {pc cfg}` -> real config paths; a `*` in a component is a one-character
    wildcard (the reference replaces `-`), matched against files of the same length (logdir_helpers.py:130-151)."""
    comps = path.basename(log_dir.rstrip(path.sep)).split(' ')
    assert is_log_date(comps[0]), 'Invalid log_dir: {}'.format(log_dir)
    comps = [c for c in comps[1:] if 'RESTORE@' not in c]
    assert len(comps) <= len(base_dirs)
    out = []
    for base, c in zip(base_dirs, comps):
        rel = c.replace('@', path.sep)
        # the component may or may not repeat the name of the base dir (ae_configs/cvpr/low vs cvpr/low)
        cands = [path.join(base, rel), path.join(path.dirname(base.rstrip(path.sep)), rel)]
        hits = []
        for cand in cands:
            hits = [g for g in glob.glob(cand) if len(g) == len(cand) and path.isfile(g)]
            if len(hits) == 1:
                break
        if len(hits) != 1:
            raise ValueError('Cannot find config on disk: {} (matches: {})'.format(cands, hits))
        out.append(hits[0])
    return out


def iter_job_dirs(log_dir_root, job_ids_str):
    for job_id in job_ids_str.strip().replace(';', ',').split(','):
        m = [d for d in glob.glob(path.join(log_dir_root, job_id + '*')) if path.isdir(d) and
             len(path.basename(d).split(' ')) >= 3]
        if len(m) != 1:
            print('*** ERR: {} matches for job {}'.format(len(m), job_id))
            continue
        yield m[0]


class MeasuresWriter(object):
    def __init__(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        self.fout = open(path.join(out_dir, _MEASURES_FILE_NAME), 'w')
        self.fout.write('img_name,bpp,ms-ssim,psnr\n')

    def append(self, img_name, otp):
        self.fout.write('{},{},{},{}\n'.format(img_name, otp['bpp'], otp['ms-ssim'], otp['psnr']))

    def close(self):
        self.fout.close()


class ValuesAggregator(object):
    def __init__(self, *tags):
        self._vals = defaultdict(list)
        self.tags = tags

    def update(self, d):
        for t in self.tags:
            assert not np.isnan(d[t]), 'nan encountered in {}'.format(d)
            self._vals[t].append(d[t])

    def averages(self):
        return {t: float(np.mean(v)) for t, v in self._vals.items()}

    def averages_str(self):
        a = self.averages()
        return ', '.join('{}: {:.3f}'.format(t, a[t]) for t in self.tags)


# ---- the per-image fetch (val.py:81-94) -------------------------------------------------------------------------

class Fetcher(object):
    """builds the networks once, then maps one padded uint8 CHW image to its measures."""

    def __init__(self, ae_config, pc_config, weights, device, host_metrics=False, plan_flags=0, share_with=None):
        self.device = torch.device(device)
        self.host_metrics = host_metrics        # True: MS-SSIM / PSNR in numpy on the host, as the reference does
        if share_with is not None:
            # another stream of the same evaluation: the first fetcher's device weights and packed filters, own workspaces
            self.ae, self.pc = share_with.ae.sharing_weights(), share_with.pc.sharing_weights()
        else:
            self.ae = autoencoder.get_network_cls(ae_config)(ae_config).load_weights(weights, self.device)
            self.pc = probclass.get_network_cls(pc_config)(pc_config, num_centers=ae_config.num_centers).load_weights(
                weights, self.device)
        self.ae.plan_flags = int(plan_flags)          # e.g. _lib.CONV3_IN_FLIGHT(n): validate() keeps n fetchers busy at once
        self.pc_config = pc_config
        self._bpp_fetcher = None
        self._streams = streams.BranchStreams(self.device)
        self._copy_stream = None
        self._metrics_ws = metrics.ValMetricsWorkspace()

    def __call__(self, img_chw_uint8, want_symbols=False, want_image=False):
        return self.collect(self.enqueue(img_chw_uint8, want_symbols, want_image))

    def enqueue(self, img_chw_uint8, want_symbols=False, want_image=False):
        """launch the whole path for one image on this fetcher's stream WITHOUT waiting for it: validate() keeps several
        fetchers busy at once (the images are independent), collect() turns the device results into Python values.
        A LIST of same-shape images is evaluated as one batch (one pass of every kernel over all of them; the inference path has no
        cross-image term -- BatchNorm is folded --, the measures are taken per image): collect() then returns a list."""
        if isinstance(img_chw_uint8, (list, tuple)):
            x_uint8 = torch.as_tensor(np.stack(img_chw_uint8))
            batched = True
        else:
            x_uint8 = torch.as_tensor(img_chw_uint8)[None]
            batched = False
        outer, main = torch.cuda.current_stream(self.device), self._streams.main
        main.wait_stream(outer)
        x_dev = None
        if self.device.type == 'cuda':
            # the image goes up on a stream of its own: a copy from pageable memory makes the host wait until its stream has reached
            # it, which on the compute stream means waiting for the image before this one (2.1 of the 2.6 ms an enqueue took,
            # tools/val_host_profile.py, tools/h2d_probe.py)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._copy_stream):
                x_dev = x_uint8.to(self.device)
            main.wait_stream(self._copy_stream)
        with torch.cuda.stream(main):
            pending = self._measure(x_uint8, want_symbols, want_image, x_dev)
        pending['batched'] = batched
        return pending

    def collect(self, pending):
        main = self._streams.main
        if 'device7xN' in pending['scalars']:
            with torch.cuda.stream(main):
                rows = pending['scalars']['device7xN'].tolist()            # ONE transfer for the whole batch
                arrays = {k: v.cpu().numpy() for k, v in pending['arrays'].items()}
            torch.cuda.current_stream(self.device).wait_stream(main)
            outs = []
            for i, v in enumerate(rows):
                otp = {'bpp': float(np.float32(v[6])), 'ms-ssim': metrics.msssim_from_scale_values(v[:5]), 'psnr': metrics.psnr_from_mse(v[5])}
                for k, arr in arrays.items():
                    otp[k] = arr[i:i + 1]
                outs.append(otp)
            return outs if pending.get('batched') else outs[0]
        with torch.cuda.stream(main):
            sc = pending['scalars']
            if 'device7' in sc:
                # [5 MS-SSIM scale values, mean squared error, bpp] in ONE device tensor: one transfer, one wait (for this stream only)
                v = sc['device7'].tolist()
                otp = {'bpp': float(np.float32(v[6])), 'ms-ssim': metrics.msssim_from_scale_values(v[:5]), 'psnr': metrics.psnr_from_mse(v[5])}
            else:
                otp = {'bpp': float(sc['bpp'])}                                   # .item(): waits for this stream only
                otp['ms-ssim'] = metrics.msssim_from_scale_values(sc['ms-ssim'].tolist()) if torch.is_tensor(sc['ms-ssim']) else float(sc['ms-ssim'])
                otp['psnr'] = metrics.psnr_from_mse(float(sc['psnr'])) if torch.is_tensor(sc['psnr']) else float(sc['psnr'])
            for k, v in pending['arrays'].items():
                otp[k] = v.cpu().numpy()
        torch.cuda.current_stream(self.device).wait_stream(main)
        return otp

    def _measure(self, x_uint8, want_symbols, want_image, x_uint8_dev=None):
        if x_uint8_dev is None:
            x_uint8_dev = x_uint8.to(self.device, non_blocking=True)
        else:
            x_uint8_dev.record_stream(torch.cuda.current_stream(self.device))        # allocated on the copy stream, used on this one
        if x_uint8.shape[0] > 1 and not self.host_metrics:
            return self._measure_batch(x_uint8_dev, want_symbols, want_image)
        x = x_uint8_dev.float()
        enc = self.ae.encode(x, is_training=False)
        # decoder and context model are independent consumers of the encoder output: with a CU-range arrangement (streams.py) the
        # context model runs on a second stream beside the decoder; in the default serial arrangement `side` is this stream
        cur = torch.cuda.current_stream(self.device)
        side = self._streams.context_model_stream(x.shape[0], x.shape[2], x.shape[3], int(self.ae.config.num_chan_bn))
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            bc = self.pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=self.pc.auto_pad_value(self.ae))
            bpp = bits.bitcost_to_bpp(bc, x)
        x_out = self.ae.decode(enc.qhard, is_training=False, plan_flags=self._streams.decode_flags(side))
        cur.wait_stream(side)
        x_out_uint8_dev = x_out.to(torch.uint8)                    # tf.cast truncates (val.py:91)
        arrays = {}
        if self.host_metrics:
            x_out_uint8 = x_out_uint8_dev.cpu().numpy()
            # python floats: collect() tells a finished host value from a device tensor by type (np.float32 is not a float)
            ms = float(metrics.msssim_nchw_uint8(x_uint8.numpy(), x_out_uint8))
            ps = float(metrics.psnr_uint8(x_uint8.numpy(), x_out_uint8))
        else:
            # the same float64 computation on the device (csrc/val_metrics.hip): 0.5 s of numpy per Kodak image would dwarf the 2.4 ms GPU path
            # (device tensors, nothing waited for here: collect() finishes them on the host)
            ms, ps = metrics.val_metrics_device(x_uint8_dev, x_out_uint8_dev, self._metrics_ws)
        if want_symbols:
            arrays['sym'] = enc.symbols
        if want_image:
            arrays['img_out'] = x_out_uint8_dev
        if not self.host_metrics and torch.is_tensor(ms) and ms.is_cuda:
            return {'scalars': {'device7': torch.cat([ms, ps.reshape(1), bpp.reshape(1).to(torch.float64)])}, 'arrays': arrays}
        return {'scalars': {'bpp': bpp, 'ms-ssim': ms, 'psnr': ps}, 'arrays': arrays}

    def _measure_batch(self, x_uint8_dev, want_symbols, want_image):
        """B same-shape images through ONE pass of the codec; bpp, MS-SSIM terms and squared error per image -> [B, 7] on the device"""
        x = x_uint8_dev.float()
        enc = self.ae.encode(x, is_training=False)
        bc = self.pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=self.pc.auto_pad_value(self.ae))
        x_out = self.ae.decode(enc.qhard, is_training=False)
        x_out_uint8_dev = x_out.to(torch.uint8)
        # per image, by the single-image kernels (bit-identical to one image per step), one library call each for the whole batch
        bpp = bits.bitcost_to_bpp_per_image(bc, x)
        m6 = metrics.val_metrics_device_per_image(x_uint8_dev, x_out_uint8_dev, self._metrics_ws)
        rows = torch.cat([m6, bpp.to(torch.float64)[:, None]], dim=1)
        arrays = {}
        if want_symbols:
            arrays['sym'] = enc.symbols
        if want_image:
            arrays['img_out'] = x_out_uint8_dev
        return {'scalars': {'device7xN': rows}, 'arrays': arrays}

    def real_bpp(self, symbols, num_pixels):
        if self._bpp_fetcher is None:
            pred = probclass.PredictionNetwork(self.pc, self.pc_config, self.ae.get_centers_variable())
            checker = probclass.ProbclassNetworkTesting(self.pc, self.ae)
            self._bpp_fetcher = bpp_helpers.BppFetcher(pred, checker)
        return self._bpp_fetcher.get_bpp(symbols, num_pixels)


def _decoded_ahead(image_paths, indices, pad, threads):
    """(index, CHW uint8 image) in order, the PNGs decoded by a few host threads a bounded distance ahead of the consumer: a
    Kodak-sized PNG takes the host ~8 ms to decode and the device ~1.5 ms to code (PIL's decoder and zlib release the GIL)."""
    if threads <= 1 or len(indices) <= 1:
        for idx in indices:
            yield idx, load_image_chw(image_paths[idx], pad)
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as pool:
        ahead, it = deque(), iter(indices)
        for idx in it:
            ahead.append((idx, pool.submit(load_image_chw, image_paths[idx], pad)))
            if len(ahead) >= 2 * threads:
                break
        while ahead:
            idx, fut = ahead.popleft()
            nxt = next(it, None)
            if nxt is not None:
                ahead.append((nxt, pool.submit(load_image_chw, image_paths[nxt], pad)))
            yield idx, fut.result()


def batch_size_for_shape(h, w, limit=8):
    """how many images of one (padded) shape go into one step: as many as keep the step near a Kodak image's work -- 8 of 256 x 256,
    2 of 384 x 512, 1 from 512 x 768 on (there 4 images in flight already fill the chip: bench.py, batch 2 x 2 in flight = batch 1 x 4)"""
    return int(max(1, min(int(limit), (512 * 1024) // max(1, int(h) * int(w)))))


def _same_shape_batches(decoded, limit):
    """(index, image) in order -> lists [(index, image), ...] of CONSECUTIVE images of one shape, at most batch_size_for_shape of them:
    the order of the images is the order of the batches' members"""
    cur = []
    for idx, img in decoded:
        if cur and (img.shape != cur[0][1].shape or len(cur) >= batch_size_for_shape(img.shape[1], img.shape[2], limit)):
            yield cur
            cur = []
        cur.append((idx, img))
    if cur:
        yield cur


def validate(ae_config, pc_config, weights, image_paths, out_dir, flags, device='cuda', verbose=True, host_metrics=False,
             in_flight=4, loader_threads=8, batch_same_shape=8):
    """-> dict of averages; writes out_dir/measures.csv (rank 0).
    in_flight: images of this rank processed concurrently, each by its own Fetcher (networks, workspace, stream): the images
    are independent (the reference runs one per sess.run, val.py:157-158), and the launches of one fill the kernel-boundary
    bubbles of the others, and a 3x3 launch need not fill the chip alone (bench.py: 158 -> 183 Mpix/s on Kodak-sized images with 4).  --real_bpp codes one image at a time.
    batch_same_shape: consecutive small images of one shape are evaluated as one batch (batch_size_for_shape: 8 of 256 x 256) -- one
    step of a fetcher then carries the batch; rows of measures.csv and their order are those of the one-image-per-step loop (values
    to the fp32 agreement of the plans the library picks for the two batch sizes: tests/test_gpu_network.py)."""
    from collections import deque
    rank, world = sharding.rank_and_world()
    n_f = 1 if (flags.real_bpp or host_metrics) else max(1, int(in_flight))
    fetcher = Fetcher(ae_config, pc_config, weights, device, host_metrics=host_metrics, plan_flags=_lib.CONV3_IN_FLIGHT(n_f) if n_f > 1 else 0)
    fetchers = [fetcher] + [Fetcher(ae_config, pc_config, weights, device, host_metrics=host_metrics,
                                    plan_flags=_lib.CONV3_IN_FLIGHT(n_f), share_with=fetcher) for _ in range(n_f - 1)]
    pad = fetcher.ae.get_subsampling_factor()
    local = []
    pending = deque()

    def finish():
        members, f, h = pending.popleft()
        outs = f.collect(h)
        for (idx, img), otp in zip(members, outs if isinstance(outs, list) else [outs]):
            finish_one(idx, image_paths[idx], img, f, otp)

    def finish_one(idx, p, img, f, otp):
        if flags.real_bpp:
            bpp_real, bpp_theory = f.real_bpp(otp.pop('sym'), bpp_helpers.num_pixels_in_image(img))
            otp['bpp_real'], otp['bpp_theory'] = bpp_real, bpp_theory
            if verbose:
                print('BPP: Real         {:.5f}\n     Theoretical: {:.5f} [{:5.1f}% of real]\n'
                      '     Loss:        {:.5f} [{:5.1f}% of real]'.format(
                          bpp_real, bpp_theory, bpp_theory / bpp_real * 100, otp['bpp'], otp['bpp'] / bpp_theory * 100))
            assert abs(bpp_theory - otp['bpp']) < 1e-3, 'Expected bpp_theory to match loss! Got {} and {}'.format(
                bpp_theory, otp['bpp'])
        if flags.save_ours:
            save_img(path.basename(p), otp.pop('img_out'), out_dir)
        local.append((idx, (path.basename(p), otp)))

    limit = 1 if (flags.real_bpp or host_metrics) else max(1, int(batch_same_shape))
    decoded = _decoded_ahead(image_paths, list(sharding.shard_indices(len(image_paths), rank, world)), pad, loader_threads)
    for k, members in enumerate(_same_shape_batches(decoded, limit)):
        f = fetchers[k % n_f]
        if len(pending) == n_f:
            finish()                                  # the oldest step ran on this fetcher: its buffers are free again
        imgs = [img for _, img in members]
        pending.append((members, f, f.enqueue(imgs if len(imgs) > 1 else imgs[0], want_symbols=flags.real_bpp, want_image=flags.save_ours)))
    while pending:
        finish()
    merged = sharding.gather_in_order(local, len(image_paths))
    agg = ValuesAggregator('bpp', 'ms-ssim', 'psnr')
    if rank == 0:
        w = MeasuresWriter(out_dir)
        for i, (name, otp) in enumerate(merged):
            w.append(name, otp)
            agg.update(otp)
            if verbose:
                print('{: 10d} {} | Mean: {}'.format(i, name, agg.averages_str()))
        w.close()
    else:
        for _, otp in merged:
            agg.update(otp)
    return agg.averages()


def save_img(img_name, img_out, out_dir):
    from PIL import Image
    assert img_out.ndim == 4 and img_out.shape[1] == 3, 'Expected NCHW, got {}'.format(img_out.shape)
    d = path.join(out_dir, 'imgs')
    os.makedirs(d, exist_ok=True)
    Image.fromarray(np.transpose(img_out[0], (1, 2, 0))).save(path.join(d, img_name))


def load_weights_for_job(job_dir, weights_arg, ae_config, pc_config, restore_itr=-1):
    if weights_arg == 'synthetic':
        return _weights.synthetic_weights(ae_config, pc_config)
    from . import tf_checkpoint
    p = weights_arg or job_dir
    if not path.exists(p) and not path.exists(p + '.index'):
        raise FileNotFoundError('{} not found (expected a TF-1 checkpoint prefix, a ckpts/ or log dir, or an .npz); '
                                'or pass --weights synthetic'.format(p))
    return tf_checkpoint.load_weights(p, itr=restore_itr)


def default_loader_threads(world=None):
    """host threads decoding PNGs ahead of the device: min(16, cores / ranks of this node), at least 1.  A Kodak-sized PNG takes one
    core 8 ms to decode and the device path 1.5 ms.  Measured on the MI355X host (round 6, tools/val_throughput.py, 96 images, 4 in
    flight): 1 / 8 / 16 / 32 threads -> 109 / 357 / 361 / 321 images/s -- beyond 16 the threads only contend for the interpreter
    lock with the loop that feeds the device (VERDICT r5 asked for min(32, ...): 32 is slower).  With the entry point's 8 hardware
    queues: 8 / 16 threads -> 486 / 484 images/s against 524 for images that are already decoded."""
    if world is None:
        world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')) or 1)
    return int(max(1, min(16, (os.cpu_count() or 1) // max(1, world))))


def main(argv=None):
    from . import ask_for_hardware_queues
    ask_for_hardware_queues(8)          # several images in flight, one stream each: before the HIP runtime starts (package __init__)
    p = argparse.ArgumentParser()
    p.add_argument('log_dir_root', help='Path to dir containing log_dirs.')
    p.add_argument('job_ids', help='Comma separated list of job_ids.')
    p.add_argument('images')
    p.add_argument('--save_ours', '-o', action='store_const', const=True)
    p.add_argument('--how_many', type=int, help='Number of images to output')
    p.add_argument('--reset', action='store_const', const=True, help='Remove previous output')
    p.add_argument('--host_metrics', action='store_const', const=True,
                   help='MS-SSIM / PSNR in numpy on the host as the reference does (default: same float64 math on the device)')
    p.add_argument('--real_bpp', action='store_const', const=True,
                   help='If given, calculate real bpp using arithmetic encoding.')
    p.add_argument('--weights', help="'synthetic', an .npz of checkpoint variables, or a TF-1 checkpoint prefix / ckpts dir "
                                     "(default: newest checkpoint in JOB_DIR/ckpts)")
    p.add_argument('--restore_itr', type=int, default=-1, help='Restore the newest checkpoint with iteration <= this '
                                                               '(val.py:215-217); -1 = newest.')
    p.add_argument('--device', default=None)
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help='process group of a multi-rank run (nccl = RCCL, one GPU per rank; gloo lets ranks share a GPU in tests)')
    p.add_argument('--in_flight', type=int, default=4, help='images processed concurrently per GPU, one stream each (1 = one at a time)')
    p.add_argument('--loader_threads', type=int, default=None,
                   help='host threads decoding PNGs ahead of the device (1 = decode in the loop; default: min(32, cores / ranks))')
    p.add_argument('--batch_same_shape', type=int, default=8,
                   help='consecutive images of one shape evaluated as ONE batch of up to this many when they are small (a 256 x 256 image '
                        'alone fills an eighth of the chip); per-image measures and their order as without; 1 = never')
    flags, unknown = p.parse_known_args(argv)
    if unknown:
        print('Unknown flags: {}'.format(unknown))

    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if flags.backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
    device = flags.device or 'cuda:{}'.format(torch.cuda.current_device())

    image_paths, dataset_name = get_image_paths(flags.images)
    image_paths = image_paths[:flags.how_many]
    here = path.dirname(path.abspath(__file__))
    bases = [os.environ.get('CONFIG_BASE_AE', path.join(here, 'ae_configs')),
             os.environ.get('CONFIG_BASE_PC', path.join(here, 'pc_configs'))]
    for job_dir in iter_job_dirs(flags.log_dir_root, flags.job_ids):
        ae_p, pc_p = config_paths_from_log_dir(job_dir, bases)
        ae_config, _ = config_parser.parse(ae_p)
        pc_config, _ = config_parser.parse(pc_p)
        weights = load_weights_for_job(job_dir, flags.weights, ae_config, pc_config, flags.restore_itr)
        out_dir = path.join(flags.log_dir_root, '{} {}'.format(log_date_from_log_dir(job_dir), dataset_name))
        if flags.reset:
            # rank 0 clears the directory; nobody writes into it before that is done (every rank passes the barrier,
            # whether or not the directory existed: no rank-asymmetric collective)
            if path.isdir(out_dir) and sharding.rank_and_world()[0] == 0:
                import shutil
                shutil.rmtree(out_dir)
            sharding.barrier()
        avgs = validate(ae_config, pc_config, weights, image_paths, out_dir,
                        OutputFlags(flags.save_ours, -1, flags.real_bpp), device, host_metrics=bool(flags.host_metrics),
                        in_flight=flags.in_flight,
                        loader_threads=default_loader_threads() if flags.loader_threads is None else flags.loader_threads,
                        batch_same_shape=flags.batch_same_shape)
        if sharding.rank_and_world()[0] == 0:
            print('Validation completed: {} | {}'.format(out_dir, avgs))
    print('*** All given job_ids validated.')


if __name__ == '__main__':
    main()
