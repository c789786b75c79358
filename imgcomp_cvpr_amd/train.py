"""Training driver -- the train.py entry point of the reference (code/train.py) on the MI355X kernels.

    python -m imgcomp_cvpr_amd.train AE_CONFIG PC_CONFIG [-o LOG_DIR_ROOT] [-dtrain DS] [-dtest DS] [--synthetic]
        [-ltrain 100] [-lsave 1000] [-ltest 1000] [--restore DIR [-i ITR] [--restore_continue] [--restore_skip_vars A,B]]
        [--from_identity DIR] [--max_itr N] [--no_sync_bn]
        (the reference's own flags, train.py:475-502; DS as in inputpipeline.get_dataset: imgnet_train / imgnet_test =
        TFRecord shards under $RECORDS_ROOT, a paths .pkl, or an image glob)

    multi-GPU (data parallel, one process per GPU, RCCL):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
           -m imgcomp_cvpr_amd.train AE_CONFIG PC_CONFIG --synthetic --max_itr 100

What is kept from the reference: the config files and the `MMDD_HHMM cfg@path cfg@path` log-dir naming
(logdir_helpers.py:34-56), the loss / optimiser recipe (training.py), random crops + horizontal flips of the
training images (inputpipeline.py:199-213), img/s on the console (train.py:201-213,256), checkpoints every
--save_interval iterations in the reference's on-disk format -- `ckpts/ckpt-<itr>.index` + `.data-00000-of-00001`
(TF-1 tensor bundle, tf_checkpoint.py) + `ckpts/var_names.pkl` (saver.py:19-43) -- so the reference's own val.py can
restore what this train.py wrote and vice versa.  What is different: the TF input queue is a plain loader;
TensorBoard / Sheets logging is out of scope.  --restore continues a run: variables, global_step (so the DECAY
schedule and the checkpoint numbering carry on) and the Adam slots when the checkpoint has them.  Under data parallelism the global batch of the config is split over the ranks; gradients are
averaged with three bucketed RCCL all-reduces per step (training.GradBuckets), and BatchNorm normalises over the WHOLE
split batch (cross-replica moments, training.py; --no_sync_bn keeps the statistics local to a rank).
"""
import argparse
import glob
import os
import time
from datetime import datetime
from os import path

import numpy as np
import torch

from . import config_parser, sharding, training
from . import weights as _weights


def create_unique_log_dir(config_rel_paths, log_dir_root):
    """`{MMDD_HHMM} {cfg paths with / -> @ and - -> *}` (logdir_helpers.py:34-56)."""
    post = ' '.join(p.replace(path.sep, '@').replace('-', '*') for p in config_rel_paths)
    os.makedirs(log_dir_root, exist_ok=True)
    t = datetime.now()
    while True:
        d = path.join(log_dir_root, '{} {}'.format(t.strftime('%m%d_%H%M'), post))
        if not any(n.split(' ')[0] == t.strftime('%m%d_%H%M') for n in os.listdir(log_dir_root)):
            os.makedirs(d)
            return d
        t = t.replace(minute=(t.minute + 1) % 60, hour=(t.hour + (t.minute + 1) // 60) % 24)


NUM_CROPS_PER_IMG = 8            # train.py:38 of the reference


class CropLoader(object):
    """random crop_size crops (+ random horizontal flip) of the images matching a glob, as (N,3,h,w) float 0..255.
    Mirrors inputpipeline.py:147-213: every decoded image yields NUM_CROPS_PER_IMG crops (one flip decision per
    image, tf_helpers.random_flip on the stacked crops), crops from different images are mixed through a shuffle pool
    (tf.train.shuffle_batch_join: capacity 1000, min_after_dequeue 800 -- scaled down for small data sets), and
    decoding runs on background threads so the GPU step (~1000 img/s) does not wait for PNG decoding.
    `synthetic=True`: seeded synthetic images, filled synchronously (deterministic; tests and benchmarks)."""

    def __init__(self, images_glob, crop_size, batch_size, seed=0, synthetic=False, num_crops_per_img=NUM_CROPS_PER_IMG,
                 num_threads=4, capacity=1000, min_after_dequeue=800):
        self.crop, self.batch = tuple(crop_size), batch_size
        self.rs = np.random.RandomState(seed)
        self.synthetic = synthetic
        self.crops_per_img = num_crops_per_img
        if synthetic:
            self.images = [_weights.synthetic_image((1, 3, 2 * self.crop[0], 2 * self.crop[1]), 'natural', seed=1000 * seed + i)[0]
                           for i in range(8)]
            self._pool = None
        else:
            from . import datasets
            self.dataset = datasets.get_dataset(images_glob)          # record shards, paths pickle or image glob
            self.images = None
            import threading
            self._pool, self._lock = [], threading.Condition()
            self._capacity = max(capacity, 2 * batch_size)
            self._min_after = min(min_after_dequeue, self._capacity - batch_size)
            self._stop = False
            self._threads = [threading.Thread(target=self._worker, args=(seed * 1000 + 17 * t + 1,), daemon=True)
                             for t in range(num_threads)]
            for t in self._threads:
                t.start()

    @property
    def num_images(self):
        return len(self.images) if self.synthetic else self.dataset.num_images

    def _image(self, i):
        return self.images[i]

    def _crops_of(self, im, rs):
        H, W = im.shape[1:]
        if H < self.crop[0] or W < self.crop[1]:
            raise ValueError('image smaller than crop size')
        flip = rs.rand() < 0.5
        out = []
        for _ in range(self.crops_per_img):
            y, x = rs.randint(H - self.crop[0] + 1), rs.randint(W - self.crop[1] + 1)
            c = im[:, y:y + self.crop[0], x:x + self.crop[1]]
            out.append(np.ascontiguousarray(c[:, :, ::-1] if flip else c))
        return out

    def _worker(self, seed):
        rs = np.random.RandomState(seed)
        for im_hwc in self.dataset.stream(rs):
            if self._stop:
                return
            H, W = im_hwc.shape[:2]
            if H < self.crop[0] or W < self.crop[1]:
                continue                                   # tf.random_crop would fail the queue runner; skip instead
            crops = self._crops_of(np.transpose(im_hwc, (2, 0, 1)), rs)
            with self._lock:
                while len(self._pool) + len(crops) > self._capacity and not self._stop:
                    self._lock.wait(0.1)
                self._pool.extend(crops)
                self._lock.notify_all()

    def close(self):
        if self._pool is not None:
            self._stop = True
            with self._lock:
                self._lock.notify_all()

    def get_batch(self):
        out = np.empty((self.batch, 3) + self.crop, np.float32)
        if self.synthetic:
            b = 0
            while b < self.batch:
                for c in self._crops_of(self._image(self.rs.randint(self.num_images)), self.rs)[:self.batch - b]:
                    out[b] = c
                    b += 1
            return out
        with self._lock:
            while len(self._pool) < self._min_after + self.batch:
                self._lock.wait(0.1)
            for b in range(self.batch):
                j = self.rs.randint(len(self._pool))
                self._pool[j], self._pool[-1] = self._pool[-1], self._pool[j]
                out[b] = self._pool.pop()
            self._lock.notify_all()
        return out


def test_in_train(tr, ae, pc, x_test):
    """the reference's `test` name scope (train.py:115-127): encode / decode(qhard) / bitcost(qhard) with is_training=False on
    the CURRENT training variables (ae / pc are bound to tr.graph), bpp and the distortions on truncated uint8 values."""
    from . import bits
    enc = ae.encode(x_test, is_training=False)
    x_out = ae.decode(enc.qhard, is_training=False)
    bc = pc.bitcost(enc.qhard, enc.symbols, is_training=False, pad_value=pc.auto_pad_value(ae))
    d = training.Distortions(ae.config, x_test, x_out, is_training=False)
    out = {'bpp': float(bits.bitcost_to_bpp(bc, x_test)), 'mse': float(d.mse), 'psnr': float(d.psnr)}
    if d.ms_ssim is not None:
        out['ms_ssim'] = float(d.ms_ssim)
    return out


def train(ae_config_path, pc_config_path, log_dir_root, loader_fn, max_itr, log_interval=100, save_interval=1000,
          restore=None, device=None, verbose=True, log_interval_test=-1, test_loader_fn=None, restore_itr=-1,
          restore_continue=False, restore_skip_vars=None, from_identity=None, sync_bn=None):
    """log_interval = --log_interval_train, save_interval = --log_interval_save of the reference (train.py:475-502);
    restore* / from_identity as restore_manager.py:23-58: --from_identity DIR = --restore DIR without global_step and
    anything matching *Adam*; restore_skip_vars = comma-separated substrings of variable names NOT to restore;
    restore_continue = keep logging / checkpointing into the restored run's log dir."""
    rank, world = sharding.rank_and_world()
    ae_config, ae_rel = config_parser.parse(ae_config_path)
    pc_config, pc_rel = config_parser.parse(pc_config_path)
    device = device or 'cuda:{}'.format(torch.cuda.current_device())
    batch_total = int(ae_config.batch_size)
    if batch_total % world:
        raise ValueError('batch_size {} not divisible by {} ranks'.format(batch_total, world))
    loader = loader_fn(ae_config, batch_total // world, rank)
    if from_identity:
        restore, restore_skip_vars = from_identity, 'global_step,Adam'          # restore_manager.py:27-30
    skip = [v.strip() for v in restore_skip_vars.split(',') if v.strip()] if restore_skip_vars else []
    # a fresh set of variables (Xavier, as slim initialises); a restore overwrites what the checkpoint holds and is not skipped
    weights = _weights.synthetic_weights(ae_config, pc_config, gain=1.0, heatmap_bias=None)
    weights[_weights.ENC + '/centers'] = np.random.RandomState(666).uniform(
        *map(float, ae_config.centers_initial_range), size=int(ae_config.num_centers)).astype(np.float32)
    ckpt = None
    if restore:
        from . import tf_checkpoint
        ckpt = tf_checkpoint.load_weights(restore, itr=restore_itr, training_state=True)
        ckpt = {k: v for k, v in ckpt.items() if not any(sv in k for sv in skip)}
        restored = {k: v for k, v in ckpt.items() if tf_checkpoint.is_model_variable(k)}
        if not skip:
            missing = sorted(set(weights) - set(restored))
            if missing:
                raise ValueError('checkpoint {} lacks {} variables of the graph, e.g. {}'.format(restore, len(missing), missing[:3]))
        weights.update(restored)
        if verbose and rank == 0:
            print('Restoring {} variables...'.format(len(ckpt)))
    # epoch length as the reference counts it (training_helpers.py:51-60): every decoded image yields NUM_CROPS_PER_IMG crops,
    # so batch 30 consumes 3 images; the DECAY schedule (x0.1 every 2 epochs, staircase) is timed in these epochs
    num_itr_per_epoch = training.get_num_itr_per_epoch(loader.num_images, batch_total, loader.crops_per_img)
    tr = training.Trainer(ae_config, pc_config, weights, device, num_itr_per_epoch, sync_bn=sync_bn)
    start_itr = tr.restore_training_state(ckpt) if ckpt else 0
    if world > 1:
        # every rank starts from identical variables (rank 0's)
        import torch.distributed as dist
        for t in tr.graph.params.values():
            dist.broadcast(t, src=0)
    # test-in-train (train.py:115-127, 265-266): plugin objects bound to the training variables, evaluated every
    # log_interval_test iterations on rank 0 (-1 skips it, as in the reference)
    ae = pc = test_loader = None
    if log_interval_test > 0 and rank == 0:
        from . import autoencoder, probclass
        ae = autoencoder.get_network_cls(ae_config)(ae_config)
        pc = probclass.get_network_cls(pc_config)(pc_config, num_centers=ae_config.num_centers)
        tr.graph.bind(ae, pc)
        test_loader = (test_loader_fn or loader_fn)(ae_config, batch_total, rank)
    log_dir = None
    if rank == 0 and log_dir_root:
        if restore_continue and restore:
            from . import tf_checkpoint
            log_dir = tf_checkpoint.log_dir_for_restore(restore)
            print('Using restore dir as log dir!')
        else:
            log_dir = create_unique_log_dir([ae_rel, pc_rel], log_dir_root)
        os.makedirs(path.join(log_dir, 'ckpts'), exist_ok=True)
        if verbose:
            print('Log dir: {}'.format(log_dir))
    t_last, n_last = time.time(), start_itr
    hist = []
    last = start_itr + max_itr
    for step in range(start_itr, last):
        x = torch.as_tensor(loader.get_batch()).to(device)
        out = tr.step(x)
        hist.append(out)
        itr = step + 1                          # the reference reads global_step AFTER the train op (train.py:245)
        if verbose and rank == 0 and (itr % log_interval == 0 or itr == last):
            torch.cuda.synchronize()
            dt = time.time() - t_last
            ips = (itr - n_last) * batch_total / max(dt, 1e-9)
            print('{: 7d} | {} | (img/s: {:.1f})'.format(itr, ' '.join('{}: {:.4f}'.format(k, v) for k, v in out.items()), ips), flush=True)
        if log_dir and (itr % save_interval == 0 or itr == last):
            if verbose:
                print('Saving...')
            save_checkpoint(path.join(log_dir, 'ckpts'), tr.state_weights(), itr)
        if ae is not None and (itr % log_interval_test == 0 or itr == last):
            res = test_in_train(tr, ae, pc, torch.as_tensor(test_loader.get_batch()).to(device))
            hist[-1] = dict(out, **{'test_' + k: v for k, v in res.items()})
            if verbose:
                print('{: 7d} | test | {}'.format(itr, ' '.join('{}: {:.4f}'.format(k, v) for k, v in res.items())), flush=True)
        if world > 1 and ((log_dir_root and (itr % save_interval == 0 or itr == last)) or
                          (log_interval_test > 0 and (itr % log_interval_test == 0 or itr == last))):
            # rank 0 alone has just saved / evaluated: the others wait here instead of inside the next step's first BatchNorm
            # exchange, whose bounded spin (sync_bn='p2p', csrc/peer_exchange.hip) would time out behind a long pause
            import torch.distributed as dist
            dist.barrier()
        if itr % log_interval == 0:             # reset after all of the above for accurate timings (train.py:268-269)
            t_last, n_last = time.time(), itr
    return tr, hist, log_dir


def save_checkpoint(ckpt_dir, variables, global_step):
    """saver.py:46-100: `ckpt-<global_step>` as a TF-1 tensor bundle + var_names.pkl (written once per directory)."""
    from . import tf_checkpoint
    os.makedirs(ckpt_dir, exist_ok=True)
    tensors = dict(variables)
    tensors['global_step'] = np.array(global_step, np.int64)          # (already there when state_weights wrote it)
    if not path.exists(path.join(ckpt_dir, 'var_names.pkl')):
        tf_checkpoint.write_var_names(ckpt_dir, sorted(tensors))
    tf_checkpoint.write_bundle(path.join(ckpt_dir, 'ckpt-{}'.format(global_step)), tensors)


def build_arg_parser():
    """command line of the reference's train.py (train.py:475-502).  Kept: every flag that concerns the path built here.
    Accepted and ignored (subsystems out of scope): --dataset_codec_distance, --log_run_metadata,
    --summarize_gradients, --ckpt_interval, --description.  Added: --synthetic, --max_itr, --no_sync_bn."""
    p = argparse.ArgumentParser()
    p.add_argument('autoencoder_config_path')
    p.add_argument('probclass_config_path')
    p.add_argument('--dataset_train', '-dtrain', default='imgnet_train', help='imgnet_train | paths .pkl | image glob (inputpipeline.get_dataset)')
    p.add_argument('--dataset_test', '-dtest', default='imgnet_test', help='same, for the test-in-train evaluation')
    p.add_argument('--dataset_codec_distance', '-dcodec', default='testset', help='ignored (codec_distance is out of scope)')
    p.add_argument('--log_dir_root', '-o', default='logs', metavar='LOG_DIR_ROOT')
    p.add_argument('--log_interval_train', '--log_interval', '-ltrain', type=int, default=100, dest='log_interval_train')
    p.add_argument('--log_interval_save', '--save_interval', '-lsave', type=int, default=1000, dest='log_interval_save')
    p.add_argument('--log_interval_test', '-ltest', type=int, default=1000, help='Set to -1 to skip testing, which saves memory.')
    p.add_argument('--log_run_metadata', '-lmeta', action='store_const', const=True, help='ignored')
    p.add_argument('--summarize_gradients', '-lgrads', action='store_const', const=True, help='ignored')
    p.add_argument('--temporary', '-t', action='store_const', const=True, help='Append _TMP to LOG_DIR_ROOT')
    p.add_argument('--from_identity', metavar='IDENTITY_CKPT_DIR',
                   help='Like --restore IDENTITY_CKPT_DIR, but global_step and any variables matching *Adam* are not restored')
    p.add_argument('--restore', '-r', metavar='RESTORE_DIR', help='ckpt dir / log dir / checkpoint prefix / .npz to restore from')
    p.add_argument('--restore_itr', '-i', type=int, default=-1,
                   help='Iteration to restore from. -1 = latest, otherwise the latest checkpoint with iteration <= restore_itr')
    p.add_argument('--restore_continue', action='store_const', const=True,
                   help='keep saving logs and checkpoints into the log dir of RESTORE_DIR')
    p.add_argument('--restore_skip_vars', type=str, help='Var names to skip, comma separated, e.g. "Adam,global_step"')
    p.add_argument('--ckpt_interval', type=float, default=1, help='ignored (every checkpoint is kept)')
    p.add_argument('--description', '-d', type=str, help='ignored (Google Sheets logging is out of scope)')
    p.add_argument('--synthetic', action='store_const', const=True, help='seeded synthetic images instead of data sets')
    p.add_argument('--max_itr', type=int, default=1000, help='iterations to run (the reference runs until interrupted)')
    p.add_argument('--no_sync_bn', action='store_const', const=True,
                   help='data parallel: BatchNorm statistics per rank instead of over the whole (split) batch')
    p.add_argument('--sync_bn_p2p', action='store_const', const=True,
                   help='data parallel: exchange the BatchNorm moments over peer-mapped memory (one small launch per layer) '
                        'instead of an RCCL all-reduce per layer')
    return p


def main(argv=None):
    p = build_arg_parser()
    flags = p.parse_args(argv)
    if flags.temporary:
        flags.log_dir_root = flags.log_dir_root.rstrip(path.sep) + '_TMP'
    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    def loader_fn(ae_config, batch, rank):
        return CropLoader(flags.dataset_train, ae_config.crop_size, batch, seed=rank, synthetic=bool(flags.synthetic))

    def test_loader_fn(ae_config, batch, rank):
        # ip_test: one crop per image, no shuffling (train.py:117-121)
        return CropLoader(flags.dataset_test, ae_config.crop_size, batch, seed=10007, synthetic=bool(flags.synthetic),
                          num_crops_per_img=1)
    here = path.dirname(path.abspath(__file__))

    def resolve(pth, base):
        return pth if path.isfile(pth) else path.join(here, base, pth)
    train(resolve(flags.autoencoder_config_path, 'ae_configs'), resolve(flags.probclass_config_path, 'pc_configs'),
          flags.log_dir_root, loader_fn, flags.max_itr, flags.log_interval_train, flags.log_interval_save, flags.restore,
          log_interval_test=flags.log_interval_test, test_loader_fn=test_loader_fn, restore_itr=flags.restore_itr,
          restore_continue=bool(flags.restore_continue), restore_skip_vars=flags.restore_skip_vars,
          from_identity=flags.from_identity, sync_bn=False if flags.no_sync_bn else ('p2p' if flags.sync_bn_p2p else None))


if __name__ == '__main__':
    main()
