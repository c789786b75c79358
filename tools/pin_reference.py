#!/usr/bin/env python
"""Pin this build to the reference's published model -- ONE command for the day the assets arrive (VERDICT r4, next-round item 7;
SURVEY 8(c) / 8(f) N1: the only TF-1.4 anchor the reference offers is README.md:47 -- 0515_1103 on Kodak: bpp 0.370, MS-SSIM 0.975).

    python tools/pin_reference.py CKPTS_ROOT KODAK_DIR [--job_id 0515_1103] [--expect_bpp 0.370] [--expect_msssim 0.975]
                                  [--decimals 3] [--out tests/golden/kodak_0515_1103.npz] [--inventory_only]

CKPTS_ROOT is the extracted ckpts.tar.gz of README.md:14-15 (the directory val.py takes as LOG_DIR_ROOT: it holds
`0515_1103 ae_configs@cvpr@low pc_configs@cvpr@res_shallow/ckpts/{ckpt-<itr>.index, .data-00000-of-00001, var_names.pkl}`).
What it does, failing with a precise message at the first thing that is off:
  1. finds the job's log dir and its configs (val.py:57-75), the newest checkpoint (saver.py:102-127);
  2. INVENTORY: every variable the two networks need (name -> shape, derived here from the configs the same way
     weights.synthetic_weights lays them out) must be in the bundle with that shape and float32; every model variable of the
     bundle and of var_names.pkl (saver.py:19-43) must be one the networks know -- a missing, mis-shaped or unknown variable is
     named; optimiser slots, beta powers and global_step are training state and are only counted;
  3. loads the bundle with its CRC-32C checks on (tf_checkpoint.read_bundle(verify=True));
  4. runs val.py's validate() over KODAK_DIR (host metrics: the reference's own numpy MS-SSIM / PSNR, val.py:96-108);
  5. asserts the means equal README's to --decimals places;
  6. writes --out: per-image name, bpp, MS-SSIM, PSNR and the CRC-32 of every image's symbol volume (int64, C-order) -- the
     fixture that pins conv / BN / conv3d arithmetic to TF-1.4 from then on (tests/test_gpu_configs.py::test_pinned_kodak_fixture replays
     it wherever IMGCOMP_CKPTS_ROOT and IMGCOMP_KODAK point at the assets, and skips elsewhere).
Steps 1-3 need no GPU (--inventory_only stops there)."""
import argparse
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class PinError(Exception):
    pass


def expected_inventory(ae_config, pc_config):
    """name -> shape of every variable the hot path reads, from the configs."""
    from imgcomp_cvpr_amd import weights as W
    w = W.synthetic_weights(ae_config, pc_config)
    return {k: tuple(np.asarray(v).shape) for k, v in w.items()}


def check_inventory(prefix, ckpt_dir, ae_config, pc_config):
    """-> (names to load, report dict); raises PinError naming the first offending variables."""
    from imgcomp_cvpr_amd import tf_checkpoint as T
    want = expected_inventory(ae_config, pc_config)
    have = T.list_variables(prefix)
    problems = []
    for name in sorted(want):
        if name not in have:
            problems.append('missing variable {} (expected float32 {})'.format(name, list(want[name])))
            continue
        dt, shape = have[name]
        if tuple(shape) != want[name]:
            problems.append('variable {} has shape {}, expected {}'.format(name, list(shape), list(want[name])))
        if dt != np.float32:
            problems.append('variable {} is {}, expected float32'.format(name, dt))
    model_in_bundle = [n for n in have if T.is_model_variable(n)]
    for name in model_in_bundle:
        if name not in want:
            problems.append('bundle holds a model variable the networks of these configs do not have: {} {}'.format(name, list(have[name][1])))
    state = [n for n in have if T.is_training_state(n)]
    other = [n for n in have if not T.is_model_variable(n) and not T.is_training_state(n)]
    n_pkl = None
    pkl = os.path.join(ckpt_dir, 'var_names.pkl')
    if os.path.isfile(pkl):
        names = T.read_var_names(ckpt_dir)
        n_pkl = len(names)
        for n in names:
            if n not in have:
                problems.append('var_names.pkl lists {} which is not in the bundle'.format(n))
        for n in want:
            if n not in names:
                problems.append('var_names.pkl does not list {} (restore_manager would not restore it)'.format(n))
    else:
        problems.append('no var_names.pkl in {} (saver.py:19-43 writes one next to every checkpoint)'.format(ckpt_dir))
    if problems:
        raise PinError('checkpoint {} does not match the inventory of SURVEY Appendix B:\n  '.format(prefix) + '\n  '.join(problems[:40]) +
                       ('\n  ... {} more'.format(len(problems) - 40) if len(problems) > 40 else ''))
    return sorted(want), {'model_variables': len(want), 'training_state_variables': len(state), 'other_variables': other,
                          'var_names_pkl': n_pkl, 'parameters': int(sum(int(np.prod(s)) for s in want.values()))}


def find_job(ckpts_root, job_id):
    from imgcomp_cvpr_amd import val, config_parser
    dirs = list(val.iter_job_dirs(ckpts_root, job_id))
    if len(dirs) != 1:
        raise PinError('expected exactly one log dir starting with {!r} in {}, found {}'.format(job_id, ckpts_root, dirs))
    job_dir = dirs[0]
    here = os.path.join(ROOT, 'imgcomp_cvpr_amd')
    ae_p, pc_p = val.config_paths_from_log_dir(job_dir, [os.path.join(here, 'ae_configs'), os.path.join(here, 'pc_configs')])
    ae_config, _ = config_parser.parse(ae_p)
    pc_config, _ = config_parser.parse(pc_p)
    return job_dir, ae_config, pc_config


def pin(ckpts_root, images, job_id='0515_1103', expect_bpp=0.370, expect_msssim=0.975, decimals=3, out=None, inventory_only=False,
        device='cuda:0', restore_itr=-1, verbose=True):
    from imgcomp_cvpr_amd import tf_checkpoint as T
    say = print if verbose else (lambda *a, **k: None)
    job_dir, ae_config, pc_config = find_job(ckpts_root, job_id)
    ckpt_dir = T.ckpt_dir_for_log_dir(job_dir)
    if not os.path.isdir(ckpt_dir):
        raise PinError('no ckpts/ directory in {}'.format(job_dir))
    itr, prefix = T.latest_checkpoint_before_itr(ckpt_dir, restore_itr)
    if not os.path.isfile(prefix + '.index'):
        raise PinError('{} is not a TF-1 tensor bundle (no .index)'.format(prefix))
    say('checkpoint {} (iteration {})'.format(prefix, itr))
    names, rep = check_inventory(prefix, ckpt_dir, ae_config, pc_config)
    say('inventory ok: {model_variables} model variables, {parameters} parameters, {training_state_variables} training-state '
        'variables, var_names.pkl lists {var_names_pkl}'.format(**rep))
    try:
        weights = dict(T.read_bundle(prefix, names=names, verify=True))
    except (ValueError, KeyError) as e:
        raise PinError('reading {}: {}'.format(prefix, e))
    for n, a in weights.items():
        if not np.all(np.isfinite(a)):
            raise PinError('variable {} holds non-finite values'.format(n))
    say('bundle read, every tensor CRC-32C verified')
    if inventory_only:
        return {'inventory': rep}

    from imgcomp_cvpr_amd import val
    image_paths, dataset_name = val.get_image_paths(images)
    if not image_paths:
        raise PinError('no images in {}'.format(images))
    out_dir = os.path.join(ckpts_root, '{} {}'.format(val.log_date_from_log_dir(job_dir), dataset_name))
    rows, crcs = [], []
    fetcher = val.Fetcher(ae_config, pc_config, weights, device, host_metrics=True)
    pad = fetcher.ae.get_subsampling_factor()
    for p in image_paths:
        img = val.load_image_chw(p, pad)
        otp = fetcher.collect(fetcher.enqueue(img, want_symbols=True, want_image=False))
        sym = np.ascontiguousarray(np.asarray(otp.pop('sym')).astype(np.int64))
        crcs.append(zlib.crc32(sym.tobytes()) & 0xffffffff)
        rows.append((os.path.basename(p), float(otp['bpp']), float(otp['ms-ssim']), float(otp['psnr'])))
        say('{:24s} bpp {:.5f}  ms-ssim {:.5f}  psnr {:.3f}'.format(*rows[-1]))
    mean_bpp = float(np.mean([r[1] for r in rows]))
    mean_ms = float(np.mean([r[2] for r in rows]))
    say('mean over {} images: bpp {:.5f}  ms-ssim {:.5f}   (README.md:47: {} / {})'.format(len(rows), mean_bpp, mean_ms, expect_bpp, expect_msssim))
    tol = 0.5 * 10 ** (-decimals) + 1e-12
    bad = []
    if expect_bpp is not None and abs(mean_bpp - expect_bpp) > tol:
        bad.append('mean bpp {:.5f} != {} to {} decimals'.format(mean_bpp, expect_bpp, decimals))
    if expect_msssim is not None and abs(mean_ms - expect_msssim) > tol:
        bad.append('mean MS-SSIM {:.5f} != {} to {} decimals'.format(mean_ms, expect_msssim, decimals))
    if bad:
        raise PinError('the published numbers are NOT reproduced: ' + '; '.join(bad))
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        np.savez(out, names=np.array([r[0] for r in rows]), bpp=np.array([r[1] for r in rows], np.float64),
                 msssim=np.array([r[2] for r in rows], np.float64), psnr=np.array([r[3] for r in rows], np.float64),
                 symbol_crc32=np.array(crcs, np.uint32), job_id=np.array(job_id), iteration=np.array(itr, np.int64),
                 mean_bpp=np.array(mean_bpp), mean_msssim=np.array(mean_ms))
        say('wrote {}'.format(out))
    return {'inventory': rep, 'rows': rows, 'symbol_crc32': crcs, 'mean_bpp': mean_bpp, 'mean_msssim': mean_ms, 'out_dir': out_dir}


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('ckpts_root')
    p.add_argument('images')
    p.add_argument('--job_id', default='0515_1103')
    p.add_argument('--expect_bpp', type=float, default=0.370)
    p.add_argument('--expect_msssim', type=float, default=0.975)
    p.add_argument('--decimals', type=int, default=3)
    p.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'kodak_0515_1103.npz'))
    p.add_argument('--inventory_only', action='store_true')
    p.add_argument('--device', default='cuda:0')
    p.add_argument('--restore_itr', type=int, default=-1)
    a = p.parse_args(argv)
    try:
        pin(a.ckpts_root, a.images, a.job_id, a.expect_bpp, a.expect_msssim, a.decimals, a.out, a.inventory_only, a.device, a.restore_itr)
    except PinError as e:
        print('pin_reference: FAILED: {}'.format(e), file=sys.stderr)
        return 1
    print('pin_reference: ok')
    return 0


if __name__ == '__main__':
    sys.exit(main())
