"""Training / validation image sources -- the dataset dispatch of the reference (code/inputpipeline.py:16-128).

    get_dataset(ds):  'imgnet_train' / 'imgnet_test' -> TFRecord shards under $RECORDS_ROOT/{train,val}/*.tfrecord,
                                                         encoded image in the feature 'image/encoded'
                      '*.pkl'                         -> pickled list of paths relative to the pickle's directory
                                                         (PathsDataset.make_paths_pickle_file_from_image_glob)
                      anything else                   -> a glob matching image files
Every dataset has .name, .num_images (used for the iterations-per-epoch of the learning-rate schedule,
training_helpers.py:51-60) and .stream(rs), an endless iterator of decoded HWC uint8 RGB images in shuffled order
(what `images_decoded(num_epochs=None, shuffle=True)` is to the TF queue runners).

The TFRecord container and tf.train.Example are restated from their published definitions
(tensorflow/core/lib/io/record_writer.cc: u64 length, masked crc32c(length), payload, masked crc32c(payload);
tensorflow/core/example/{example,feature}.proto) -- TensorFlow is not installable here, so parity is pinned only
by round trips through the writer below and the published CRC/mask constants (see tf_checkpoint.py).
"""
import glob
import io
import os
import pickle
import struct

import numpy as np

from .tf_checkpoint import _field, _parse_proto, _put_varint, crc32c, mask_crc

IMAGENET_NUM_TRAIN = 1281167          # inputpipeline.py:72


def _decode_image(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'), dtype=np.uint8)


# ---- TFRecord container + tf.train.Example ------------------------------------------------------------------------------

def iter_tfrecord(path, verify=False):
    """payload bytes of every record of one .tfrecord file."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError('{}: truncated record header'.format(path))
            n, crc_len = struct.unpack('<QI', head)
            if verify and mask_crc(crc32c(head[:8])) != crc_len:
                raise ValueError('{}: corrupted record length'.format(path))
            data = f.read(n)
            tail = f.read(4)
            if len(data) != n or len(tail) != 4:
                raise ValueError('{}: truncated record'.format(path))
            if verify and mask_crc(crc32c(data)) != struct.unpack('<I', tail)[0]:
                raise ValueError('{}: corrupted record payload'.format(path))
            yield data


def example_bytes_feature(example, key):
    """first bytes value of feature `key` of a serialised tf.train.Example (Example.features = 1, Features.feature = 1
    map<string, Feature>, Feature.bytes_list = 1, BytesList.value = 1)."""
    want = key.encode('utf-8')
    for f1, _, features in _parse_proto(example):
        if f1 != 1:
            continue
        for f2, _, entry in _parse_proto(features):
            if f2 != 1:
                continue
            k, v = None, None
            for f3, _, x in _parse_proto(entry):
                if f3 == 1:
                    k = x
                elif f3 == 2:
                    v = x
            if k != want or v is None:
                continue
            for f4, _, blist in _parse_proto(v):
                if f4 == 1:
                    for f5, _, val in _parse_proto(blist):
                        if f5 == 1:
                            return val
    raise KeyError('feature {!r} not in example'.format(key))


def _delimited(num, payload):
    return _field(num, 2, _put_varint(len(payload)) + payload)


def make_example(features):
    """{key: bytes} -> serialised tf.train.Example (bytes features only)."""
    body = b''
    for k, v in features.items():
        feature = _delimited(1, _delimited(1, v))                      # Feature{bytes_list{value}}
        body += _delimited(1, _delimited(1, k.encode('utf-8')) + _delimited(2, feature))
    return _delimited(1, body)


def write_tfrecord(path, payloads):
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data))))


# ---- datasets -----------------------------------------------------------------------------------------------------------

class RecordsDataset(object):
    """inputpipeline.py:44-86."""

    def __init__(self, name, records_glob, feature_key, num_images, no_matches_hint=None):
        self.files = sorted(glob.glob(records_glob))
        if not self.files:
            raise ValueError('No matches for {}. ({})'.format(records_glob, no_matches_hint or ''))
        self.name, self.records_glob, self.feature_key = name, records_glob, feature_key
        self._num_images = num_images

    @property
    def num_images(self):
        if self._num_images is None:                                   # imgnet_test: count once
            self._num_images = sum(1 for f in self.files for _ in iter_tfrecord(f))
        return self._num_images

    def stream(self, rs):
        while True:
            for i in rs.permutation(len(self.files)):                  # shard order shuffled per epoch, records in file order
                for rec in iter_tfrecord(self.files[i]):
                    yield _decode_image(example_bytes_feature(rec, self.feature_key))


class PathsDataset(object):
    """inputpipeline.py:89-145."""

    def __init__(self, name, paths):
        if not paths:
            raise ValueError('no images in {}'.format(name))
        self.name, self.paths = name, list(paths)

    @property
    def num_images(self):
        return len(self.paths)

    def stream(self, rs):
        while True:
            for i in rs.permutation(len(self.paths)):
                with open(self.paths[i], 'rb') as f:
                    yield _decode_image(f.read())

    @staticmethod
    def from_img_glob(img_glob):
        paths = sorted(glob.glob(img_glob))
        if not paths:
            raise ValueError('glob not matching any files: {}'.format(img_glob))
        return PathsDataset('glob_' + img_glob.replace('/', '_').replace('*', '_'), paths)

    @staticmethod
    def from_paths_pickle_file(paths_pickle_file):
        if not paths_pickle_file.endswith('.pkl'):
            raise ValueError('Not a .pkl file: {}'.format(paths_pickle_file))
        base = os.path.dirname(os.path.abspath(paths_pickle_file))
        with open(paths_pickle_file, 'rb') as f:
            rel = pickle.load(f)
        return PathsDataset('pickle_{}'.format(paths_pickle_file), [os.path.join(base, p) for p in rel])


def get_dataset(ds, records_root=None):
    """inputpipeline.py:16-37: named record sets, then a paths pickle, then an image glob."""
    root = records_root or os.environ.get('RECORDS_ROOT', '')
    named = {'imgnet_train': ('train', IMAGENET_NUM_TRAIN), 'imgnet_test': ('val', None)}
    if ds in named:
        sub, n = named[ds]
        return RecordsDataset(ds, os.path.join(root, sub, '*.tfrecord'), 'image/encoded', n,
                              'Make sure $RECORDS_ROOT is set correctly.')
    if ds.endswith('.tfrecord') or ds.endswith('.tfrecords'):            # convenience: a glob of record files
        return RecordsDataset('records_' + ds.replace('/', '_').replace('*', '_'), ds, 'image/encoded', None)
    if ds.endswith('.pkl'):
        return PathsDataset.from_paths_pickle_file(ds)
    try:
        return PathsDataset.from_img_glob(ds)
    except ValueError:
        raise ValueError('Invalid dataset: {}'.format(ds))
