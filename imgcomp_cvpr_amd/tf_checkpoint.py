"""TF-1 checkpoint ("tensor bundle", V2 format) reader and writer -- row N1 of SURVEY.md section 8.

The reference stores its models with tf.train.Saver (code/saver.py:46-100): per checkpoint
    ckpts/ckpt-<itr>.index                   an SSTable (the LevelDB table format) mapping
                                              ""            -> BundleHeaderProto
                                              <tensor name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
    ckpts/ckpt-<itr>.data-00000-of-00001     the raw little-endian tensor bytes, addressed by (offset, size)
    ckpts/var_names.pkl                      pickled list of variable names ("...:0"), saver.py:19-43
and finds them by file name (saver.py:115-142: everything containing 'ckpt', iteration = first '-<digits>').

TensorFlow is not installable here, so the format is restated from its published definition
(tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensorflow/core/lib/io/{table,block,format}.cc,
tensorflow/core/protobuf/tensor_bundle.proto, pinned version tensorflow-gpu 1.4.1, requirements.txt:9).
PARITY UNPINNED: no reference checkpoint exists in this environment; what is tested is that the writer's files
re-read exactly, that block/footer/CRC rules hold, and the published constants (magic number, CRC-32C check values).

Only what a Saver checkpoint of dense variables needs is implemented: one or more shards, uncompressed blocks
(BundleWriter always sets kNoCompression), no tensor slices.
"""
import os
import pickle
import re
import struct
from collections import OrderedDict

import numpy as np

_TABLE_MAGIC = 0xdb4775248b80fb57          # table/format.h kTableMagicNumber
_FOOTER_LEN = 48                           # 2 block handles padded to 40 bytes + 8 bytes magic
_BLOCK_TRAILER = 5                         # 1 byte compression type + 4 bytes masked crc32c
_RESTART_INTERVAL = 16
_CKPT_DIR_NAME = 'ckpts'
_CKPT_FN = 'ckpt'

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_ENUM = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- CRC-32C (Castagnoli), masked as in lib/hash/crc32c.h ------------------------------------------------------------

def _make_crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC_TABLE = _make_crc_table()


def _crc32c_py(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_native_crc = None


def crc32c(data, crc=0):
    """CRC-32C; large buffers go through the library's host routine (ic_crc32c) when it can be loaded -- pure Python
    manages ~6 MB/s, a 40 MB checkpoint has to be summed on every save."""
    global _native_crc
    data = bytes(data)
    if len(data) >= 4096 and _native_crc is not False:
        if _native_crc is None:
            try:
                from . import _lib
                _native_crc = _lib.lib.ic_crc32c
            except Exception:                       # no ROCm runtime on this host: the table-driven loop still works
                _native_crc = False
        if _native_crc:
            return int(_native_crc(data, len(data), crc))
    return _crc32c_py(data, crc)


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints and the three protobuf messages ---------------------------------------------------------------------------

def _get_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field number, wire type, value); value is int (varint / fixed) or bytes (length-delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wt))
        out.append((field, wt, v))
    return out


def _parse_shape(buf):
    dims = []
    for field, _, v in _parse_proto(buf):
        if field == 2:                                  # repeated Dim dim = 2
            size = 0
            for f2, _, v2 in _parse_proto(v):
                if f2 == 1:                             # int64 size = 1
                    size = v2 - (1 << 64) if v2 >= 1 << 63 else v2
            dims.append(size)
        elif field == 3 and v:                          # unknown_rank
            raise ValueError('tensor of unknown rank in a checkpoint')
    return tuple(dims)


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for field, _, v in _parse_proto(buf):
        if field == 1:
            e['dtype'] = v
        elif field == 2:
            e['shape'] = _parse_shape(v)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = v
        elif field == 5:
            e['size'] = v
        elif field == 6:
            e['crc32c'] = v
        elif field == 7:
            e['slices'] += 1
    return e


def _field(num, wt, payload):
    return _put_varint((num << 3) | wt) + payload


def _encode_shape(shape):
    out = b''
    for d in shape:
        dim = _field(1, 0, _put_varint(int(d)))
        out += _field(2, 2, _put_varint(len(dim)) + dim)
    return out


def _encode_entry(dtype_enum, shape, shard_id, offset, size, crc_masked):
    sh = _encode_shape(shape)
    out = _field(1, 0, _put_varint(dtype_enum))
    out += _field(2, 2, _put_varint(len(sh)) + sh)
    if shard_id:
        out += _field(3, 0, _put_varint(shard_id))
    if offset:
        out += _field(4, 0, _put_varint(offset))
    out += _field(5, 0, _put_varint(size))
    out += _field(6, 5, struct.pack('<I', crc_masked))
    return out


def _encode_header(num_shards):
    version = _field(1, 0, _put_varint(1))              # VersionDef.producer = kTensorBundleVersion
    return _field(1, 0, _put_varint(num_shards)) + _field(3, 2, _put_varint(len(version)) + version)


# ---- SSTable ------------------------------------------------------------------------------------------------------------

def _read_block(buf, offset, size, what):
    raw = buf[offset:offset + size + _BLOCK_TRAILER]
    if len(raw) != size + _BLOCK_TRAILER:
        raise ValueError('{}: truncated table block at {}'.format(what, offset))
    ctype = raw[size]
    want = struct.unpack_from('<I', raw, size + 1)[0]
    if mask_crc(crc32c(raw[:size + 1])) != want:
        raise ValueError('{}: block checksum mismatch at offset {}'.format(what, offset))
    if ctype != 0:
        raise ValueError('{}: compressed table block (type {}); Saver checkpoints are written uncompressed'.format(what, ctype))
    return raw[:size]


def _iter_block(block):
    """entries of one table block in order: (key bytes, value bytes)."""
    if len(block) < 4:
        raise ValueError('table block too small')
    num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise ValueError('bad restart array')
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key):
            raise ValueError('bad key prefix length')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _read_table(path):
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < _FOOTER_LEN:
        raise ValueError('{}: too short for a table'.format(path))
    footer = buf[-_FOOTER_LEN:]
    if struct.unpack_from('<Q', footer, 40)[0] != _TABLE_MAGIC:
        raise ValueError('{}: not an SSTable (bad magic number)'.format(path))
    pos = 0
    _, pos = _get_varint(footer, pos)                   # metaindex handle: offset, size
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries = OrderedDict()
    for _, handle in _iter_block(_read_block(buf, ioff, isize, path)):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        for k, v in _iter_block(_read_block(buf, boff, bsize, path)):
            entries[k] = v
    return entries


class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''
        self.empty = True

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            n = min(len(key), len(self.last_key))
            while shared < n and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1
        self.empty = False

    def finish(self):
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts)
        return out + struct.pack('<I', len(self.restarts))

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _write_table(path, items, block_size=262144):
    """items: sorted list of (key bytes, value bytes)."""
    out = bytearray()

    def emit(block):
        handle = _put_varint(len(out)) + _put_varint(len(block))
        out.extend(block)
        out.extend(b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return handle

    index = _BlockBuilder(1)
    cur = _BlockBuilder(_RESTART_INTERVAL)
    for key, value in items:
        cur.add(key, value)
        if cur.size_estimate() >= block_size:
            index.add(cur.last_key, emit(cur.finish()))
            cur = _BlockBuilder(_RESTART_INTERVAL)
    if not cur.empty:
        index.add(cur.last_key, emit(cur.finish()))
    meta_handle = emit(_BlockBuilder(1).finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _TABLE_MAGIC)
    out.extend(footer)
    with open(path, 'wb') as f:
        f.write(out)


# ---- bundle -------------------------------------------------------------------------------------------------------------

def _data_path(prefix, shard, num_shards):
    return '{}.data-{:05d}-of-{:05d}'.format(prefix, shard, num_shards)


def list_variables(prefix):
    """-> OrderedDict name -> (numpy dtype, shape), in key order."""
    out = OrderedDict()
    for k, v in _read_table(prefix + '.index').items():
        if k == b'':
            continue
        e = _parse_entry(v)
        if e['dtype'] not in _DTYPES:
            raise ValueError('{}: unsupported dtype enum {}'.format(k.decode(), e['dtype']))
        out[k.decode('utf-8')] = (np.dtype(_DTYPES[e['dtype']]), e['shape'])
    return out


def read_bundle(prefix, names=None, skip=(), verify=False):
    """Reads the checkpoint `prefix` (path without .index / .data-*).
    names: only these variables (exact names); skip: substrings of names to leave out (saver.py:28-37 semantics);
    verify: also check every tensor's CRC-32C (slow in pure Python, off by default).
    -> OrderedDict name -> np.ndarray in the stored (TF) layout."""
    table = _read_table(prefix + '.index')
    if b'' not in table:
        raise ValueError('{}.index: no bundle header'.format(prefix))
    num_shards, endianness = 1, 0
    for field, _, v in _parse_proto(table[b'']):
        if field == 1:
            num_shards = v
        elif field == 2:
            endianness = v
    if endianness != 0:
        raise ValueError('big-endian bundle')
    shards = {}
    out = OrderedDict()
    for k, v in table.items():
        if k == b'':
            continue
        name = k.decode('utf-8')
        if names is not None and name not in names:
            continue
        if any(s in name for s in skip):
            continue
        e = _parse_entry(v)
        if e['slices']:
            raise ValueError('{}: partitioned (sliced) variables are not supported'.format(name))
        if e['dtype'] not in _DTYPES:
            raise ValueError('{}: unsupported dtype enum {}'.format(name, e['dtype']))
        dt = np.dtype(_DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise ValueError('{}: entry size {} does not match shape {} of {}'.format(name, e['size'], e['shape'], dt))
        if e['shard_id'] not in shards:
            shards[e['shard_id']] = np.memmap(_data_path(prefix, e['shard_id'], num_shards), dtype=np.uint8, mode='r')
        raw = shards[e['shard_id']][e['offset']:e['offset'] + e['size']]
        if raw.size != e['size']:
            raise ValueError('{}: data shard too short'.format(name))
        if verify and e['crc32c'] is not None and mask_crc(crc32c(raw.tobytes())) != e['crc32c']:
            raise ValueError('{}: tensor checksum mismatch'.format(name))
        out[name] = np.frombuffer(raw.tobytes(), dtype=dt).reshape(e['shape']).copy()
    if names is not None:
        missing = [n for n in names if n not in out]
        if missing:
            raise KeyError('not in checkpoint {}: {}'.format(prefix, ', '.join(missing[:5])))
    return out


def write_bundle(prefix, tensors):
    """Writes {name: array} as a single-shard V2 checkpoint a tf.train.Saver can restore."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = [(b'', _encode_header(1))]
    offset = 0
    with open(_data_path(prefix, 0, 1), 'wb') as f:
        for name in sorted(tensors, key=lambda n: n.encode('utf-8')):
            a = np.asarray(tensors[name], order='C')             # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_ENUM:
                raise ValueError('{}: dtype {} has no checkpoint representation here'.format(name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode('utf-8'),
                          _encode_entry(_DTYPE_ENUM[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    _write_table(prefix + '.index', items)


# ---- the reference's checkpoint-directory conventions (saver.py, restore_manager.py) ------------------------------------

def ckpt_dir_for_log_dir(log_dir):
    return os.path.join(log_dir, _CKPT_DIR_NAME)


def log_dir_for_restore(restore):
    """--restore_continue: the log dir a restore path belongs to (saver.py `log_dir_from_ckpt_dir`: the ckpts/ dir's parent).
    `restore` may be the ckpts/ dir, the log dir itself, or a checkpoint prefix / file inside ckpts/."""
    restore = os.path.normpath(restore)
    if os.path.isdir(restore):
        return os.path.dirname(restore) if os.path.basename(restore) == _CKPT_DIR_NAME else restore
    d = os.path.dirname(restore)
    return os.path.dirname(d) if os.path.basename(d) == _CKPT_DIR_NAME else d


def iteration_of_checkpoint(ckpt_path):
    m = re.search(r'-(\d+)', os.path.basename(ckpt_path))        # saver.py:130-135
    assert m is not None, 'Expected -(\\d+), got {}'.format(ckpt_path)
    return int(m.group(1))


def all_ckpts_with_iterations(ckpt_dir):
    """saver.py:115-142: every file whose name contains 'ckpt', extension stripped -> sorted (iteration, prefix)."""
    prefixes = set(os.path.join(ckpt_dir, os.path.splitext(fn)[0]) for fn in os.listdir(ckpt_dir)
                   if _CKPT_FN in fn and re.search(r'-(\d+)', fn) and fn.endswith(('.index', '.npz')))
    return sorted((iteration_of_checkpoint(p), p) for p in prefixes)


def latest_checkpoint_before_itr(ckpt_dir, itr=-1):
    """saver.py:102-127: the newest checkpoint, or the newest one with iteration <= itr."""
    allc = all_ckpts_with_iterations(ckpt_dir)
    if not allc:
        raise ValueError('no checkpoints in {}'.format(ckpt_dir))
    if itr == -1:
        return allc[-1]
    for it, p in reversed(allc):
        if itr >= it:
            return it, p
    raise ValueError('*** Cannot find ckpt with iter <= {} in {}'.format(itr, allc))


def read_var_names(ckpt_dir, skip_var_names=None):
    """var_names.pkl (saver.py:19-43): list of 'name:0' -> names without the output suffix."""
    with open(os.path.join(ckpt_dir, 'var_names.pkl'), 'rb') as f:
        all_v = pickle.load(f)
    skip = skip_var_names or []
    return [re.sub(r':\d+$', '', v) for v in all_v if not any(s in v for s in skip)]


def write_var_names(ckpt_dir, names):
    with open(os.path.join(ckpt_dir, 'var_names.pkl'), 'wb') as f:
        pickle.dump([n + ':0' for n in names], f)


def is_model_variable(name):
    """what the hot path needs from a training checkpoint: no optimiser slots, counters or summaries."""
    if not (name.startswith('autoencoder/') or name.startswith('probclass3d/')):
        return False
    return not re.search(r'/(Adam\w*|ExponentialMovingAverage|Momentum)$', name) and not re.search(r'beta[12]_power(_\d+)?$', name)


def is_training_state(name):
    """what continuing a training run needs on top of the model variables (the reference's Saver restores every variable
    of the graph: restore_manager.py:52-58): the step counter and the two Adam optimisers' slots and beta powers."""
    return name == 'global_step' or bool(re.search(r'/(Adam_AE|Adam_AE_1|Adam_PC|Adam_PC_1)$', name)) or \
        bool(re.match(r'beta[12]_power(_1)?$', name)) or \
        bool(re.match(r'(Adam_AE|Adam_PC)/beta[12]_power$', name))          # TF's names; second form: files of round-2 builds


def load_weights(path, itr=-1, training_state=False):
    """path: a checkpoint prefix, a .index file, a ckpts/ directory or a log dir containing one (restore_manager.py:52-58),
    or an .npz written by this package's train.py.  -> dict name -> array with the variables of the two networks
    (training_state=True: plus global_step and the optimiser slots when the checkpoint has them)."""
    keep = (lambda n: is_model_variable(n) or is_training_state(n)) if training_state else is_model_variable
    if path.endswith('.npz') and os.path.isfile(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files if keep(k)}
    if os.path.isdir(path):
        ckpt_dir = path if os.path.basename(os.path.normpath(path)) == _CKPT_DIR_NAME else ckpt_dir_for_log_dir(path)
        if not os.path.isdir(ckpt_dir):
            raise ValueError('Invalid ckpt dir: {}'.format(path))
        _, prefix = latest_checkpoint_before_itr(ckpt_dir, itr)
        if os.path.isfile(prefix + '.npz') and not os.path.isfile(prefix + '.index'):
            return load_weights(prefix + '.npz', training_state=training_state)
    else:
        prefix = path[:-len('.index')] if path.endswith('.index') else path
    if not os.path.isfile(prefix + '.index'):
        raise ValueError('Invalid ckpt dir: {}'.format(path))                   # restore_manager.py:52-58
    names = [n for n in list_variables(prefix) if keep(n)]
    return dict(read_bundle(prefix, names=names))
