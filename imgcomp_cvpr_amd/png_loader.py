"""PNG -> padded CHW uint8, in this process or in decoder PROCESSES (reference code/images_iterator.py:28-59 reads one image per step on
the session's thread).

A Kodak-sized PNG costs a host core 8 ms (zlib + un-filtering inside PIL) and the device path 1.5 ms, so val.py decodes ahead of the
device.  Decoder THREADS (round 5) share the interpreter lock with the loop that feeds the device: 8-16 of them reach 360 images/s where
the loop alone, on decoded images, does 514.  This module is the other arrangement: `python -m
imgcomp_cvpr_amd.png_loader` is a worker that imports numpy and PIL only (no torch: 0.15 s to start), reads `pad<TAB>path` lines on stdin and
answers each with a 16-byte header (status, C, H, W as uint32) on stdout; the pixels go through a shared-memory slot per worker (a
64 KB-per-read pipe costs the parent 19 interpreter-lock round trips per Kodak image: 125 images/s) and the parent's reader thread
copies them into a buffer of its choice -- page-locked memory in val.py -- with the lock released."""
import struct
import sys

import numpy as np


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


def decode_chw_view(p, pad):
    """-> (3, H, W) uint8 VIEW (not contiguous) of the decoded, padded image"""
    from PIL import Image
    im = np.asarray(Image.open(p).convert('RGB'), dtype=np.uint8)
    im, _ = add_padding(im, pad)
    return np.transpose(im, (2, 0, 1))


_HEADER = struct.Struct('<4I')
SLOT_BYTES = 64 << 20          # one shared-memory slot per worker: any image up to 64 MB of CHW bytes (a 4K RGB frame is 25 MB)


def _serve(stdin, stdout, slot_path=None):
    """worker loop.  With a slot (a file under /dev/shm the parent created and maps too) the pixels go there and only the 16-byte
    header crosses the pipe -- a pipe hands over 64 KB per read, i.e. 19 interpreter-lock round trips of the parent per Kodak image
    (measured: 125 images/s with 8 workers, worse with more); without one, or for an image larger than the slot, header + bytes."""
    slot = None
    if slot_path:
        import mmap
        with open(slot_path, 'r+b') as f:
            slot = np.frombuffer(mmap.mmap(f.fileno(), SLOT_BYTES), dtype=np.uint8)
        stdout.write(_HEADER.pack(2, 0, 0, 0))                          # mapped: the parent may unlink the file now
        stdout.flush()
    for line in iter(stdin.readline, b''):
        pad, _, p = line.rstrip(b'\n').partition(b'\t')
        try:
            view = decode_chw_view(p.decode('utf-8', 'surrogateescape'), int(pad))
            n = int(np.prod(view.shape))
            if slot is not None and n <= SLOT_BYTES:
                np.copyto(slot[:n].reshape(view.shape), view)            # HWC -> CHW straight into the shared slot
                stdout.write(_HEADER.pack(3, *view.shape))
            else:
                chw = np.ascontiguousarray(view)
                stdout.write(_HEADER.pack(1, *chw.shape))
                stdout.write(memoryview(chw).cast('B'))
        except Exception as ex:                                        # the parent raises it with the path
            msg = '{}: {}'.format(type(ex).__name__, ex).encode('utf-8', 'replace')
            stdout.write(_HEADER.pack(0, len(msg), 0, 0))
            stdout.write(msg)
        stdout.flush()


class PngWorkers(object):
    """n decoder processes behind n reader threads.  submit(path, pad, alloc) -> Future of the CHW uint8 array; alloc(shape) returns the
    writable uint8 numpy array the pixels are copied into (default: a fresh array).  Every worker owns one shared-memory slot (a file
    under /dev/shm, unlinked as soon as both sides have mapped it: nothing is left behind whatever happens to either process); the
    reader thread copies slot -> destination (a large numpy copy releases the interpreter lock) and the slot is free again."""

    def __init__(self, n, shared=True):
        import mmap
        import os
        import queue
        import subprocess
        import tempfile
        from concurrent.futures import ThreadPoolExecutor
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env['PYTHONPATH'] = root + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
        self.procs, self.slots = [], {}
        shared = shared and os.path.isdir('/dev/shm')
        for _ in range(int(n)):
            args, slot, path = [sys.executable, '-m', 'imgcomp_cvpr_amd.png_loader'], None, None
            if shared:
                fd, path = tempfile.mkstemp(prefix='imgcomp_png_', dir='/dev/shm')
                try:
                    os.ftruncate(fd, SLOT_BYTES)                       # sparse: pages exist once touched
                    slot = np.frombuffer(mmap.mmap(fd, SLOT_BYTES), dtype=np.uint8)
                finally:
                    os.close(fd)
                args.append(path)
            try:
                pr = subprocess.Popen(args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=0, env=env)
                if path:
                    head = bytearray(_HEADER.size)
                    self._read_exact(pr.stdout, memoryview(head))      # the worker has mapped the slot
                    if _HEADER.unpack(bytes(head))[0] != 2:
                        raise IOError('PNG decoder process did not map its slot')
            finally:
                if path:
                    os.unlink(path)
            self.procs.append(pr)
            self.slots[pr.pid] = slot
        self.free = queue.Queue()
        for pr in self.procs:
            self.free.put(pr)
        self.pool = ThreadPoolExecutor(max_workers=len(self.procs))

    @staticmethod
    def _read_exact(f, view):
        got = 0
        while got < len(view):
            k = f.readinto(view[got:])
            if not k:
                raise IOError('PNG decoder process closed its pipe')
            got += k

    def _one(self, p, pad, alloc):
        pr = self.free.get()
        try:
            pr.stdin.write('{}\t'.format(int(pad)).encode() + p.encode('utf-8', 'surrogateescape') + b'\n')
            head = bytearray(_HEADER.size)
            self._read_exact(pr.stdout, memoryview(head))
            ok, c, h, w = _HEADER.unpack(bytes(head))
            if not ok:
                msg = bytearray(c)
                self._read_exact(pr.stdout, memoryview(msg))
                raise IOError('{}: {}'.format(p, msg.decode('utf-8', 'replace')))
            out = alloc((c, h, w)) if alloc else np.empty((c, h, w), np.uint8)
            if ok == 3:
                np.copyto(out, self.slots[pr.pid][:c * h * w].reshape(c, h, w))
            else:
                self._read_exact(pr.stdout, memoryview(out).cast('B'))
            return out
        finally:
            self.free.put(pr)

    def submit(self, p, pad, alloc=None):
        return self.pool.submit(self._one, p, pad, alloc)

    def close(self):
        self.pool.shutdown(wait=True)
        for pr in self.procs:
            try:
                pr.stdin.close()
                pr.stdout.close()
                pr.wait(timeout=5)
            except Exception:
                pr.kill()
        self.slots = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


if __name__ == '__main__':
    _serve(sys.stdin.buffer, sys.stdout.buffer, sys.argv[1] if len(sys.argv) > 1 else None)
