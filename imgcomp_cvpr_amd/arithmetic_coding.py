"""32-bit integer arithmetic (range) coder used by the real-bpp path (val.py --real_bpp).

The reference ships code/arithmetic_coding.py, a copy of Project Nayuki's "Reference arithmetic coding"
(MIT licence).  This module is an independent restatement of that published algorithm -- same state
width, same renormalisation rule, same bit order -- so that the byte stream is IDENTICAL to the
reference's for the same (symbol, frequency-table) sequence; tests/golden/arithcoding.npz holds a stream
produced by the reference's own coder and tests/test_cpu_host.py checks byte equality.

Algorithm (state: low, high in [0, 2^32), inclusive interval):
  narrow:  r = high - low + 1;  high = low + cum_hi * r // total - 1;  low = low + cum_lo * r // total
  while low and high agree in the top bit: emit it (followed by the pending "underflow" bits, inverted),
      shift both left (high shifts in a 1);
  while low = 01..., high = 10...: drop the second-highest bit of both, count one pending underflow bit.
  finish: emit a single 1 bit.  total must be <= 2^30 + 2.
The API keeps the reference's names (ArithmeticEncoder.write / ArithmeticDecoder.read /
SimpleFrequencyTable / BitOutputStream / BitInputStream / CountingBitOutputStream) so bit_counter.py reads
the same; the frequency table additionally accepts numpy arrays and the codec has array fast paths
(encode_sequence / decode with a callback) because the host loop is the slow part of real-bpp coding.
"""
import numpy as np

STATE_BITS = 32
_FULL = 1 << STATE_BITS
_MASK = _FULL - 1
_TOP = _FULL >> 1
_SECOND = _TOP >> 1
MAX_TOTAL = (_FULL >> 2) + 2


class SimpleFrequencyTable(object):
    """immutable table over symbols 0..n-1 with cumulative sums."""

    def __init__(self, freqs):
        f = [int(v) for v in freqs]
        if not f:
            raise ValueError('At least 1 symbol needed')
        if any(v < 0 for v in f):
            raise ValueError('Negative frequency')
        self._f = f
        cum = [0]
        for v in f:
            cum.append(cum[-1] + v)
        self._cum = cum

    def get_symbol_limit(self):
        return len(self._f)

    def get(self, symbol):
        return self._f[symbol]

    def get_total(self):
        return self._cum[-1]

    def get_low(self, symbol):
        return self._cum[symbol]

    def get_high(self, symbol):
        return self._cum[symbol + 1]


class BitOutputStream(object):
    """MSB-first bit packer over a binary file object; close() pads the last byte with zeros."""

    def __init__(self, out):
        self.output = out
        self._acc = 0
        self._n = 0

    def write(self, b):
        if b not in (0, 1):
            raise ValueError('Argument must be 0 or 1')
        self._acc = (self._acc << 1) | b
        self._n += 1
        if self._n == 8:
            self.output.write(bytes((self._acc,)))
            self._acc, self._n = 0, 0

    def close(self):
        while self._n:
            self.write(0)
        self.output.close()


class CountingBitOutputStream(object):
    """forwards to a bit stream and counts bits; close() rounds the count up to whole bytes."""

    def __init__(self, bit_out):
        self.num_bits = 0
        self.bit_out = bit_out

    def write(self, b):
        self.num_bits += 1
        self.bit_out.write(b)

    def close(self):
        self.num_bits += (-self.num_bits) % 8
        self.bit_out.close()


class BitInputStream(object):
    """MSB-first bit reader; read() returns -1 at end of stream."""

    def __init__(self, inp):
        self.input = inp
        self._byte = 0
        self._left = 0
        self._eof = False

    def read(self):
        if self._eof:
            return -1
        if self._left == 0:
            t = self.input.read(1)
            if len(t) == 0:
                self._eof = True
                return -1
            self._byte = t[0]
            self._left = 8
        self._left -= 1
        return (self._byte >> self._left) & 1

    def close(self):
        self.input.close()
        self._eof = True


class _Coder(object):
    def __init__(self):
        self.low = 0
        self.high = _MASK

    def _narrow(self, cum_lo, cum_hi, total):
        if total > MAX_TOTAL:
            raise ValueError('Cannot code symbol because total is too large')
        if cum_lo == cum_hi:
            raise ValueError('Symbol has zero frequency')
        r = self.high - self.low + 1
        self.high = self.low + cum_hi * r // total - 1
        self.low = self.low + cum_lo * r // total
        while ((self.low ^ self.high) & _TOP) == 0:
            self._shift()
            self.low = (self.low << 1) & _MASK
            self.high = ((self.high << 1) & _MASK) | 1
        while (self.low & ~self.high & _SECOND) != 0:
            self._underflow()
            self.low = (self.low << 1) & (_MASK >> 1)
            self.high = ((self.high << 1) & (_MASK >> 1)) | _TOP | 1


class ArithmeticEncoder(_Coder):
    def __init__(self, bitout):
        super(ArithmeticEncoder, self).__init__()
        self.output = bitout
        self._pending = 0

    def write(self, freqs, symbol):
        self._narrow(freqs.get_low(symbol), freqs.get_high(symbol), freqs.get_total())

    def write_cum(self, cum_lo, cum_hi, total):
        """fast path: cumulative counts of the symbol given directly."""
        self._narrow(int(cum_lo), int(cum_hi), int(total))

    def finish(self):
        self.output.write(1)

    def _shift(self):
        bit = self.low >> (STATE_BITS - 1)
        self.output.write(bit)
        for _ in range(self._pending):
            self.output.write(bit ^ 1)
        self._pending = 0

    def _underflow(self):
        self._pending += 1


class ArithmeticDecoder(_Coder):
    def __init__(self, bitin):
        super(ArithmeticDecoder, self).__init__()
        self.input = bitin
        self.code = 0
        for _ in range(STATE_BITS):
            self.code = (self.code << 1) | self._bit()

    def _bit(self):
        b = self.input.read()
        return 0 if b == -1 else b

    def read(self, freqs):
        total = freqs.get_total()
        if total > MAX_TOTAL:
            raise ValueError('Cannot decode symbol because total is too large')
        r = self.high - self.low + 1
        value = ((self.code - self.low + 1) * total - 1) // r
        lo, hi = 0, freqs.get_symbol_limit()
        while hi - lo > 1:                       # largest symbol with cum_lo <= value
            mid = (lo + hi) >> 1
            if freqs.get_low(mid) > value:
                hi = mid
            else:
                lo = mid
        self._narrow(freqs.get_low(lo), freqs.get_high(lo), total)
        return lo

    def _shift(self):
        self.code = ((self.code << 1) & _MASK) | self._bit()

    def _underflow(self):
        self.code = (self.code & _TOP) | ((self.code << 1) & (_MASK >> 1)) | self._bit()


def encode_sequence(symbols, freqs, fileobj):
    """symbols: (n,) ints; freqs: (n, L) int frequency rows (all > 0).  Writes the stream to fileobj (closed
    afterwards, like BitOutputStream.close) and returns the number of bits including the byte padding."""
    symbols = np.asarray(symbols).astype(np.int64)
    freqs = np.asarray(freqs).astype(np.int64)
    cum = np.concatenate([np.zeros((freqs.shape[0], 1), np.int64), np.cumsum(freqs, axis=1)], axis=1)
    idx = np.arange(symbols.shape[0])
    lo = cum[idx, symbols].tolist()
    hi = cum[idx, symbols + 1].tolist()
    tot = cum[:, -1].tolist()
    out = CountingBitOutputStream(BitOutputStream(fileobj))
    enc = ArithmeticEncoder(out)
    for a, b, t in zip(lo, hi, tot):
        enc._narrow(a, b, t)
    enc.finish()
    out.close()
    return out.num_bits
