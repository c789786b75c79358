"""HIP streams for the two branches that hang off the encoder output.

The reference evaluates bitcost and reconstruction in one session.run (val.py:85-89); they share nothing but the encoder
output.  On the MI355X the decoder is a chain of 3x3 launches that keep ONE 512-register work-group per CU and -- for a
Kodak-sized feature map -- occupy 192 of the 256 CUs.  A context-model work-group that lands on one of those CUs takes
registers the next 3x3 work-group needs, which then waits for the whole SIMD: a plain second stream buys ~1 %.  A stream
restricted to the CUs the decoder leaves idle (ic_stream_create_cu_range) removes the interference and hides the
context model completely (round-1 A/B: 3.15 -> 2.98 ms per Kodak image).

    bs = BranchStreams(device)
    with torch.cuda.stream(bs.main):                 # CU-range streams are blocking w.r.t. the legacy default stream
        enc = ae.encode(x)
        side = bs.context_model_stream(N, H, W)
        side.wait_stream(bs.main)
        with torch.cuda.stream(side): ... pc.bitcost(...)
        x_out = ae.decode(enc.qhard, False, plan_flags=bs.decode_flags(side))     # per-call: no library state is touched
        bs.main.wait_stream(side)

Which arrangement is used is a constructor argument (`share`):
  'serial' (the default, alias 'auto') -- no second stream: the context model runs ahead of the decoder on the caller's stream
             and every launch of both takes the whole chip;
  'cu_range' -- as above: the context model on the CUs the decoder's one-work-group-per-CU launches leave idle;
  'full_chip' -- the decoder's 3x3 launches take the form that fills the chip and the context model runs on a plain second
             stream, filling whatever the decoder's launch boundaries leave.
Round 1 measured 'cu_range' 5 % ahead (the context model took 0.24 ms, the full-chip 3x3 form did not exist).  At the end of
round 2 -- context model 0.135 ms, NB-segment 3x3 jobs that fill the chip for any map of >= 32 tile groups -- 'serial' is
level on a Kodak image (157.3 / 157.8 against 157.0 / 157.9 Mpix/s, then 158.7 against 155.8) and AHEAD on a 256 x 256 image
(65.4 against 60.9 Mpix/s: next to a CU-range stream the decoder has to keep to the forms that stay off those CUs).  The other
two stay available for callers whose side branch is heavier than this context model.
"""
import ctypes

import torch

from . import _lib


DEFAULT_SHARE = 'serial'
DEFAULT_IDLE_LAYERS = None       # None = sized from the context model's work (BranchStreams.auto_idle_layers); 0 = the whole stack


class BranchStreams(object):
    MIN_CUS = 32          # fewer than this and the context model becomes the critical path of a Kodak-sized image

    def __init__(self, device, share=None, idle_layers=None):
        share = share or DEFAULT_SHARE
        # 3x3 launches of a decode call that leave the side stream's CUs alone; the rest of the stack takes the whole chip
        # again (the context model is done long before the decoder: 0.4 ms of a 1.3 ms decode on a Kodak-sized image)
        self.idle_layers = DEFAULT_IDLE_LAYERS if idle_layers is None else int(idle_layers)
        self._auto_layers = 0
        share = 'serial' if share == 'auto' else share
        assert share in ('serial', 'cu_range', 'full_chip')
        self.share = share
        self.device = torch.device(device)
        self.main = torch.cuda.Stream(device=self.device)
        self._plain = torch.cuda.Stream(device=self.device)
        self._ranged = {}
        self._handles = []
        self.n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count

    def idle_cus(self, N, H, W):
        """CUs the decoder's 3x3 launches leave idle for an (N, 3, H, W) image, rounded down to whole CUs per XCD."""
        if self.share != 'cu_range':
            return 0
        # the plan the decoder runs when it is asked to leave its idle CUs alone
        wgs = int(_lib.lib.ic_wino3x3_c128_workgroups(N, H // 4, W // 4, _lib.CONV3_LEAVE_IDLE_CUS))
        if wgs <= 0 or wgs >= self.n_cus:
            return 0
        return min((self.n_cus - wgs) // 8 * 8, self.n_cus // 2)

    # Measured on the MI355X (bench.py --idle_layers sweep, Kodak image): the context model's 196,608 symbols take ~0.53 ms on
    # 64 CUs = 2.7 ns per symbol per 64 CUs, a one-work-group-per-CU 3x3 launch 36.4 us.  The decoder leaves the side
    # stream's CUs alone for that many of its 3x3 launches and takes the whole chip for the rest (sweep at the round's end:
    # 13 launches 158.2 Mpix/s, 14: 158.5 / 157.9, 15: 158.1 / 157.9, 16: 157.3; with too few the context model is not done
    # and the step waits for it -- 12 launches cost 4 Mpix/s with the round's first kernels).
    NS_PER_SYMBOL_64CU = 2.7
    US_PER_LAYER = 36.4

    def auto_idle_layers(self, N, H, W, n_cus, C=32):
        import math
        symbols = N * C * (H // 8) * (W // 8)
        t_pc_us = symbols * self.NS_PER_SYMBOL_64CU * 1e-3 * 64.0 / max(n_cus, 1)
        n = int(math.ceil(t_pc_us / self.US_PER_LAYER))
        return 0 if n >= 60 else max(n, 4)

    def context_model_stream(self, N, H, W, C=32):
        n = self.idle_cus(N, H, W)
        self._auto_layers = self.auto_idle_layers(N, H, W, n, C) if n >= self.MIN_CUS else 0
        if self.share == 'serial':
            return self.main             # no second stream: the context model runs ahead of the decoder, both on the whole chip
        if n < self.MIN_CUS:
            return self._plain           # the decoder fills the chip in rounds: nothing to partition
        if n not in self._ranged:
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                rc = _lib.lib.ic_stream_create_cu_range(self.n_cus - n, n, ctypes.byref(h))
            if rc != 0:                  # runtime without CU masks: fall back to the plain side stream
                self._ranged[n] = self._plain
            else:
                self._handles.append(h)
                self._ranged[n] = torch.cuda.ExternalStream(h.value, device=self.device)
        return self._ranged[n]

    def decode_flags(self, side):
        """plan flags for the ae.decode call that shares the chip with `side`: next to a CU-range stream its partly filled 3x3
        rounds stay one work-group per CU (the idle CUs untouched) instead of being spread over every CU."""
        if side is self._plain or side is self.main:
            return 0
        layers = self._auto_layers if self.idle_layers is None else self.idle_layers
        return _lib.CONV3_LEAVE_IDLE_CUS | _lib.conv3_leave_idle_layers(layers)

    def close(self):
        for h in self._handles:
            _lib.lib.ic_stream_destroy(h)
        self._handles, self._ranged = [], {}
