"""Cross-replica sums over peer-mapped device memory (csrc/peer_exchange.hip) -- the exchange under cross-replica BatchNorm.

The reference has no multi-GPU code; BASELINE.json's north_star splits one batch over the 8 GPUs of a node.  BatchNorm then
needs the per-channel moments of ALL ranks once per layer and direction (training.py): 140 exchanges of at most 2 KB per step,
each on the critical path.  `PeerExchange.allreduce_f64(t)` does one in ONE small launch: every rank stores its values into
every peer's region (xGMI stores), raises a flag, waits for the flags in its own region and sums in rank order (bit-identical
on all ranks).  RCCL stays the fallback (`sync_bn=True`): this path is opt-in (`sync_bn='p2p'`).

Set-up (once): each rank creates a fine-grained region, the 64-byte inter-process handles travel through the process group as
Python objects, every rank opens the others'.  One process per GPU on one node (hipIpc); world size <= 8."""
import ctypes

import torch

from . import _lib
from ._lib import lib, check, ptr


class PeerExchange(object):
    def __init__(self, device, process_group=None, spin_limit=0):
        """spin_limit: polls of one peer flag before an exchange gives up and raises the status word (0 = the library's default,
        a few seconds -- longer than any rank-0-only pause train.py does not bracket with a barrier)"""
        import torch.distributed as dist
        self.spin_limit = int(spin_limit)
        if not (dist.is_available() and dist.is_initialized()):
            raise _lib.HipLibraryError('PeerExchange needs an initialised process group (the handles travel through it)')
        self.pg = process_group
        self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        if self.world > lib.ic_peer_max_world():
            raise _lib.HipLibraryError('PeerExchange: world size {} > {}'.format(self.world, lib.ic_peer_max_world()))
        self.dev = torch.device(device)
        self.max_values = int(lib.ic_peer_max_values())
        with torch.cuda.device(self.dev):
            own = ctypes.c_void_p()
            handle = (ctypes.c_ubyte * 64)()
            check(lib.ic_peer_region_create(ctypes.byref(own), handle), 'ic_peer_region_create')
            self._own = own
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle), group=process_group)
            self._mapped = []
            regions = (ctypes.c_void_p * self.world)()
            for r in range(self.world):
                if r == self.rank:
                    regions[r] = own
                    continue
                m = ctypes.c_void_p()
                buf = (ctypes.c_ubyte * 64).from_buffer_copy(handles[r])
                check(lib.ic_peer_region_open(buf, ctypes.byref(m)), 'ic_peer_region_open (rank {})'.format(r))
                self._mapped.append(m)
                regions[r] = m
            self._regions = regions
        self._status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.seq = 0
        dist.barrier(group=process_group)            # nobody pushes before every region is mapped everywhere

    def allreduce_f64(self, t):
        """t: contiguous float64 device tensor of <= max_values elements; summed over the ranks in place (rank order)."""
        assert t.dtype == torch.float64 and t.is_contiguous() and t.numel() <= self.max_values
        self.seq += 1
        check(lib.ic_peer_allreduce_f64_bounded(ptr(t), t.numel(), self._regions, self.rank, self.world, self.seq & 0xffffffff or 1,
                                                self.spin_limit, ptr(self._status), _lib.current_stream(self.dev)),
              'ic_peer_allreduce_f64')
        return t

    def check_status(self):
        """synchronises; raises if a peer's contribution did not arrive in time since the last check"""
        if int(self._status.item()) != 0:
            self._status.zero_()
            raise _lib.HipLibraryError('PeerExchange: a peer exchange timed out (a rank is missing or stalled)')

    def close(self):
        for m in self._mapped:
            lib.ic_peer_region_close(m)
        self._mapped = []
        if self._own is not None:
            lib.ic_peer_region_destroy(self._own)
            self._own = None
