"""Peer-to-peer SyncBatchNorm statistics exchange between the ranks of one node (csrc/p2p.hip).

Opt-in replacement (SM3_SYNCBN_P2P=1) for the torch.distributed all-reduce behind the reference's
`nn.SyncBatchNorm.convert_sync_batchnorm` (tools/backbone_train.py:510): one kernel on the calling stream per exchange, no
cross-stream event, the result bit-identical on every rank.  One mailbox set per execution lane (each lane has its own
stream and therefore its own exchange sequence).  torch.distributed is used once, to pass the hipIpc handles around."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from .ops import check, _stream


class P2PStatSync:
    def __init__(self, lanes, device, timeout_s=20.0):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("P2PStatSync needs an initialised torch.distributed process group (to exchange the handles)")
        lib = _lib.load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.max_elems = lib.sm3_p2p_max_elems()
        self.timeout_s = float(timeout_s)
        self.device = device
        self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()  # poll(): last value copied back, no host wait
        self._err_event = None
        self._own, self._peers, self._boxes, self._seq = {}, {}, {}, {}
        kinds = set()
        with torch.cuda.device(device):
            handles = {}
            for lane in lanes:
                ptr = C.c_void_p()
                h = (C.c_ubyte * 64)()
                kind = C.c_int(-1)
                check(lib.sm3_p2p_alloc(C.byref(ptr), h, C.byref(kind)), "sm3_p2p_alloc")
                self._own[lane] = ptr.value
                handles[lane] = bytes(h)
                kinds.add(kind.value)
        # what the mailboxes of THIS rank are made of: "finegrained" (hipExtMallocWithFlags, as RCCL's IPC buffers),
        # "coarse" (plain hipMalloc: the fallback, or SM3_P2P_FINEGRAINED=0) -- bench.py records it next to its numbers
        self.memory_kind = "finegrained" if kinds == {1} else ("coarse" if kinds == {0} else "mixed")
        with torch.cuda.device(device):
            gathered = [None] * self.world
            dist.all_gather_object(gathered, handles)  # every rank's {lane: handle}
            for lane in lanes:
                arr = (C.c_void_p * self.world)()
                opened = []
                for r in range(self.world):
                    if r == self.rank:
                        arr[r] = self._own[lane]
                        continue
                    p = C.c_void_p()
                    hb = (C.c_ubyte * 64).from_buffer_copy(gathered[r][lane])
                    check(lib.sm3_p2p_open(hb, C.byref(p)), "sm3_p2p_open")
                    arr[r] = p.value
                    opened.append(p.value)
                self._boxes[lane] = arr
                self._peers[lane] = opened
                self._seq[lane] = 0
        dist.barrier()  # every mailbox is mapped everywhere before the first exchange

    def __call__(self, lane, t):
        """t (fp64, contiguous, on this device) <- sum over ranks, in place, on the current stream."""
        if t.dtype != torch.float64 or not t.is_contiguous() or t.numel() > self.max_elems:
            raise ValueError("P2PStatSync: fp64 contiguous tensors of at most sm3_p2p_max_elems() elements")
        self._seq[lane] += 1
        check(_lib.load().sm3_p2p_allreduce_f64(t.data_ptr(), t.numel(), self._boxes[lane], self.rank, self.world,
                                                self._seq[lane], self.err.data_ptr(), self.timeout_s, _stream()),
              "sm3_p2p_allreduce_f64")

    _MSG = ("P2PStatSync: a peer's statistics did not arrive in time (rank died or ranks out of step); every exchange "
            "since then returned NaN.  SM3Trainer skipped the optimizer update of every step whose gradients were poisoned "
            "(parameters and AdamW moments are those of the last good step; BatchNorm running statistics are not)")

    def check(self):
        """Raise if any exchange so far timed out (synchronises)."""
        if int(self.err.item()):
            raise RuntimeError(self._MSG)

    def poll(self):
        """The same without a host wait: raises if the flag copied back by the PREVIOUS poll() reads 1, then enqueues a new
        copy on the current stream.  A trainer calls it once per step, so a failed exchange raises one step late at the
        latest (the step's loss is NaN in the meantime: the kernel poisons the sums)."""
        if self._err_event is not None and self._err_event.query() and int(self._err_host[0]):
            raise RuntimeError(self._MSG)
        if self._err_event is None or self._err_event.query():
            self._err_host.copy_(self.err, non_blocking=True)
            self._err_event = torch.cuda.Event()
            self._err_event.record()

    def close(self, barrier=True):
        """Unmap the peers' mailboxes and free one's own.  barrier=False: at interpreter teardown or after a failed
        exchange, when the peers can no longer be counted on to arrive."""
        if not self._own:
            return
        lib = _lib.load()
        torch.cuda.synchronize(self.device)
        if barrier and dist.is_initialized():
            dist.barrier()  # nobody unmaps while a peer may still write
        for lane, opened in self._peers.items():
            for p in opened:
                lib.sm3_p2p_close(p)
        for p in self._own.values():
            lib.sm3_p2p_free(p)
        self._peers, self._own = {}, {}
