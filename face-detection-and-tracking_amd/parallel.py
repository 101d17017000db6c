"""Frame-parallel sharding of the detect+track path.

Detection has no cross-frame state (reference pyramid.py:218-351, detection.py:52), so frame
`f = step * world + rank` goes to GPU `rank`; the IoU tracker is strictly sequential over frames
(reference iouTracke_cal.py:117-156) but consumes only the fixed-size Detect record of each frame.
The one exchange step of the path is therefore ONE all-gather per step of `[num_classes*top_k*5]`
f32 per rank (15 KB; latency-bound over xGMI), after which every rank runs the association for the
`world` frames of the step in rank order == frame order.  No other collective exists on the path.

One process per GPU.  The collective itself is RCCL behind the C ABI (`fdt_comm_*`, `fdt_allgather_dets`,
include/fdt.h): `RcclExchange`.  torch.distributed is used only as the host channel that ships rank 0's
128-byte RCCL id to the other ranks.  `FrameParallel` is the torch.distributed form of the same exchange,
kept for the CPU (gloo) tests and for rehearsing several ranks on one GPU.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class FrameParallel:
    def __init__(self, rank=0, world=1, record_floats=2 * 750 * 5, device=None):
        self.rank, self.world, self.record_floats = rank, world, record_floats
        self.device = device if device is not None else torch.device("cpu")
        self.gathered = torch.zeros((world, record_floats), dtype=torch.float32, device=self.device)
        # with one rank the local record *is* the gathered row: no copy, no collective
        self.mine = self.gathered[0] if world == 1 else torch.zeros(record_floats, dtype=torch.float32,
                                                                     device=self.device)

    def frame_of(self, step):
        """Global frame index this rank detects in `step`."""
        return step * self.world + self.rank

    def exchange(self, stream=None):
        """All-gather `self.mine` of every rank into `self.gathered` ([world, record], frame order)."""
        if self.world > 1:
            if self.device.type == "cuda" and dist.get_backend() == "gloo":
                # rehearsal of N ranks on one GPU: gloo moves host memory
                host = self.mine.cpu()
                out = torch.empty((self.world, self.record_floats), dtype=torch.float32)
                dist.all_gather_into_tensor(out.view(-1), host)
                self.gathered.copy_(out)
            else:
                dist.all_gather_into_tensor(self.gathered.view(-1), self.mine)
        return self.gathered

    def frames_of_step(self, step):
        return [step * self.world + r for r in range(self.world)]


def make_rccl_comm(rank, world, device_index):
    """fdt_comm_init_rank with the id broadcast over the already-initialised torch.distributed group (any backend:
    it only carries 128 bytes of host data)."""
    L = _lib.lib()
    buf = C.create_string_buffer(128)
    if rank == 0:
        _lib.check(L.fdt_comm_unique_id(buf))
    if world > 1:
        obj = [bytes(buf.raw)]
        dist.broadcast_object_list(obj, src=0)
        buf = C.create_string_buffer(obj[0], 128)
    h = L.fdt_comm_init_rank(world, rank, buf, device_index)
    if not h:
        raise _lib.FdtError(_lib.FDT_ERR_HIP, (L.fdt_last_error() or b"").decode())
    return h


class RcclExchange(FrameParallel):
    """Same buffers as FrameParallel; the all-gather is `fdt_allgather_dets` (RCCL through the C ABI) enqueued on
    the caller's HIP stream."""

    def __init__(self, rank, world, record_floats, device, comm=None):
        super().__init__(rank, world, record_floats, device)
        if world > 1 and self.mine.data_ptr() == self.gathered.data_ptr():
            raise AssertionError
        self._own = comm is None
        self._comm = comm if comm is not None else make_rccl_comm(rank, world, device.index or 0)

    def exchange(self, stream=None):
        if self.world > 1:
            _lib.check(_lib.lib().fdt_allgather_dets(self._comm, 0, C.c_void_p(self.mine.data_ptr()),
                                                     C.c_void_p(self.gathered.data_ptr()), self.record_floats, stream))
        return self.gathered

    def close(self):
        if self._own and self._comm:
            _lib.lib().fdt_comm_destroy(self._comm)
        self._comm = None
