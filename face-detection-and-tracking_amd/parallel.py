"""Frame-parallel sharding of the detect+track path.

Detection has no cross-frame state (reference pyramid.py:218-351, detection.py:52), so frame
`f = step * world + rank` goes to GPU `rank`; the IoU tracker is strictly sequential over frames
(reference iouTracke_cal.py:117-156) but consumes only the fixed-size Detect record of each frame.
The one exchange step of the path is therefore ONE all-gather per step of `[num_classes*top_k*5]`
f32 per rank (15 KB; latency-bound over xGMI), after which every rank runs the association for the
`world` frames of the step in rank order == frame order.  No other collective exists on the path.

One process per GPU, `torch.distributed` (backend "nccl" is RCCL on ROCm; "gloo" for CPU tests).
"""
import torch
import torch.distributed as dist


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

    def exchange(self):
        """All-gather `self.mine` of every rank into `self.gathered` ([world, record], frame order)."""
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered.view(-1), self.mine)
        return self.gathered

    def frames_of_step(self, step):
        return [step * self.world + r for r in range(self.world)]
