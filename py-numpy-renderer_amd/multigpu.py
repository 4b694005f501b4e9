"""Screen-tile split of one frame over the GPUs of a node (one process per GPU).

The path shards by pixels: every mutable buffer of the reference (frame, z, stencil) is
per-pixel and primitives interact only through them (obj/triangular.py:101-118,356-368), so
disjoint row bands are independent.  Geometry, textures and per-frame constants are replicated;
each rank rasterises and shades only its band of H/world output rows and ONE all-gather (RCCL
over xGMI when the backend is "nccl") assembles the uint8 frame on every rank.  Row bands in
output order make the gathered buffer the final frame with no reorder.
"""
import torch
import torch.distributed as dist


def row_band(height, rank, world):
    """Output rows ``[begin, end)`` of *rank*; bands are equal so the all-gather is regular."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    if height % world:
        raise ValueError(f"{height} rows do not split evenly over {world} ranks")
    rows = height // world
    return rank * rows, (rank + 1) * rows


def all_gather_frame(part, frame=None, group=None):
    """Assemble the frame from every rank's band with a single collective.

    *part* is this rank's ``(rows, W, 3)`` uint8 band (any device); returns ``(rows*world, W, 3)``.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if frame is None:
        frame = torch.empty((part.shape[0] * world,) + tuple(part.shape[1:]), dtype=part.dtype,
                            device=part.device)
    if world == 1:
        if frame.data_ptr() != part.data_ptr():
            frame.copy_(part)
        return frame
    dist.all_gather_into_tensor(frame.view(-1), part.contiguous().view(-1), group=group)
    return frame


class BandRenderer:
    """Per-rank driver: render this rank's band into HBM, then all-gather the frame."""

    def __init__(self, scene, rank=0, world=1, shadows=True, light_timing=False):
        height, width = (int(v) for v in scene.resolution)
        self.rank, self.world = rank, world
        self.band = row_band(height, rank, world)
        self.backend = scene._backend()
        self.stream = torch.cuda.Stream()
        self.frame = torch.empty((height, width, 3), dtype=torch.uint8, device="cuda")
        rows = self.band[1] - self.band[0]
        self.part = self.frame if world == 1 else torch.empty((rows, width, 3), dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(self.stream):
            self.desc = self.backend.render_device(scene, self.part.data_ptr(), self.stream.cuda_stream,
                                                   shadows=shadows, row_band=self.band, light_timing=light_timing)
        self.stream.synchronize()

    def step(self):
        """Enqueue one frame (no host synchronisation)."""
        with torch.cuda.stream(self.stream):
            self.backend.enqueue(self.desc, self.part.data_ptr(), self.stream.cuda_stream)
            if self.world > 1:
                all_gather_frame(self.part, self.frame)
        return self.frame
