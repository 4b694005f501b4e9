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
    """Per-rank driver: render this rank's band into HBM, then all-gather the frame.

    ``frames_in_flight`` > 1 renders successive frames on different HIP streams (each with
    its own output buffers and its own work buffers inside the library), so the short,
    latency-bound stages of one frame overlap with the next frame's: throughput mode.  With 1
    every frame runs alone on one stream: latency mode.
    """

    def __init__(self, scene, rank=0, world=1, shadows=True, light_timing=False, frames_in_flight=1):
        height, width = (int(v) for v in scene.resolution)
        self.rank, self.world = rank, world
        self.band = row_band(height, rank, world)
        self.backend = scene._backend()
        rows = self.band[1] - self.band[0]
        self.lanes = []
        self.count = 0
        self.desc = None
        for _ in range(max(1, int(frames_in_flight))):
            stream = torch.cuda.Stream()
            frame = torch.empty((height, width, 3), dtype=torch.uint8, device="cuda")
            part = frame if world == 1 else torch.empty((rows, width, 3), dtype=torch.uint8, device="cuda")
            with torch.cuda.stream(stream):
                self.desc = self.backend.render_device(scene, part.data_ptr(), stream.cuda_stream, shadows=shadows,
                                                       row_band=self.band, light_timing=light_timing)
            stream.synchronize()
            self.lanes.append((stream, frame, part))
        self.frame = self.lanes[0][1]

    def step(self):
        """Enqueue one frame (no host synchronisation); returns the tensor it will land in."""
        stream, frame, part = self.lanes[self.count % len(self.lanes)]
        self.count += 1
        with torch.cuda.stream(stream):
            self.backend.enqueue(self.desc, part.data_ptr(), stream.cuda_stream)
            if self.world > 1:
                all_gather_frame(part, frame)
        self.frame = frame
        return frame

    def synchronize(self):
        for stream, _, _ in self.lanes:
            stream.synchronize()
