"""Screen-tile split of one frame over the GPUs of a node (one process per GPU).

The path shards by pixels: every mutable buffer of the reference (frame, z, stencil) is
per-pixel and primitives interact only through them (obj/triangular.py:101-118,356-368), so
disjoint sets of screen tiles are independent.  Geometry, textures and per-frame constants are
replicated; each rank rasterises and shades only its tiles and ONE all-gather (RCCL over xGMI
when the backend is "nccl") assembles the uint8 frame on every rank.  Two partitions:

* ``"bands"``   rank r renders the contiguous band of H/world output rows; bands in output order
  make the gathered buffer the final frame with no reorder, but the load is uneven (sky bands are
  empty, the band holding the mesh and its shadow is the whole frame's critical path);
* ``"stripes"`` rank r renders the tile rows (16 screen rows) t with t mod world == r -- every rank
  gets an interleaved sample of the screen, so the load is balanced -- and one row gather
  (``unstripe``) after the all-gather puts the rows in frame order.
* ``"weighted"`` contiguous bands again, cut on tile rows where the COST is equal instead of the row count
  (``weighted_bands``; the cost of a tile row is what the tile kernel's own model makes of its list lengths in a
  whole frame rendered once while priming): every rank sends as many rows as the tallest band holds (the all-gather
  is regular) and one row gather (``unband_index``) drops the padding.
"""
import torch
import torch.distributed as dist

TILE_ROWS = 16


def row_band(height, rank, world):
    """Output rows ``[begin, end)`` of *rank*; bands are equal so the all-gather is regular."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    if height % world:
        raise ValueError(f"{height} rows do not split evenly over {world} ranks")
    rows = height // world
    return rank * rows, (rank + 1) * rows


RESIDENT_WORKGROUPS = 256 * 6      # tile workgroups an MI355X holds at once (256 CUs, six wavefronts per SIMD, four per workgroup)


def tile_row_costs(records, tiles_x, peaks=False):
    """Cost of every tile row of a whole frame (bottom row first), from the tile kernel's per-tile records of that
    frame (``DeviceRenderer.read_tile_records``: words 5-7 are the lengths of a tile's three lists) with the kernel's
    own cost model (``tile_cost`` in csrc/kernels_tile.h: 20 + 2 small pairs + 30 big pairs + 3 shadow quads, in
    units of ~0.1 us).  With *peaks* also the cost of every row's most expensive tile."""
    import numpy as np
    rec = np.asarray(records, dtype=np.int64)
    cost = (20 + 2 * rec[:, 5] + 30 * rec[:, 6] + 3 * rec[:, 7]).reshape(-1, int(tiles_x))
    return (cost.sum(axis=1), cost.max(axis=1)) if peaks else cost.sum(axis=1)


def weighted_bands(row_cost, height, world, row_peak=None, resident=RESIDENT_WORKGROUPS):
    """Contiguous bands of output rows ``[(begin, end)] * world`` (rank 0 on top, every band at least one tile row, cut
    on tile rows) that minimise the cost of the most expensive band; *row_cost* lists the tile rows bottom first
    (screen y up, like the reference's buffers and the device's tile grid).

    The cost of a band is the sum of its tiles' costs -- or, with *row_peak* (the most expensive tile of every row),
    a model of what the tile kernel's launch on that band lasts: the device works on *resident* tiles at once, so a
    band takes its summed cost / *resident*, plus a tail behind its most expensive tile (a fifth of that tile's cost:
    what c3's and c4's whole frames show, 48.5 + 0.2 x 117 = 73 us and 45.5 + 0.2 x 81 = 60 us).  A thin band under
    the heaviest tiles of the frame is mostly tail, and taking rows from it does not make it faster."""
    cost = [int(c) for c in row_cost][::-1]                  # top tile row first: bands are handed out in output order
    peak = [int(c) for c in row_peak][::-1] if row_peak is not None else None
    n = len(cost)
    if n != -(-int(height) // TILE_ROWS):
        raise ValueError(f"{n} tile-row costs for {height} rows")
    if not 1 <= world <= n:
        raise ValueError(f"{world} ranks for {n} tile rows")
    if peak is not None and len(peak) != n:
        raise ValueError("row_peak and row_cost differ in length")

    def band_cost(total, top):
        return total if peak is None else total + top * int(resident) // 5     # (in units of cost / resident)

    def cuts_for(cap):
        """Greedy: every band takes tile rows while it stays under *cap* and leaves one for each rank behind it."""
        cuts, at = [], 0
        for r in range(world):
            last = n - (world - 1 - r)                       # rows [at, last) are this band's to choose from
            acc, top, end = 0, 0, at
            while end < last:
                nxt_top = max(top, peak[end]) if peak is not None else 0
                if end > at and band_cost(acc + cost[end], nxt_top) > cap:
                    break
                acc += cost[end]
                top = nxt_top
                end += 1
            if r == world - 1 and end < n:
                return None
            cuts.append((at, end))
            at = end
        return cuts

    lo = max(band_cost(c, p) for c, p in zip(cost, peak if peak is not None else [0] * n))
    hi = band_cost(sum(cost), max(peak) if peak is not None else 0)
    while lo < hi:                                           # smallest cap the greedy split fits under
        mid = (lo + hi) // 2
        if cuts_for(mid) is None:
            lo = mid + 1
        else:
            hi = mid
    # Among the splits whose most expensive band costs no more than that, the one with the least sum of squared band
    # costs: the greedy split fills every band up to the cap, and where one band is dear whatever its size (a heavy tile's
    # tail) that would starve the bands behind it for no gain.  (rows^2 x ranks steps, once per renderer.)
    cap = lo
    inf = float("inf")
    best = [[inf] * (n + 1) for _ in range(world + 1)]
    cut = [[0] * (n + 1) for _ in range(world + 1)]
    best[0][0] = 0
    for r in range(1, world + 1):
        for i in range(r, n - (world - r) + 1):
            acc, top = 0, 0
            for j in range(i - 1, r - 2, -1):                 # band = rows [j, i)
                acc += cost[j]
                if peak is not None:
                    top = max(top, peak[j])
                c = band_cost(acc, top)
                if c > cap:
                    break
                if best[r - 1][j] + c * c < best[r][i]:
                    best[r][i], cut[r][i] = best[r - 1][j] + c * c, j
    cuts, i = [], n
    for r in range(world, 0, -1):
        cuts.append((cut[r][i], i))
        i = cut[r][i]
    cuts.reverse()
    top_rows = int(height) - (n - 1) * TILE_ROWS            # the top tile row may be a partial one
    def first_row(t):                                        # output row where tile row t (from the top) begins
        return 0 if t == 0 else top_rows + (t - 1) * TILE_ROWS
    return [(first_row(a), int(height) if b == n else first_row(b)) for a, b in cuts]


def unband_index(bands, device=None):
    """For every output row of the frame, its row in the all-gathered buffer of padded bands (every rank sends
    ``max(end - begin)`` rows, its own first)."""
    per = max(e - b for b, e in bands)
    idx = torch.cat([torch.arange(b, e, dtype=torch.long) - b + r * per for r, (b, e) in enumerate(bands)])
    return idx.to(device) if device is not None else idx


def stripe_rows(height, world):
    """Rows of one rank's output buffer in the striped layout: its share of the frame's tile rows,
    rounded up so that every rank sends the same amount (``mr_frame_desc.stripe_count``)."""
    tile_rows = -(-int(height) // TILE_ROWS)
    return -(-tile_rows // int(world)) * TILE_ROWS


def unstripe_index(height, world, device=None):
    """For every output row of the frame, its row in the all-gathered striped buffer.

    Rank r holds the frame's tile rows g = r, r + world, ... (counted from the BOTTOM of the frame,
    like the reference's buffers), highest first, rows inside a tile row top-down
    (``include/mi355rast.h``, stripe_count)."""
    per = stripe_rows(height, world)
    out_row = torch.arange(height, dtype=torch.long)
    py = height - 1 - out_row                               # screen row, y up
    g = py // TILE_ROWS
    rank, local = g % world, g // world
    idx = rank * per + (per // TILE_ROWS - 1 - local) * TILE_ROWS + (TILE_ROWS * g + TILE_ROWS - 1 - py)
    return idx.to(device) if device is not None else idx


def unstripe(gathered, height, world, out=None, index=None):
    """Frame ``(height, W, 3)`` from the all-gathered striped buffer ``(world * stripe_rows, W, 3)``."""
    if index is None:
        index = unstripe_index(height, world, gathered.device)
    if out is None:
        return gathered.index_select(0, index)
    torch.index_select(gathered, 0, index, out=out)
    return out


def all_gather_frame(part, frame=None, group=None):
    """Assemble every rank's rows with a single collective.

    *part* is this rank's ``(rows, W, 3)`` uint8 buffer (any device); returns ``(rows*world, W, 3)``:
    the frame itself for row bands, the striped buffer (see ``unstripe``) for stripes."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if frame is None:
        frame = torch.empty((part.shape[0] * world,) + tuple(part.shape[1:]), dtype=part.dtype,
                            device=part.device)
    if world == 1:
        if frame.data_ptr() != part.data_ptr():
            frame.copy_(part)
        return frame
    if part.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a machine without RCCL peers (several ranks sharing one GPU): gloo moves host memory
        host = torch.empty(frame.shape, dtype=frame.dtype)
        torch.cuda.current_stream().synchronize()
        dist.all_gather_into_tensor(host.view(-1), part.contiguous().view(-1).cpu(), group=group)
        frame.copy_(host)
        return frame
    dist.all_gather_into_tensor(frame.view(-1), part.contiguous().view(-1), group=group)
    return frame


class BandRenderer:
    """Per-rank driver: render this rank's tiles into HBM, then all-gather the frame.

    ``frames_in_flight`` > 1 renders successive frames on different HIP streams (each with
    its own output buffers and its own work buffers inside the library), so the short,
    latency-bound stages of one frame overlap with the next frame's: throughput mode.  With 1
    every frame runs alone on one stream: latency mode.

    Frames are enqueued without host synchronisation, so a work list that overflows cannot be
    retried on the spot the way ``mr_render`` does it: the priming frames are checked here (and
    rendered again with the grown lists), and ``verify()`` does the same for the frames since.
    """

    def __init__(self, scene, rank=0, world=1, shadows=True, light_timing=False, frames_in_flight=1,
                 partition="bands", timing_every=1, overlay=False, streams=None):
        height, width = (int(v) for v in scene.resolution)
        if partition not in ("bands", "stripes", "weighted"):
            raise ValueError(f"unknown partition {partition!r}")
        self.rank, self.world, self.partition = rank, world, partition
        self.height, self.width = height, width
        self.backend = scene._backend()
        self.scene, self.shadows, self.light_timing = scene, shadows, light_timing
        # HIP event marks cost a few microseconds each between two kernels: only every timing_every-th
        # frame carries them (0: none does); mr_get_kernel_times averages over those
        self.timing_every = int(timing_every)
        striped = partition == "stripes" and world > 1
        self.stripe = (rank, world) if striped else None
        self.bands = None
        if partition == "weighted" and world > 1:
            if overlay:
                raise ValueError("the overlay of a split frame needs equal bands or stripes (mr_overlay_apply)")
            self.bands = self._cost_bands(scene, shadows, height, width, world)
            self.band = self.bands[rank]
            rows = max(e - b for b, e in self.bands)         # every rank sends the tallest band's rows
        else:
            self.band = (0, height) if striped else row_band(height, rank, world)
            rows = stripe_rows(height, world) if striped else self.band[1] - self.band[0]
        # The debug-frustum overlay (obj/core.py:638) of a split frame: every rank appends the state (z, float colour)
        # of the touched pixels it owns to its rows, the ONE all-gather carries rows and state, and every rank replays
        # the overlay on the assembled frame (include/mi355rast.h, mr_overlay_apply): the lines test z at pixels other
        # ranks own, so no rank could draw its share alone.  On one device the frame's own kernel draws it.
        self.overlay = bool(overlay)
        self.rows_bytes = rows * width * 3
        self.state_offset = self.state_bytes = 0
        if self.overlay and world > 1:
            self.backend.sync_scene(scene)
            self.backend.sync_overlay(scene)
            self.state_offset = -(-self.rows_bytes // 16) * 16
            self.state_bytes = -(-self.backend.overlay_state_bytes() // 16) * 16
        self.part_bytes = self.state_offset + self.state_bytes if self.state_bytes else self.rows_bytes
        self.lanes = []
        self.count = 0
        self.descs = []
        for lane in range(max(1, int(frames_in_flight))):
            # (a scene keeps work buffers per stream it has been rendered on, 32 at most: a caller that builds many
            # renderers for one scene hands them the same streams)
            stream = streams[lane] if streams else torch.cuda.Stream()
            frame = torch.empty((height, width, 3), dtype=torch.uint8, device="cuda")
            if self.state_bytes:            # rows + state travel as one flat buffer; the rows are copied out after the gather
                part = torch.zeros(self.part_bytes, dtype=torch.uint8, device="cuda")
                gathered = torch.empty(self.part_bytes * world, dtype=torch.uint8, device="cuda")
            else:
                part = frame if world == 1 else torch.empty((rows, width, 3), dtype=torch.uint8, device="cuda")
                gathered = (torch.empty((rows * world, width, 3), dtype=torch.uint8, device="cuda")
                            if striped or self.bands else frame)
            self.lanes.append((stream, frame, part, gathered))
        self.rows_all = (torch.empty((rows * world, width, 3), dtype=torch.uint8, device="cuda")
                         if self.state_bytes and striped else None)
        self.index = (unstripe_index(height, world, "cuda") if striped else
                      unband_index(self.bands, "cuda") if self.bands else None)
        # the collective's operands, flattened once (step() is the host's per-frame cost: keep it to the calls)
        self._flat = [(g.view(-1), p.view(-1)) for _, _, p, g in self.lanes]
        self._direct = world > 1 and dist.is_initialized() and dist.get_backend() != "gloo"
        self.desc = None
        self.descs, self.descs_untimed = [], []
        self.prime()
        self.frame = self.lanes[0][1]

    def _cost_bands(self, scene, shadows, height, width, world):
        """One whole frame on this device (every rank renders the same one and reads the same list lengths, so all
        arrive at the same cuts without talking), then the bands of equal cost."""
        tmp = torch.empty((height, width, 3), dtype=torch.uint8, device="cuda")
        for _ in range(6):
            self.backend.render_device(scene, tmp.data_ptr(), torch.cuda.current_stream().cuda_stream, shadows=shadows)
            torch.cuda.current_stream().synchronize()
            if not self.backend.overflowed():
                break
        costs, peaks = tile_row_costs(self.backend.read_tile_records(), -(-width // 16), peaks=True)
        return weighted_bands(costs, height, world, row_peak=peaks)

    def prime(self):
        """One frame per lane through the full host path (scene sync, frame packing); repeated while
        a work list overflows, so that the frames enqueued afterwards find lists that are large enough."""
        for _ in range(6):
            for stream, _, part, _ in self.lanes:
                with torch.cuda.stream(stream):
                    self.desc = self.backend.render_device(self.scene, part.data_ptr(), stream.cuda_stream,
                                                           shadows=self.shadows, row_band=self.band,
                                                           light_timing=self.light_timing, stripe=self.stripe,
                                                           overlay=self.overlay)
                stream.synchronize()
            if not self.backend.overflowed():
                self.set_descriptors([self.desc])
                return
        raise RuntimeError("work lists kept overflowing while priming")

    def set_descriptors(self, descs):
        """Frame descriptors to cycle through in ``step`` (e.g. a camera path), instead of repeating
        the priming frame's."""
        mode = "light" if self.light_timing else "all"
        self.descs = [self.backend.with_timing(d, mode) for d in descs]
        self.descs_untimed = [self.backend.with_timing(d, "none") for d in descs]

    def step(self):
        """Enqueue one frame (no host synchronisation); returns the tensor it will land in."""
        stream, frame, part, gathered = self.lanes[self.count % len(self.lanes)]
        timed = self.timing_every > 0 and self.count % self.timing_every == 0
        pool = self.descs if timed else self.descs_untimed
        desc = pool[self.count % len(pool)]
        self.count += 1
        with torch.cuda.stream(stream):
            self.backend.enqueue(desc, part.data_ptr(), stream.cuda_stream)
            if self.world > 1:
                if self._direct:
                    gflat, pflat = self._flat[(self.count - 1) % len(self.lanes)]
                    dist.all_gather_into_tensor(gflat, pflat)
                else:
                    all_gather_frame(part, gathered)
                if self.state_bytes:
                    self._assemble_with_overlay(gathered, frame, stream)
                elif self.index is not None:
                    unstripe(gathered, self.height, self.world, out=frame, index=self.index)
        self.frame = frame
        return frame

    def _assemble_with_overlay(self, gathered, frame, stream):
        """Rows out of the gathered parts into *frame*, then the overlay replayed on it from the gathered state."""
        parts = gathered.view(self.world, self.part_bytes)
        rows = parts[:, :self.rows_bytes]
        if self.index is None:              # bands in output order: the rows of rank r are rows r * H/N ... of the frame
            frame.view(self.world, self.rows_bytes).copy_(rows)
        else:
            self.rows_all.view(self.world, self.rows_bytes).copy_(rows)
            unstripe(self.rows_all, self.height, self.world, out=frame, index=self.index)
        self.backend.overlay_apply(gathered.data_ptr(), self.part_bytes, self.state_offset, self.world, self.index is not None,
                                   int(self.scene.system), frame.data_ptr(), stream.cuda_stream)

    def verify(self):
        """Synchronise and check that no frame since the last check overflowed a work list.  Returns
        True when all was well; otherwise the lists have been grown, the lanes primed again, and the
        frames enqueued since the last check must be considered invalid."""
        self.synchronize()
        if not self.backend.overflowed():
            return True
        self.prime()
        return False

    def kernel_times(self, n_frames=64):
        """Average device milliseconds per stage over this renderer's own marked frames (at most *n_frames*
        per stream); returns (dict, frames averaged).  Synchronises the device."""
        total, used = {}, 0
        for stream, *_ in self.lanes:
            times, n = self.backend.stream_kernel_times(stream.cuda_stream, n_frames)
            for k, v in times.items():
                total[k] = total.get(k, 0.0) + v * n
            used += n
        return {k: (v / used if used else 0.0) for k, v in total.items()}, used

    def synchronize(self):
        for stream, *_ in self.lanes:
            stream.synchronize()
