"""GPU box: exercises the RCCL all-gather path of multigpu.py with a world of ONE rank
(launch: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_selftest.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import torch.distributed as dist
import scenes
from py_numpy_renderer_amd.multigpu import BandRenderer

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
api = scenes.product_api()
scene = scenes.build(api, "diablo_floor_small")
scene.device = local_rank
want = scene.render()
br = BandRenderer(scene, dist.get_rank(), dist.get_world_size(), frames_in_flight=2)
# force the collective even though world == 1
part = br.lanes[0][2]
out = torch.empty_like(part)
for _ in range(3):
    br.step()
br.synchronize()
dist.all_gather_into_tensor(out.view(-1), part.view(-1))
torch.cuda.synchronize()
assert np.array_equal(out.cpu().numpy(), want), "all-gathered frame differs"
assert np.array_equal(br.frame.cpu().numpy(), want)
print("nccl selftest ok: world", dist.get_world_size(), "backend", dist.get_backend())
dist.barrier()
dist.destroy_process_group()
