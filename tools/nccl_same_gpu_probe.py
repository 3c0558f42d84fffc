"""Probe: can two ranks share GPU 0 under RCCL (for testing the distributed path on a 1-GPU box)?"""
import os
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(4, device="cuda", dtype=torch.int32) + 10 * rank
out = torch.empty(4, device="cuda", dtype=torch.int32)
dist.all_to_all_single(out, x, output_split_sizes=[2, 2], input_split_sizes=[2, 2])
objs = [None] * world
dist.all_gather_object(objs, ("r", rank))
print("rank", rank, out.tolist(), objs, flush=True)
dist.destroy_process_group()
