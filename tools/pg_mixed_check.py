import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("cpu:gloo,cuda:nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
dist.all_reduce(t)
box = [b"x" * 128 if dist.get_rank() == 0 else None]
dist.broadcast_object_list(box, src=0, device=torch.device("cpu"))
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from admm_amd import dist as adist
adist.barrier()
print("backend", dist.get_backend())
print(dist.get_rank(), t.item(), len(box[0]))
dist.destroy_process_group()
