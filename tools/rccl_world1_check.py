import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dev=torch.device("cuda",0)
dist.init_process_group("nccl", device_id=dev)
full=torch.arange(16,dtype=torch.float32,device=dev)
mine=full[0:16]
dist.all_gather_into_tensor(full, mine)
r=torch.ones(8,dtype=torch.float64,device=dev); dist.all_reduce(r)
flag=torch.tensor([1],dtype=torch.int32,device=dev); dist.broadcast(flag,0)
dist.barrier(); torch.cuda.synchronize()
print("rccl world-1 ok", full.sum().item(), r.sum().item(), dist.get_backend())
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torchpdlp_amd as tp
c=tp.Comm(); print("Comm", c.rank, c.world, c.backend)
dist.destroy_process_group()
