import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29512")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
torch.cuda.set_device(0)
t=torch.tensor([1.5],dtype=torch.float64,device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
print("nccl ok", t.item()); dist.destroy_process_group()
