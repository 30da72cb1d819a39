import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
x = torch.ones(1024*1024*3, device="cuda")
dist.reduce(x, dst=0, op=dist.ReduceOp.SUM); dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.0,2.0], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("ok", float(x.sum()), t.tolist()); dist.destroy_process_group()
