import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY","0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g=[torch.empty(4,2,device="cuda") for _ in range(1)]
dist.all_gather(g, torch.ones(4,2,device="cuda"))
x = torch.zeros(1, dtype=torch.int64, device="cuda"); out=[torch.zeros(1,dtype=torch.int64,device="cuda")]
dist.all_gather(out, x)
print("nccl ok", float(t.item()), dist.get_backend())
dist.destroy_process_group()
