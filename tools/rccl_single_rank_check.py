import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("rccl single-rank ok", float(t.sum()), dist.get_backend())
dist.destroy_process_group()
