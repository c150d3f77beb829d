import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t1 = time.time()
x = torch.ones(1 << 20, device="cuda")
out = [torch.empty_like(x)]
dist.gather(x, out, dst=0)
torch.cuda.synchronize(); t2 = time.time()
dist.barrier(); torch.cuda.synchronize(); t3 = time.time()
print("init %.2f s, first gather %.2f s, barrier %.2f s" % (t1 - t0, t2 - t1, t3 - t2))
dist.destroy_process_group()
