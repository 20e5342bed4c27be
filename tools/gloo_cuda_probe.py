import os, sys, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
def log(m):
    print(f"[{r}] {m}", flush=True)
x = torch.full((12,), float(r), device="cuda", dtype=torch.float64)
out = torch.zeros(24, device="cuda", dtype=torch.float64)
log("all_gather_into_tensor cuda ...")
try:
    dist.all_gather_into_tensor(out, x)
    log(f"ok {out.tolist()[::12]}")
except Exception as e:
    log(f"ERR {type(e).__name__}: {str(e)[:200]}")
log("barrier ...")
dist.barrier()
log("all_reduce cuda ...")
t = torch.tensor([float(r)], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
log(f"ok {t.item()}")
dist.destroy_process_group()
