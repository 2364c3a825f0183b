import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
x = torch.arange(12, device="cuda", dtype=torch.bfloat16).view(6,2)
y = torch.empty_like(x)
dist.all_to_all_single(y, x, output_split_sizes=[6], input_split_sizes=[6])
torch.cuda.synchronize(); print("eager a2a ok", torch.equal(x,y))
z = torch.empty(6,2, device="cuda", dtype=torch.bfloat16)
dist.all_gather_into_tensor(z, x); print("allgather ok")
try:
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dist.all_to_all_single(y, x, output_split_sizes=[6], input_split_sizes=[6])
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y.zero_()
        dist.all_to_all_single(y, x*2, output_split_sizes=[6], input_split_sizes=[6])
        w = y + 1
    g.replay(); torch.cuda.synchronize()
    print("graph a2a ok", torch.equal(w, x*2+1))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
dist.barrier(); dist.destroy_process_group(); print("done")
