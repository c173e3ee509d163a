import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.tensor([5, -7, 1 << 62], dtype=torch.int64, device="cuda:0")
for op in (dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX):
    y = x.clone(); dist.all_reduce(y, op=op); assert torch.equal(y, x), op
h = dist.all_reduce(x.clone(), async_op=True); h.wait()
send = torch.arange(10, dtype=torch.int64, device="cuda:0"); recv = torch.empty(10, dtype=torch.int64, device="cuda:0")
dist.all_to_all_single(recv, send, output_split_sizes=[10], input_split_sizes=[10]); assert torch.equal(recv, send)
cnt = torch.tensor([10], dtype=torch.int64, device="cuda:0"); out = [torch.empty_like(cnt)]
dist.all_gather(out, cnt); assert int(out[0]) == 10
torch.cuda.synchronize(); print("RCCL int64 all_reduce SUM/MIN/MAX, async all_reduce, all_to_all_single with split sizes, all_gather: ok (world 1)")
dist.destroy_process_group()
