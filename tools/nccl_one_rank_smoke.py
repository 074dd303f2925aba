"""One-rank RCCL smoke (tests/test_bench_two_ranks_gpu.py): the calls bench.py's N > 1 branch makes on a multi-GPU node -- process group on the
"nccl" backend bound to the device, MAX all-reduce of a float64 scalar, all_gather_into_tensor of a result shard, all_gather_object, barrier -- with
world size 1 on the one GPU a lease has.  Proves the RCCL communicator of this image initialises and moves device tensors; says nothing about xGMI."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("all_reduce", t.item())
x=torch.rand(1024,8192,device=dev); out=torch.empty_like(x); dist.all_gather_into_tensor(out, x); torch.cuda.synchronize(); print("all_gather ok", bool((out==x).all()))
o=[None]; dist.all_gather_object(o, "rank 0 desc"); print(o)
dist.barrier(); dist.destroy_process_group(); print("done")
