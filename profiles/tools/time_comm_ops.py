"""GPU-side cost per call of the two communication points as torch.distributed issues them (1-rank RCCL group on
one GPU, ghost buffers sent to itself): what the N>1 loop adds to a step besides wire time."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cnt = 7 * 4096 * 3
s_in, r_in, s_out, r_out = (torch.zeros(cnt, dtype=torch.float64, device=dev) for _ in range(4))
dt = torch.zeros(1, dtype=torch.float64, device=dev)
x = torch.zeros(1 << 20, dtype=torch.float64, device=dev)


def timed(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print(f"{name:46s} host {1e6 * th / n:7.1f} us  wall {1e6 * tw / n:7.1f} us", flush=True)


def k():
    x.add_(1.0)   # a ~10 us kernel on the current stream standing for the step's kernels


def p2p():
    ops = [dist.P2POp(dist.isend, s_in, 0), dist.P2POp(dist.irecv, r_in, 0),
           dist.P2POp(dist.isend, s_out, 0), dist.P2POp(dist.irecv, r_out, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()


timed("kernel only", k)
timed("kernel + all_reduce(MIN, 1 double)", lambda: (k(), dist.all_reduce(dt, op=dist.ReduceOp.MIN)))
timed("kernel + all_reduce async_op + wait", lambda: (k(), dist.all_reduce(dt, op=dist.ReduceOp.MIN, async_op=True).wait()))
timed("kernel + batch_isend_irecv (4 x 688 KB)", lambda: (k(), p2p()))
timed("kernel + all_reduce + kernel + p2p", lambda: (k(), dist.all_reduce(dt, op=dist.ReduceOp.MIN), k(), p2p()))
g = torch.zeros(8, dtype=torch.float64, device=dev)
timed("kernel + all_gather_into_tensor(1 double)", lambda: (k(), dist.all_gather_into_tensor(g[:1], dt)))
dist.destroy_process_group()
