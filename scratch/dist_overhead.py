"""Host + device cost of the torch.distributed (RCCL) calls of one W>1 step, measured with a ONE-rank nccl group on one
GPU: the message never leaves the device, so what is left is the per-call overhead every rank pays."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
send = torch.zeros((264, 768), dtype=torch.bfloat16, device=dev)
recv = torch.zeros((264, 768), dtype=torch.bfloat16, device=dev)
dC = torch.zeros((264, 768), dtype=torch.float32, device=dev)
mine = torch.zeros((264, 768), dtype=torch.float32, device=dev)
loss = torch.zeros(1, device=dev)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("all_gather_into_tensor  host %.1f us  total %.1f us" % t(lambda: dist.all_gather_into_tensor(recv, send)))
print("reduce_scatter_tensor   host %.1f us  total %.1f us" % t(lambda: dist.reduce_scatter_tensor(mine, dC)))
print("all_reduce (sync op)    host %.1f us  total %.1f us" % t(lambda: dist.all_reduce(loss)))
def ar():
    h = dist.all_reduce(loss, async_op=True); h.wait()
print("all_reduce async+wait   host %.1f us  total %.1f us" % t(ar))
def three():
    dist.all_gather_into_tensor(recv, send); h = dist.all_reduce(loss, async_op=True); dist.reduce_scatter_tensor(mine, dC); h.wait()
print("AG + AR(async) + RS     host %.1f us  total %.1f us" % t(three))
def two():
    dist.all_gather_into_tensor(recv, send); dist.reduce_scatter_tensor(mine, dC)
print("AG + RS                 host %.1f us  total %.1f us" % t(two))
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpr_scale_amd import dist as D
comm = D.try_direct_comm(dev)
print("direct communicator:", "ok" if comm is not None else "unavailable")
if comm is not None:
    print("direct all_gather       host %.1f us  total %.1f us" % t(lambda: comm.all_gather_rows(send, recv)))
    print("direct reduce_scatter   host %.1f us  total %.1f us" % t(lambda: comm.reduce_scatter_rows(dC, mine)))
    print("direct all_reduce       host %.1f us  total %.1f us" % t(lambda: comm.all_reduce_sum(loss)))
    def three_d():
        comm.all_gather_rows(send, recv); comm.reduce_scatter_rows(dC, mine); comm.all_reduce_sum(loss)
    print("direct AG + RS + AR     host %.1f us  total %.1f us" % t(three_d))
    comm.close()
dist.barrier(); dist.destroy_process_group()
