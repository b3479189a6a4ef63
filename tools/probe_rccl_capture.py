"""Can an RCCL all-reduce be captured INSIDE a HIP graph on this stack (torch 2.10 + ROCm 7.x)?  One-rank group on one GPU: capture
cast -> all_reduce(AVG) -> cast on a forked comm stream between two compute kernels, replay it, compare with the eager result.
If it captures, the data-parallel step needs no host-side replay boundaries (trainer.GraphedTrainStep cuts the step into 7 graphs
because the collectives stay outside the captured regions).  Prints one RESULT line either way (recorded in DESIGN.md section 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 24_000_000
g = torch.randn(n, device=dev)
s = torch.empty(n, dtype=torch.bfloat16, device=dev)
out = torch.empty(n, device=dev)
comm = torch.cuda.Stream()

def body():
    a = g * 2.0                      # "backward" producing the range
    comm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
        s.copy_(a)
        dist.all_reduce(s, op=dist.ReduceOp.AVG)
        out.copy_(s)
    b = a + 1.0                      # more "backward" overlapping the exchange
    torch.cuda.current_stream().wait_stream(comm)
    return b

try:
    dist.all_reduce(s)               # communicator set-up outside the capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = out.clone()
    out.zero_()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    ok = torch.equal(out, ref)
    print(f"RESULT rccl_all_reduce_captured_in_hip_graph=yes replay_matches_eager={ok} ms_per_replay={dt * 1e3:.3f}")
except Exception as e:  # noqa: BLE001
    print(f"RESULT rccl_all_reduce_captured_in_hip_graph=no error={type(e).__name__}: {str(e)[:300]}")
finally:
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
