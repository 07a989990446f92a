"""GPU-side cost of a dependent kernel launch: 1000 tiny element-wise kernels replayed from a HIP graph
(no host in the loop) and issued eagerly.  Development tool (GPU box)."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
n = 1000
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            x.add_(1.0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("graph replay: %.2f us per kernel" % (e0.elapsed_time(e1) * 1e3 / (5 * n)))
t0 = time.perf_counter()
e0.record()
for _ in range(n):
    x.add_(1.0)
e1.record()
torch.cuda.synchronize()
print("eager: %.2f us per kernel (GPU timeline), host %.2f us per launch" % (e0.elapsed_time(e1) * 1e3 / n, (time.perf_counter() - t0) * 1e6 / n))
