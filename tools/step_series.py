import sys, os, time
sys.path[:0] = ["/root/repo", "/root/repo/glow-tts-train_amd"]
import torch, bench
from glow_tts_train.train import train_batch
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
print("reserved MB", torch.cuda.memory_reserved() / 1e6, "allocated", torch.cuda.memory_allocated() / 1e6)
