import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/glow-tts-train_amd"]
import torch, bench
from glow_tts_train.train import train_batch
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
for i in range(201):
    loss = train_batch(model, opt, batch, cfg.grad_clip, None)
    if i % 50 == 0:
        torch.cuda.synchronize()
        print(i, "reserved MB", round(torch.cuda.memory_reserved() / 1e6), "allocated", round(torch.cuda.memory_allocated() / 1e6), "loss", float(loss))
