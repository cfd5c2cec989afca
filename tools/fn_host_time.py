import os, sys, time, collections
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch
import bench
from glow_tts_train import convops, ops, attentions
from glow_tts_train.train import train_batch
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, which):
    orig = getattr(cls, which)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            e = acc[f"{cls.__name__}.{which}"]; e[0] += time.perf_counter() - t0; e[1] += 1
    setattr(cls, which, staticmethod(timed))
import inspect
for mod in (convops, ops, attentions):
    for name, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls.__module__ == mod.__name__:
            wrap(cls, "forward"); wrap(cls, "backward")
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
for _ in range(5): train_batch(model, opt, batch, cfg.grad_clip, None)
acc.clear()
N = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): train_batch(model, opt, batch, cfg.grad_clip, None)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host enqueue {1e3*(t1-t0)/N:.2f} ms/step")
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:34s} {1e3*t/N:7.3f} ms/step  {n/N:6.1f} calls  {1e6*t/n:7.1f} us each")
