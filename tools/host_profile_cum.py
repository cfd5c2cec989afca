import cProfile, os, pstats, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch
import bench
from glow_tts_train.train import train_batch
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
for _ in range(5):
    train_batch(model, opt, batch, cfg.grad_clip, None)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    train_batch(model, opt, batch, cfg.grad_clip, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
