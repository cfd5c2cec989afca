import sys
sys.path[:0] = ["/root/repo/glow-tts-train_amd", "/root/repo"]
import torch
from glow_tts_train import convops
B, C, T = 32, 192, 160
x = torch.randn(B, C, T, device="cuda", requires_grad=True)
r = torch.randn(B, C, T, device="cuda", requires_grad=True)
g = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
go = torch.randn(B, C, T, device="cuda")
def step():
    y = convops.ChanLayerNormFn.apply(x, r, g, b, 1e-4, False, False, 0.0)
    y.backward(go)
for _ in range(5): step()
torch.cuda.synchronize()
from glow_tts_train import _hip
_hip.enable_timing()
for _ in range(20): step()
t = _hip.disable_timing()
for k, v in t.items(): print(k, round(1e3 * sum(v) / len(v), 2), "us")
