import sys
sys.path[:0] = ["/root/repo/glow-tts-train_amd", "/root/repo"]
import torch
from glow_tts_train import convops, attentions
torch.manual_seed(0)
blk = attentions.CouplingBlock(32, 48, kernel_size=3, dilation_rate=1, n_layers=2, p_dropout=0.0).cuda()
fired = []
for n, p in blk.named_parameters():
    p.grad = torch.zeros_like(p)
    p.register_post_accumulate_grad_hook(lambda p_, n=n: fired.append(n))
notified = []
convops.add_grad_ready_listener(lambda ps: notified.extend(id(p) for p in ps))
x = torch.randn(2, 32, 40, device="cuda", requires_grad=True)
z, ld = blk(x, torch.ones(2, 1, 40, device="cuda"))
(z.sum() + ld.sum()).backward()
convops.flush_groups()
print("hooks fired:", len(fired), sorted(set(fired)))
print("notified:", len(notified), "params:", len(list(blk.parameters())))
