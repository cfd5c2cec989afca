import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from glow_tts_train import models, utils, optimize
from oracle import glow_oracle as O

hp = O.HParams(n_vocab=40, hidden_channels=32, filter_channels=64, filter_channels_dp=32, n_layers_enc=1, n_blocks_dec=2, n_block_layers=2)
sd = O.init_state_dict(hp, seed=9)
for k in list(sd):
    if k.endswith(".end.weight"):
        sd[k] = 0.05 * torch.randn_like(sd[k])
m = models.FlowGenerator(n_vocab=40, hidden_channels=32, filter_channels=64, filter_channels_dp=32, out_channels=80, kernel_size=3, n_heads=2, n_layers_enc=1, p_dropout=0.0, n_blocks_dec=2, kernel_size_dec=5, dilation_rate=1, n_block_layers=2, p_dropout_dec=0.0, n_split=4, n_sqz=2, window_size=4, mean_only=True, prenet=True)
m.load_state_dict(sd)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
m.cuda().train()
torch.manual_seed(3)
b, tx, ty = 2, 8, 32
x = torch.randint(1, 40, (b, tx)).cuda(); xl = torch.tensor([8, 5]).cuda()
y = torch.randn(b, 80, ty).cuda(); yl = torch.tensor([32, 20]).cuda()

def fwd():
    (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_) = m(x, xl, y, yl)
    l1 = utils.mle_loss(z, z_m, z_logs, logdet, z_mask); l2 = utils.duration_loss(logw, logw_, xl)
    return dict(z=z, z_m=z_m, logdet=logdet, x_m=x_m, attn=attn, logw=logw, logw_=logw_, l1=l1, l2=l2)

with torch.no_grad():
    e = {k: v.clone() for k, v in fwd().items()}
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        print("replay", rep, {k: float((out[k] - e[k]).abs().max()) for k in e})
