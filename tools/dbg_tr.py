import sys, os, collections, torch
sys.path.insert(0, '.')
import bench
from uni_renderer_amd import backward as bw
from uni_renderer_amd.train_step import train_step
cnt = collections.Counter(); byt = collections.Counter(); tim = collections.Counter()
orig = bw.transpose2d
def spy(x):
    k = tuple(x.shape)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(x); e1.record(); torch.cuda.synchronize()
    cnt[k] += 1; byt[k] += x.numel() * 4; tim[k] += e0.elapsed_time(e1)
    return r
dev = torch.device('cuda:0')
nets = bench.build_models(dev, torch.float32)
for m in nets: m.train(); m.requires_grad_(True)
B, L = 4, 64
g = torch.Generator(device=dev).manual_seed(7)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5, t_img=torch.randint(0, 1000, (B,), device=dev, generator=g),
             t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g), target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
train_step(nets, batch, dtype=torch.bfloat16)
bw.transpose2d = spy
train_step(nets, batch, dtype=torch.bfloat16)
print('calls', sum(cnt.values()), 'GB', sum(byt.values()) / 1e9, 'ms', sum(tim.values()))
for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:25]:
    print(k, cnt[k], round(v, 2), 'ms', round(byt[k] / v / 1e6, 1), 'GB/s')
