import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from uni_renderer_amd.train_step import _forward_backward
dev = torch.device("cuda:0")
nets = bench.build_models(dev, torch.float32)
for m in nets:
    m.train(); m.requires_grad_(True)
g = torch.Generator(device=dev).manual_seed(7)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
B, L = 2, 32
batch = dict(x_t=mk(B,4,L,L), cond=mk(B,28,L,L), ehs=mk(B,77,768)*0.5, t_img=torch.randint(0,1000,(B,),device=dev,generator=g),
             t_attr=torch.randint(0,1000,(B,),device=dev,generator=g), target_img=mk(B,4,L,L), target_attr=mk(B,28,L,L))
from uni_renderer_amd import backward as BW
traces = []
def run(ck):
    BW.wgrad_queue.trace = {}
    for m in nets:
        (m.enable_gradient_checkpointing if ck else m.disable_gradient_checkpointing)()
        for p in m.parameters(): p.grad = None
    loss = _forward_backward(nets, batch, None, None, torch.bfloat16, None)
    torch.cuda.synchronize()
    traces.append(dict(BW.wgrad_queue.trace))
    return float(loss), {n: p.grad.clone() for m in nets for n, p in m.named_parameters() if p.grad is not None}
l0, g0 = run(False); l0b, g0b = run(False); l1, g1 = run(True)
print("loss", l0, l0b, l1)
nd_rep = [n for n in g0 if not torch.equal(g0[n], g0b[n])]
nd = [n for n in g0 if not torch.equal(g0[n], g1[n])]
print("params", len(g0), "differ between two PLAIN runs:", len(nd_rep), " plain vs checkpointed:", len(nd))
for n in nd[:12]:
    print(n, float((g0[n]-g1[n]).norm()/g0[n].norm().clamp_min(1e-30)))

a, b = traces[0], traces[2]
print("wgrad group keys only in plain:", {k: a[k] for k in a if a.get(k) != b.get(k)})
print("wgrad group keys only in checkpointed:", {k: b[k] for k in b if a.get(k) != b.get(k)})
