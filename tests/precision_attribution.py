"""Where does the fp16 error come from?  (analysis script, CPU only, not collected by pytest; test infrastructure --
it imports the oracle.)  fp16 STORAGE rounding is injected into the fp32 oracle at the points where the HIP path
materialises an activation, separately for the residual stream (block outputs and the x + f(x) adds) and for the
branches; the numbers are quoted in DESIGN.md section 5.

    python tests/precision_attribution.py [latent_side=32]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.unirenderer_oracle as O
torch.set_num_threads(8)
MODE = {"res": False, "br": False}
def q(x):  # fp16 storage rounding
    return x.to(torch.float16).to(torch.float32)
def Rr(x): return q(x) if MODE["res"] else x
def Rb(x): return q(x) if MODE["br"] else x

def resnet_fwd(self, x, temb):
    h = Rb(F.silu(self.norm1(x)))
    h = Rb(self.conv1(h) + self.time_emb_proj(F.silu(temb))[:, :, None, None])
    h = Rb(F.silu(self.norm2(h)))
    h = self.conv2(h)
    if self.conv_shortcut is not None:
        x = Rb(self.conv_shortcut(x))
    return Rr((x + h) / self.output_scale_factor)
def attn_fwd(self, x, context=None):
    context = x if context is None else context
    b, t, _ = x.shape
    qq, k, v = Rb(self.to_q(x)), Rb(self.to_k(context)), Rb(self.to_v(context))
    d = qq.shape[-1] // self.heads
    qq = qq.view(b, -1, self.heads, d).transpose(1, 2); k = k.view(b, -1, self.heads, d).transpose(1, 2); v = v.view(b, -1, self.heads, d).transpose(1, 2)
    w = torch.softmax((qq @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    o = Rb((Rb(w) @ v).transpose(1, 2).reshape(b, t, self.heads * d))
    return self.to_out[0](o)
def geglu_fwd(self, x):
    h, gate = self.proj(x).chunk(2, dim=-1)
    return Rb(h * F.gelu(gate))
def btb_fwd(self, x, context):
    x = Rr(x + self.attn1(Rb(self.norm1(x))))
    x = Rr(x + self.attn2(Rb(self.norm2(x)), context))
    x = Rr(x + self.ff(Rb(self.norm3(x))))
    return x
def t2d_fwd(self, x, context):
    b, c, h, w = x.shape
    res = x
    x = Rb(self.proj_in(Rb(self.norm(x))))
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
    for blk in self.transformer_blocks:
        x = blk(x, context)
    x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
    return Rr(self.proj_out(x) + res)
O.ResnetBlock2D.forward = resnet_fwd
O.Attention.forward = attn_fwd
O.GEGLU.forward = geglu_fwd
O.BasicTransformerBlock.forward = btb_fwd
O.Transformer2DModel.forward = t2d_fwd
_ds = O.Downsample2D.forward; _us = O.Upsample2D.forward
O.Downsample2D.forward = lambda self, x: Rr(_ds(self, x))
O.Upsample2D.forward = lambda self, x, output_size=None: Rr(_us(self, x, output_size))

unet, enc, dec = O.build_triplet(O.SD15_CONFIG, seed=1234)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x, c, ehs, ti, ta = O.make_inputs(1, L, 768, seed=8, t_img=0)
def rel(a, b): return float((a - b).norm() / b.norm())
with torch.no_grad():
    t0 = time.time(); ref = O.dual_stream_step(unet, enc, dec, x, c, ehs, ti, ta); print("fp32", time.time() - t0, flush=True)
    for name, m in [("all", dict(res=True, br=True)), ("residual-stream only", dict(res=True, br=False)), ("branches only", dict(res=False, br=True))]:
        MODE.update(m)
        out = O.dual_stream_step(unet, enc, dec, x, c, ehs, ti, ta)
        print(name, "img", rel(out["img_pred"], ref["img_pred"]), "attr", rel(out["attr_pred"], ref["attr_pred"]), flush=True)
