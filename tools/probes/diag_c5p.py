import sys; sys.path.insert(0,'/root/repo')
import torch
from tests import encoder_cases as E
a = dict(B=16, Ns=2048, Nq=100, d=256, H=8, L=6, T=32, memories=["mv", "pc", "voxel", "prompt"], p=float(sys.argv[1]) if len(sys.argv)>1 else 0.6, seed=0, data_seed=1234)
_e,_g,sd = E.f17_modules(a)
q_o,l_o,loss_o,g_o,gin_o = E.f17_oracle(a, sd)
gmax = max(float(v.norm()) for v in g_o.values())
res={}
for fused in (True, False):
    q,lg,loss,g,gin = E.f17_hip(a,"fp32",fused)
    res[fused]=g
    errs = sorted(((float((g[n].cpu()-g_o[n]).norm()/max(float(g_o[n].norm()),1e-3*gmax)), n, float(g_o[n].norm())/gmax) for n in g_o), reverse=True)
    print("fused" if fused else "modular", "q err", float((q.cpu()-q_o).abs().max()))
    for e in errs[:6]: print("   %.2e %s (norm/gmax %.1e)"%e)
d = sorted(((float((res[True][n]-res[False][n]).norm()/max(float(res[False][n].norm()),1e-3*gmax)), n) for n in g_o), reverse=True)
print("fused vs modular:", d[:4])
