import sys; sys.path.insert(0,'/root/repo')
import torch
from tests import encoder_cases as E
a = dict(B=4, Ns=int(sys.argv[1]) if len(sys.argv)>1 else 2048, Nq=120, d=int(sys.argv[2]) if len(sys.argv)>2 else 768, H=12, L=4, nb=3, C=201, foc=(0, 2), memories=["voxel", "mv", "pc"], seed=0, data_seed=1234, wscale=float(sys.argv[3]) if len(sys.argv)>3 else None)
_e,_m,sd = E.f13_state(a)
q_o, pc_o, pm_o, loss_o, g_o, gin_o = E.f13_oracle(a, sd)
for fused in (True, False):
    q, pc, pm, loss, g, gin = E.f13_hip(a, "fp32", fused)
    print("fused" if fused else "modular")
    for i,(m,r) in enumerate(zip(pm, pm_o)):
        mm = m.detach().cpu(); fin = r > -1e5
        print(i, "flip %.2e"%float(((mm<0)!=(r<0)).float().mean()), "maxabs err %.2e"%float((mm[fin]-r[fin]).abs().max()), "scale %.2e"%float(r[fin].abs().max()),
              "cls err %.2e"%float((pc[i].detach().cpu()-pc_o[i])[torch.isfinite(pc_o[i])].abs().max()))
