"""Audit of the forward / data-gradient launch grids of one train step: workgroups vs resident slots,
and the flops a strict round model would lose to a partially filled last round."""
import os, sys, collections, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import conv as C
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step
dev="cuda:0"; torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
m = MaskRCNN(cfg).to(dev); opt = set_optimizer(m, cfg.TRAIN)
b = synthetic_batch(4, 1024, device=dev, seed=2000); m.external_proposals = SyntheticProposals(b[2], 1024, seed=7)
m.generator = torch.Generator(device=dev).manual_seed(1)
train_step(m, opt, list(b))
C.SHAPE_LOG = []
train_step(m, opt, list(b)); torch.cuda.synchronize()
cnt = collections.Counter(C.SHAPE_LOG)
rows = []
def grid(P, Cout):
    t128 = math.ceil(P/128)*math.ceil(Cout/128)
    bm = 64 if (Cout <= 64 or t128 < 512) else 128
    ny = math.ceil(Cout/bm); nx = math.ceil(P/128); bn = 128
    if bm == 64 and nx*ny < 1024 and P >= 64: bn = 64; nx = math.ceil(P/64)
    slots = 1024 if bm == 128 else 1536
    return bm, bn, nx*ny, slots
for (N,Cin,H,W,Cout,R,S,st,pd), c in cnt.items():
    OH = (H+2*pd[0]-R)//st[0]+1; OW = (W+2*pd[1]-S)//st[1]+1
    P = N*OH*OW; fl = 2*P*Cout*Cin*R*S
    for kind,(PP,CO) in (("fwd",(P,Cout)),("dgrad",(N*H*W,Cin))):
        if kind=="dgrad" and (st!=(1,1) or Cin<16): continue
        bm,bn,wgs,slots = grid(PP,CO)
        rounds = wgs/slots
        eff = rounds/math.ceil(rounds) if rounds>1 else 1.0
        rows.append((fl*c*(1-eff), kind,(N,Cin,H,W,Cout,R,S,st[0]),c,bm,bn,wgs,round(rounds,2),round(eff,2), round(fl*c/1e9)))
rows.sort(reverse=True)
for r in rows[:14]: print(r[1:], 'lostGF', round(r[0]/1e9))
