import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.conv import _conv_fwd
DEV="cuda:0"
def run(N, Cin, H, W, Cout, iters=30):
    x=torch.randn(N,Cin,H,W,device=DEV); w=(torch.randn(Cout,Cin,3,3,device=DEV)*0.05).contiguous(memory_format=torch.channels_last)
    sc=torch.rand(Cout,device=DEV)+0.5; b=torch.randn(Cout,device=DEV)
    for _ in range(3): _conv_fwd(x,w,b,(1,1),(1,1),relu=True,scale=sc)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(iters): _conv_fwd(x,w,b,(1,1),(1,1),relu=True,scale=sc)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    n,ms=_lib.prof_get("conv3x3_patch"); us=ms/max(n,1)*1e3; fl=2.0*N*H*W*Cin*Cout*9
    print(json.dumps({"N":N,"Cin":Cin,"HW":H,"Cout":Cout,"us":round(us,1),"TFLOPs":round(fl/us/1e6,1),"ideal_us_at_154":round(fl/154e6,1)}))
for cin in (64,128,256,512,1024):
    run(4,cin,64,64,256)
for cin in (64,128,256,512):
    run(16,cin,64,64,256)
