import sys; sys.path[:0]=["/root/repo/dusty-gan-v2_amd"]
import torch, torch.nn.functional as F
from gans.models.ops import native
torch.manual_seed(0)
for (kh,kw,s,C,O,H,W) in [(4,4,2,16,32,18,66),(4,16,1,32,1,4,16),(4,4,2,2,64,66,258)]:
    for dt in (torch.float32, torch.bfloat16):
        x=torch.randn(4,H,W,C,device="cuda",dtype=dt,requires_grad=True); w=torch.randn(O,kh,kw,C,device="cuda",dtype=dt,requires_grad=True)
        g=native.ConvGeom(kh,kw,s,0,False)
        try:
            y=native.conv_ring(x,w,g)
        except Exception as e:
            print(kh,kw,s,C,O,dt,"FWD FAIL",repr(e)[:200]); continue
        xr=x.detach().float().permute(0,3,1,2).requires_grad_(True); wr=w.detach().float().permute(0,3,1,2).requires_grad_(True)
        yr=F.conv2d(xr,wr,stride=s)
        print(kh,kw,s,C,O,dt,"fwd err",float((y.float().permute(0,3,1,2)-yr).abs().max()/yr.abs().max()))
        gy=torch.randn_like(y)
        try:
            gx,gw=torch.autograd.grad(y,(x,w),gy,create_graph=(dt==torch.float32))
            gxr,gwr=torch.autograd.grad(yr,(xr,wr),gy.float().permute(0,3,1,2))
            print("   bwd err",float((gx.float().permute(0,3,1,2)-gxr).abs().max()/gxr.abs().max()),float((gw.float().permute(0,3,1,2)-gwr).abs().max()/gwr.abs().max()))
            if dt==torch.float32:
                l=(gx**2).sum(); gg=torch.autograd.grad(l,w)[0]
                l2=(gxr_:=torch.autograd.grad(F.conv2d(xr,wr,stride=s),xr,gy.float().permute(0,3,1,2),create_graph=True)[0]); gg2=torch.autograd.grad((l2**2).sum(),wr)[0]
                print("   double-bwd err",float((gg.permute(0,3,1,2)-gg2).abs().max()/gg2.abs().max()))
        except Exception as e:
            print("   BWD FAIL",repr(e)[:300])
