"""CPU study of the rounding error of 1-D Winograd forms for the 64->64 3x3 layer (fp32 transforms and accumulation over 64 channels x 3 rows,
filter transform in fp64) against an fp64 direct convolution: F(2,3) (built), F(4,3) (candidate), direct fp32.  python tools/wino_error_study.py"""
import numpy as np, torch
torch.manual_seed(0)
C=64; H=16; W=96
x=torch.randn(C,H+2,W+2+4,dtype=torch.float64)          # padded input (generic interior)
w=torch.randn(64,C,3,3,dtype=torch.float64)*np.sqrt(2.0/(C*9))
def direct(x,w,dt):
    x=x.to(dt); w=w.to(dt)
    out=torch.zeros(64,H,W,dtype=dt)
    for c in range(C):             # sequential accumulation over channels like an MFMA chain
        for dy in range(3):
            for dx in range(3):
                out+= w[:,c,dy,dx].view(64,1,1)*x[c,dy:dy+H,dx:dx+W].unsqueeze(0)
    return out
ref=direct(x,w,torch.float64)
d32=direct(x,w,torch.float32).double()
def wino(x,w,m):
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=np.float64)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]])
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],dtype=np.float64)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=np.float64)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]])
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],dtype=np.float64)
    a=m+2
    BT32=torch.tensor(BT,dtype=torch.float32); AT32=torch.tensor(AT,dtype=torch.float32)
    U=torch.einsum('pk,ocyk->ocyp',torch.tensor(G),w).float()     # filter transform in fp64, rounded once
    x32=x.float()
    nt=W//m
    out=torch.zeros(64,H,W,dtype=torch.float32)
    # input tiles: d[c,y,t,0..a)
    idx=(torch.arange(nt)*m).view(-1,1)+torch.arange(a).view(1,-1)
    d=x32[:,:,idx]                               # [C,H+2,nt,a]
    V=torch.einsum('pk,cytk->cytp',BT32,d)       # fp32 transform
    M=torch.zeros(64,H,nt,a,dtype=torch.float32)
    for c in range(C):
        for dy in range(3):
            M+= U[:,c,dy,:].view(64,1,1,a)*V[c,dy:dy+H].unsqueeze(0)
    o=torch.einsum('jp,oytp->oytj',AT32,M)       # [64,H,nt,m]
    return o.reshape(64,H,nt*m).double()
for m in (2,4):
    o=wino(x,w,m)
    e=(o-ref).abs()
    print('F(%d,3): max %.3e mean %.3e'%(m,e.max(),e.mean()))
e=(d32-ref).abs(); print('direct fp32: max %.3e mean %.3e'%(e.max(),e.mean()))
print('output std %.3f'%ref.std())
