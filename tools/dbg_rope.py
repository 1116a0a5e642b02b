import torch, sys
sys.path.insert(0,'.')
from sglang_custom_amd import ops
dev='cuda'; torch.manual_seed(0)
T,hq,hkv,hd=37,4,2,64; rd=64; dt=torch.bfloat16
inv = 1.0 / (10000 ** (torch.arange(0, rd, 2, dtype=torch.float) / rd))
t = torch.arange(512, dtype=torch.float); fr = torch.einsum('i,j->ij', t, inv)
cache = torch.cat((fr.cos(), fr.sin()), -1).to(dev).to(dt)
pos = torch.randint(0, 512, (T,), device=dev)
q = torch.randn(T, hq*hd, device=dev).to(dt); k = torch.randn(T, hkv*hd, device=dev).to(dt)
cs = cache.index_select(0,pos); cos,sin = cs.chunk(2,-1)
x = q.view(T,hq,hd); x1,x2 = torch.chunk(x,2,-1); c=cos.unsqueeze(-2); s=sin.unsqueeze(-2)
o1_gpu = x1*c - x2*s
r=lambda t: t.bfloat16().float()
o1_emu = r(r(x1.float()*c.float()) - r(x2.float()*s.float())).bfloat16()
o1_cpu = (x1.cpu()*c.cpu() - x2.cpu()*s.cpu()).to(dev)
print('gpu-vs-emu', (o1_gpu.float()-o1_emu.float()).abs().max().item(), 'gpu-vs-cpu', (o1_gpu.float()-o1_cpu.float()).abs().max().item())
q2=q.clone(); k2=k.clone()
ops.rope_inplace(pos,q2,k2,hd,cache,True)
mine=q2.view(T,hq,hd)[...,:hd//2]
d=(mine.float()-o1_cpu.float()).abs()
print('kernel-vs-cpu', d.max().item(), (d>0).sum().item(), 'of', d.numel())
idx=(d>0).nonzero()[:5]
for i in idx:
    i=tuple(i.tolist()); print(i, x1[i].item(), c[i[0],0,i[2]].item(), x2[i].item(), s[i[0],0,i[2]].item(), mine[i].item(), o1_cpu[i].item())
