import torch
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(200000, 4, device=dev) * torch.rand(200000, 1, device=dev) * 3
n = x.norm(2, dim=1)
sq = x * x
cands = {
 "seq": ((sq[:,0]+sq[:,1])+sq[:,2])+sq[:,3],
 "tree02_13": (sq[:,0]+sq[:,2])+(sq[:,1]+sq[:,3]),
 "tree01_23": (sq[:,0]+sq[:,1])+(sq[:,2]+sq[:,3]),
 "seq_rev": ((sq[:,3]+sq[:,2])+sq[:,1])+sq[:,0],
}
for k, v in cands.items():
    print(k, int((v.sqrt() != n).sum()))
# fma variants: acc = fma(x,x,acc) sequential
acc = torch.zeros(200000, device=dev, dtype=torch.float64)
xd = x.double()
a = sq[:,0]
for i in (1,2,3):
    a = (a.double() + xd[:,i]*xd[:,i]).float()   # fma(x,x,acc): exact product + single rounding
print("fma_seq", int((a.sqrt() != n).sum()))
y = torch.nn.functional.normalize(x)
print("normalize == x / n[:,None]:", int((y != x / n.clamp_min(1e-12)[:, None]).sum()))
z = torch.randn(100000, device=dev) * 4
print("sigmoid composite mismatch:", int((torch.sigmoid(z) != 1.0 / (1.0 + torch.exp(-z))).sum()))
