import torch, sys
sys.path.insert(0, '.')
from distributedpytorch_amd.models.resnet import tiny_resnet
from distributedpytorch_amd.parallel.flat import FlatParams
from distributedpytorch_amd import ops
from distributedpytorch_amd.nn import conv_bn
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = tiny_resnet().to(dev).train()
flat = FlatParams(m)
x = torch.randn(16, 3, 64, 64, device=dev)
def st(name, t):
    t = t.float()
    print(f"{name:30s} shape={tuple(t.shape)} nan={bool(torch.isnan(t).any())} absmax={float(t.abs().max()):.4g} mean={float(t.mean()):.4g}")
st('master', flat.master); st('lp', flat.lp)
h = ops.to_nhwc(x, 8); st('nhwc', h)
y, s = m.stem.conv1(h, want_stats=True); st('stem conv', y); st('stem stats', s)
z = m.stem.bn1(y, stats=s, relu=True); st('stem bn', z)
z2 = m.stem.bn1(y, stats=None, relu=True); st('stem bn (own stats)', z2)
p = ops.max_pool_3x3_s2_nhwc(z2); st('pool', p)
b = m.block1
y, s = b.conv1(p, want_stats=True); st('b1 conv1', y); st('b1 stats', s)
st('b1 stats ref sum', y.float().reshape(-1, y.shape[-1]).sum(0)); st('b1 stats sum', s.sum(0)[0])
z = b.bn1(y, stats=s, relu=True); st('b1 bn1', z)
out = m(x); st('logits', out)
y = torch.randint(0, 16, (16,), device=dev)
loss = ops.softmax_cross_entropy(out, y, 16); st('loss', loss.detach())
print('ref loss', float(torch.nn.functional.cross_entropy(out.detach().float(), y)))
loss2 = m.loss(x, y); st('loss2', loss2.detach())
loss2.backward(); st('grad', flat.grad)
for n_, p_ in zip(flat.names, flat.params):
    if torch.isnan(p_.grad).any(): print('nan grad in', n_)
