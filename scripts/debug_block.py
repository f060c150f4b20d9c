import torch, sys
sys.path.insert(0, '.')
from distributedpytorch_amd.models import resnet
from distributedpytorch_amd.ops import block, sink
from distributedpytorch_amd.parallel.flat import FlatParams
DEV='cuda'
x = torch.randn(8, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (8,), device=DEV)
def run(enabled, use_sink=True):
    torch.manual_seed(1)
    m = resnet.tiny_resnet().to(DEV).train()
    for mod in m.modules():
        if hasattr(mod, "running_mean"):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0.0, 0.2)
    flat = FlatParams(m)
    if use_sink: flat.install_sink(None)
    block.ENABLED = enabled
    loss = m.loss(x, y); loss.backward()
    block.ENABLED = True; sink.uninstall()
    return float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters()}
def cmp(a, b, tag):
    worst = max(((a[1][n]-b[1][n]).abs().max().item() / (b[1][n].abs().max().item()+1e-9), n) for n in a[1])
    print(tag, 'loss', a[0], b[0], 'worst rel diff', worst)
A = run(True); B = run(True); Cc = run(False); D = run(False); E = run(False, False)
cmp(A, B, 'fused vs fused      ')
cmp(Cc, D, 'unfused vs unfused  ')
cmp(A, Cc, 'fused vs unfused    ')
cmp(Cc, E, 'unfused sink vs none')
for n in A[1]:
    r = (A[1][n]-Cc[1][n]).abs().max().item() / (Cc[1][n].abs().max().item()+1e-9)
    print(f'{n:32s} {r:.4f}')
print('---- repeat study')
Fs = [run(True) for _ in range(4)]; Us = [run(False) for _ in range(4)]
for i in range(1,4):
    cmp(Fs[i], Fs[0], f'fused {i} vs fused 0    ')
for i in range(1,4):
    cmp(Us[i], Us[0], f'unfused {i} vs unfused 0')
