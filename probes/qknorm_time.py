import sys, torch
sys.path.insert(0, '/root/repo')
from iggt_official_amd import _C
from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D
_C.load()
S, g, psi = 32, 37, 5
P = psi + g * g
T = S * P
qkv = torch.randn(T, 3072, device='cuda').half()
qw = torch.ones(64, device='cuda'); qb = torch.zeros(64, device='cuda')
cos, sin = RotaryPositionEmbedding2D(100).tables(64, g, torch.device('cuda'))
qkmax = torch.zeros(_C.QKMAX_NUMEL, device='cuda')
def f(): _C.qknorm_rope(qkv, qkv, qkv[:, 1024:], None, qw, qb, qw, qb, cos, sin, T, P, g, psi, 1e-5, q_scale=0.18, qkmax=qkmax)
for _ in range(5): f()
torch.cuda.synchronize()
ts=[]
for _ in range(7):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/20)
print("qknorm_rope + qkmax_reduce, 32 views: %.1f us" % (sorted(ts)[3]*1e3))
