import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import fused, make_mlp
L=256; M=2_000_000
torch.manual_seed(0)
net = make_mlp(3*L, 2*L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
tiny = torch.randn(1000, L, device="cuda")
big = torch.randn(M, L, device="cuda")
nodes = torch.randn(120_000, L, device="cuda")
idx = [torch.randint(0, 1000, (M,), device="cuda") for _ in range(3)]
nidx = [torch.randint(0, 120_000, (M,), device="cuda") for _ in range(2)]
flop = 2*(3*L*2*L+2*L*L)*M
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
with torch.no_grad():
    a=t(lambda: fused.fused_concat_mlp(net, [(tiny,idx[0]),(tiny,idx[1]),(tiny,idx[2])], None))
    b=t(lambda: fused.fused_concat_mlp(net, [(nodes,nidx[0]),(nodes,nidx[1]),(big,None)], big))
    c=t(lambda: fused.fused_concat_mlp(net, [(tiny,idx[0]),(tiny,idx[1]),(tiny,idx[2])], big))
print(f"all-L2 inputs, no skip: {a:.2f} ms {flop/a/1e9:.1f} TF | real inputs+skip: {b:.2f} ms {flop/b/1e9:.1f} TF | L2 inputs + skip/out stream: {c:.2f} ms")
