"""Probe of the EXPERIMENTAL 64-row / 4-wave / two-workgroups-per-CU tile of hgnn_mlp_forward_f32_split3
(hgnn_set_option("mlp_split3_rows128", 2), csrc/mlp_split3_khalf.h with HGNN_KH_NW = 4): K = 256 -> 512 -> 256 on a direct
segment, against the 64-row / 8-wave kernel, through the training forward's pre-LayerNorm dumps.  In the faulty build the
layer-1 GEMM dump is exact, ~4 % of the rows have a wrong pre-LayerNorm output (one hidden element of lanes 48-63 wrong)
and ~6 % more a wrong output element; with hgnn_set_option("mlp_split3_one_wg", 1) every row is right.
Usage: python tools/experimental/split3_r64x2_probe.py [label]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HGNN_EXPERIMENTAL", "1")   # option value 2 (the experimental two-workgroup tile) is refused without it
import torch
from hierarchicalgnn_amd import _lib, fused, make_mlp
lib = _lib.load()
fused.set_fp32_split3(True)
fused.set_fp32_split3_training(True)
L = 256
M = 64 * 512 * 6
torch.manual_seed(M)
net = make_mlp(L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
for p in net.parameters():
    if p.dim() == 1:
        p.data.add_(0.2 * torch.randn_like(p))
direct = torch.randn(M, L, device="cuda")
segs = [(direct, None)]

def run(v):
    _lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", v))
    d, keep, M_, n_out = fused._descriptor(net, segs, direct)
    zs = [torch.empty((M, int(d.width[l + 1])), dtype=torch.float32, device="cuda") for l in range(2)]
    for l in range(2):
        d.save_pre[l] = zs[l].data_ptr()
    out = torch.empty((M, n_out), dtype=torch.float32, device="cuda")
    assert fused._try_split3(net, segs, d, keep, training=True)
    _lib.check(lib.hgnn_mlp_forward_f32_split3(ctypes.byref(d), _lib.ptr(out), _lib.current_stream(direct.device)))
    torch.cuda.synchronize()
    return zs[0], zs[1], out

ln2 = [m for m in net if isinstance(m, torch.nn.LayerNorm)][-1]
with torch.no_grad():
    r0, r1, ro = run(0)
    z0, z1, o = run(2)
    print("rows with a wrong layer-1 GEMM dump:", int(((z0 - r0).abs() > 1e-5 * float(r0.abs().max())).any(dim=1).sum()))
    good_z1 = ((z1 - r1).abs() <= 1e-5 * float(r1.abs().max())).all(dim=1)
    bad_o = ((o - ro).abs() > 1e-4).any(dim=1)
    rows = torch.nonzero(good_z1 & bad_o).flatten()
    print("rows with a correct pre-LN output but a wrong output:", len(rows))
    z = r1.double()
    mean = z.mean(dim=1)
    var = z.var(dim=1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + ln2.eps)
    sh = -mean * rstd
    w, b = ln2.weight.double(), ln2.bias.double()
    print(sys.argv[1] if len(sys.argv) > 1 else "", "rows with wrong pre-LN output:", int((~good_z1).sum()), " rows with a correct pre-LN output but a wrong output:", len(rows))
