import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import _lib, fused, make_mlp, mlp
lib = _lib.load()
fused.set_fp32_split3(True)
L = 256
res = {}
for nseg, M in [(2, 40000), (2, 120000), (2, 240000), (2, 480000), (3, 65536), (3, 131072), (3, 262144), (3, 500000), (3, 1000000)]:
    torch.manual_seed(0)
    net = make_mlp(nseg * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh" if nseg == 3 else "GELU", hidden_activation="GELU").cuda()
    n_tab = 120000
    table = torch.randn(n_tab, L, device="cuda")
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.sort(torch.randint(0, n_tab, (M,), device="cuda")).values
    direct = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)] if nseg == 3 else [(direct, None), (torch.randn(M, L, device="cuda"), None)]
    row = {}
    for v in (0, 1):
        _lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", v))
        with torch.no_grad():
            for _ in range(3):
                mlp.concat_mlp(net, segs, skip=direct)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); mlp.concat_mlp(net, segs, skip=direct); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
        row["rows128" if v else "rows64"] = round(sorted(ts)[len(ts) // 2], 4)
    res[f"nseg{nseg}_M{M}"] = row
    print(f"nseg{nseg}_M{M}", row, flush=True)
_lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", 1))
