# rocprofv3 per-kernel summary of an EC-IN inference forward (fused path only): bash tools/prof_model_fwd.sh [L] [bf16]
set -e
cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out
L=${1:-128}
cat > /tmp/fwd_once.py <<PY
import sys, torch
sys.path.insert(0, "/root/repo")
from hierarchicalgnn_amd import synth
from hierarchicalgnn_amd.models import EC_InteractionGNN
L = $L
hp = dict(spatial_channels=3, latent=L, hidden=2 * L, n_interaction_graph_iters=14, nb_node_layer=3, nb_edge_layer=2,
          output_layers=3, hidden_output_activation="GELU", hidden_activation="GELU", layernorm=True, share_weight=False,
          feature_dtype="${2:-fp32}")
torch.manual_seed(1236)
model = EC_InteractionGNN(hp).cuda().eval()
x, ei = synth.trackml_event()
x, ei = x.cuda(), ei.cuda()
with torch.no_grad():
    for _ in range(6):
        model(x, ei)
torch.cuda.synchronize()
PY
cd /tmp
rm -rf $O/prof_fwd
rocprofv3 --kernel-trace --stats -d $O/prof_fwd -o fwd -- python3 /tmp/fwd_once.py > $O/fwd_prof.out 2> $O/fwd_prof.err
DB=$(find $O/prof_fwd -name "*.db" | head -1)
python3 /root/repo/tools/rocpd_summary.py stats $DB $O/fwd_kernel_stats_L${L}_${2:-fp32}.csv
head -16 $O/fwd_kernel_stats_L${L}_${2:-fp32}.csv | cut -c1-150
