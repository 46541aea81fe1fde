"""a few full-size DCAE encode + decode calls in bf16x3 mode (for rocprofv3 --kernel-trace --stats): frames"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
from ladcast_amd.models import AutoencoderDC
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(1234)
g = AutoencoderDC.from_config(bench.CONFIG_DCAE_84).cuda().eval().set_gemm_precision("bf16x3")
x = torch.randn(frames, 84, 120, 240, device="cuda"); st = torch.randn(1, 5, 120, 240, device="cuda")
if os.environ.get("DCAE_GRAPH"):
    g.enable_hip_graph(True)
for _ in range(4):
    z = g.encode(x, static_conditioning_tensor=st).latent
    y = g.decode(z).sample
torch.cuda.synchronize()
