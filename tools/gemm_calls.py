"""Per-call-signature GEMM time inside the real forward (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import defaultdict
import ladcast_amd.hip as hip
from ladcast_amd.models import LaDCastTransformer3DModel
from bench import CONFIGS
torch.manual_seed(1234)
m = LaDCastTransformer3DModel.from_config(CONFIGS["375M"]).cuda().eval().set_gemm_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = torch.randn(B, 84, 4, 15, 30, device="cuda"); known = torch.randn(B, 84, 1, 15, 30, device="cuda"); t = torch.tensor([0.3], device="cuda"); ts = torch.tensor([2018010100], device="cuda")
for _ in range(3): m(x, t, known, time_elapsed=ts)
recs = []
orig = hip.gemm_grouped
def hooked(problems, split_bf16=False):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(problems, split_bf16=split_bf16); e.record()
    recs.append((tuple((p[0].d.M, p[0].d.N, p[0].d.K, p[0].d.batch) for p in problems), s, e))
hip.gemm_grouped = hooked
for _ in range(5): m(x, t, known, time_elapsed=ts)
torch.cuda.synchronize()
agg = defaultdict(list)
for sig, s, e in recs: agg[sig].append(s.elapsed_time(e))
tot = 0
for sig, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    ms = sum(v) / 5; n = len(v) // 5; fl = sum(2.0 * M * N * K * b for (M, N, K, b) in sig)
    tot += ms
    print(f"{str(sig):75s} x{n:2d}  {ms/n*1e3:7.1f} us each  {fl/(ms/n)/1e9:6.1f} TF/s   {ms:6.3f} ms/fwd")
print("total GEMM ms per forward", tot)
