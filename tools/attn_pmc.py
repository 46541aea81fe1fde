"""20 launches each of the packed (2nd generation) and split (3rd generation) attention at (B, S, H) = (1, 2250, 12) for a rocprofv3 --pmc pass:
cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d <dir> -- python3 tools/attn_pmc.py ; python tools/attn_pmc.py summarize <dir>"""
import csv, glob, os, sys
from collections import defaultdict

if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(sys.argv[2], "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "split" if "attn_fwd_split" in k else "packed" if "attn_fwd_packed" in k else "pack_pass" if "attn_pack" in k else None
            if name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, c in agg.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        line = f"{name:10s} " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items()))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
            line += f"  | mfma_busy/(4 SIMD x SQ_BUSY_CYCLES)={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CYCLES']):.3f}"
        print(line)
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

B, S, H = 1, 2250, 12
D = H * 128
qkv = torch.randn(B, S, 3 * D, device="cuda")
out = torch.empty(B, S, D, device="cuda")
pk = torch.empty(hip.attn_packed_bytes(B, S, H) // 4, device="cuda", dtype=torch.float32)
kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)
hip.attn_pack(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], pk, split_row=S, **kw)
sp = qkv.clone()
hip.attn_qkv_prepare_split(sp[:, :, :D], sp[:, :, D : 2 * D], sp[:, :, 2 * D :], split_row=S, **kw)
for _ in range(20):
    hip.attn_fwd_packed(pk, out, B=B, S=S, H=H, ldo=D, o_bs=S * D, out_split=True)
    hip.attn_fwd_split(sp[:, :, :D], sp[:, :, D : 2 * D], sp[:, :, 2 * D :], out, ldo=D, o_bs=S * D, out_split=True, **kw)
torch.cuda.synchronize()
