"""a few TAIL-schedule attention calls at (1, 2250, 16) and plain ones, for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
S, H, B = 2250, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1
D = H * 128
qkv = torch.randn(B, S, 3 * D, device="cuda")
out = torch.empty(B, S, D, device="cuda")
kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)
hip.attn_qkv_prepare_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], split_row=S, **kw)
for ws in (True, False):
    for _ in range(300):
        hip.attn_fwd_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, ldo=D, o_bs=S * D, out_split=True, use_workspace=ws, **kw)
    torch.cuda.synchronize()
