import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
B, S, H = [int(v) for v in sys.argv[1:4]]
D = H * 128
qkv = torch.randn(B, S, 3 * D, device="cuda"); O = torch.empty(B, S, D, device="cuda")
for _ in range(6):
    hip.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2*D], qkv[:, :, 2*D:], O, B=B, S=S, H=H, ld_qkv=3*D, qkv_bs=S*3*D, ldo=D, o_bs=S*D, split_bf16=True)
torch.cuda.synchronize()
