"""victim / aggressor experiment: one process runs the AdaLN modulation GEMV (linear_small, 2 rows x 58368 x 1536: the launch the bisection
flagged) in a loop and checks every result against its first; N - 1 others loop ONE kernel of the library.
usage: python tools/canary/run_victim.py <aggressor op> [N] [seconds]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
op = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
victim = sys.argv[4] if len(sys.argv) > 4 else "gemv"
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
runner = os.path.join(ROOT, "tools", "canary", "run_canary.py")
sys.path.insert(0, ROOT)
import torch
import ladcast_amd.hip as hip
pad_mb = int(os.environ.get("VICTIM_VA_PAD_MB", "0"))  # shift this process's device allocations in the virtual address space
pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device="cuda") if pad_mb else None
g = torch.Generator().manual_seed(1)
x, W, b = torch.randn(2, 1536, generator=g).cuda(), (torch.randn(58368, 1536, generator=g) / 39).cuda(), torch.randn(58368, generator=g).cuda()
xl = torch.randn(2, 2250, 1536, generator=g).cuda()
x20 = torch.randn(20, 1536, generator=g).cuda()
sc = (0.1 * torch.randn(2, 3072, generator=g)).cuda()


def run_victim(y):
    if victim == "gemv":
        hip.linear_small(x, W, y, rows=2, N=58368, K=1536, bias=b, act_in=hip.ACT_SILU)
    elif victim == "gemv_rows20":  # 20 rows: linear_rows_mfma_kernel (x from LDS straight into MFMA operands)
        hip.linear_small(x20, W, y, rows=20, N=58368, K=1536, bias=b, act_in=hip.ACT_SILU)
    elif victim == "gemv_narrow":  # the one-column-per-wave instantiation
        hip.linear_small(x, W, y[:, :1536], rows=2, N=1536, K=1536, bias=b, act_in=hip.ACT_SILU)
    elif victim == "torch_mv":
        torch.matmul(x, W.t(), out=y)
    elif victim == "torch_layernorm":
        y.view(-1)[: xl.numel()].copy_(torch.nn.functional.layer_norm(xl, (1536,)).view(-1))
    elif victim == "torch_softmax":
        y.view(-1)[: xl.numel()].copy_(torch.softmax(xl, dim=-1).view(-1))
    elif victim == "torch_elementwise":
        torch.mul(W[:2, :].repeat(1, 38), 1.0009765625, out=y)
    elif victim == "ln":
        hip.layernorm_mod(xl, y.view(-1)[: xl.numel()].view_as(xl) if y.numel() >= xl.numel() else xl, B=2, rows=2250, D=1536, ldx=1536, x_bs=2250 * 1536, ldy=1536,
                          y_bs=2250 * 1536, scale=sc[:, 1536:], shift=sc, mod_bs=3072, mode=0, eps=1e-6)


shape = (2, 2250 * 1536) if victim in ("ln", "torch_layernorm", "torch_softmax") else (20, 58368) if victim == "gemv_rows20" else (2, 58368)
ref = torch.zeros(shape, device="cuda")
run_victim(ref)
torch.cuda.synchronize()  # the reference result is computed on an idle GPU, the aggressors start after it
if op.startswith("synth"):  # tools/canary/synthetic_aggressor.hip (mode = the digit), built to /tmp/synth_aggr
    workers = [subprocess.Popen(["/tmp/synth_aggr", op[5:], str(seconds + 8)], env=env, cwd=ROOT, stdout=subprocess.DEVNULL) for _ in range(N - 1)]
else:
    workers = [subprocess.Popen([sys.executable, runner, "worker", op, str(seconds + 8)], env={**env, "VICTIM_VA_PAD_MB": "0"}, cwd=ROOT) for _ in range(N - 1)]
time.sleep(8)
calls = bad = words = 0
worst = 0.0
t_end = time.time() + seconds
while time.time() < t_end:
    ys = []
    for _ in range(16):
        y = torch.zeros(shape, device="cuda")
        run_victim(y)
        ys.append(y)
    torch.cuda.synchronize()
    for y in ys:
        calls += 1
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            words += int((d > 0).sum())
            worst = max(worst, d.max().item())
print(f"victim [{victim}] (W at 0x{W.data_ptr():x}, VA pad {pad_mb} MiB) next to {N - 1} x [{op}]: {bad} of {calls} results differ ({words} words, worst abs diff {worst:.3e})", flush=True)
for w in workers:
    w.wait()
