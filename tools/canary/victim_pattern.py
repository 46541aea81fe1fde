"""What do the wrong words of the GPU-sharing effect look like?  The victim (the wide AdaLN GEMV, linear_small_kernel<4>: 2 rows x 58368
columns x K = 1536) runs on ANALYTIC inputs next to N - 1 processes looping the 4-wave split attention, so that every wrong word can be
read: with x = 1, W = 1 every output is exactly 1536 (integers: no rounding, any order), and a wrong value says how many terms were
lost or doubled; with W[n][k] = (k == n mod K) the output is x[n mod K] - which element of the staged x was read.
usage: python tools/canary/victim_pattern.py [N] [seconds] [ones|pick|coded|coded2] [aggressor: attn_b2 | ... | synth0..synth6]      (A/B library: LDC_LINEAR_SMALL_ITERS=1 walks one column group)"""
import collections, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
kind = sys.argv[3] if len(sys.argv) > 3 else "ones"
aggressor = sys.argv[4] if len(sys.argv) > 4 else "attn_b2"
sys.path.insert(0, ROOT)
import torch
import ladcast_amd.hip as hip

K, NC = 1536, 58368
if kind == "ones":
    x = torch.ones(2, K)
    W = torch.ones(NC, K)
    want = torch.full((2, NC), float(K))
elif kind == "coded":  # x[k] = 1 + (index of the 64-byte piece of a W row that holds k): a lost piece q shows as -16 (q + 1)
    x = (1 + torch.arange(K) // 16).float().repeat(2, 1)
    W = torch.ones(NC, K)
    want = torch.full((2, NC), float(16 * (96 * 97 // 2)))
elif kind == "coded2":  # x[k] = 1 + 1000 (k mod 4) + lane that loads k: which lanes / which float4 components are lost
    kk = torch.arange(K)
    x = (1 + 1000 * (kk % 4) + (kk % 256) // 4).float().repeat(2, 1)
    W = torch.ones(NC, K)
    want = torch.full((2, NC), float(x[0].sum().item()))
else:
    x = torch.arange(1, K + 1, dtype=torch.float32).repeat(2, 1)
    x[1] += 4096
    W = torch.zeros(NC, K)
    W[torch.arange(NC), torch.arange(NC) % K] = 1.0
    want = x[:, torch.arange(NC) % K]
x, W, want = x.cuda(), W.cuda(), want.cuda()
y0 = torch.zeros(2, NC, device="cuda")
hip.linear_small(x, W, y0, rows=2, N=NC, K=K)
torch.cuda.synchronize()
assert torch.equal(y0, want), "the victim is wrong on an idle GPU"
if hasattr(hip.lib, "ldc_debug_ls_counters"):
    import ctypes
    c0 = (ctypes.c_uint * 16)()
    hip.lib.ldc_debug_ls_counters(c0)
    print("  in-kernel check on the idle GPU:", list(c0))
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
if os.environ.get("AGGRESSOR_LIB_PATH"):  # the aggressors load another build of the library than the victim
    env["LDC_LIB_PATH"] = os.environ["AGGRESSOR_LIB_PATH"]
    print("aggressors load", env["LDC_LIB_PATH"])
runner = os.path.join(ROOT, "tools", "canary", "run_canary.py")
local = None
if aggressor.startswith("local"):  # the synthetic aggressor on a second stream of THIS process (/tmp/libsynth.so): co-residency without a second process
    import ctypes
    local = ctypes.CDLL("/tmp/libsynth.so")
    local.synth_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    side = torch.cuda.Stream()
    workers = []
elif aggressor.startswith("synth"):  # tools/canary/synthetic_aggressor.hip, mode = the digit: one instruction class in the attention's footprint
    workers = [subprocess.Popen(["/tmp/synth_aggr", aggressor[5:], str(seconds + 8)], env=env, cwd=ROOT, stdout=subprocess.DEVNULL) for _ in range(N - 1)]
else:
    workers = [subprocess.Popen([sys.executable, runner, "worker", aggressor, str(seconds + 8)], env=env, cwd=ROOT, stdout=subprocess.DEVNULL) for _ in range(N - 1)]
time.sleep(8)
calls = bad = 0
vals = collections.Counter()
pos64 = collections.Counter()
runs = collections.Counter()
pieces = collections.Counter()
addr_4k = collections.Counter()
pairs = []
rows = collections.Counter()
shown = 0
t_end = time.time() + seconds
while time.time() < t_end:
    ys = []
    for _ in range(16):
        y = torch.zeros(2, NC, device="cuda")
        if local is not None:  # ~1 ms of aggressor on the side stream (256 workgroups: one per CU), the victim lands in the middle of it
            local.synth_launch(int(aggressor[5:]), 2000, 256, ctypes.c_void_p(side.cuda_stream))
        hip.linear_small(x, W, y, rows=2, N=NC, K=K)
        ys.append(y)
    torch.cuda.synchronize()
    for y in ys:
        calls += 1
        if torch.equal(y, want):
            continue
        bad += 1
        d = (y != want).cpu()
        yc, wc = y.cpu(), want.cpu()
        for r in range(2):
            idx = torch.nonzero(d[r]).flatten().tolist()
            rows[r] += len(idx)
            run = 0
            for q, n in enumerate(idx):
                vals[(yc[r, n] - wc[r, n]).item()] += 1
                if kind == "coded":
                    pc = int((wc[r, n] - yc[r, n]).item()) // 16 - 1
                    if 0 <= pc < 96 and (wc[r, n] - yc[r, n]).item() == 16 * (pc + 1):
                        pieces[pc] += 1
                        addr_4k[((n * K * 4 + 64 * pc) % 4096) // 64] += 1
                        if len(pairs) < 400:
                            pairs.append((n, pc))
                pos64[n % 64] += 1
                run += 1
                if q + 1 == len(idx) or idx[q + 1] != n + 1:
                    runs[run] += 1
                    run = 0
            if idx and shown < 6:
                shown += 1
                print(f"  call {calls} row {r}: {len(idx)} wrong words; first: " + ", ".join(f"n={n} (n%64={n % 64}) got {yc[r, n].item():g} want {wc[r, n].item():g}" for n in idx[:12]), flush=True)
print(f"victim [{kind}] next to {N - 1} x [{aggressor}] (LDC_LINEAR_SMALL_ITERS={os.environ.get('LDC_LINEAR_SMALL_ITERS', 'default')}): {bad} of {calls} results wrong")
print("  wrong words per row:", dict(rows))
print("  got - want, most common:", vals.most_common(16))
print("  column mod 64 (one workgroup walks 64 consecutive columns: 4 iterations x 4 waves x 4):", sorted(pos64.items()))
print("  lengths of runs of consecutive wrong columns:", sorted(runs.items()))
if kind == "coded":
    print("  lost 64-byte piece of the W row (0..95), if one piece was lost:", sorted(pieces.items()))
    print("  its byte offset inside a 4 KiB page, in 64-byte units:", sorted(addr_4k.items()))
    print("  (column, piece) of the first wrong words:", pairs[:400])
    print(f"  W at 0x{W.data_ptr():x}")
if hasattr(hip.lib, "ldc_debug_ls_counters"):  # A/B build with -DLDC_LS_CHECK: loaded words that were zero when used
    import ctypes
    c = (ctypes.c_uint * 16)()
    hip.lib.ldc_debug_ls_counters(c)
    c = list(c)
    print(f"  in-kernel check, loaded words == 0 at their use: by component x/y/z/w {c[0:4]}, .y|.w zero by column j of the wave {c[4:8]}, by 16-lane group {c[8:12]}")
for w in workers:
    w.wait()
