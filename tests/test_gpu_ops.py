"""Parity of every C-ABI kernel against a CPU restatement on the same seeded inputs.
All calls go through the C ABI (ladcast_amd.hip -> libladcast_hip.so)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import layers as L  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    import ladcast_amd.hip as h

    return h


def dev(t):
    return t.to("cuda")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize(
    "M,N,K,batch",
    [(128, 128, 32, 1), (2250, 1536, 1536, 1), (450, 84, 1536, 2), (1800, 1536, 84, 1), (37, 200, 260, 3), (2250, 1536, 7680, 1), (1, 5, 4, 1)],
)
def test_gemm_plain(hip, M, N, K, batch):
    A, W, b = rnd(batch, M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    C = torch.full((batch, M, N), float("nan"), device="cuda")
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, batch=batch, a_bs=M * K, c_bs=M * N, bias=dev(b))
    want = A.double() @ W.double().T + b.double()
    assert rel(C, want) < 2e-6
    assert torch.isfinite(C).all()


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogue_and_strides(hip, act):
    """bias + act + per-batch gate + residual, written in place into a strided slab of a wider buffer"""
    B, M, N, K, LDC = 2, 300, 260, 132, 520
    A, W, b = rnd(B, M + 7, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    gate, buf = rnd(B, 3 * N, seed=4), rnd(B, M, LDC, seed=5)
    d_buf = dev(buf)
    Cv = d_buf[:, :, 100:]
    hip.gemm(dev(A), dev(W), Cv, M=M, N=N, K=K, batch=B, a_bs=(M + 7) * K, ldc=LDC, c_bs=M * LDC, bias=dev(b),
             gate=dev(gate)[:, N:], gate_bs=3 * N, R=Cv, ldr=LDC, r_bs=M * LDC, act=act)
    v = A[:, :M].double() @ W.double().T + b.double()
    v = [v, F.silu(v), F.gelu(v, approximate="tanh"), F.relu(v)][act]
    want = buf.clone().double()
    want[:, :, 100 : 100 + N] = buf[:, :, 100 : 100 + N].double() + v * gate[:, None, N : 2 * N].double()
    assert rel(d_buf, want) < 2e-6
    assert torch.equal(d_buf[:, :, :100].cpu(), buf[:, :, :100])  # untouched columns


@pytest.mark.parametrize(
    "M,N,K,batch",
    [(128, 128, 32, 1), (2250, 1536, 1536, 1), (450, 84, 1536, 2), (1800, 1536, 84, 1), (37, 200, 260, 3), (2250, 1536, 7680, 1), (1, 5, 4, 1),
     (450, 1536, 6144, 1), (4500, 4608, 1536, 1)],
)
def test_gemm_streamk_single(hip, M, N, K, batch):
    A, W, b = rnd(batch, M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    C = torch.full((batch, M, N), float("nan"), device="cuda")
    hip.gemm_sk(dev(A), dev(W), C, M=M, N=N, K=K, batch=batch, a_bs=M * K, c_bs=M * N, bias=dev(b))
    want = A.double() @ W.double().T + b.double()
    assert torch.isfinite(C).all()
    assert rel(C, want) < 2e-6


def test_gemm_streamk_grouped_with_epilogues(hip):
    """four different problems (shapes, weights, epilogues, in-place residuals) in one persistent launch"""
    specs = [(1800, 1536, 1536, 2, 1), (450, 1536, 1536, 2, 2), (300, 260, 132, 1, 0), (77, 96, 6144, 1, 3)]
    probs, checks = [], []
    for i, (M, N, K, B, act) in enumerate(specs):
        A, W, b = rnd(B, M, K, seed=10 + i), rnd(N, K, seed=20 + i) / math.sqrt(K), rnd(N, seed=30 + i)
        gate, res = rnd(B, N, seed=40 + i), rnd(B, M, N, seed=50 + i)
        Cd = dev(res.clone())
        probs.append(hip.gemm_problem(dev(A), dev(W), Cd, M=M, N=N, K=K, batch=B, a_bs=M * K, c_bs=M * N, bias=dev(b), gate=dev(gate), gate_bs=N,
                                      R=Cd, ldr=N, r_bs=M * N, act=act))
        v = A.double() @ W.double().T + b.double()
        v = [v, F.silu(v), F.gelu(v, approximate="tanh"), F.relu(v)][act]
        checks.append((Cd, res.double() + v * gate[:, None, :].double()))
    hip.gemm_grouped(probs)
    for Cd, want in checks:
        assert rel(Cd, want) < 2e-6
    # determinism: the same launch twice gives bit-identical results
    A, W = dev(rnd(2250, 1536, seed=1)), dev(rnd(1536, 1536, seed=2))
    c1, c2 = torch.empty(2250, 1536, device="cuda"), torch.empty(2250, 1536, device="cuda")
    hip.gemm_sk(A, W, c1, M=2250, N=1536, K=1536)
    hip.gemm_sk(A, W, c2, M=2250, N=1536, K=1536)
    assert torch.equal(c1, c2)


def _bf16_split(x):
    hi = x.bfloat16().float()
    lo = (x - hi).bfloat16().float()
    return hi.double(), lo.double()


@pytest.mark.parametrize("M,N,K,batch", [(128, 128, 32, 1), (2250, 1536, 1536, 1), (450, 84, 1536, 2), (37, 200, 264, 3), (2250, 1536, 7680, 1), (1, 5, 8, 1)])
def test_gemm_bf16x3(hip, M, N, K, batch):
    """split-bf16 GEMM: (a) equals its own definition Ah.Wh + Ah.Wl + Al.Wh almost exactly (only fp32 accumulation
    order differs), (b) is within 1e-5 rel-L2 of the exact product (vs 2e-3 for plain bf16)."""
    A, W, b = rnd(batch, M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    Wp = hip.pack_weight_bf16x2(dev(W))
    C = torch.full((batch, M, N), float("nan"), device="cuda")
    hip.gemm_sk(dev(A), Wp, C, split_bf16=True, M=M, N=N, K=K, batch=batch, a_bs=M * K, c_bs=M * N, bias=dev(b))
    ah, al = _bf16_split(A)
    wh, wl = _bf16_split(W)
    defn = ah @ wh.T + ah @ wl.T + al @ wh.T + b.double()
    exact = A.double() @ W.double().T + b.double()
    assert torch.isfinite(C).all()
    assert rel(C, defn) < 1e-6
    assert rel(C, exact) < 1e-5


def test_gemm_bf16x3_grouped_epilogue(hip):
    specs = [(1800, 1536, 1536, 2, 2), (450, 1536, 1536, 2, 1), (300, 264, 136, 1, 0)]
    probs, checks = [], []
    for i, (M, N, K, B, act) in enumerate(specs):
        A, W, b = rnd(B, M, K, seed=10 + i), rnd(N, K, seed=20 + i) / math.sqrt(K), rnd(N, seed=30 + i)
        gate, res = rnd(B, N, seed=40 + i), rnd(B, M, N, seed=50 + i)
        Cd = dev(res.clone())
        probs.append(hip.gemm_problem(dev(A), hip.pack_weight_bf16x2(dev(W)), Cd, M=M, N=N, K=K, batch=B, a_bs=M * K, c_bs=M * N, bias=dev(b),
                                      gate=dev(gate), gate_bs=N, R=Cd, ldr=N, r_bs=M * N, act=act))
        v = A.double() @ W.double().T + b.double()
        v = [v, F.silu(v), F.gelu(v, approximate="tanh")][act]
        checks.append((Cd, res.double() + v * gate[:, None, :].double()))
    hip.gemm_grouped(probs, split_bf16=True)
    for Cd, want in checks:
        assert rel(Cd, want) < 1e-5


def _unsplit(buf, rows, cols):
    """decode the split activation format: every 32 bytes = [hi x8 | lo x8] bf16 -> (hi, lo) as fp32 [rows, cols]"""
    w = buf.detach().cpu().contiguous().view(torch.int16).reshape(rows, cols // 8, 2, 8)
    f = (w.to(torch.int32) << 16).view(torch.float32)
    return f[:, :, 0].reshape(rows, cols), f[:, :, 1].reshape(rows, cols)


@pytest.mark.parametrize("M,N,K", [(2250, 1536, 1536), (450, 4608, 1536), (2250, 1536, 7680), (300, 264, 160), (5000, 6144, 1536), (1, 8, 32),
                                   (37, 200, 96), (18000, 1536, 1536), (4500, 1536, 2048), (2250, 1528, 1024)])
def test_gemm_bf16x3_split_activation_formats(hip, M, N, K):
    """LDC_GEMM_A_SPLIT: activations pre-split by a producer give the same product (the split is the same operation,
    moved; the 16x16x32 kernel that serves this format sums in a different order than the 32x32x16 one, hence 1e-6
    and not bit equality); LDC_GEMM_C_SPLIT: the output is the hi / lo split of exactly the fp32 values the same
    kernel writes otherwise (both heights of the tile are covered)."""
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    gate, res = rnd(N, seed=4), rnd(M, N, seed=5)
    Wp = hip.pack_weight_bf16x2(dev(W))
    Ap = hip.pack_weight_bf16x2(dev(A))  # same [rows][K/8][hi|lo] layout
    kw = dict(M=M, N=N, K=K, bias=dev(b), gate=dev(gate), R=dev(res), ldr=N, act=2)
    C0 = torch.empty(M, N, device="cuda")
    hip.gemm_grouped([hip.gemm_problem(dev(A), Wp, C0, **kw)], split_bf16=True)
    C1 = torch.empty(M, N, device="cuda")
    hip.gemm_grouped([hip.gemm_problem(Ap, Wp, C1, flags=hip.GEMM_A_SPLIT, **kw)], split_bf16=True)
    assert rel(C1, C0.cpu()) < 1e-6
    C2 = torch.empty(M, N, device="cuda")
    hip.gemm_grouped([hip.gemm_problem(Ap, Wp, C2, flags=hip.GEMM_A_SPLIT | hip.GEMM_C_SPLIT, **kw)], split_bf16=True)
    hi, lo = _unsplit(C2, M, N)
    want_hi = C1.cpu().bfloat16().float()
    want_lo = (C1.cpu() - want_hi).bfloat16().float()
    assert torch.equal(hi, want_hi) and torch.equal(lo, want_lo)
    v = A.double() @ W.double().T + b.double()
    want = res.double() + F.gelu(v, approximate="tanh") * gate.double()
    assert rel(C0, want) < 1e-5 and rel(C1, want) < 1e-5


def test_gemm_bf16x3_v3_grouped_batched_deterministic(hip):
    """the pre-split-activation kernel on a grouped, batched launch with split tiles: matches the exact product, is
    bitwise reproducible call to call (fixed reduction order) and leaves the arrival counters re-armed"""
    specs = [(1800, 1536, 1536, 2), (450, 1536, 1536, 2), (300, 264, 160, 1)]
    probs, wants, outs = [], [], []
    for i, (M, N, K, B) in enumerate(specs):
        A, W, b = rnd(B, M, K, seed=10 + i), rnd(N, K, seed=20 + i) / math.sqrt(K), rnd(N, seed=30 + i)
        Ap = hip.pack_weight_bf16x2(dev(A).reshape(B * M, K))
        C = torch.empty(B, M, N, device="cuda")
        probs.append(hip.gemm_problem(Ap, hip.pack_weight_bf16x2(dev(W)), C, M=M, N=N, K=K, batch=B, a_bs=M * K, c_bs=M * N, bias=dev(b),
                                      flags=hip.GEMM_A_SPLIT))
        wants.append(A.double() @ W.double().T + b.double())
        outs.append(C)
    hip.gemm_grouped(probs, split_bf16=True)
    first = [c.clone() for c in outs]
    for c, w in zip(outs, wants):
        assert rel(c, w) < 1e-5
    for _ in range(3):
        for c in outs:
            c.fill_(float("nan"))
        hip.gemm_grouped(probs, split_bf16=True)
        for c, f0 in zip(outs, first):
            assert torch.equal(c, f0)


def test_gemm_split_flags_rejected_elsewhere(hip):
    a = torch.zeros(64, 64, device="cuda")
    with pytest.raises(RuntimeError):
        hip.gemm_grouped([hip.gemm_problem(a, a, a, M=64, N=64, K=64, flags=hip.GEMM_A_SPLIT)], split_bf16=False)
    with pytest.raises(RuntimeError):  # mixed activation formats in one launch
        w = hip.pack_weight_bf16x2(a)
        hip.gemm_grouped([hip.gemm_problem(a, w, torch.empty_like(a), M=64, N=64, K=64, flags=hip.GEMM_A_SPLIT),
                          hip.gemm_problem(a, w, torch.empty_like(a), M=64, N=64, K=64)], split_bf16=True)


@pytest.mark.parametrize("split", [False, True])
def test_streamk_handoff_stress(hip, split):
    """Split-tile slabs under changing data: back-to-back calls on shapes whose ranges split almost every tile, with NEW
    inputs each time (a stale slab would show as a mismatch), checking every word, interleaved with different-shaped
    calls that re-use the same workspace slots."""
    shapes = [(2250, 1536, 1536), (450, 1536, 6144), (1800, 4608, 1536), (2250, 1536, 7680)]
    Ws = {}
    for (M, N, K) in shapes:
        W = rnd(N, K, seed=K + N) / math.sqrt(K)
        Ws[(M, N, K)] = (W, hip.pack_weight_bf16x2(dev(W)) if split else dev(W))
    tol = 1e-5 if split else 2e-6
    for it in range(12):
        M, N, K = shapes[it % len(shapes)]
        W, Wd = Ws[(M, N, K)]
        A = rnd(M, K, seed=100 + it)
        C = torch.full((M, N), float("nan"), device="cuda")
        hip.gemm_sk(dev(A), Wd, C, split_bf16=split, M=M, N=N, K=K)
        got = C.cpu().double()
        want = A.double() @ W.double().T
        assert torch.isfinite(got).all(), it
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err < 50 * tol, (it, err)  # every word, max-norm
        assert rel(C, want) < tol, it


def test_gemm_bf16x3_inlaunch_reduction_is_deterministic_and_rearmed(hip):
    """The last-arriver reduction re-reads every slab in fixed order: repeated launches are bitwise identical, and the
    arrival counters are back to zero after every call (checked through the workspace)."""
    M, N, K = 2250, 1536, 7680
    A, W = dev(rnd(M, K, seed=1)), rnd(N, K, seed=2) / math.sqrt(K)
    Wp = hip.pack_weight_bf16x2(dev(W))
    outs = []
    for _ in range(4):
        C = torch.empty(M, N, device="cuda")
        hip.gemm_sk(A, Wp, C, split_bf16=True, M=M, N=N, K=K)
        outs.append(C)
    for c in outs[1:]:
        assert torch.equal(outs[0], c)
    ws = hip._grouped_workspace(A.device)
    assert int(ws.view(torch.int32)[: (1 << 20) // 4].abs().sum().item()) == 0


def test_gemm_fp32_streamk_inlaunch_reduction_is_deterministic_and_rearmed(hip):
    """round 3: the exact-fp32 stream-K kernel reduces its split tiles INSIDE the launch (last arriver, fixed order; the second
    "fix-up" launch is gone): repeated launches are bitwise identical - with other launches in between that leave different data in
    the slabs -, the arrival counters are back to zero after every call, ragged shapes whose tiles are cut into 3+ pieces included,
    and a grouped launch of different k-depths."""
    for (M, N, K) in ((2250, 1536, 7680), (450, 1536, 6144), (129, 130, 36), (2250, 4608, 1536)):
        A, W, b = dev(rnd(M, K, seed=1)), dev(rnd(N, K, seed=2) / math.sqrt(K)), dev(rnd(N, seed=3))
        other = dev(rnd(M, K, seed=9) * 3)
        outs = []
        for it in range(5):
            C = torch.full((M, N), float("nan"), device="cuda")
            hip.gemm_sk(A, W, C, M=M, N=N, K=K, bias=b, act=hip.ACT_GELU_TANH)
            outs.append(C)
            if it % 2 == 0:
                hip.gemm_sk(other, W, torch.empty(M, N, device="cuda"), M=M, N=N, K=K)
        for c in outs[1:]:
            assert torch.equal(outs[0], c), (M, N, K)
        want = F.gelu(A.cpu().double() @ W.cpu().double().T + b.cpu().double(), approximate="tanh")
        assert rel(outs[0], want) < 2e-6
        ws = hip._grouped_workspace(A.device)
        assert int(ws.view(torch.int32)[: (1 << 20) // 4].abs().sum().item()) == 0


def test_gemm_fp32_ring_kernel_vs_register_staged_kernel(hip):
    """round 3: exact-fp32 problems with K % 32 == 0, contiguous weights and 16-byte rows run on the LDS-DMA ring kernel (fp32 operand
    rows, v_mfma_f32_16x16x4_f32); the descriptor flag LDC_GEMM_F32_REGSTAGE keeps them on the register-staged stream-K kernel (round 4:
    a caller's choice in the ABI - the shipped library reads no environment variable any more).  Both against
    fp64 and against each other: ragged M / N, batches, every epilogue term, C as a strided slab, in-place residual."""
    cases = [(2250, 1536, 1536, 1, 0), (300, 260, 128, 2, 2), (4500, 4608, 1536, 1, 1), (77, 96, 6144, 1, 3), (1800, 84, 1536, 2, 0),
             (9000, 6144, 1536, 1, 2)]
    for i, (M, N, K, B, act) in enumerate(cases):
        LDC = N + 12
        A, W, b = rnd(B, M, K, seed=10 + i), rnd(N, K, seed=20 + i) / math.sqrt(K), rnd(N, seed=30 + i)
        gate, buf = rnd(B, N, seed=40 + i), rnd(B, M, LDC, seed=50 + i)
        v = A.double() @ W.double().T + b.double()
        v = [v, F.silu(v), F.gelu(v, approximate="tanh"), F.relu(v)][act]
        want = buf.clone().double()
        want[:, :, 8 : 8 + N] += v * gate[:, None, :].double()
        got = {}
        for ring in ("1", "0"):
            d_buf = dev(buf)
            Cv = d_buf[:, :, 8:]
            hip.gemm_grouped([hip.gemm_problem(dev(A), dev(W), Cv, M=M, N=N, K=K, batch=B, a_bs=M * K, ldc=LDC, c_bs=M * LDC, bias=dev(b),
                                               gate=dev(gate), gate_bs=N, R=Cv, ldr=LDC, r_bs=M * LDC, act=act,
                                               flags=0 if ring == "1" else hip.GEMM_F32_REGSTAGE)])
            got[ring] = d_buf.cpu()
            assert rel(d_buf, want) < 2e-6, (M, N, K, ring)
            assert torch.equal(got[ring][:, :, :8], buf[:, :, :8]) and torch.equal(got[ring][:, :, 8 + N :], buf[:, :, 8 + N :])
        assert rel(got["1"], got["0"]) < 1e-6
        assert not torch.equal(got["1"], got["0"]) or M * N < 50000  # two kernels, two summation orders: the flag does select the other one
    # grouped: the dual block's two streams in one launch, different row counts; bitwise repeatable
    A, W0, W1 = dev(rnd(2250, 1536, seed=1)), dev(rnd(6144, 1536, seed=2) / 39.0), dev(rnd(6144, 1536, seed=3) / 39.0)
    outs = []
    for _ in range(3):
        C = torch.full((2250, 6144), float("nan"), device="cuda")
        hip.gemm_grouped([hip.gemm_problem(A, W0, C, M=1800, N=6144, K=1536, act=hip.ACT_GELU_TANH),
                          hip.gemm_problem(A[1800:], W1, C[1800:], M=450, N=6144, K=1536, act=hip.ACT_GELU_TANH)])
        outs.append(C)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = torch.cat([F.gelu(A[:1800].cpu().double() @ W0.cpu().double().T, approximate="tanh"),
                      F.gelu(A[1800:].cpu().double() @ W1.cpu().double().T, approximate="tanh")])
    assert rel(outs[0], want) < 2e-6


def test_gemm_rejects_bad_arguments(hip):
    a = torch.zeros(8, 6, device="cuda")
    with pytest.raises(RuntimeError):
        hip.gemm(a, a, a, M=8, N=8, K=6)  # K % 4 != 0
    with pytest.raises(RuntimeError):
        hip.gemm(torch.zeros(4, 4), a, a, M=4, N=4, K=4)  # host tensor


@pytest.mark.parametrize("B,S,H", [(1, 2250, 12), (2, 450, 2), (1, 33, 1), (1, 128, 3), (3, 70, 2), (1, 1, 1)])
def test_attention(hip, B, S, H):
    D = H * 128
    qkv = rnd(B, S, 3 * D, seed=11)
    qkv[..., :D] *= 2.0  # wider logits than unit variance
    d_qkv = dev(qkv)
    out = torch.full((B, S, D + 64), float("nan"), device="cuda")
    hip.attn_fwd(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D,
                 ldo=D + 64, o_bs=S * (D + 64))
    q, k, v = [t.reshape(B, S, H, 128).transpose(1, 2).double() for t in qkv.split(D, dim=-1)]
    want = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, D)
    assert rel(out[:, :, :D], want) < 2e-6
    assert torch.isnan(out[:, :, D:]).all()  # pad columns untouched


@pytest.mark.parametrize("B,S,H,bias", [(1, 2250, 12, False), (1, 2250, 16, True), (2, 700, 13, True), (1, 450, 12, False), (5, 300, 4, True), (8, 129, 7, False),
                                         (3, 1000, 20, False)])
def test_attention_f32_balanced_schedule(hip, B, S, H, bias):
    """exact-fp32 attention with its workspace (ldc_attn_fwd_ws, round 5): the launch's (unit, key tile) items in one contiguous range per
    CU, pieces of a unit combined by the last arriver.  216 units (one 375M member: every unit has 2 or 3 pieces), 288 (1.6B), 13.4-tile
    ranges over 22-tile units (2 x 700 x 13), ranges that hold a whole unit between two partial ones (3 x 1000 x 20: 60 tiles per range,
    32 per unit), ranges of 2 - 3 tiles (450 tokens: a unit's 15 tiles over ~5 workgroups; 5 x 300 x 4), two query blocks with one valid
    row in the second (S = 129).  Against fp64 sdpa, bitwise repeatable (the
    counters return to zero after every launch: 4 launches on the same workspace), and equal to fp32 rounding to the plain grids."""
    D = H * 128
    units, nt = -(-S // 128) * H * B, -(-S // 32)
    assert units % 256 != 0 and units <= 1024 and units * nt >= 512, "shape does not take the balanced schedule"
    qkv = rnd(B, S, 3 * D, seed=31)
    qkv[..., :D] *= 2.0
    kb = dev(0.5 * rnd(S, seed=32)) if bias else None
    d_qkv = dev(qkv)
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D + 64, o_bs=S * (D + 64), key_bias=kb)
    outs = []
    for rep in range(4):
        o = torch.full((B, S, D + 64), float("nan"), device="cuda")
        hip.attn_fwd(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], o, **kw)
        outs.append(o)
    assert all(torch.equal(outs[0][:, :, :D], o[:, :, :D]) for o in outs[1:])
    assert torch.isnan(outs[0][:, :, D:]).all() and torch.isfinite(outs[0][:, :, :D]).all()  # pad columns untouched, every row written
    plain = torch.full((B, S, D + 64), float("nan"), device="cuda")
    hip.attn_fwd(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], plain, use_workspace=False, **kw)
    q, k, v = [t.reshape(B, S, H, 128).transpose(1, 2).double() for t in qkv.split(D, dim=-1)]
    mask = None if kb is None else kb.cpu().double().view(1, 1, 1, S)
    want = F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(B, S, D)
    assert rel(outs[0][:, :, :D], want) < 2e-6 and rel(plain[:, :, :D], want) < 2e-6
    assert rel(outs[0][:, :, :D], plain[:, :, :D].cpu()) < 2e-6
    assert not torch.equal(outs[0][:, :, :D], plain[:, :, :D])  # (it IS another schedule: the key tiles are summed in another grouping)
    # the workspace's counter block is all zero again
    ws = hip._attn_f32_workspace(d_qkv.device)
    torch.cuda.synchronize()
    assert int(ws[: 65536].view(torch.int32).abs().sum().item()) == 0


def test_split_activation_producers(hip):
    """LayerNorm and the split attention can write their output in the split activation format: it must be the
    hi / lo split of exactly the fp32 values they write otherwise"""
    B, rows, D = 2, 53, 1536
    x = rnd(B, rows, D, seed=1) * 3 + 0.5
    mod = rnd(B, 3 * D, seed=2) * 0.3
    kw = dict(B=B, rows=rows, D=D, ldx=D, x_bs=rows * D, ldy=D, y_bs=rows * D, scale=dev(mod)[:, D:], shift=dev(mod), mod_bs=3 * D, mode=0, eps=1e-6)
    y0, y1 = torch.empty(B, rows, D, device="cuda"), torch.empty(B, rows, D, device="cuda")
    hip.layernorm_mod(dev(x), y0, **kw)
    hip.layernorm_mod(dev(x), y1, out_split=True, **kw)
    hi, lo = _unsplit(y1, B * rows, D)
    w = y0.cpu().reshape(B * rows, D)
    assert torch.equal(hi, w.bfloat16().float()) and torch.equal(lo, (w - w.bfloat16().float()).bfloat16().float())

    Bq, S, H = 1, 300, 3
    Dh = H * 128
    qkv = dev(rnd(Bq, S, 3 * Dh, seed=11))
    kwa = dict(B=Bq, S=S, H=H, ld_qkv=3 * Dh, qkv_bs=S * 3 * Dh)
    hip.attn_qkv_prepare_split(qkv[:, :, :Dh], qkv[:, :, Dh : 2 * Dh], qkv[:, :, 2 * Dh :], split_row=S, **kwa)
    o0, o1 = torch.empty(Bq, S, Dh + 64, device="cuda"), torch.zeros(Bq, S, Dh + 64, device="cuda")
    hip.attn_fwd_split(qkv[:, :, :Dh], qkv[:, :, Dh : 2 * Dh], qkv[:, :, 2 * Dh :], o0, ldo=Dh + 64, o_bs=S * (Dh + 64), **kwa)
    hip.attn_fwd_split(qkv[:, :, :Dh], qkv[:, :, Dh : 2 * Dh], qkv[:, :, 2 * Dh :], o1, ldo=Dh + 64, o_bs=S * (Dh + 64), out_split=True, **kwa)
    hi, lo = _unsplit(o1, S, Dh + 64)
    w = o0.cpu().reshape(S, Dh + 64)[:, :Dh]
    assert torch.equal(hi[:, :Dh], w.bfloat16().float()) and torch.equal(lo[:, :Dh], (w - w.bfloat16().float()).bfloat16().float())
    assert (o1[:, :, Dh:] == 0).all()  # pad columns untouched


def test_split_format_from_the_transpose_and_the_pooling_pass(hip):
    """the channel->token transpose and the token-mean pass can leave their rows in the split format (patch-embed and
    refiner proj_in inputs): exactly the hi / lo split of the fp32 values; the mean itself is bit-identical"""
    B, C, N, ld = 2, 84, 450, 96
    x = rnd(B, C, N, seed=1) * 2
    t0 = torch.full((B, N, ld), float("nan"), device="cuda")
    t1 = torch.full((B, N, ld), float("nan"), device="cuda")
    hip.chan_to_token(dev(x), t0, B=B, C=C, N=N, ldo=ld, fill_cols=ld)
    hip.chan_to_token(dev(x), t1, B=B, C=C, N=N, ldo=ld, fill_cols=ld, out_split=True)
    hi, lo = _unsplit(t1, B * N, ld)
    w = t0.cpu().reshape(B * N, ld)
    assert torch.equal(hi, w.bfloat16().float()) and torch.equal(lo, (w - w.bfloat16().float()).bfloat16().float())

    rows, D, S = 450, 256, 470  # the split copy lands inside a longer [B, S, D] buffer (ws.nh[:, Nx:])
    h = dev(rnd(B, rows, D, seed=2) * 3 + 0.25)
    m0, m1 = torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda")
    big = torch.zeros(B, S, D, device="cuda")
    hip.mean_rows(h, m0, B=B, rows=rows, D=D, ldx=D, x_bs=rows * D)
    hip.mean_rows(h, m1, B=B, rows=rows, D=D, ldx=D, x_bs=rows * D, x_split=big[:, S - rows :], lds=D, s_bs=S * D)
    assert torch.equal(m0, m1)
    assert (big[:, : S - rows] == 0).all()
    hi, lo = _unsplit(big[:, S - rows :].contiguous(), B * rows, D)
    w = h.cpu().reshape(B * rows, D)
    assert torch.equal(hi, w.bfloat16().float()) and torch.equal(lo, (w - w.bfloat16().float()).bfloat16().float())


def test_attention_online_softmax_rescale_branch(hip):
    """Force the running max to jump at a late key tile (a spiked key row) and force early tiles to be
    negligible: exercises the O/l rescale path and the skip-rescale fast path (guide rule 26)."""
    S, H = 200, 1
    qkv = rnd(1, S, 3 * 128, seed=5) * 0.1
    qkv[0, 150, 128:256] = qkv[0, 7, 0:128] * 400.0  # key 150 aligned with query 7
    d_qkv = dev(qkv)
    out = torch.empty(1, S, 128, device="cuda")
    hip.attn_fwd(d_qkv[:, :, :128], d_qkv[:, :, 128:256], d_qkv[:, :, 256:], out, B=1, S=S, H=H, ld_qkv=384, qkv_bs=S * 384, ldo=128, o_bs=S * 128)
    q, k, v = [t.reshape(1, S, 1, 128).transpose(1, 2).double() for t in qkv.split(128, dim=-1)]
    want = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(1, S, 128)
    assert rel(out, want) < 2e-6
    assert rel(out[0, 7], want[0, 7]) < 2e-6


def test_qk_rmsnorm_rope(hip):
    B, Nx, Nc, H = 2, 37, 11, 3
    D, S = H * 128, 48
    qkv = rnd(B, S, 3 * D, seed=1)
    wq, wk = 1 + 0.1 * rnd(128, seed=2), 1 + 0.1 * rnd(128, seed=3)
    cos, sin = L.get_1d_rotary_pos_embed(128, torch.arange(Nx).float() * 0.37, 256.0)
    d = dev(qkv.clone())
    hip.qk_rmsnorm_rope(d[:, :, :D], d[:, :, D : 2 * D], B=B, row0=0, rows=Nx, H=H, ld=3 * D, bs=S * 3 * D, wq=dev(wq), wk=dev(wk), eps=1e-7,
                        cos=dev(cos), sin=dev(sin))
    hip.qk_rmsnorm_rope(d[:, :, :D], d[:, :, D : 2 * D], B=B, row0=Nx, rows=Nc, H=H, ld=3 * D, bs=S * 3 * D, wq=dev(wk), wk=dev(wq), eps=1e-7)
    nq, nk = L.RMSNorm(128, 1e-7), L.RMSNorm(128, 1e-7)
    nq.weight.data, nk.weight.data = wq, wk
    want = qkv.clone()
    with torch.no_grad():
        for j, (norm_a, norm_b) in enumerate([(nq, nk), (nk, nq)]):  # j=0 -> q columns, j=1 -> k columns
            x = qkv[:, :, j * D : (j + 1) * D].reshape(B, S, H, 128).transpose(1, 2)
            a = L.apply_rotary_emb(norm_a(x[:, :, :Nx]), (cos, sin))
            b = norm_b(x[:, :, Nx:])
            want[:, :, j * D : (j + 1) * D] = torch.cat([a, b], dim=2).transpose(1, 2).reshape(B, S, D)
    assert rel(d, want) < 1e-6
    assert torch.equal(d[:, :, 2 * D :].cpu(), qkv[:, :, 2 * D :])  # v untouched


@pytest.mark.parametrize("D", [1536, 2048, 256])
def test_layernorm_mod(hip, D):
    B, rows = 2, 53
    x = rnd(B, rows + 3, D, seed=1) * 3 + 0.5
    mod = rnd(B, 3 * D, seed=2) * 0.3
    y = torch.empty(B, rows, D, device="cuda")
    hip.layernorm_mod(dev(x), y, B=B, rows=rows, D=D, ldx=D, x_bs=(rows + 3) * D, ldy=D, y_bs=rows * D, scale=dev(mod)[:, D:], shift=dev(mod),
                      mod_bs=3 * D, mode=0, eps=1e-6)
    want = F.layer_norm(x[:, :rows].double(), (D,), eps=1e-6) * (1 + mod[:, None, D : 2 * D].double()) + mod[:, None, :D].double()
    assert rel(y, want) < 1e-6
    w, b = rnd(D, seed=3), rnd(D, seed=4)
    hip.layernorm_mod(dev(x), y, B=B, rows=rows, D=D, ldx=D, x_bs=(rows + 3) * D, ldy=D, y_bs=rows * D, scale=dev(w), shift=dev(b), mode=1, eps=1e-7)
    assert rel(y, F.layer_norm(x[:, :rows].double(), (D,), w.double(), b.double(), eps=1e-7)) < 1e-6


def test_small_row_ops(hip):
    B, rows, D = 3, 450, 1536
    x = rnd(B, rows, D, seed=1)
    y = torch.empty(B, D, device="cuda")
    hip.mean_rows(dev(x), y, B=B, rows=rows, D=D, ldx=D, x_bs=rows * D)
    assert rel(y, x.double().mean(dim=1)) < 1e-6
    g, a = rnd(B, 2 * D, seed=2), rnd(B, rows, D, seed=3)
    out = dev(x.clone())
    hip.gate_residual(out, dev(a), dev(g)[:, D:], out, B=B, rows=rows, D=D, ld_res=D, res_bs=rows * D, ld_y=D, y_bs=rows * D, gate_bs=2 * D)
    assert rel(out, x.double() + a.double() * g[:, None, D:].double()) < 1e-6


@pytest.mark.parametrize("rows,x_rows,add_rows", [(1, 1, 1), (3, 3, 1), (11, 1, 11), (64, 64, 64)])
def test_linear_small(hip, rows, x_rows, add_rows):
    N, K = 300, 256
    x, W, b, add = rnd(x_rows, K, seed=1), rnd(N, K, seed=2) / 16, rnd(N, seed=3), rnd(add_rows, N, seed=4)
    y = torch.empty(rows, N, device="cuda")
    hip.linear_small(dev(x), dev(W), y, rows=rows, N=N, K=K, x_rows=x_rows, bias=dev(b), add=dev(add), add_rows=add_rows, act_in=1, act_out=1)
    xi = F.silu(x.double())[torch.arange(rows) % x_rows]
    want = F.silu(xi @ W.double().T + b.double()) + add.double()[torch.arange(rows) % add_rows]
    assert rel(y, want) < 1e-6


@pytest.mark.parametrize("rows,x_rows,add_rows,N,K", [(20, 20, 20, 58368, 1536), (9, 9, 1, 4100, 256), (16, 1, 16, 4096, 1536), (17, 17, 17, 4111, 64),
                                                    (32, 32, 4, 6000, 576), (40, 40, 40, 5000, 1536), (70, 7, 70, 4097, 512)])
def test_linear_small_many_rows_on_the_matrix_cores(hip, rows, x_rows, add_rows, N, K):
    """more than 8 rows and a wide output (the AdaLN modulation of a sampler chunk's conditioning batch: 20 rows per member x 38 D) go to
    linear_rows_mfma_kernel - exact fp32 products on v_mfma_f32_16x16x4_f32: every epilogue option of the plain path, ragged column tails,
    1 / 2 row tiles per workgroup, several row groups, K chunks (K > 512) and a partial last chunk; each row's result is independent of the
    rows it shares a launch with (bit for bit against a launch of the first 9 rows)"""
    x, W, b, add = rnd(x_rows, K, seed=1), rnd(N, K, seed=2) / K**0.5, rnd(N, seed=3), rnd(add_rows, N, seed=4)
    y = torch.empty(rows, N, device="cuda")
    hip.linear_small(dev(x), dev(W), y, rows=rows, N=N, K=K, x_rows=x_rows, bias=dev(b), add=dev(add), add_rows=add_rows, act_in=1, act_out=1)
    xi = F.silu(x.double())[torch.arange(rows) % x_rows]
    want = F.silu(xi @ W.double().T + b.double()) + add.double()[torch.arange(rows) % add_rows]
    assert rel(y, want) < 1e-6
    y2 = torch.empty(rows, N, device="cuda")
    hip.linear_small(dev(x), dev(W), y2, rows=rows, N=N, K=K, x_rows=x_rows)  # no bias / activation / addend
    assert rel(y2, x.double()[torch.arange(rows) % x_rows] @ W.double().T) < 1e-6
    y9 = torch.empty(9, N, device="cuda")
    hip.linear_small(dev(x), dev(W), y9, rows=9, N=N, K=K, x_rows=x_rows, bias=dev(b), add=dev(add), add_rows=add_rows, act_in=1, act_out=1)
    assert torch.equal(y9, y[:9])
    # the 8-row VALU kernel computes the same rows to fp32 rounding (other summation order)
    y8 = torch.empty(8, N, device="cuda")
    hip.linear_small(dev(x), dev(W), y8, rows=8, N=N, K=K, x_rows=x_rows, bias=dev(b), add=dev(add), add_rows=add_rows, act_in=1, act_out=1)
    assert rel(y8, y[:8]) < 1e-6


def test_linear_small_grouped_is_bitwise_the_single_launches(hip):
    """three independent small linears of different K / rows / activations in one launch == three launches, bit for bit;
    a problem that reads another's output is refused (the problems of one launch run concurrently)"""
    D = 384
    tsin, pooled = dev(rnd(1, 256, seed=1)), dev(rnd(3, D, seed=2))
    W = [dev(rnd(D, 256, seed=3) / 16), dev(rnd(D - 4, 256, seed=4) / 16), dev(rnd(D, D, seed=5) / 16)]
    b = [dev(rnd(D, seed=6)), None, dev(rnd(D, seed=7))]
    add = dev(rnd(1, D, seed=8))
    specs = [dict(x=tsin, W=W[0], rows=1, N=D, K=256, bias=b[0], act_out=1),
             dict(x=tsin, W=W[1], rows=1, N=D - 4, K=256, bias=b[1], act_in=1),
             dict(x=pooled, W=W[2], rows=3, N=D, K=D, bias=b[2], add=add, add_rows=1, act_out=1)]
    single, grouped = [], []
    for sp in specs:
        sp = dict(sp)
        y = torch.full((sp["rows"], sp["N"]), float("nan"), device="cuda")
        hip.linear_small(sp.pop("x"), sp.pop("W"), y, **sp)
        single.append(y)
    probs = []
    for sp in specs:
        sp = dict(sp)
        y = torch.full((sp["rows"], sp["N"]), float("nan"), device="cuda")
        grouped.append(y)
        probs.append(hip.linear_small_problem(sp.pop("x"), sp.pop("W"), y, **sp))
    hip.linear_small_grouped(probs)
    for a, g in zip(single, grouped):
        assert torch.equal(a, g)
    chained = [hip.linear_small_problem(tsin, W[0], grouped[0], rows=1, N=D, K=256), hip.linear_small_problem(grouped[0], W[2], grouped[2], rows=1, N=D, K=D)]
    with pytest.raises(RuntimeError):
        hip.linear_small_grouped(chained)
    with pytest.raises(ValueError):
        hip.linear_small_grouped([])


def test_layernorm_two_row_segments_is_bitwise_two_launches(hip):
    """rows [0, split) with one modulation set and [split, rows) with another in one launch (the dual block's
    norm1 + norm1_context) == two ldc_layernorm_mod launches, bit for bit, in fp32 and in the split format"""
    B, S, Nx, D = 2, 45, 36, 256
    x = dev(rnd(B, S, D, seed=1))
    mods = dev(rnd(B, 4 * D, seed=2))
    sx, hx, sc, hc = mods[:, :D], mods[:, D : 2 * D], mods[:, 2 * D : 3 * D], mods[:, 3 * D :]
    for split in (False, True):
        two = torch.full((B, S, D), float("nan"), device="cuda")
        kw = dict(B=B, D=D, ldx=D, x_bs=S * D, ldy=D, y_bs=S * D, mod_bs=4 * D, mode=0, eps=1e-6, out_split=split)
        hip.layernorm_mod(x[:, :Nx], two[:, :Nx], rows=Nx, scale=sx, shift=hx, **kw)
        hip.layernorm_mod(x[:, Nx:], two[:, Nx:], rows=S - Nx, scale=sc, shift=hc, **kw)
        one = torch.full((B, S, D), float("nan"), device="cuda")
        hip.layernorm_mod(x, one, rows=S, scale=sx, shift=hx, split_row=Nx, scale2=sc, shift2=hc, **kw)
        assert torch.equal(one.view(torch.int32), two.view(torch.int32))


def test_layout_and_embedding_kernels(hip):
    B, C, N = 2, 84, 1800
    x = rnd(B, C, N, seed=1)
    tok = torch.full((B, N, 96), float("nan"), device="cuda")
    hip.chan_to_token(dev(x), tok, B=B, C=C, N=N, ldo=96)
    assert torch.equal(tok[:, :, :C].cpu(), x.transpose(1, 2))
    assert (tok[:, :, C:] == 0).all()
    back = torch.empty(B, C, N, device="cuda")
    hip.token_to_chan(tok, back, B=B, C=C, N=N, ldi=96)
    assert torch.equal(back.cpu(), x)
    t = torch.tensor([1.0955067, -1.553652, 0.0])
    e = torch.empty(3, 256, device="cuda")
    hip.timestep_embedding(dev(t), e, 3)
    assert (e.cpu() - L.get_timestep_embedding(t, 256)).abs().max() < 2e-6
    temb, te = rnd(4, 512, seed=2), rnd(1, 1024, seed=3)
    d = dev(temb.clone())
    hip.temb_modulate(d, dev(te), B=4, D=512, te_rows=1)
    assert rel(d, temb * (1 + te[:, :512]) + te[:, 512:]) < 1e-6
    lat, mu, sd = rnd(3, 84, 4, 15, 30, seed=4), rnd(84, seed=5), rnd(84, seed=6).abs() + 0.3
    y = torch.empty(3, 84, 4, 15, 30, device="cuda")
    hip.chan_affine(dev(lat), y, dev(mu), dev(sd), 0.5, outer=3, C=84, inner=4 * 450, inverse=False)
    want = (lat - mu[None, :, None, None, None]) / sd[None, :, None, None, None] * 0.5
    assert torch.equal(y.cpu(), want)
    z = torch.empty_like(y)
    hip.chan_affine(y, z, dev(mu), dev(sd), 0.5, outer=3, C=84, inner=4 * 450, inverse=True)
    assert torch.equal(z.cpu(), (want / 0.5) * sd[None, :, None, None, None] + mu[None, :, None, None, None])


def test_sampler_state_kernels_are_bit_exact(hip):
    """fp64 Heun state and fp32 DPM++ updates reproduce torch's elementwise rounding exactly."""
    n = 84 * 4 * 450
    noise, Fm = rnd(n, seed=1), rnd(n, seed=2)
    s0, s1 = torch.tensor(79.999985, dtype=torch.float32), torch.tensor(59.657501, dtype=torch.float32)
    sd = 0.5
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    hip.edm_init_state(dev(noise), float(s0), x)
    x_ref = noise.double() * s0
    assert torch.equal(x.cpu(), x_ref)
    c_in = 1 / ((s0**2 + sd**2) ** 0.5)
    xin = torch.empty(n, device="cuda")
    hip.edm_scale_f64_to_f32(x, float(c_in), xin)
    assert torch.equal(xin.cpu(), (x_ref * c_in).float())
    c_skip, c_out = sd**2 / (s0**2 + sd**2), s0 * sd / (s0**2 + sd**2) ** 0.5
    xn, dc = torch.empty_like(x), torch.empty_like(x)
    hip.edm_euler(x, dev(Fm), float(c_skip), float(c_out), float(s0), float(s1 - s0), xn, dc)
    den = c_skip * x_ref + c_out * Fm.double()
    d_ref = (x_ref - den) / s0
    xn_ref = x_ref + (s1 - s0) * d_ref
    assert torch.equal(dc.cpu(), d_ref) and torch.equal(xn.cpu(), xn_ref)
    c_skip1, c_out1 = sd**2 / (s1**2 + sd**2), s1 * sd / (s1**2 + sd**2) ** 0.5
    hip.edm_heun(x, xn, dev(Fm), dc, float(c_skip1), float(c_out1), float(s1), float(s1 - s0))
    den = c_skip1 * xn_ref + c_out1 * Fm.double()
    dp = (xn_ref - den) / s1
    assert torch.equal(xn.cpu(), x_ref + (s1 - s0) * (0.5 * d_ref + 0.5 * dp))
    # DPM-Solver++ (2M) update in fp32
    smp, m1 = rnd(n, seed=3), rnd(n, seed=4)
    x0, prev = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    a, b, inv_r0 = torch.tensor(0.7457), torch.tensor(-0.2543), torch.tensor(1.25)
    hip.dpm_step(dev(smp), dev(Fm), dev(m1), x0, prev, float(c_skip), float(c_out), float(a), float(b), float(inv_r0), 2)
    m0 = c_skip * smp + c_out * Fm
    want = a * smp - b * m0 - (0.5 * b) * (inv_r0 * (m0 - m1))
    assert torch.equal(x0.cpu(), m0) and torch.equal(prev.cpu(), want)


# -- single-term bf16 mode (LDC_GEMM_BF16_1TERM / LDC_ATTN_BF16_1TERM; BASELINE configs[4] "fp16/bf16 mixed") -------------------
def _bf16_rows(x):
    """fp32 [..., K] -> a buffer of the same fp32 shape whose rows hold the K values as plain bf16 in their first 2 K bytes (LDC_FMT_BF16)"""
    buf = torch.zeros_like(x)
    buf.view(torch.bfloat16)[..., : x.shape[-1]] = x.bfloat16()
    return buf


def _from_bf16_rows(buf, K):
    return buf.detach().cpu().view(torch.bfloat16)[..., :K].float()


@pytest.mark.parametrize("M,N,K,batch", [(2250, 1536, 1536, 1), (450, 4608, 1536, 2), (2250, 1536, 7680, 1), (300, 264, 192, 1), (5000, 6144, 1536, 1),
                                         (1, 8, 64, 1), (1800, 84, 1536, 1)])
def test_gemm_bf16_single_term(hip, M, N, K, batch):
    """C = bf16(A) . bf16(W)^T with fp32 accumulation on PLAIN bf16 operand rows (half the bytes of the split format, 64 k per k-step):
    against the fp64 product of the ROUNDED operands only the accumulation order differs (1e-5); against the unrounded product it is
    bf16-accurate (stated tolerance 6e-3).  LDC_GEMM_C_SPLIT writes C as bf16 rows too."""
    A, W, b = rnd(batch, M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    Ap = dev(_bf16_rows(A))
    Wp = hip.pack_weight_bf16(dev(W))
    assert torch.equal(Wp.cpu().view(N, K).float(), W.bfloat16().float())
    C = torch.empty(batch, M, N, device="cuda")
    f1 = hip.GEMM_A_SPLIT | hip.GEMM_BF16_1TERM
    kw = dict(M=M, N=N, K=K, batch=batch, a_bs=M * K, c_bs=M * N, bias=dev(b))
    hip.gemm_grouped([hip.gemm_problem(Ap, Wp, C, flags=f1, **kw)], split_bf16=True)
    want_r = A.bfloat16().double() @ W.bfloat16().double().T + b.double()
    assert rel(C, want_r) < 1e-5
    e = rel(C, A.double() @ W.double().T + b.double())
    assert 1e-4 < e < 6e-3, e  # it really is the single-term product, not the compensated one
    if N % 8 == 0:
        C2 = torch.full((batch, M, N), float("nan"), device="cuda")
        hip.gemm_grouped([hip.gemm_problem(Ap, Wp, C2, flags=f1 | hip.GEMM_C_SPLIT, **kw)], split_bf16=True)
        assert torch.equal(_from_bf16_rows(C2, N), C.cpu().bfloat16().float())  # the bf16 rounding of exactly the fp32 output
    with pytest.raises(RuntimeError):  # one arithmetic per launch
        hip.gemm_grouped([hip.gemm_problem(Ap, Wp, C, flags=f1, **kw),
                          hip.gemm_problem(Ap, hip.pack_weight_bf16x2(dev(W)), C, flags=hip.GEMM_A_SPLIT, **kw)], split_bf16=True)


def test_bf16_row_producers(hip):
    """LayerNorm, the transpose, the pooling pass and the attention write their output as plain bf16 rows (LDC_FMT_BF16) = the bf16
    rounding of exactly the fp32 values they write otherwise"""
    B, rows, D = 2, 450, 1536
    x, sc, sh = dev(rnd(B, rows, D, seed=1)), dev(rnd(B, D, seed=2) * 0.1), dev(rnd(B, D, seed=3) * 0.1)
    y0, y2 = torch.empty(B, rows, D, device="cuda"), torch.zeros(B, rows, D, device="cuda")
    kw = dict(B=B, rows=rows, D=D, ldx=D, x_bs=rows * D, ldy=D, y_bs=rows * D, scale=sc, shift=sh, mod_bs=D, mode=0, eps=1e-6)
    hip.layernorm_mod(x, y0, **kw)
    hip.layernorm_mod(x, y2, out_split=hip.FMT_BF16, **kw)
    assert torch.equal(_from_bf16_rows(y2, D), y0.cpu().bfloat16().float())
    img = dev(rnd(B, 84, 900, seed=4))
    t0, t2 = torch.empty(B, 900, 128, device="cuda"), torch.zeros(B, 900, 128, device="cuda")
    hip.chan_to_token(img, t0, B=B, C=84, N=900, ldo=128, fill_cols=128)
    hip.chan_to_token(img, t2, B=B, C=84, N=900, ldo=128, fill_cols=128, out_split=hip.FMT_BF16)
    assert torch.equal(_from_bf16_rows(t2, 128), t0.cpu().bfloat16().float())
    m0, m2, xs = torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda"), torch.zeros(B, rows, D, device="cuda")
    hip.mean_rows(x, m0, B=B, rows=rows, D=D, ldx=D, x_bs=rows * D)
    hip.mean_rows(x, m2, B=B, rows=rows, D=D, ldx=D, x_bs=rows * D, x_split=xs, lds=D, s_bs=rows * D, fmt=hip.FMT_BF16)
    assert torch.equal(m2, m0) and torch.equal(_from_bf16_rows(xs, D), x.cpu().bfloat16().float())
    S, H = 200, 3
    Dh = H * 128
    qkv = dev(rnd(1, S, 3 * Dh, seed=5))
    _prep(hip, qkv, 1, S, H, Dh, split_row=S)
    o0, o2 = torch.empty(1, S, Dh + 128, device="cuda"), torch.zeros(1, S, Dh + 128, device="cuda")
    akw = dict(B=1, S=S, H=H, ld_qkv=3 * Dh, qkv_bs=S * 3 * Dh, ldo=Dh + 128, o_bs=S * (Dh + 128))
    hip.attn_fwd_split(qkv[:, :, :Dh], qkv[:, :, Dh : 2 * Dh], qkv[:, :, 2 * Dh :], o0, **akw)
    hip.attn_fwd_split(qkv[:, :, :Dh], qkv[:, :, Dh : 2 * Dh], qkv[:, :, 2 * Dh :], o2, out_split=hip.FMT_BF16, **akw)
    assert torch.equal(_from_bf16_rows(o2, Dh + 128)[..., :Dh], o0.cpu()[..., :Dh].bfloat16().float())


# -- third-generation attention: row-major split-bf16 operand rows (attn_split.hip) + the fused QKV-projection epilogue ----------------
def _prep(hip, d_qkv, B, S, H, D, **kw):
    hip.attn_qkv_prepare_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=d_qkv.shape[1] * 3 * D, **kw)


@pytest.mark.parametrize("B,S,H", [(1, 2250, 12), (2, 450, 2), (1, 33, 1), (1, 128, 3), (3, 70, 2), (1, 1, 1), (2, 2250, 12), (1, 2250, 16), (1, 97, 2),
                                   (1, 32, 1), (1, 64, 2), (1, 4500, 12)])
@pytest.mark.parametrize("one_term", [False, True])
def test_attention_split(hip, B, S, H, one_term):
    """prepare (no norm / RoPE: scale + split, in place) + attention on the row-major split rows == sdpa on the raw operands; both key-range
    layouts (one / two wave groups), ragged last tile, single-term mode at its stated tolerance"""
    D = H * 128
    qkv = rnd(B, S, 3 * D, seed=11)
    qkv[..., :D] *= 4.0
    d_qkv = dev(qkv)
    _prep(hip, d_qkv, B, S, H, D, split_row=S)
    out = torch.full((B, S, D + 64), float("nan"), device="cuda")
    hip.attn_fwd_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D + 64,
                       o_bs=S * (D + 64), one_term=one_term)
    q, k, v = [t.reshape(B, S, H, 128).transpose(1, 2).double() for t in qkv.split(D, dim=-1)]
    want = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, D)
    assert torch.isfinite(out[:, :, :D]).all()
    assert rel(out[:, :, :D], want) < (1e-2 if one_term else 2e-5)
    assert torch.isnan(out[:, :, D:]).all()
    # the prepared rows: q / k groups of [hi x8 | lo x8], v as [hi x128 | lo x128]; hi + lo reproduces the operand to 2^-17
    got = d_qkv.cpu()
    scale = 0.08838834764831845 * 1.4426950408889634
    hi, lo = _unsplit(got[..., : 2 * D].reshape(B * S, 2 * D), B * S, 2 * D)
    assert rel((hi + lo)[:, :D] / scale, qkv[..., :D].reshape(B * S, D)) < 1e-5 and rel((hi + lo)[:, D:], qkv[..., D : 2 * D].reshape(B * S, D)) < 1e-5
    vw = got[..., 2 * D :].contiguous().view(torch.int16).reshape(B, S, H, 2, 128)
    vf = (vw.to(torch.int32) << 16).view(torch.float32)
    assert rel((vf[:, :, :, 0] + vf[:, :, :, 1]).reshape(B, S, D), qkv[..., 2 * D :]) < 1e-5


@pytest.mark.parametrize("B,S,H,one_term,bias", [(1, 2250, 16, False, False), (1, 2250, 16, True, True), (1, 3300, 10, False, True), (2, 2250, 8, False, False),
                                                  (1, 1200, 29, False, False), (3, 1100, 11, True, False)])
def test_attention_split_tail_schedule(hip, B, S, H, one_term, bias):
    """more (query block, head, batch) units than CUs by at most half a round - the 1.6B model's 16 heads x 18 query blocks = 288: a
    persistent workgroup per CU runs whole units, then one key slice of a left-over unit; a second launch merges the slices.  Against
    fp64 sdpa, bitwise repeatable, and equal to fp32 rounding to the plain grids (`use_workspace=False`: the same call without the
    workspace).  Shapes: 288 units (32 left over, 8 slices), 260 (4 left over: slice count capped at 16), 288 as 2 x 144, 290 (34 -> 7
    slices of 38 key tiles), 297 with 35-tile key ranges."""
    D = H * 128
    assert hip.lib.ldc_attn_fwd_split_workspace_bytes(B, S, H) > 0, "shape does not take the tail schedule"
    qkv = rnd(B, S, 3 * D, seed=21)
    qkv[..., :D] *= 4.0
    kb = None
    if bias:
        kb = hip.pad_key_bias(dev(0.5 * rnd(S, seed=22)))
    d_qkv = dev(qkv)
    _prep(hip, d_qkv, B, S, H, D, split_row=S)
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D, one_term=one_term, key_bias=kb)
    outs = []
    for rep in range(3):
        o = torch.full((B, S, D), float("nan"), device="cuda")
        hip.attn_fwd_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], o, **kw)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    plain = torch.full((B, S, D), float("nan"), device="cuda")
    hip.attn_fwd_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], plain, use_workspace=False, **kw)
    q, k, v = [t.reshape(B, S, H, 128).transpose(1, 2).double() for t in qkv.split(D, dim=-1)]
    mask = None if kb is None else kb[:S].cpu().double().view(1, 1, 1, S)
    want = F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(B, S, D)
    assert torch.isfinite(outs[0]).all()
    tol = 1e-2 if one_term else 2e-5
    assert rel(outs[0], want) < tol and rel(plain, want) < tol
    assert rel(outs[0], plain) < (1e-5 if not one_term else 5e-3)  # same products, another order of the key tiles (bf16 P: another rounding point)
    o2 = torch.empty(B, S, D, device="cuda")  # split-format output through the merge kernel
    hip.attn_fwd_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], o2, **dict(kw, out_split=hip.FMT_BF16 if one_term else True))
    if one_term:
        got = (o2.view(torch.int16)[..., :D].to(torch.int32) << 16).view(torch.float32)
        assert torch.equal(got.cpu(), outs[0].cpu().bfloat16().float())
    else:
        hi, lo = _unsplit(o2.reshape(B * S, D), B * S, D)
        wh = outs[0].cpu().reshape(B * S, D).bfloat16().float()
        assert torch.equal(hi, wh) and torch.equal(lo, (outs[0].cpu().reshape(B * S, D) - wh).bfloat16().float())


@pytest.mark.parametrize("Nx,Nc,rope1", [(37, 11, False), (1800, 450, True), (64, 0, False), (40, 57, True)])
def test_attention_split_norm_rope_two_segments(hip, Nx, Nc, rope1):
    """the stand-alone producer's q/k RMSNorm + RoPE per row segment, against the oracle layers + sdpa (reference
    models/LaDCast_3D_model.py:103-203); 5 extra token rows in the buffer: batch stride != S * ld"""
    B, H = 2 if Nx < 100 else 1, 3
    D, S = H * 128, Nx + Nc
    qkv = rnd(B, S + 5, 3 * D, seed=1)
    wq0, wk0, wq1, wk1 = [1 + 0.1 * rnd(128, seed=s_) for s_ in (2, 3, 4, 5)]
    cos0, sin0 = L.get_1d_rotary_pos_embed(128, torch.arange(Nx).float() * 0.37, 256.0)
    cos1, sin1 = L.get_1d_rotary_pos_embed(128, torch.arange(max(Nc, 1)).float() * 0.11 - 3.0, 256.0)
    d = dev(qkv)
    out = torch.empty(B, S, D, device="cuda")
    seg1 = (dev(wq1), dev(wk1), dev(cos1) if rope1 else None, dev(sin1) if rope1 else None)
    hip.attn_qkv_prepare_split(d[:, :, :D], d[:, :, D : 2 * D], d[:, :, 2 * D :], B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=(S + 5) * 3 * D, split_row=Nx,
                               seg0=(dev(wq0), dev(wk0), dev(cos0), dev(sin0)), seg1=seg1, eps=1e-7)
    hip.attn_fwd_split(d[:, :, :D], d[:, :, D : 2 * D], d[:, :, 2 * D :], out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=(S + 5) * 3 * D, ldo=D, o_bs=S * D)
    assert torch.equal(d[:, S:].cpu(), qkv[:, S:])  # rows past S untouched
    want = _norm_rope_sdpa(qkv[:, :S], B, S, H, Nx, Nc, (wq0, wk0, wq1, wk1), (cos0, sin0), (cos1, sin1) if rope1 else None)
    assert rel(out, want) < 2e-5


def _norm_rope_sdpa(x, B, S, H, Nx, Nc, ws_, rope0, rope1):
    D = H * 128
    norms = []
    for w in ws_:
        n = L.RMSNorm(128, 1e-7)
        n.weight.data = w
        norms.append(n)
    with torch.no_grad():
        x = x.double()
        qk = []
        for j in range(2):
            t = x[:, :, j * D : (j + 1) * D].reshape(B, S, H, 128).transpose(1, 2).float()
            a = L.apply_rotary_emb(norms[j](t[:, :, :Nx]), rope0)
            b = norms[2 + j](t[:, :, Nx:])
            if rope1 is not None and Nc:
                b = L.apply_rotary_emb(b, rope1)
            qk.append(torch.cat([a, b], dim=2).double())
        v = x[:, :, 2 * D :].reshape(B, S, H, 128).transpose(1, 2)
        return F.scaled_dot_product_attention(qk[0], qk[1], v).transpose(1, 2).reshape(B, S, D)


def test_attention_split_rescale_branch_and_split_output(hip):
    """a key that makes one query's running max jump far past the lazy-max threshold in a late tile (rule 26: force the rare branch), and
    the LDC_ATTN_OUT_SPLIT output = the hi / lo split of exactly the fp32 output"""
    S = 200
    qkv = rnd(1, S, 3 * 128, seed=5) * 0.1
    qkv[0, 150, 128:256] = qkv[0, 7, 0:128] * 400.0
    d_qkv = dev(qkv)
    _prep(hip, d_qkv, 1, S, 1, 128, split_row=S)
    out = torch.empty(1, S, 128, device="cuda")
    kw = dict(B=1, S=S, H=1, ld_qkv=384, qkv_bs=S * 384, ldo=128, o_bs=S * 128)
    hip.attn_fwd_split(d_qkv[:, :, :128], d_qkv[:, :, 128:256], d_qkv[:, :, 256:], out, **kw)
    q, k, v = [t.reshape(1, S, 1, 128).transpose(1, 2).double() for t in qkv.split(128, dim=-1)]
    want = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(1, S, 128)
    assert rel(out, want) < 2e-5 and rel(out[0, 7], want[0, 7]) < 2e-5
    out2 = torch.empty(1, S, 128, device="cuda")
    hip.attn_fwd_split(d_qkv[:, :, :128], d_qkv[:, :, 128:256], d_qkv[:, :, 256:], out2, out_split=True, **kw)
    hi, lo = _unsplit(out2.reshape(S, 128), S, 128)
    want_hi = out.cpu().reshape(S, 128).bfloat16().float()
    assert torch.equal(hi, want_hi) and torch.equal(lo, (out.cpu().reshape(S, 128) - want_hi).bfloat16().float())


@pytest.mark.parametrize("Nx,Nc,H,batch", [(1800, 450, 12, 1), (37, 11, 2, 2), (450, 0, 12, 1), (4500, 0, 3, 1)])
def test_gemm_qkv_epilogue_feeds_the_attention(hip, Nx, Nc, H, batch):
    """ldc_gemm_grouped_bf16x3_qkv: the QKV projection (two grouped problems: a RoPE'd stream and a plain one, plus - as in the single
    blocks - an ordinary GELU problem in the same launch) writes the attention operand rows itself; attention on them == oracle layers
    (RMSNorm, RoPE) + sdpa on the fp64 projection.  Both tile heights, split tiles (stream-K reduction) included."""
    D, K = H * 128, 256
    S = Nx + Nc
    A = rnd(batch, S, K, seed=1)
    Wx, Wc = rnd(3 * D, K, seed=2) / math.sqrt(K), rnd(3 * D, K, seed=3) / math.sqrt(K)
    bx, bc = rnd(3 * D, seed=4) * 0.1, rnd(3 * D, seed=5) * 0.1
    Wm, bm = rnd(512, K, seed=6) / math.sqrt(K), rnd(512, seed=7)
    wq0, wk0, wq1, wk1 = [1 + 0.1 * rnd(128, seed=s_) for s_ in (12, 13, 14, 15)]
    cos0, sin0 = L.get_1d_rotary_pos_embed(128, torch.arange(Nx).float() * 0.37, 256.0)
    Ap = hip.pack_weight_bf16x2(dev(A).reshape(batch * S, K)).view(batch, S, K)
    qkv = torch.full((batch, S, 3 * D), float("nan"), device="cuda")
    mlp = torch.empty(batch, S, 512, device="cuda")
    f = hip.GEMM_A_SPLIT
    probs = [hip.gemm_problem(Ap, hip.pack_weight_bf16x2(dev(Wx)), qkv, M=Nx, N=3 * D, K=K, batch=batch, a_bs=S * K, c_bs=S * 3 * D, bias=dev(bx), flags=f)]
    epis = [hip.qkv_epilogue(dev(wq0), dev(wk0), hip.compact_rope_table(dev(cos0), dev(sin0)), eps=1e-7, heads=H)]
    if Nc:
        probs.append(hip.gemm_problem(Ap[:, Nx:], hip.pack_weight_bf16x2(dev(Wc)), qkv[:, Nx:], M=Nc, N=3 * D, K=K, batch=batch, a_bs=S * K, c_bs=S * 3 * D,
                                      bias=dev(bc), flags=f))
        epis.append(hip.qkv_epilogue(dev(wq1), dev(wk1), None, eps=1e-7, heads=H))
    probs.append(hip.gemm_problem(Ap, hip.pack_weight_bf16x2(dev(Wm)), mlp, M=S, N=512, K=K, batch=batch, a_bs=S * K, c_bs=S * 512, bias=dev(bm), act=2, flags=f))
    epis.append(None)
    hip.gemm_grouped_qkv(probs, epis)
    out = torch.empty(batch, S, D, device="cuda")
    hip.attn_fwd_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, B=batch, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    proj = torch.cat([A[:, :Nx].double() @ Wx.double().T + bx.double(), A[:, Nx:].double() @ Wc.double().T + bc.double()], dim=1)
    want = _norm_rope_sdpa(proj, batch, S, H, Nx, Nc, (wq0, wk0, wq1, wk1), (cos0, sin0), None)
    assert rel(out, want) < 2e-5
    assert rel(mlp, F.gelu(A.double() @ Wm.double().T + bm.double(), approximate="tanh")) < 1e-5  # the ordinary problem of the same launch
    # against the stand-alone producer on the fp32 projection: the same rows up to the projection's rounding
    qkv2 = proj.float().cuda().contiguous()
    seg1 = (dev(wq1), dev(wk1), None, None)
    hip.attn_qkv_prepare_split(qkv2[:, :, :D], qkv2[:, :, D : 2 * D], qkv2[:, :, 2 * D :], B=batch, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, split_row=Nx,
                               seg0=(dev(wq0), dev(wk0), dev(cos0), dev(sin0)), seg1=seg1, eps=1e-7)
    h1, l1 = _unsplit(qkv[..., : 2 * D].reshape(batch * S, 2 * D), batch * S, 2 * D)
    h2, l2 = _unsplit(qkv2[..., : 2 * D].reshape(batch * S, 2 * D), batch * S, 2 * D)
    assert rel(h1 + l1, h2 + l2) < 1e-5
    with pytest.raises(RuntimeError):  # N must be 3 * heads * 128
        hip.gemm_grouped_qkv(probs[:1], [hip.qkv_epilogue(dev(wq0), dev(wk0), None, heads=H + 1)])
    # the route ladcast_hip.h documents for LDC_ERR_UNSUPPORTED (plain GEMM, then ldc_attn_qkv_prepare_split in place): same attention
    qkv.fill_(float("nan"))
    mlp.zero_()
    hip.gemm_grouped_qkv(probs, epis, force_fallback=True)
    out_fb = torch.empty(batch, S, D, device="cuda")
    hip.attn_fwd_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out_fb, B=batch, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    assert rel(out_fb, want) < 2e-5 and rel(out_fb, out.double().cpu()) < 1e-5
    assert rel(mlp, F.gelu(A.double() @ Wm.double().T + bm.double(), approximate="tanh")) < 1e-5


@pytest.mark.parametrize("Nx,Nc,H,batch", [(1800, 450, 12, 1), (37, 11, 2, 2), (450, 0, 12, 1), (300, 100, 3, 3)])
def test_gemm_qkv_epilogue_exact_fp32(hip, Nx, Nc, H, batch):
    """ldc_gemm_grouped_qkv_f32 (round 5): the exact-fp32 QKV projection applies per-head RMSNorm + rotary embedding in its epilogue and
    writes plain fp32 rows - against the two-step route it replaces in the model (ldc_gemm_grouped, then ldc_qk_rmsnorm_rope in place: same
    arithmetic in the same order), and the attention on those rows against oracle layers + sdpa on the fp64 projection.  A RoPE'd stream,
    a plain one and an ordinary GELU problem in one launch, as in the model's blocks."""
    D, K = H * 128, 256
    S = Nx + Nc
    A = rnd(batch, S, K, seed=1)
    Wx, Wc = rnd(3 * D, K, seed=2) / math.sqrt(K), rnd(3 * D, K, seed=3) / math.sqrt(K)
    bx, bc = rnd(3 * D, seed=4) * 0.1, rnd(3 * D, seed=5) * 0.1
    Wm, bm = rnd(512, K, seed=6) / math.sqrt(K), rnd(512, seed=7)
    wq0, wk0, wq1, wk1 = [1 + 0.1 * rnd(128, seed=s_) for s_ in (12, 13, 14, 15)]
    cos0, sin0 = L.get_1d_rotary_pos_embed(128, torch.arange(Nx).float() * 0.37, 256.0)
    dA = dev(A)
    qkv = torch.full((batch, S, 3 * D), float("nan"), device="cuda")
    mlp = torch.empty(batch, S, 512, device="cuda")

    def problems(qkv_, mlp_):
        pr = [hip.gemm_problem(dA, dev(Wx), qkv_, M=Nx, N=3 * D, K=K, batch=batch, a_bs=S * K, c_bs=S * 3 * D, bias=dev(bx))]
        if Nc:
            pr.append(hip.gemm_problem(dA[:, Nx:], dev(Wc), qkv_[:, Nx:], M=Nc, N=3 * D, K=K, batch=batch, a_bs=S * K, c_bs=S * 3 * D, bias=dev(bc)))
        pr.append(hip.gemm_problem(dA, dev(Wm), mlp_, M=S, N=512, K=K, batch=batch, a_bs=S * K, c_bs=S * 512, bias=dev(bm), act=2))
        return pr

    epis = [hip.qkv_epilogue(dev(wq0), dev(wk0), hip.compact_rope_table(dev(cos0), dev(sin0)), eps=1e-7, heads=H, qscale=1.0)]
    if Nc:
        epis.append(hip.qkv_epilogue(dev(wq1), dev(wk1), None, eps=1e-7, heads=H, qscale=1.0))
    epis.append(None)
    assert hip.gemm_grouped_qkv_f32(problems(qkv, mlp), epis)
    # the route it replaces: plain projection, then the q / k norm + rotary kernel in place, segment by segment
    qkv2 = torch.full((batch, S, 3 * D), float("nan"), device="cuda")
    mlp2 = torch.empty(batch, S, 512, device="cuda")
    hip.gemm_grouped(problems(qkv2, mlp2))
    hip.qk_rmsnorm_rope(qkv2[:, :, :D], qkv2[:, :, D : 2 * D], B=batch, row0=0, rows=Nx, H=H, ld=3 * D, bs=S * 3 * D, wq=dev(wq0), wk=dev(wk0), eps=1e-7,
                        cos=dev(cos0), sin=dev(sin0))
    if Nc:
        hip.qk_rmsnorm_rope(qkv2[:, :, :D], qkv2[:, :, D : 2 * D], B=batch, row0=Nx, rows=Nc, H=H, ld=3 * D, bs=S * 3 * D, wq=dev(wq1), wk=dev(wk1), eps=1e-7)
    assert torch.isfinite(qkv).all() and torch.equal(mlp, mlp2)
    assert rel(qkv, qkv2) < 3e-7 and torch.equal(qkv[..., 2 * D :], qkv2[..., 2 * D :])  # v: bias only - the same bits
    out = torch.empty(batch, S, D, device="cuda")
    hip.attn_fwd(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, B=batch, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    proj = torch.cat([A[:, :Nx].double() @ Wx.double().T + bx.double(), A[:, Nx:].double() @ Wc.double().T + bc.double()], dim=1)
    want = _norm_rope_sdpa(proj, batch, S, H, Nx, Nc, (wq0, wk0, wq1, wk1), (cos0, sin0), None)
    assert rel(out, want) < 3e-6
    with pytest.raises(RuntimeError):  # N must be 3 * heads * 128
        hip.gemm_grouped_qkv_f32(problems(qkv, mlp)[:1], [hip.qkv_epilogue(dev(wq0), dev(wk0), None, heads=H + 1, qscale=1.0)])
    # a shape the ring kernel does not serve (K % 32 != 0): nothing is launched, the caller takes the two-step route
    A2, W2 = dev(rnd(1, 40, 24, seed=8)), dev(rnd(3 * 128, 24, seed=9))
    c2 = torch.full((1, 40, 384), float("nan"), device="cuda")
    assert not hip.gemm_grouped_qkv_f32([hip.gemm_problem(A2, W2, c2, M=40, N=384, K=24)], [hip.qkv_epilogue(dev(wq0), dev(wk0), None, heads=1, qscale=1.0)])
    torch.cuda.synchronize()
    assert torch.isnan(c2).all()


# -- launch merges: each fused launch is bit-identical to the two launches it replaces ---------------------------------------------------
def test_timestep_sinusoid_inside_the_first_linear_is_bitwise(hip):
    D = 1536
    t = dev(torch.tensor([0.3, -1.553652, 1.0955067, 0.0, 0.77]))[1:4]  # 4-byte-aligned view, as the sampler's c_noise[i : i + 1]
    W, b = dev(rnd(D, 256, seed=1) / 16), dev(rnd(D, seed=2))
    tsin = torch.empty(3, 256, device="cuda")
    hip.timestep_embedding(t.contiguous(), tsin, 3)
    y0, y1, y2 = [torch.empty(3, D, device="cuda") for _ in range(3)]
    hip.linear_small(tsin, W, y0, rows=3, N=D, K=256, bias=b, act_out=hip.ACT_SILU)
    hip.linear_small(t, W, y1, rows=3, N=D, K=256, bias=b, act_in=hip.ACT_IN_TIMESTEP_SINCOS, act_out=hip.ACT_SILU)
    hip.linear_small_grouped([hip.linear_small_problem(t, W, y2, rows=3, N=D, K=256, bias=b, act_in=hip.ACT_IN_TIMESTEP_SINCOS, act_out=hip.ACT_SILU)])
    assert torch.equal(y1, y0) and torch.equal(y2, y0)
    want = F.silu(L.get_timestep_embedding(t.cpu(), 256).double() @ W.cpu().double().T + b.cpu().double())
    assert rel(y1, want) < 1e-5
    with pytest.raises(RuntimeError):
        hip.linear_small(t, W[:, :128].contiguous(), y1, rows=3, N=D, K=128, act_in=hip.ACT_IN_TIMESTEP_SINCOS)  # K must be 256


@pytest.mark.parametrize("B,te_rows", [(1, 1), (3, 1), (3, 3)])
def test_linear_small_modulation_epilogue_is_bitwise(hip, B, te_rows):
    D = 1536
    x, W, b, add = dev(rnd(B, D, seed=1)), dev(rnd(D, D, seed=2) / 40), dev(rnd(D, seed=3)), dev(rnd(1, D, seed=4))
    te = dev(rnd(te_rows, 2 * D, seed=5) * 0.3)
    y0, y1 = torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda")
    hip.linear_small(x, W, y0, rows=B, N=D, K=D, bias=b, add=add, add_rows=1)
    hip.temb_modulate(y0, te, B=B, D=D, te_rows=te_rows)
    hip.linear_small(x, W, y1, rows=B, N=D, K=D, bias=b, add=add, add_rows=1, mod=te, mod_rows=te_rows)
    assert torch.equal(y1, y0)
    v = x.cpu().double() @ W.cpu().double().T + b.cpu().double() + add.cpu().double()
    t = te.cpu().double().expand(B, -1) if te_rows == 1 else te.cpu().double()
    assert rel(y1, v * (1 + t[:, :D]) + t[:, D:]) < 1e-5


@pytest.mark.parametrize("split", [False, True])
def test_gate_residual_layernorm_is_bitwise(hip, split):
    B, rows, D = 2, 450, 1536
    h, y, gate = rnd(B, rows + 3, D, seed=1), dev(rnd(B, rows, D, seed=2)), dev(rnd(B, 2 * D, seed=3))
    w, b = dev(1 + 0.1 * rnd(D, seed=4)), dev(0.1 * rnd(D, seed=5))
    h0, h1 = dev(h), dev(h)
    o0, o1 = torch.full((B, rows, D), float("nan"), device="cuda"), torch.full((B, rows, D), float("nan"), device="cuda")
    kw = dict(B=B, rows=rows, D=D, ld_res=D, res_bs=(rows + 3) * D, ld_y=D, y_bs=rows * D, gate_bs=2 * D)
    hip.gate_residual(h0, y, gate, h0, **kw)
    hip.layernorm_mod(h0, o0, B=B, rows=rows, D=D, ldx=D, x_bs=(rows + 3) * D, ldy=D, y_bs=rows * D, scale=w, shift=b, mode=1, eps=1e-7, out_split=split)
    hip.gate_residual_layernorm(h1, y, gate, o1, ld_out=D, out_bs=rows * D, weight=w, bias=b, eps=1e-7, out_split=split, **kw)
    assert torch.equal(h1, h0) and torch.equal(o1.view(torch.int32), o0.view(torch.int32))
    assert torch.equal(h1[:, rows:].cpu(), h[:, rows:])  # rows past `rows` untouched
    want_h = h[:, :rows].double() + gate.cpu()[:, None, :D].double() * y.cpu().double()
    assert rel(h1[:, :rows], want_h) < 1e-6
    if not split:
        assert rel(o1, F.layer_norm(want_h, (D,), w.cpu().double(), b.cpu().double(), 1e-7)) < 1e-5


@pytest.mark.parametrize("B,S,H", [(1, 2250, 12), (2, 450, 2), (1, 97, 2), (1, 4500, 12), (1, 1, 1)])
def test_attention_key_bias(hip, B, S, H):
    """per-key additive score bias (`scale_attn_by_lat`: a (1, 1, 1, keys) float attention mask) in the fp32 kernel and in the split
    kernel (both wave-group layouts, both arithmetic modes), against sdpa(attn_mask=bias)"""
    D = H * 128
    qkv = rnd(B, S, 3 * D, seed=11)
    qkv[..., :D] *= 2.0
    bias = rnd(S, seed=12) * 3.0  # large enough to reorder the softmax
    q, k, v = [t.reshape(B, S, H, 128).transpose(1, 2).double() for t in qkv.split(D, dim=-1)]
    want = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.double().view(1, 1, 1, S)).transpose(1, 2).reshape(B, S, D)
    plain = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, D)
    assert S == 1 or rel(plain, want) > 1e-2  # the bias matters
    d_qkv = dev(qkv)
    out = torch.empty(B, S, D, device="cuda")
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    hip.attn_fwd(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], out, key_bias=dev(bias), **kw)
    assert rel(out, want) < 2e-6
    _prep(hip, d_qkv, B, S, H, D, split_row=S)
    kb = hip.pad_key_bias(dev(bias))
    assert kb.numel() % 32 == 0 and torch.equal(kb[:S].cpu(), bias)
    for one_term, tol in ((False, 2e-5), (True, 1e-2)):
        out.fill_(float("nan"))
        hip.attn_fwd_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], out, key_bias=kb, one_term=one_term, **kw)
        assert rel(out, want) < tol, one_term
    with pytest.raises(ValueError):
        hip.attn_fwd_split(d_qkv[:, :, :D], d_qkv[:, :, D : 2 * D], d_qkv[:, :, 2 * D :], out, key_bias=dev(bias)[: max(S - 1, 0)], **kw)


def test_attention_split_ignores_stale_lds(hip):
    """The last iteration of a key-range group computes S_next from a ring stage nothing was loaded into: whatever an earlier kernel left in
    LDS - NaN and inf bit patterns included - must not reach the running max.  (Found in round 2: an inf there passed the lazy-max
    threshold; random rows came out NaN whenever a group's first tile was also its last.)  Poison LDS with a kernel fed NaN / inf, then run
    the shapes whose groups have a single tile."""
    for poison in (float("nan"), float("inf"), -float("inf")):
        x = torch.full((1, 512, 384), poison, device="cuda")
        o2 = torch.empty(1, 512, 128, device="cuda")
        hip.attn_fwd(x[:, :, :128], x[:, :, 128:256], x[:, :, 256:], o2, B=1, S=512, H=1, ld_qkv=384, qkv_bs=512 * 384, ldo=128, o_bs=512 * 128)
        for (B, S, H) in ((1, 33, 1), (1, 1, 1), (3, 70, 2), (1, 40, 3)):
            D = H * 128
            qkv = rnd(B, S, 3 * D, seed=11)
            d = dev(qkv)
            _prep(hip, d, B, S, H, D, split_row=S)
            out = torch.full((B, S, D), float("nan"), device="cuda")
            for one_term in (False, True):
                hip.attn_fwd_split(d[:, :, :D], d[:, :, D : 2 * D], d[:, :, 2 * D :], out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D,
                                   one_term=one_term)
                q, k, v = [t.reshape(B, S, H, 128).transpose(1, 2).double() for t in qkv.split(D, dim=-1)]
                want = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, D)
                assert torch.isfinite(out).all(), (poison, B, S, H, one_term)
                assert rel(out, want) < (1e-2 if one_term else 2e-5)
