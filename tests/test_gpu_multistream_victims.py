"""Guard rail for the "first-read effect" (DESIGN.md, section on GPU sharing): a VALU read whose
upper dwords came back as 0 in lanes 48..63 when an MFMA-streaming wave shared the SIMD.  The one victim found in round 4 (the AdaLN GEMV)
carries a guard (csrc/rowops.hip: ls_first_read); nothing proved the other kernels immune, and the contract topology (one process per GPU,
one stream) can never show it.  Here EVERY row / VALU kernel of the library runs as the victim: >= 500 launches on a second stream while an
MFMA-streaming aggressor (csrc/test_aggressor.hip: 224 live VGPRs, 64 KiB LDS, one workgroup per CU - half of every SIMD's registers stay
free for the victim's waves) runs on the first, and every result must equal the solo run bit for bit.  The list in `VICTIMS` is the list
INTEGRATION.md section 3d cites: multi-stream use of the library is supported for these kernels on the strength of this test."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import ladcast_amd.hip as hip  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHES = 512


def _aggressor():
    lib = ctypes.CDLL(os.path.join(ROOT, "ladcast_amd", "libladcast_test_aggressor.so"))
    lib.ldc_test_aggressor_launch.restype = ctypes.c_int
    lib.ldc_test_aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).cuda()


def _victims():
    """name -> (make_output, run(out)): every call writes a fresh output tensor, inputs are shared"""
    v = {}
    D = 1536
    x3, mod = rnd(2, 450, D, seed=1), 0.1 * rnd(2, 3 * D, seed=2)
    v["layernorm_mod"] = (lambda: torch.empty(2, 450, D, device="cuda"),
                          lambda y: hip.layernorm_mod(x3, y, B=2, rows=450, D=D, ldx=D, x_bs=450 * D, ldy=D, y_bs=450 * D, scale=mod[:, D:], shift=mod, mod_bs=3 * D, mode=0, eps=1e-6))
    v["layernorm_mod_split_out"] = (lambda: torch.empty(2, 450, D, device="cuda"),
                                    lambda y: hip.layernorm_mod(x3, y, B=2, rows=450, D=D, ldx=D, x_bs=450 * D, ldy=D, y_bs=450 * D, scale=mod[:, D:], shift=mod, mod_bs=3 * D, mode=0,
                                                                eps=1e-6, out_split=True))
    v["mean_rows"] = (lambda: torch.empty(2, D, device="cuda"), lambda y: hip.mean_rows(x3, y, B=2, rows=450, D=D, ldx=D, x_bs=450 * D))
    a3, g3 = rnd(2, 450, D, seed=3), rnd(2, 2 * D, seed=4)
    v["gate_residual"] = (lambda: torch.empty(2, 450, D, device="cuda"),
                          lambda y: hip.gate_residual(x3, a3, g3[:, D:], y, B=2, rows=450, D=D, ld_res=D, res_bs=450 * D, ld_y=D, y_bs=450 * D, gate_bs=2 * D))
    xs, Wg, bg = rnd(2, D, seed=5), rnd(58368, D, seed=6) / 39, rnd(58368, seed=7)  # the AdaLN modulation GEMV of the 375M model: the launch where the effect was found
    v["linear_small_gemv_2rows"] = (lambda: torch.empty(2, 58368, device="cuda"), lambda y: hip.linear_small(xs, Wg, y, rows=2, N=58368, K=D, bias=bg, act_in=hip.ACT_SILU))
    x20 = rnd(20, D, seed=8)
    v["linear_small_mfma_20rows"] = (lambda: torch.empty(20, 58368, device="cuda"), lambda y: hip.linear_small(x20, Wg, y, rows=20, N=58368, K=D, bias=bg, act_in=hip.ACT_SILU))
    xt = rnd(2, 84, 1800, seed=9)
    v["chan_to_token"] = (lambda: torch.empty(2, 1800, 96, device="cuda"), lambda y: hip.chan_to_token(xt, y, B=2, C=84, N=1800, ldo=96, fill_cols=96))
    tk = rnd(2, 1800, 96, seed=10)
    v["token_to_chan"] = (lambda: torch.empty(2, 84, 1800, device="cuda"), lambda y: hip.token_to_chan(tk, y, B=2, C=84, N=1800, ldi=96))
    # DCAE row kernels
    xr, w, b, r = rnd(900, 252, seed=11), rnd(252, seed=12), rnd(252, seed=13), rnd(900, 252, seed=14)
    v["rmsnorm_rows"] = (lambda: torch.empty(900, 252, device="cuda"), lambda y: hip.rmsnorm_rows(xr, w, y, rows=900, C=252, eps=1e-5, b=b, resid=r, act=hip.ACT_RELU))
    v["split_rows"] = (lambda: torch.empty(900, 256, device="cuda"), lambda y: hip.split_rows(xr, y, rows=900, C=252))
    xd, wd3, wd5, bd = rnd(2 * 15 * 30, 128, seed=15), rnd(9, 128, seed=16) / 3, rnd(25, 128, seed=17) / 5, rnd(128, seed=18)
    v["sphere_dwconv_3x3"] = (lambda: torch.empty(900, 128, device="cuda"), lambda y: hip.sphere_dwconv_nhwc(xd, wd3, y, B=2, H=15, W=30, C=128, bias=bd, ksize=3))
    v["sphere_dwconv_5x5"] = (lambda: torch.empty(900, 128, device="cuda"), lambda y: hip.sphere_dwconv_nhwc(xd, wd5, y, B=2, H=15, W=30, C=128, ksize=5))
    v["sphere_dwconv_3x3_glu"] = (lambda: torch.empty(900, 64, device="cuda"), lambda y: hip.sphere_dwconv_nhwc(xd, wd3, y, B=2, H=15, W=30, C=128, bias=bd, ksize=3, glu=True))
    xg, wg = rnd(900, 160, seed=19), rnd(160, 32, seed=20)
    v["grouped_conv1x1"] = (lambda: torch.empty(900, 160, device="cuda"), lambda y: hip.grouped_conv1x1_nhwc(xg, wg, y, M=900, groups=5, ldx=160, ldy=160))
    qkv = rnd(2, 450, 6 * 96, seed=21)
    v["relu_linear_attn"] = (lambda: torch.empty(2, 450, 6 * 32, device="cuda"), lambda y: hip.relu_linear_attn_nhwc(qkv, y, B=2, P=450, groups=6, ldq=6 * 96, ldy=6 * 32, eps=1e-15))
    xp, cu, cdn = rnd(2, 12, 16, 16, seed=22), rnd(2, 12, 16, 32, seed=23), rnd(2, 12, 16, 8, seed=24)
    v["pixel_shuffle_shortcut"] = (lambda: torch.empty(2, 24, 32, 8, device="cuda"), lambda y: hip.pixel_shuffle_shortcut(cu, xp, y, B=2, H=12, W=16, cout=8, cin=16))
    v["pixel_unshuffle_shortcut"] = (lambda: torch.empty(2, 6, 8, 32, device="cuda"), lambda y: hip.pixel_unshuffle_shortcut(cdn, xp, y, B=2, H2=6, W2=8, cout=32, cin=16))
    mreg = rnd(900, 48, seed=25)
    v["chan_regroup"] = (lambda: torch.empty(900, 12, device="cuda"), lambda y: hip.chan_regroup(mreg, y, M=900, cin=48, cout=12))
    # sampler state updates (fp64 state)
    n = 84 * 4 * 15 * 30
    xcur, Fm = (79.999985 * rnd(n, seed=26)).double(), rnd(n, seed=27)
    v["edm_scale_f64_to_f32"] = (lambda: torch.empty(n, device="cuda"), lambda y: hip.edm_scale_f64_to_f32(xcur, 0.0125, y))
    dcur = torch.empty(n, dtype=torch.float64, device="cuda")
    v["edm_euler"] = (lambda: torch.empty(n, dtype=torch.float64, device="cuda"), lambda y: hip.edm_euler(xcur, Fm, 3.9e-5, 0.49999, 79.999985, -20.342484, y, dcur))
    lat, mu, sd = rnd(2, 84, 4, 15, 30, seed=28), rnd(84, seed=29), rnd(84, seed=30).abs() + 0.5
    v["chan_affine"] = (lambda: torch.empty(2, 84, 4, 15, 30, device="cuda"), lambda y: hip.chan_affine(lat, y, mu, sd, 0.5, outer=2, C=84, inner=4 * 15 * 30, inverse=False))
    tt = torch.linspace(-1.5, 1.1, 20).cuda()
    v["timestep_embedding"] = (lambda: torch.empty(20, 256, device="cuda"), lambda y: hip.timestep_embedding(tt, y, 20))
    return v


VICTIMS = sorted([
    "layernorm_mod", "layernorm_mod_split_out", "mean_rows", "gate_residual", "linear_small_gemv_2rows", "linear_small_mfma_20rows", "chan_to_token",
    "token_to_chan", "rmsnorm_rows", "split_rows", "sphere_dwconv_3x3", "sphere_dwconv_5x5", "sphere_dwconv_3x3_glu", "grouped_conv1x1", "relu_linear_attn",
    "pixel_shuffle_shortcut", "pixel_unshuffle_shortcut", "chan_regroup", "edm_scale_f64_to_f32", "edm_euler", "chan_affine", "timestep_embedding"])


def test_row_kernels_next_to_an_mfma_aggressor_on_a_second_stream():
    aggr = _aggressor()
    victims = _victims()
    assert sorted(victims) == VICTIMS
    sink = torch.zeros(256, device="cuda")
    s_aggr, s_vict = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    overlap_seen = 0
    for name in VICTIMS:
        make, run = victims[name]
        ref = make()
        run(ref)  # solo, idle GPU
        torch.cuda.synchronize()
        outs = [make() for _ in range(LAUNCHES)]
        for o in outs:
            o.view(torch.uint8).fill_(0xFF)  # (NaN patterns: an unwritten word cannot pass for a result)
        torch.cuda.synchronize()
        ev0, ev1, va, vb = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(s_aggr):
            ev0.record()
            for _ in range(12):  # ~12 x 4 ms of MFMA streaming on every CU
                assert aggr.ldc_test_aggressor_launch(256, 9000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            ev1.record()
        with torch.cuda.stream(s_vict):
            va.record()
            for o in outs:
                run(o)
            vb.record()
        torch.cuda.synchronize()
        # the victim's launches really ran while the aggressor was running (both windows overlap)
        if ev0.elapsed_time(vb) > 0 and va.elapsed_time(ev1) > 0:
            overlap_seen += 1
        bad = [i for i, o in enumerate(outs) if not torch.equal(o.view(torch.uint8), ref.view(torch.uint8))]
        assert not bad, f"{name}: {len(bad)} of {LAUNCHES} launches next to the MFMA aggressor differ from the solo run (first: launch {bad[0]})"
    assert overlap_seen >= len(VICTIMS) - 2, f"only {overlap_seen} of {len(VICTIMS)} victim windows overlapped the aggressor's"


def test_first_read_reproducer_guarded_build_is_clean():
    """tools/canary/first_read_repro.hip - ONE translation unit holding the library's GEMV source and the aggressor, one
    process, two streams - as a regression test.  Built twice by `make -C ladcast_amd/csrc repro` (part of build()): as shipped, and with the
    guard of rowops.hip (ls_first_read) compiled out.  The shipped build must be clean next to the MFMA-streaming aggressor; the unguarded
    build's count is printed (round 4 measured 2880 of 2880 launches wrong on the MI355X it was found on): should a toolchain change make
    the guard unnecessary - or defeat it - this is where it shows."""
    import re
    import subprocess

    d = os.path.join(ROOT, "tools", "canary", "build")
    g, u = os.path.join(d, "first_read_repro_guarded"), os.path.join(d, "first_read_repro_unguarded")
    if not (os.path.exists(g) and os.path.exists(u)):
        pytest.skip("reproducer binaries not built (make -C ladcast_amd/csrc repro)")
    out = {}
    for name, exe in (("guarded", g), ("unguarded", u)):
        r = subprocess.run([exe, "1.5"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"next to the (\S+) aggressor: (\d+) of (\d+) launches wrong", r.stdout)}
        print(f"\n[{name}] " + r.stdout.strip().replace("\n", "\n[" + name + "] "))
    for mode in ("VALU-only", "MFMA-streaming"):
        wrong, launches = out["guarded"][mode]
        assert launches > 50 and wrong == 0, (mode, out["guarded"])
    assert out["unguarded"]["VALU-only"][0] == 0  # the control: without MFMAs next to it even the unguarded GEMV is right
