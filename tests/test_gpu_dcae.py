"""Parity of the HIP DCAE path (NHWC kernels behind the reference's AutoencoderDC / SphereConv2d API)
against the reference-pinned sphere-conv fixtures and the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dcae as OD  # noqa: E402
from tests.synth import make_dcae, rel_l2, synth_field, tiny_dcae_config  # noqa: E402


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_sphere_conv_matches_reference_fixtures(golden_dir):
    """fixtures = outputs of ladcast.models.sphere_conv.SphereConv2d (tests/golden/make_golden.py)"""
    from ladcast_amd.models import SphereConv2d

    g = np.load(os.path.join(golden_dir, "sphere_conv_ref.npz"))
    seen = 0
    i = 0
    while f"c{i}_meta" in g:
        ci, co, k, grp, b, H, W = [int(v) for v in g[f"c{i}_meta"]]
        i += 1
        if grp not in (1, ci) or (grp == ci and (ci != co or ci % 4)):
            continue  # grouped-but-not-depthwise cases are not used by the DCAE
        j = i - 1
        m = SphereConv2d(ci, co, k, 1, k // 2, groups=grp, bias=bool(b))
        m.weight.data = torch.from_numpy(g[f"c{j}_w"])
        if b:
            m.bias.data = torch.from_numpy(g[f"c{j}_b"])
        m = m.cuda()
        y = m(torch.from_numpy(g[f"c{j}_x"]).cuda())
        assert rel_l2(y.cpu(), torch.from_numpy(g[f"c{j}_y"])) < 2e-6, f"case {j}"
        seen += 1
    assert seen >= 3
    # docstring KAT (models/sphere_conv.py:142-172): exact small integers
    c = SphereConv2d(1, 1, 5, 1, 2)
    c.weight.data = torch.tensor([[[[0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 1, 0]]]], dtype=torch.float32)
    c.bias.data = torch.tensor([0.0])
    y = c.cuda()(torch.arange(0, 24).view(1, 1, 3, 8).float().cuda())
    assert torch.equal(y.cpu(), torch.from_numpy(g["kat_y"]))


@pytest.mark.parametrize("k,depthwise", [(3, False), (3, True), (5, True)])
def test_sphere_conv_pole_rows_vs_oracle(k, depthwise):
    from ladcast_amd.models import SphereConv2d
    from oracle.sphere_conv import SphereConv2d as OSC

    ci = co = 8
    o = OSC(ci, co, k, 1, k // 2, groups=ci if depthwise else 1, bias=True)
    with torch.no_grad():
        o.weight.copy_(rnd(*o.weight.shape, seed=1))
        o.bias.copy_(rnd(co, seed=2))
    m = SphereConv2d(ci, co, k, 1, k // 2, groups=ci if depthwise else 1, bias=True)
    m.load_state_dict(o.state_dict())
    x = rnd(2, ci, 9, 16, seed=3)
    with torch.no_grad():
        want = o(x)
    got = m.cuda()(x.cuda()).cpu()
    assert rel_l2(got, want) < 2e-6
    assert rel_l2(got[:, :, 0], want[:, :, 0]) < 2e-6 and rel_l2(got[:, :, -1], want[:, :, -1]) < 2e-6  # pole rows use the flipped kernel rows
    assert rel_l2(got[:, :, 1], want[:, :, 1]) < 2e-6  # row 1 of a 5x5 sees reflected input with the unflipped kernel


def test_dcae_elementwise_kernels():
    import ladcast_amd.hip as hip

    B, H, W, C = 2, 6, 8, 16
    x = rnd(B, C, H, W, seed=1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    # down: conv output has cout/4 channels at full res
    cout = 32
    cv = rnd(B, cout // 4, H, W, seed=2)
    want = F.pixel_unshuffle(cv, 2) + F.pixel_unshuffle(x, 2).unflatten(1, (-1, C * 4 // cout)).mean(dim=2)
    y = torch.empty(B, H // 2, W // 2, cout, device="cuda")
    hip.pixel_unshuffle_shortcut(cv.permute(0, 2, 3, 1).contiguous().cuda(), xn.cuda(), y, B=B, H2=H // 2, W2=W // 2, cout=cout, cin=C)
    assert rel_l2(y.cpu().permute(0, 3, 1, 2), want) < 1e-6
    # up
    cout = 8
    cv = rnd(B, cout * 4, H, W, seed=3)
    want = F.pixel_shuffle(cv, 2) + F.pixel_shuffle(x.repeat_interleave(cout * 4 // C, dim=1), 2)
    y = torch.empty(B, 2 * H, 2 * W, cout, device="cuda")
    hip.pixel_shuffle_shortcut(cv.permute(0, 2, 3, 1).contiguous().cuda(), xn.cuda(), y, B=B, H=H, W=W, cout=cout, cin=C)
    assert rel_l2(y.cpu().permute(0, 3, 1, 2), want) < 1e-6
    # regroup both ways
    m = rnd(50, 48, seed=4)
    y = torch.empty(50, 12, device="cuda")
    hip.chan_regroup(m.cuda(), y, M=50, cin=48, cout=12)
    assert rel_l2(y.cpu(), m.unflatten(1, (-1, 4)).mean(dim=2)) < 1e-6
    y = torch.empty(50, 144, device="cuda")
    hip.chan_regroup(m.cuda(), y, M=50, cin=48, cout=144)
    assert torch.equal(y.cpu(), m.repeat_interleave(3, dim=1))
    # rmsnorm rows with residual + relu
    xr, w, b, r = rnd(37, 252, seed=5), rnd(252, seed=6), rnd(252, seed=7), rnd(37, 252, seed=8)
    y = torch.empty(37, 252, device="cuda")
    hip.rmsnorm_rows(xr.cuda(), w.cuda(), y, rows=37, C=252, eps=1e-5, b=b.cuda(), resid=r.cuda(), act=hip.ACT_RELU)
    want = F.relu(xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * w + b + r)
    assert rel_l2(y.cpu(), want) < 1e-6


def test_linear_attention_and_grouped_conv():
    import ladcast_amd.hip as hip

    B, P, groups = 2, 450, 6
    qkv = rnd(B, P, groups * 96, seed=1)
    y = torch.empty(B, P, groups * 32, device="cuda")
    hip.relu_linear_attn_nhwc(qkv.cuda(), y, B=B, P=P, groups=groups, ldq=groups * 96, ldy=groups * 32, eps=1e-15)
    hs = qkv.permute(0, 2, 1).reshape(B, groups, 96, P).double()
    q, k, v = hs.chunk(3, dim=2)
    q, k = F.relu(q), F.relu(k)
    v = F.pad(v, (0, 0, 0, 1), value=1.0)
    out = (v @ k.transpose(-1, -2)) @ q
    want = (out[:, :, :-1] / (out[:, :, -1:] + 1e-15)).reshape(B, groups * 32, P).permute(0, 2, 1)
    assert rel_l2(y.cpu(), want) < 2e-6
    # round 3: the default form cuts a group's pixels into 128-pixel slices (two launches, partial KV matrices in scratch); the one-launch
    # form (a caller without scratch) sums in another order: equal to fp32 rounding.  The sliced order only depends on P: a frame's
    # result is bitwise the same alone and inside a batch; P not a multiple of 128 / of 32; below 1024 pixels both calls are the one-launch form
    for B2, P2, g2 in ((1, 1800, 5), (3, 1100, 4), (2, 1025, 3), (2, 450, 3)):
        q2 = rnd(B2, P2, g2 * 96, seed=5).cuda()
        ya = torch.full((B2, P2, g2 * 32), float("nan"), device="cuda")
        yb = torch.full((B2, P2, g2 * 32), float("nan"), device="cuda")
        hip.relu_linear_attn_nhwc(q2, ya, B=B2, P=P2, groups=g2, ldq=g2 * 96, ldy=g2 * 32, eps=1e-15)
        hip.relu_linear_attn_nhwc(q2, yb, B=B2, P=P2, groups=g2, ldq=g2 * 96, ldy=g2 * 32, eps=1e-15, sliced=False)
        assert torch.isfinite(ya).all() and rel_l2(ya.cpu(), yb.cpu()) < 2e-6
        y1 = torch.empty(1, P2, g2 * 32, device="cuda")
        hip.relu_linear_attn_nhwc(q2[B2 - 1 :].contiguous(), y1, B=1, P=P2, groups=g2, ldq=g2 * 96, ldy=g2 * 32, eps=1e-15)
        assert torch.equal(y1[0], ya[B2 - 1])
    M, G = 77, 5
    x, w = rnd(M, G * 32, seed=2), rnd(G * 32, 32, seed=3)
    y = torch.empty(M, G * 32, device="cuda")
    hip.grouped_conv1x1_nhwc(x.cuda(), w.cuda(), y, M=M, groups=G, ldx=G * 32, ldy=G * 32)
    want = F.conv2d(x.t().reshape(1, G * 32, M, 1), w.reshape(G * 32, 32, 1, 1), groups=G).reshape(G * 32, M).t()
    assert rel_l2(y.cpu(), want) < 2e-6


@pytest.mark.parametrize("k,B,H,W,C", [(5, 2, 7, 12, 96), (5, 1, 15, 30, 192), (3, 2, 9, 16, 64), (5, 1, 30, 60, 1440), (5, 1, 3, 8, 32)])
def test_multiscale_projection(k, B, H, W, C):
    """SanaMultiscaleAttentionProjection (models/DCAE.py:77-88): depthwise sphere conv (register-tiled along the image row; W = 8: the
    untiled kernel) + grouped 1x1 conv vs the oracle's two modules - pole rows (flipped kernel rows, W/2 roll), wrap, row widths that
    are not a multiple of the 8-pixel segment, strided source and destination (the qkv concat)"""
    import ladcast_amd.hip as hip
    from oracle.sphere_conv import SphereConv2d as OSC

    dw = OSC(C, C, k, 1, k // 2, groups=C, bias=False)
    with torch.no_grad():
        dw.weight.copy_(rnd(*dw.weight.shape, seed=1) / k)
    wg = rnd(C, 32, 1, 1, seed=2) / 32 ** 0.5
    x = rnd(B, C, H, W, seed=3)
    with torch.no_grad():
        want = F.conv2d(dw(x), wg, groups=C // 32)
    M = B * H * W
    xn = torch.full((M, C + 8), float("nan"), device="cuda")
    xn[:, :C] = x.cuda().permute(0, 2, 3, 1).reshape(M, C)
    y = torch.full((M, 2 * C), float("nan"), device="cuda")
    d = torch.empty(M, C, device="cuda")
    hip.sphere_dwconv_nhwc(xn, dw.weight.detach().reshape(C, k * k).t().contiguous().cuda(), d, B=B, H=H, W=W, C=C, ldx=C + 8, ksize=k)
    hip.grouped_conv1x1_nhwc(d, wg.reshape(C, 32).contiguous().cuda(), y[:, C:], M=M, groups=C // 32, ldx=C, ldy=2 * C)
    got = y[:, C:].reshape(B, H, W, C).permute(0, 3, 1, 2).cpu()
    assert torch.isnan(y[:, :C]).all()
    assert rel_l2(got, want) < 2e-6
    assert rel_l2(got[:, :, 0], want[:, :, 0]) < 2e-6 and rel_l2(got[:, :, -1], want[:, :, -1]) < 2e-6 and rel_l2(got[:, :, 1], want[:, :, 1]) < 2e-6


@pytest.mark.parametrize("k,B,H,W,C", [(3, 2, 9, 16, 64), (3, 1, 15, 30, 128), (3, 1, 7, 12, 192), (5, 1, 6, 8, 64)])
def test_glu_depthwise(k, B, H, W, C):
    """GLUMBConv's depthwise conv + gate (models/DCAE.py:311-313) vs the oracle (register-tiled kernel; W = 8 with k = 5: the untiled one)"""
    import ladcast_amd.hip as hip
    from oracle.sphere_conv import SphereConv2d as OSC

    dw = OSC(C, C, k, 1, k // 2, groups=C, bias=True)
    with torch.no_grad():
        dw.weight.copy_(rnd(*dw.weight.shape, seed=1) / k)
        dw.bias.copy_(rnd(C, seed=2))
    x = rnd(B, C, H, W, seed=3)
    with torch.no_grad():
        d = dw(x)
        want = d[:, : C // 2] * F.silu(d[:, C // 2 :])
    M = B * H * W
    xn = x.cuda().permute(0, 2, 3, 1).reshape(M, C).contiguous()
    y = torch.full((M, C // 2), float("nan"), device="cuda")
    hip.sphere_dwconv_nhwc(xn, dw.weight.detach().reshape(C, k * k).t().contiguous().cuda(), y, B=B, H=H, W=W, C=C, bias=dw.bias.detach().cuda(), ksize=k,
                           glu=True)
    got = y.reshape(B, H, W, C // 2).permute(0, 3, 1, 2).cpu()
    assert rel_l2(got, want) < 2e-6
    assert rel_l2(got[:, :, 0], want[:, :, 0]) < 2e-6 and rel_l2(got[:, :, -1], want[:, :, -1]) < 2e-6


def _pair(cfg):
    from ladcast_amd.models import AutoencoderDC

    o = make_dcae(cfg)
    g = AutoencoderDC.from_config(cfg)
    g.load_state_dict(o.state_dict(), strict=True)
    return o, g.cuda().eval()


def test_tiny_dcae_matches_oracle_and_pin(golden_dir):
    o, g = _pair(tiny_dcae_config())
    f, st = synth_field(2, 8, 48, 64), synth_field(1, 5, 48, 64, seed=1)
    with torch.no_grad():
        zo = o.encode(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        yo = o.decode(zo).sample
    zg = g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent
    assert zg.shape == zo.shape and rel_l2(zg.cpu(), zo) < 2e-5
    yg = g.decode(zo.cuda()).sample
    assert yg.shape == yo.shape == (2, 8, 48, 64) and rel_l2(yg.cpu(), yo) < 2e-5
    full = g(f.cuda(), static_conditioning_tensor=st.cuda(), return_static=True).sample
    assert full.shape == (2, 13, 48, 64)
    pin = np.load(os.path.join(golden_dir, "oracle_pins.npz"))
    z1 = g.encode(f[:1].cuda(), static_conditioning_tensor=st.cuda()).latent
    got = z1.cpu().double().flatten()[::13][:4096]
    want = torch.from_numpy(pin["tiny_dcae_z"])
    assert ((got - want).norm() / want.norm()).item() < 2e-5


def _hip_of(o, cfg):
    from ladcast_amd.models import AutoencoderDC

    g = AutoencoderDC.from_config(cfg)
    g.load_state_dict(o.state_dict(), strict=True)
    return g.cuda().eval()


def test_full_dcae_single_frame_all_modes(full_dcae_oracle):
    """BASELINE configs[0]: one 240x121(->120)x84 frame, encode + decode, at full size in the three arithmetic modes, each at its
    stated tolerance (ladcast_amd/precision.py); the single-term `bf16` mode also against the oracle under the reference's own mixed
    precision (oracle/autocast.py: torch.autocast + the fp32 islands of models/DCAE.py:162,180)."""
    from ladcast_amd.precision import tolerance
    from oracle import autocast as OA

    d = full_dcae_oracle
    g = _hip_of(d.model, d.cfg)
    with torch.no_grad(), OA.reference_autocast("cuda"):
        a_enc = rel_l2(d.model.encode(d.f, static_conditioning_tensor=d.st).latent.float(), d.z)
        a_dec = rel_l2(d.model.decode(d.z).sample.float(), d.y)
    for mode in ("fp32", "bf16x3", "bf16"):
        g.set_gemm_precision(mode)
        zg = g.encode(d.f.cuda(), static_conditioning_tensor=d.st.cuda()).latent
        yg = g.decode(d.z.cuda()).sample
        assert zg.shape == (1, 84, 15, 30) and yg.shape == (1, 84, 120, 240)
        ez, ey = rel_l2(zg.cpu(), d.z), rel_l2(yg.cpu(), d.y)
        print(f"\nfull DCAE, one frame [{mode}]: encode rel-L2 {ez:.2e}, decode rel-L2 {ey:.2e}" +
              (f" (oracle under autocast: {a_enc:.2e} / {a_dec:.2e})" if mode == "bf16" else ""))
        if mode == "bf16":
            assert 1e-5 < ez < tolerance("bf16", "dcae_encode") and 1e-5 < ey < tolerance("bf16", "dcae_decode"), (ez, ey)
            assert ez <= a_enc and ey <= a_dec, (ez, a_enc, ey, a_dec)
        else:
            assert ez < 5e-5 and ey < tolerance(mode, "dcae"), (mode, ez, ey)
        if mode == "fp32":  # decode of its own latent, the round trip of configs[0]
            assert rel_l2(g.decode(zg).sample.cpu(), d.y) < 1e-4


def test_decode_latent_ens_and_error_conventions():
    from ladcast_amd.pipelines import decode_latent_ens
    from oracle.pipelines import decode_latent_ens as o_decode

    o, g = _pair(tiny_dcae_config())
    lat = rnd(2, 8, 3, 6, 8, seed=4)
    mu, sd = rnd(8, seed=5), rnd(8, seed=6).abs() + 0.5
    with torch.no_grad():
        want = o_decode(o, lat, mu, sd, extract_first=2)
    got = decode_latent_ens(g, lat.cuda(), mu, sd, extract_first=2)
    assert got.shape == want.shape == (2, 8, 2, 48, 64)
    assert rel_l2(got.cpu(), want) < 2e-5
    g.enable_slicing()
    with pytest.raises(NotImplementedError):
        g.decode(lat[:, :, 0].cuda())
    with pytest.raises(NotImplementedError):
        g.encode(synth_field(2, 13, 48, 64).cuda())
    g.disable_slicing()


def test_bulk_encoder_matches_frame_by_frame_oracle():
    """SURVEY 8(f) rank 3 (preprocecss/encode_data.py:20-100): the reference encodes one frame per call; the batched
    path must give the same latents, for a tensor and for a callable source, incl. a ragged last batch"""
    from ladcast_amd.pipelines import encode_latents

    o, g = _pair(tiny_dcae_config())
    frames, st = synth_field(5, 8, 48, 64, seed=3), synth_field(1, 5, 48, 64, seed=1)[0]
    with torch.no_grad():
        want = torch.cat([o.encode(frames[i : i + 1], static_conditioning_tensor=st.unsqueeze(0)).latent for i in range(5)], dim=0)
    got = encode_latents(g, frames, static_conditioning_tensor=st, batch_size=2)
    assert got.device.type == "cpu" and got.shape == want.shape and rel_l2(got, want) < 2e-5
    got2 = encode_latents(g, lambda i: frames[i], total_samples=5, static_conditioning_tensor=st, batch_size=4)
    # the batch split changes a frame's result at fp32 rounding level only (round 4: the fp32 convs run on the stream-K ring kernel, whose
    # unit-range cut - i.e. the order in which a split tile's k-ranges are added - depends on the number of frames in the launch)
    assert rel_l2(got2, got) < 1e-6
    with pytest.raises(ValueError):
        encode_latents(g, lambda i: frames[i], static_conditioning_tensor=st)


def _unsplit(buf, rows, cols):
    """split rows (every 32 bytes = [hi x8 | lo x8] bf16) -> (hi, lo) fp32 [rows, cols], cols = the padded width"""
    w = buf.detach().cpu().contiguous().view(torch.int16).reshape(rows, cols // 8, 2, 8)
    f = (w.to(torch.int32) << 16).view(torch.float32)
    return f[:, :, 0].reshape(rows, cols), f[:, :, 1].reshape(rows, cols)


def _split_ref(x):
    hi = x.bfloat16().float()
    return hi, (x - hi).bfloat16().float()


def _check_split_rows(ys, want, C):
    """ys: device split rows [rows, C8]; want: fp32 [rows, C] the producer's fp32 output - bitwise the hi / lo split of it, pad zero"""
    rows, c8 = ys.shape
    hi, lo = _unsplit(ys, rows, c8)
    wh, wl = _split_ref(want.cpu())
    assert torch.equal(hi[:, :C], wh) and torch.equal(lo[:, :C], wl)
    assert (hi[:, C:] == 0).all() and (lo[:, C:] == 0).all()


@pytest.mark.parametrize("ci,co,k,B,H,W", [(8, 8, 3, 2, 9, 16), (252, 252, 3, 1, 30, 60), (96, 252, 3, 1, 12, 24), (504, 252, 3, 2, 15, 30),
                                            (12, 20, 5, 1, 7, 12), (1008, 1008, 3, 1, 15, 30), (252, 252, 3, 1, 120, 240), (84, 1008, 3, 3, 15, 30),
                                            (1008, 3024, 1, 1, 15, 30), (2016, 504, 1, 2, 30, 60), (252, 92, 3, 5, 16, 32)])
def test_sphere_conv_split_vs_oracle(ci, co, k, B, H, W):
    """the pre-split implicit-GEMM SphereConv2d (ldc_sphere_conv_nhwc_split, the DCAE's bf16x3 path): split rows in (channel counts
    4 mod 8 with zero pad columns), pole rows, wrap, several images per launch, bias + activation + residual, fp32 rows and split
    rows out; 1e-5 like the bf16x3 GEMM.  k = 1: the pointwise convs / Linears."""
    import ladcast_amd.hip as hip
    from ladcast_amd.models.sphere_conv import pack_dense_weight_bf16x3
    from oracle.sphere_conv import SphereConv2d as OSC

    M = B * H * W
    x, res = rnd(B, ci, H, W, seed=3), rnd(B, co, H, W, seed=4)
    if k > 1:
        o = OSC(ci, co, k, 1, k // 2, bias=True)
        with torch.no_grad():
            o.weight.copy_(rnd(*o.weight.shape, seed=1) / (ci * k * k) ** 0.5)
            o.bias.copy_(rnd(co, seed=2))
            want = torch.nn.functional.silu(o(x)) + res
        w4, bias = o.weight, o.bias
    else:
        w4, bias = rnd(co, ci, 1, 1, seed=1) / ci ** 0.5, rnd(co, seed=2)
        want = torch.nn.functional.silu(torch.einsum("oc,bchw->bohw", w4[:, :, 0, 0].double(), x.double()) + bias.double()[None, :, None, None]).float() + res
    c8 = -(-ci // 8) * 8
    xs = torch.full((M, c8), float("nan"), device="cuda")
    hip.split_rows(x.cuda().permute(0, 2, 3, 1).reshape(M, ci).contiguous(), xs, rows=M, C=ci)
    r = res.cuda().permute(0, 2, 3, 1).reshape(M, co).contiguous()
    wp = pack_dense_weight_bf16x3(w4.detach().cuda())
    y = torch.full((M, co), float("nan"), device="cuda")
    hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=bias.detach().cuda(), R=r, ldr=co, ksize=k, act=hip.ACT_SILU)
    got = y.reshape(B, H, W, co).permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) < 1e-5
    assert rel_l2(got[:, :, 0], want[:, :, 0]) < 1e-5 and rel_l2(got[:, :, -1], want[:, :, -1]) < 1e-5  # pole rows
    # split rows out: bitwise the split of the fp32 rows (same accumulators, same epilogue), pad columns zero
    o8 = -(-co // 8) * 8
    ys = torch.full((M, o8), float("nan"), device="cuda")
    hip.sphere_conv_nhwc_split(xs, wp, ys, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, ldy=o8, bias=bias.detach().cuda(), R=r, ldr=co, ksize=k,
                               act=hip.ACT_SILU, out_fmt=hip.FMT_SPLIT)
    _check_split_rows(ys, y, co)


def _from_bf16_rows(buf, rows, cols):
    """plain bf16 rows (a row's values in the first 2 * cols bytes of its fp32-sized row) -> fp32 [rows, cols]"""
    return buf.detach().cpu().contiguous().view(torch.bfloat16).reshape(rows, -1)[:, :cols].float()


@pytest.mark.parametrize("ci,co,k,B,H,W", [(8, 8, 3, 2, 9, 16), (252, 252, 3, 1, 30, 60), (96, 252, 3, 1, 12, 24), (504, 252, 3, 2, 15, 30),
                                            (12, 20, 5, 1, 7, 12), (1008, 1008, 3, 1, 15, 30), (1008, 3024, 1, 1, 15, 30), (252, 92, 3, 5, 16, 32)])
def test_sphere_conv_bf16_single_term(ci, co, k, B, H, W):
    """the same conv in the single-term `bf16` mode (in_fmt = LDC_FMT_BF16: plain bf16 operand rows, 64 channels per k-step): equal to
    the fp64 conv of the bf16-ROUNDED input and weight to 1e-5 (fp32 accumulation, nothing else is rounded); bf16 rows out = the
    rounding of the fp32 rows"""
    import ladcast_amd.hip as hip
    from ladcast_amd.models.sphere_conv import pack_dense_weight_bf16
    from oracle.sphere_conv import SphereConv2d as OSC

    M = B * H * W
    x, res = rnd(B, ci, H, W, seed=3), rnd(B, co, H, W, seed=4)
    w4 = rnd(co, ci, k, k, seed=1) / (ci * k * k) ** 0.5
    bias = rnd(co, seed=2)
    xr, wr = x.bfloat16().double(), w4.bfloat16().double()
    if k > 1:
        o = OSC(ci, co, k, 1, k // 2, bias=True).double()
        with torch.no_grad():
            o.weight.copy_(wr)
            o.bias.copy_(bias.double())
            want = (torch.nn.functional.silu(o(xr)) + res.double()).float()
    else:
        want = (torch.nn.functional.silu(torch.einsum("oc,bchw->bohw", wr[:, :, 0, 0], xr) + bias.double()[None, :, None, None]) + res.double()).float()
    c8 = -(-ci // 8) * 8
    xs = torch.full((M, c8), float("nan"), device="cuda")
    hip.split_rows(x.cuda().permute(0, 2, 3, 1).reshape(M, ci).contiguous(), xs, rows=M, C=ci, fmt=hip.FMT_BF16)
    assert torch.equal(_from_bf16_rows(xs, M, ci), x.permute(0, 2, 3, 1).reshape(M, ci).bfloat16().float())
    r = res.cuda().permute(0, 2, 3, 1).reshape(M, co).contiguous()
    wp = pack_dense_weight_bf16(w4.cuda())
    y = torch.full((M, co), float("nan"), device="cuda")
    hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=bias.cuda(), R=r, ldr=co, ksize=k, act=hip.ACT_SILU,
                               in_fmt=hip.FMT_BF16)
    got = y.reshape(B, H, W, co).permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) < 1e-5
    assert rel_l2(got[:, :, 0], want[:, :, 0]) < 1e-5 and rel_l2(got[:, :, -1], want[:, :, -1]) < 1e-5  # pole rows
    o8 = -(-co // 8) * 8
    ys = torch.full((M, o8), float("nan"), device="cuda")
    hip.sphere_conv_nhwc_split(xs, wp, ys, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, ldy=o8, bias=bias.cuda(), R=r, ldr=co, ksize=k,
                               act=hip.ACT_SILU, in_fmt=hip.FMT_BF16, out_fmt=hip.FMT_BF16)
    rows = ys.detach().cpu().contiguous().view(torch.bfloat16).reshape(M, -1)
    assert torch.equal(rows[:, :co].float(), y.cpu().bfloat16().float())
    assert (rows[:, co:o8] == 0).all()


def test_dcae_bf16_row_producers():
    """the producers with fmt = LDC_FMT_BF16: bitwise the bf16 rounding of their fp32 rows, pad columns zero"""
    import ladcast_amd.hip as hip

    def check(ys, y, C):
        rows = ys.detach().cpu().contiguous().view(torch.bfloat16).reshape(ys.shape[0], -1)
        c8 = -(-C // 8) * 8
        assert torch.equal(rows[:, :C].float(), y.cpu().bfloat16().float()) and (rows[:, C:c8] == 0).all()

    C = 252
    xr, w, b, r = rnd(37, C, seed=5), rnd(C, seed=6), rnd(C, seed=7), rnd(37, C, seed=8)
    y, ys = torch.empty(37, C, device="cuda"), torch.full((37, 256), float("nan"), device="cuda")
    hip.rmsnorm_rows(xr.cuda(), w.cuda(), y, rows=37, C=C, eps=1e-5, b=b.cuda(), resid=r.cuda(), act=hip.ACT_RELU, ys=ys, fmt=hip.FMT_BF16)
    check(ys, y, C)
    B, H, W = 2, 6, 8
    cv, xx = rnd(B, 5, H, W, seed=2), rnd(B, 20, H, W, seed=11)
    M2 = B * (H // 2) * (W // 2)
    y, ys = torch.empty(M2, 20, device="cuda"), torch.full((M2, 24), float("nan"), device="cuda")
    hip.pixel_unshuffle_shortcut(cv.permute(0, 2, 3, 1).contiguous().cuda(), xx.permute(0, 2, 3, 1).contiguous().cuda(), y, B=B, H2=H // 2, W2=W // 2,
                                 cout=20, cin=20, ys=ys, fmt=hip.FMT_BF16)
    check(ys, y, 20)
    cv, xx = rnd(B, 48, H, W, seed=3), rnd(B, 24, H, W, seed=12)
    M4 = B * 4 * H * W
    y, ys = torch.empty(M4, 12, device="cuda"), torch.full((M4, 16), float("nan"), device="cuda")
    hip.pixel_shuffle_shortcut(cv.permute(0, 2, 3, 1).contiguous().cuda(), xx.permute(0, 2, 3, 1).contiguous().cuda(), y, B=B, H=H, W=W, cout=12, cin=24,
                               ys=ys, fmt=hip.FMT_BF16)
    check(ys, y, 12)
    Bq, P, groups = 2, 450, 6
    qkv = rnd(Bq, P, groups * 96, seed=1).cuda()
    y, ys = torch.empty(Bq * P, groups * 32, device="cuda"), torch.full((Bq * P, groups * 32), float("nan"), device="cuda")
    hip.relu_linear_attn_nhwc(qkv, y, B=Bq, P=P, groups=groups, ldq=groups * 96, ldy=groups * 32, eps=1e-15)
    hip.relu_linear_attn_nhwc(qkv, ys, B=Bq, P=P, groups=groups, ldq=groups * 96, ldy=groups * 32, eps=1e-15, out_fmt=hip.FMT_BF16)
    check(ys, y, groups * 32)
    Cd = 64
    xd, wd, bd = rnd(2 * 9 * 16, Cd, seed=4).cuda(), rnd(9, Cd, seed=5).cuda(), rnd(Cd, seed=6).cuda()
    y, ys = torch.empty(2 * 9 * 16, Cd // 2, device="cuda"), torch.full((2 * 9 * 16, Cd // 2), float("nan"), device="cuda")
    hip.sphere_dwconv_nhwc(xd, wd, y, B=2, H=9, W=16, C=Cd, bias=bd, ksize=3, glu=True)
    hip.sphere_dwconv_nhwc(xd, wd, ys, B=2, H=9, W=16, C=Cd, bias=bd, ksize=3, glu=True, out_fmt=hip.FMT_BF16)
    check(ys, y, Cd // 2)


def test_dcae_bf16_mode_matches_oracle_at_its_tolerance():
    """AutoencoderDC.set_gemm_precision('bf16') (BASELINE configs[4], mixed precision): dense / 1x1 convs and Linears with ONE bf16 MFMA
    per product on plain bf16 operand rows, everything else as in the other modes.  Stated tolerance vs the fp32 oracle per encode /
    decode: ladcast_amd/precision.py (the 1e-4 budget does not apply to this mode); comparator: the oracle under the reference's mixed
    precision.  Tiny config here, the full 84 x 120 x 240 frame in test_full_dcae_single_frame_all_modes."""
    from ladcast_amd.precision import tolerance
    from oracle import autocast as OA

    cfg, shape = tiny_dcae_config(), (2, 8, 48, 64)
    o, g = _pair(cfg)
    g.set_gemm_precision("bf16")
    f, st = synth_field(*shape), synth_field(1, 5, shape[2], shape[3], seed=1)
    with torch.no_grad():
        zo = o.encode(f, static_conditioning_tensor=st.expand(shape[0], -1, -1, -1)).latent
        yo = o.decode(zo).sample
        with OA.reference_autocast("cuda"):
            a_enc = rel_l2(o.encode(f, static_conditioning_tensor=st.expand(shape[0], -1, -1, -1)).latent.float(), zo)
            a_dec = rel_l2(o.decode(zo).sample.float(), yo)
    zg = g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent
    yg = g.decode(zo.cuda()).sample
    ez, ey = rel_l2(zg.cpu(), zo), rel_l2(yg.cpu(), yo)
    print(f"\nDCAE bf16 (single-term) {shape}: encode rel-L2 {ez:.2e}, decode rel-L2 {ey:.2e}; oracle under autocast {a_enc:.2e} / {a_dec:.2e}")
    assert 1e-5 < ez < tolerance("bf16", "dcae_encode") and 1e-5 < ey < tolerance("bf16", "dcae_decode"), (ez, ey)  # lower bound: the mode is really on
    assert ez <= a_enc and ey <= a_dec, (ez, a_enc, ey, a_dec)


def test_dcae_split_row_producers():
    """every producer of a conv operand writes the split rows itself: bitwise the hi / lo split of its fp32 output, pad columns
    (C = 4 mod 8) zero"""
    import ladcast_amd.hip as hip

    # RMSNorm rows: fp32 + split, split only
    for C in (252, 504, 8):
        xr, w, b, r = rnd(37, C, seed=5), rnd(C, seed=6), rnd(C, seed=7), rnd(37, C, seed=8)
        c8 = -(-C // 8) * 8
        y, ys, ys2 = torch.empty(37, C, device="cuda"), torch.full((37, c8), float("nan"), device="cuda"), torch.full((37, c8), float("nan"), device="cuda")
        hip.rmsnorm_rows(xr.cuda(), w.cuda(), y, rows=37, C=C, eps=1e-5, b=b.cuda(), resid=r.cuda(), act=hip.ACT_RELU, ys=ys)
        want = F.relu(xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * w + b + r)
        assert rel_l2(y.cpu(), want) < 1e-6
        _check_split_rows(ys, y, C)
        hip.rmsnorm_rows(xr.cuda(), w.cuda(), None, rows=37, C=C, eps=1e-5, b=b.cuda(), resid=r.cuda(), act=hip.ACT_RELU, ys=ys2)
        assert torch.equal(ys2.view(torch.int32), ys.view(torch.int32))
    # pixel (un)shuffle shortcuts
    B, H, W, C = 2, 6, 8, 16
    x = rnd(B, C, H, W, seed=1)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    for cout in (32, 20):  # 20: 4 mod 8 (pad), cin * 4 % cout == 0 needs C = 20 g / 4 -> use C = 20 for it
        Cx = 16 if cout == 32 else 20
        xx = rnd(B, Cx, H, W, seed=11)
        cv = rnd(B, cout // 4, H, W, seed=2)
        want = F.pixel_unshuffle(cv, 2) + F.pixel_unshuffle(xx, 2).unflatten(1, (-1, Cx * 4 // cout)).mean(dim=2)
        M2 = B * (H // 2) * (W // 2)
        y, ys = torch.empty(M2, cout, device="cuda"), torch.full((M2, -(-cout // 8) * 8), float("nan"), device="cuda")
        hip.pixel_unshuffle_shortcut(cv.permute(0, 2, 3, 1).contiguous().cuda(), xx.permute(0, 2, 3, 1).contiguous().cuda(), y, B=B, H2=H // 2,
                                     W2=W // 2, cout=cout, cin=Cx, ys=ys)
        assert rel_l2(y.cpu().reshape(B, H // 2, W // 2, cout).permute(0, 3, 1, 2), want) < 1e-6
        _check_split_rows(ys, y, cout)
    for cout in (8, 12):
        cv = rnd(B, cout * 4, H, W, seed=3)
        Cx = 16 if cout == 8 else 24
        xx = rnd(B, Cx, H, W, seed=12)
        want = F.pixel_shuffle(cv, 2) + F.pixel_shuffle(xx.repeat_interleave(cout * 4 // Cx, dim=1), 2)
        M4 = B * 4 * H * W
        y, ys = torch.empty(M4, cout, device="cuda"), torch.full((M4, -(-cout // 8) * 8), float("nan"), device="cuda")
        hip.pixel_shuffle_shortcut(cv.permute(0, 2, 3, 1).contiguous().cuda(), xx.permute(0, 2, 3, 1).contiguous().cuda(), y, B=B, H=H, W=W,
                                   cout=cout, cin=Cx, ys=ys)
        assert rel_l2(y.cpu().reshape(B, 2 * H, 2 * W, cout).permute(0, 3, 1, 2), want) < 1e-6
        _check_split_rows(ys, y, cout)
    # ReLU linear attention and the GLU depthwise conv: split rows = split of their fp32 rows
    Bq, P, groups = 2, 450, 6
    qkv = rnd(Bq, P, groups * 96, seed=1).cuda()
    y, ys = torch.empty(Bq * P, groups * 32, device="cuda"), torch.full((Bq * P, groups * 32), float("nan"), device="cuda")
    hip.relu_linear_attn_nhwc(qkv, y, B=Bq, P=P, groups=groups, ldq=groups * 96, ldy=groups * 32, eps=1e-15)
    hip.relu_linear_attn_nhwc(qkv, ys, B=Bq, P=P, groups=groups, ldq=groups * 96, ldy=groups * 32, eps=1e-15, out_fmt=hip.FMT_SPLIT)
    _check_split_rows(ys, y, groups * 32)
    Cd = 64
    xd, wd, bd = rnd(2 * 9 * 16, Cd, seed=4).cuda(), rnd(9, Cd, seed=5).cuda(), rnd(Cd, seed=6).cuda()
    y, ys = torch.empty(2 * 9 * 16, Cd // 2, device="cuda"), torch.full((2 * 9 * 16, Cd // 2), float("nan"), device="cuda")
    hip.sphere_dwconv_nhwc(xd, wd, y, B=2, H=9, W=16, C=Cd, bias=bd, ksize=3, glu=True)
    hip.sphere_dwconv_nhwc(xd, wd, ys, B=2, H=9, W=16, C=Cd, bias=bd, ksize=3, glu=True, out_fmt=hip.FMT_SPLIT)
    _check_split_rows(ys, y, Cd // 2)


def test_dcae_bf16x3_mode_matches_oracle():
    """AutoencoderDC.set_gemm_precision('bf16x3'): dense 3x3 convs as split-bf16 implicit GEMMs; tiny config (both paths of
    the encoder / decoder) and one full 84 x 120 x 240 frame; tolerance 5e-5 (~40 conv layers at ~4e-6 each, fp32 budget 1e-4)"""
    o, g = _pair(tiny_dcae_config())
    g.set_gemm_precision("bf16x3")
    f, st = synth_field(2, 8, 48, 64), synth_field(1, 5, 48, 64, seed=1)
    with torch.no_grad():
        zo = o.encode(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        yo = o.decode(zo).sample
    zg = g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent
    yg = g.decode(zo.cuda()).sample
    assert rel_l2(zg.cpu(), zo) < 5e-5 and rel_l2(yg.cpu(), yo) < 5e-5
    g.set_gemm_precision("fp32")
    assert rel_l2(g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent.cpu(), zo) < 2e-5  # switching back re-packs


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_dcae_hip_graph_is_bitwise_equal_to_eager(prec):
    """enable_hip_graph: one captured graph per (direction, shape), replayed with new inputs - the same launches, so encode and
    decode are bit-identical to eager launching; a second shape and a precision switch get their own graphs"""
    _, g = _pair(tiny_dcae_config())
    g.set_gemm_precision(prec)
    cases = [(synth_field(2, 8, 48, 64).cuda(), synth_field(1, 5, 48, 64, seed=1).cuda()), (synth_field(2, 8, 48, 64, seed=5).cuda() * 0.5, synth_field(1, 5, 48, 64, seed=6).cuda()),
             (synth_field(1, 8, 32, 64, seed=7).cuda(), synth_field(1, 5, 32, 64, seed=8).cuda())]
    runs = {}
    for mode in (False, True, True):
        g.enable_hip_graph(mode)
        res = []
        for f, st in cases:
            z = g.encode(f, static_conditioning_tensor=st).latent
            res.append((z.clone(), g.decode(z).sample.clone(), g.decode(z, return_static=True).sample.clone()))
        runs.setdefault(mode, []).append(res)
        if mode:
            assert len(g._graphs) == 6  # (encode, decode, decode with static) x two shapes
    g.enable_hip_graph(False)
    assert g._graphs == {}
    for a, b, c in zip(runs[False][0], runs[True][0], runs[True][1]):
        for x, y, z in zip(a, b, c):
            assert torch.equal(x, y) and torch.equal(x, z)
    assert runs[False][0][0][2].shape[1] == runs[False][0][0][1].shape[1] + 5


def test_dc_ae_ray_1024_shape_one_frame():
    """The reference's OTHER shipped autoencoder config, configs/DC_AE_ray_1024.yaml:1-50 - 90 in / out channels (84 + 6 static fields),
    1024 latent channels, widths 512 / 1024 / 1024 / 2048 (64 linear-attention heads at the deepest stage), 1.09 G parameters - is
    inside the supported set: instantiated at full width, one frame encoded and decoded against the oracle in the exact-fp32 and the
    split-bf16 mode.  The frame is 48 x 96 (latent 6 x 12: still the linear-attention branch) so that the CPU oracle needs seconds."""
    cfg = dict(OD.CONFIG_DCAE_84, in_channels=90, out_channels=90, latent_channels=1024, encoder_block_out_channels=(512, 1024, 1024, 2048),
               decoder_block_out_channels=(512, 1024, 1024, 2048), static_channels=6)
    o = make_dcae(cfg)
    g = _hip_of(o, cfg)
    f, st = synth_field(1, 84, 48, 96), synth_field(1, 6, 48, 96, seed=1)
    with torch.no_grad():
        zo = o.encode(f, static_conditioning_tensor=st).latent
        yo = o.decode(zo).sample
    assert zo.shape == (1, 1024, 6, 12) and yo.shape == (1, 84, 48, 96)
    for mode in ("fp32", "bf16x3"):
        g.set_gemm_precision(mode)
        zg = g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent
        yg = g.decode(zo.cuda()).sample
        ez, ey = rel_l2(zg.cpu(), zo), rel_l2(yg.cpu(), yo)
        print(f"\nDC_AE_ray_1024 shape, one 48 x 96 frame [{mode}]: encode rel-L2 {ez:.2e}, decode rel-L2 {ey:.2e}")
        assert zg.shape == zo.shape and yg.shape == yo.shape
        assert ez < 1e-4 and ey < 1e-4, (mode, ez, ey)



def test_interpolate_upsampling_variant_matches_oracle(golden_dir):
    """VERDICT r04 missing 3 / item 8: `upsample_block_type="interpolate"` (models/DCAE.py:498-525,677-682: nearest x2 up-sampling, a conv at
    the output width, the same shortcut) used to raise NotImplementedError.  (a) the up block alone against the fixture made by the
    REFERENCE's own DCUpBlock2d code (tests/golden/pieces_ref.npz `up_interp`, make_golden.py::piece_fixtures); (b) a tiny DC-AE with
    interpolate decoders, decode in the three arithmetic modes against the oracle (whose block is pinned to the same fixture in
    tests/test_oracle_reference_pins.py)."""
    from ladcast_amd.precision import tolerance
    from tests.synth import piece_inputs, piece_modules

    import ladcast_amd.hip as hip

    z = np.load(os.path.join(golden_dir, "pieces_ref.npz"))
    om, x = piece_modules()["up_interp"], piece_inputs()["up_x"]
    B, ci, H, W = x.shape
    co = om.conv.out_channels
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, ci).contiguous().cuda()
    up = torch.empty(B * 4 * H * W, ci, device="cuda")
    hip.upsample_nearest2x_rows(rows, up, B=B, H=H, W=W, C=ci)
    assert torch.equal(up.reshape(B, 2 * H, 2 * W, ci).permute(0, 3, 1, 2).cpu(), F.interpolate(x, scale_factor=2, mode="nearest"))
    sc = torch.empty(B * 4 * H * W, co, device="cuda")
    hip.pixel_shuffle_shortcut(None, rows, sc, B=B, H=H, W=W, cout=co, cin=ci)
    assert torch.equal(sc.reshape(B, 2 * H, 2 * W, co).permute(0, 3, 1, 2).cpu(), F.pixel_shuffle(x.repeat_interleave(om.repeats, dim=1), 2))
    from ladcast_amd.models.sphere_conv import pack_dense_weight

    y = torch.empty(B * 4 * H * W, co, device="cuda")
    hip.sphere_conv_nhwc(up, pack_dense_weight(om.conv.weight.detach().cuda()), y, B=B, H=2 * H, W=2 * W, cin=ci, cout=co, bias=om.conv.bias.detach().cuda(), R=sc, ldr=co,
                         ksize=3)
    got = y.reshape(B, 2 * H, 2 * W, co).permute(0, 3, 1, 2).cpu()
    assert rel_l2(got, torch.from_numpy(z["up_interp"])) < 2e-6
    # (b) the whole autoencoder with interpolate decoders
    cfg = dict(tiny_dcae_config(), upsample_block_type="interpolate")
    o, g = _pair(cfg)
    assert all(b.interpolate for b in g.decoder.up_blocks if hasattr(b, "interpolate"))
    f, st = synth_field(2, 8, 48, 64), synth_field(1, 5, 48, 64, seed=1)
    with torch.no_grad():
        zz = o.encode(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        want = o.decode(zz).sample
    for mode in ("fp32", "bf16x3", "bf16"):
        g.set_gemm_precision(mode)
        got = g.decode(zz.cuda()).sample
        e = rel_l2(got.cpu(), want)
        print(f"\ntiny DC-AE, interpolate up-sampling, decode [{mode}]: rel-L2 {e:.2e}")
        assert e < (tolerance("bf16", "dcae_decode") if mode == "bf16" else 2e-5 if mode == "fp32" else 5e-5), (mode, e)
    g.set_gemm_precision("fp32")
    g.enable_hip_graph(True)
    assert torch.equal(g.decode(zz.cuda()).sample, g.decode(zz.cuda()).sample)
    g.enable_hip_graph(False)


class _ReferenceProtocolProcessor:
    """A user-supplied DC-AE attention processor written against the reference's protocol (models/DCAE.py:205-267: `proc(attn, hidden_states NCHW,
    gate=None)`, reading `attn.to_q / to_k / to_v / to_qkv_multiscale / nonlinearity / apply_linear_attention / to_out / norm_type / norm_out /
    residual_connection / attention_head_dim`), in plain torch on whatever device the tensors live."""

    def __init__(self):
        self.seen = []

    def __call__(self, attn, hidden_states, gate=None):
        self.seen.append((tuple(hidden_states.shape), None if gate is None else tuple(gate.shape), hidden_states.is_cuda))
        B, _, H, W = hidden_states.shape
        residual = hidden_states
        hl = hidden_states.movedim(1, -1)
        qkv = torch.cat([attn.to_q(hl), attn.to_k(hl), attn.to_v(hl)], dim=3).movedim(-1, 1)
        hs = torch.cat([qkv] + [blk(qkv) for blk in attn.to_qkv_multiscale], dim=1).to(torch.float32)
        q, k, v = hs.reshape(B, -1, 3 * attn.attention_head_dim, H * W).chunk(3, dim=2)
        q, k = attn.nonlinearity(q), attn.nonlinearity(k)
        out = attn.apply_linear_attention(q, k, v) if H * W > attn.attention_head_dim else attn.apply_quadratic_attention(q, k, v)
        out = attn.to_out(out.reshape(B, -1, H, W).movedim(1, -1)).movedim(-1, 1)
        if gate is not None:
            out = out * gate
        assert attn.norm_type == "rms_norm"
        out = attn.norm_out(out.movedim(1, -1)).movedim(-1, 1)
        return out + residual if attn.residual_connection else out


def test_foreign_dcae_attention_processor_is_called_not_ignored():
    """`attn.processor` of the DC-AE's SanaMultiscaleLinearAttention (models/DCAE.py:156,205-210; SURVEY 8(b) operator plug-points): an object that
    is not the built-in processor is CALLED with the reference's protocol `(attn, hidden_states, gate=gate_msa)` - round 5 refused it.
    (1) the reference-protocol processor above, installed on every attention block of the HIP autoencoder, reproduces the oracle and the fused path;
    (2) a processor with other arithmetic changes the result - it really ran; (3) the timestep-conditioned model hands it the AdaLN gate; (4) the
    split modes and hipGraph capture refuse it; (5) re-installing the built-in processor restores the fused path bit for bit."""
    from ladcast_amd.models.DCAE import SanaMultiscaleAttnProcessor2_0, SanaMultiscaleLinearAttention

    o, g = _pair(tiny_dcae_config())
    f, st = synth_field(2, 8, 48, 64), synth_field(1, 5, 48, 64, seed=1)
    with torch.no_grad():
        zo = o.encode(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        yo = o.decode(zo).sample
    fused_z = g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent
    fused_y = g.decode(zo.cuda()).sample
    attns = [m for m in g.modules() if isinstance(m, SanaMultiscaleLinearAttention)]
    assert len(attns) == 4 and all(a.foreign_processor is None for a in attns)  # 2 EfficientViT stages x 1 layer, encoder + decoder
    proc = _ReferenceProtocolProcessor()
    for a in attns:
        a.processor = proc  # the reference's plug-point: a plain attribute (models/DCAE.py:156)
    try:
        z = g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent
        y = g.decode(zo.cuda()).sample
        assert len(proc.seen) == 4 and all(s[1] is None and s[2] for s in proc.seen)
        assert sorted(s[0] for s in proc.seen) == sorted([(2, 64, 12, 16), (2, 128, 6, 8)] * 2)  # NCHW at the two EfficientViT stages
        ez, ey = rel_l2(z.cpu(), zo), rel_l2(y.cpu(), yo)
        print(f"\nforeign DC-AE attention processor vs oracle: encode {ez:.2e}, decode {ey:.2e}; vs fused path {rel_l2(z, fused_z):.2e} / {rel_l2(y, fused_y):.2e}")
        assert ez < 2e-5 and ey < 2e-5 and rel_l2(z, fused_z) < 2e-5 and rel_l2(y, fused_y) < 2e-5

        class DropAttention(_ReferenceProtocolProcessor):  # other arithmetic: the attention branch is dropped, only the residual passes
            def __call__(self, attn, hidden_states, gate=None):
                super().__call__(attn, hidden_states, gate)
                return hidden_states

        attns[0].processor = DropAttention()
        assert rel_l2(g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent, fused_z) > 1e-3
        attns[0].processor = proc
        g.set_gemm_precision("bf16x3")
        with pytest.raises(NotImplementedError):
            g.encode(f.cuda(), static_conditioning_tensor=st.cuda())
        g.set_gemm_precision("fp32")
        with pytest.raises(NotImplementedError):
            g.enable_hip_graph(True)
    finally:
        g.set_gemm_precision("fp32")
        for a in attns:
            a.processor = SanaMultiscaleAttnProcessor2_0()
    assert torch.equal(g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent, fused_z)
    g.enable_hip_graph(True)  # ... and graph capture works again
    assert torch.equal(g.encode(f.cuda(), static_conditioning_tensor=st.cuda()).latent, fused_z)
    g.enable_hip_graph(False)

    # the timestep-conditioned variant: the processor receives AdaLayerNormZeroSingle4Sana's gate as (B, C, 1, 1) and the NORMALISED tensor
    cfg = dict(tiny_dcae_config(), temb_channels=48)
    ot, gt = _pair(cfg)
    tt = torch.tensor([0.3, 1.7])
    from oracle.layers import get_timestep_embedding

    with torch.no_grad():
        emb = ot.timestep_embedder(get_timestep_embedding(tt, 256))
        zt = ot.encode(f, temb=emb, embedded_t=True, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
    fused_t = gt.encode(f.cuda(), temb=tt.cuda(), static_conditioning_tensor=st.cuda()).latent
    proc_t = _ReferenceProtocolProcessor()
    for a in (m for m in gt.modules() if isinstance(m, SanaMultiscaleLinearAttention)):
        a.processor = proc_t
    got_t = gt.encode(f.cuda(), temb=tt.cuda(), static_conditioning_tensor=st.cuda()).latent
    assert sorted(s[1] for s in proc_t.seen) == [(2, 64, 1, 1), (2, 128, 1, 1)]
    et = rel_l2(got_t.cpu(), zt)
    print(f"timestep-conditioned: foreign processor vs oracle {et:.2e}, vs fused {rel_l2(got_t, fused_t):.2e}")
    assert et < 2e-5 and rel_l2(got_t, fused_t) < 2e-5


def test_decoder_relu_stages_match_oracle():
    """`decoder_act_fns` per stage (models/DCAE.py:647,663-664,688: the activation of the decoder's ResBlocks): "relu" on the ResBlock stages, the
    rest silu - round 5 refused anything but "silu".  HIP decode against the oracle in the three arithmetic modes."""
    cfg = dict(tiny_dcae_config(), decoder_act_fns=("relu", "relu", "silu", "silu"))
    o, g = _pair(cfg)
    z = synth_field(2, 8, 6, 8, seed=5)
    with torch.no_grad():
        want = o.decode(z, return_static=True).sample
        assert rel_l2(make_dcae(tiny_dcae_config()).decode(z, return_static=True).sample, want) > 1e-2  # the activation really matters
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5), ("bf16", 2e-2)):
        g.set_gemm_precision(prec)
        e = rel_l2(g.decode(z.cuda(), return_static=True).sample.cpu(), want)
        print(f"\ndecoder with relu ResBlock stages [{prec}]: rel-L2 {e:.2e}")
        assert e < tol, (prec, e)
