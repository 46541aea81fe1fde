"""The DC-AE legs: BASELINE configs[0] (one 240 x 121 x 84 frame -> encode + decode; the grid is 120 x 240 after the reference's pole-row crop)
and the conv kernel's roofline row.  `dcae_block` is part of the default `--gpus 1` line, `dcae_workload` the stand-alone sweep
(`--workload dcae`)."""
import json
import os
import time

import torch

from .configs import CONFIG_DCAE_84, PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS

ENC_TFLOP, DEC_TFLOP = 0.6950, 0.7814  # algorithmic 2*MAC per frame
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _time_pair(g, x, st, n):
    """(encode ms, decode ms) per call: `n` back-to-back calls each, after one warm-up pair, inputs resident in HBM"""
    z = g.encode(x, static_conditioning_tensor=st).latent
    g.decode(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        z = g.encode(x, static_conditioning_tensor=st).latent
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        g.decode(z)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / n
    return te, td


def _row(te, td, frames):
    return dict(encode_ms=round(te * 1e3, 2), decode_ms=round(td * 1e3, 2), encode_tflops=round(ENC_TFLOP * frames / te, 1),
                decode_tflops=round(DEC_TFLOP * frames / td, 1))


class ConvTimer:
    """HIP-event bracket around every dense conv launch (ldc_sphere_conv_nhwc_split: 3 x 3 SphereConv2d and the 1 x 1 convs / Linears of
    the DC-AE) of one eager pass, keyed by shape"""

    def __init__(self, hip):
        self.hip, self.records = hip, {}
        self._orig = hip.sphere_conv_nhwc_split

    def __enter__(self):
        timer = self

        def conv(X, Wp, Y, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig(X, Wp, Y, **kw)
            e.record()
            k = kw.get("ksize", 3)
            key = (kw["B"], kw["H"], kw["W"], kw["cin"], kw["cout"], k, int(kw.get("in_fmt", timer.hip.FMT_SPLIT)))
            timer.records.setdefault(key, []).append((s, e))

        self.hip.sphere_conv_nhwc_split = conv
        return self

    def __exit__(self, *a):
        self.hip.sphere_conv_nhwc_split = self._orig

    def rows(self):
        out = []
        for (B, H, W, cin, cout, k, fmt), recs in self.records.items():
            us = [1e3 * s.elapsed_time(e) for s, e in recs]
            flops = 2.0 * B * H * W * cout * cin * k * k  # cin as launched (padded to the operand-row group)
            halo, bm, tw = self.hip.sphere_conv_plan(B, H, W, cin, cout, k, fmt) if k == 3 else (False, 0, 0)
            out.append(dict(shape=f"{cin}->{cout} {k}x{k} at {H}x{W} x{B}", launches=len(us), avg_launch_us=round(sum(us) / len(us), 2), total_us=round(sum(us), 1),
                            flops_per_launch=flops, kernel="conv_halo_kernel<256, 3>" if halo else "gemm_bf16x3_v3_kernel<*, 3, true>" if k > 1 else "gemm_bf16x3_v3_kernel<*, 3, true> (1x1)"))
        return sorted(out, key=lambda r: -r["total_us"])


def _conv_traffic(kernel):
    """fabric bytes per launch of the DC-AE's conv kernel from the committed PMC passes of tools/collect_dcae_pmc.sh (newest
    profiles/r*_dcae_pmc_summary_1frame.json), stamped with the kernel-source hash they were taken on"""
    import glob

    from ladcast_amd.build_id import csrc_sha16

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dcae_pmc_summary_1frame.json")))
    if not cands:
        return None, None, None
    try:
        doc = json.load(open(cands[-1]))
        ent = doc.get(kernel)
        if not ent:
            return None, None, None
        built = (doc.get("_build") or {}).get("csrc_sha16")
        src = (f"{os.path.relpath(cands[-1], ROOT)}: two separate rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) of tools/dcae_one.py 1, average over the "
               f"{ent['launches']} launches of this kernel in a one-frame encode + decode; kernel sources {built}, this run's {csrc_sha16()}; not re-measured in this run")
        return ent["hbm_bytes_per_launch"], src, built != csrc_sha16()
    except Exception:
        return None, None, None


def dcae_block(hip, dev, cpu_passes=3, frames_list=(1, 8)):
    """The `dcae` block of the default line: BASELINE configs[0] at full size on the MI355X - encode + decode ms per call at 1 and 8 frames in
    the three arithmetic modes (eager launches; one frame also through one hipGraph per direction, as the rollout runs it), the dominant
    conv's roofline row (HIP events around every conv launch of one instrumented one-frame bf16x3 pass), and configs[0]'s cpu_baseline."""
    from ladcast_amd.models import AutoencoderDC

    torch.manual_seed(1234)
    g = AutoencoderDC.from_config(CONFIG_DCAE_84).to(dev).eval()
    gen = torch.Generator().manual_seed(3)
    x8, st = torch.randn(max(frames_list), 84, 120, 240, generator=gen).to(dev), torch.randn(1, 5, 120, 240, generator=gen).to(dev)
    res = {"workload": "BASELINE configs[0]: DC_AE_84 (configs/DC_AE_84_pretrain.yaml) encode + decode of 84 x 120 x 240 frames + 5 static channels, random-init "
                       "seed 1234, inputs resident in HBM; ms per call", "tflop_per_frame": dict(encode=ENC_TFLOP, decode=DEC_TFLOP)}
    for prec in ("fp32", "bf16x3", "bf16"):
        g.set_gemm_precision(prec)
        mode = {}
        for frames in frames_list:
            te, td = _time_pair(g, x8[:frames], st, 3)
            mode[f"frames_{frames}"] = _row(te, td, frames)
        g.enable_hip_graph(True)
        te, td = _time_pair(g, x8[:1], st, 5)
        g.enable_hip_graph(False)
        mode["frames_1_graph"] = _row(te, td, 1)
        res[prec] = mode
    # the dominant conv launch, bf16x3 (the mode the decoded rollout uses), one frame
    g.set_gemm_precision("bf16x3")
    with ConvTimer(hip) as ct:
        z = g.encode(x8[:1], static_conditioning_tensor=st).latent
        g.decode(z)
        torch.cuda.synchronize()
    rows = ct.rows()
    if rows:
        top = rows[0]
        tf = top["flops_per_launch"] / (top["avg_launch_us"] * 1e-6) / 1e12
        traffic, src, stale = _conv_traffic(top["kernel"]) if top["kernel"].startswith("conv_halo") else (None, None, None)
        res["roofline"] = dict(bound="mfma", kernel=top["kernel"], shape=top["shape"], achieved=round(tf, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                               frac=round(tf / PEAK_BF16_MFMA_TFLOPS, 4), frac_of_attainable=round(3 * tf / PEAK_BF16_MFMA_TFLOPS, 4), launches=top["launches"],
                               avg_launch_us=top["avg_launch_us"], flops_per_launch=top["flops_per_launch"], traffic=traffic, traffic_source=src, traffic_stale=stale,
                               note="the conv shape with the most time in a one-frame bf16x3 encode + decode; HIP events around each launch (eager pass); split-bf16 issues 3 "
                                    "bf16 MFMAs per product: frac_of_attainable = achieved / (peak / 3).  traffic: average over ALL launches of this kernel in that pass",
                               conv_launches=rows[:6])
    g.set_gemm_precision("fp32")
    with ConvTimer(hip) as ct:
        z = g.encode(x8[:1], static_conditioning_tensor=st).latent
        g.decode(z)
        torch.cuda.synchronize()
    rows = ct.rows()
    if rows:
        top = rows[0]
        tf = top["flops_per_launch"] / (top["avg_launch_us"] * 1e-6) / 1e12
        res["roofline_fp32"] = dict(bound="mfma", kernel="gemm_bf16x3_v3_kernel<128, 0, true>", shape=top["shape"], achieved=round(tf, 1), peak=PEAK_F32_MFMA_TFLOPS,
                                    unit="TFLOP/s", frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4), launches=top["launches"], avg_launch_us=top["avg_launch_us"],
                                    flops_per_launch=top["flops_per_launch"])
    del g, x8
    torch.cuda.empty_cache()
    if cpu_passes > 0:
        from .cpu import dcae_baseline

        res["cpu_baseline"] = dcae_baseline(cpu_passes)
        gpu = res["fp32"]["frames_1_graph"]
        res["gpu_over_cpu_fp32"] = round((res["cpu_baseline"]["encode_ms"] + res["cpu_baseline"]["decode_ms"]) / (gpu["encode_ms"] + gpu["decode_ms"]), 1)
    return res


def dcae_workload(args):
    """`--workload dcae` (secondary sweep; the default line's `dcae` block carries the judged figures): full-size DCAE encode + decode for
    1 / 4 / 8 / 32 frames in every arithmetic mode.  Prints one JSON object."""
    from ladcast_amd.models import AutoencoderDC

    torch.manual_seed(1234)
    g = AutoencoderDC.from_config(CONFIG_DCAE_84).cuda().eval()
    res = {"workload": "DCAE (DC_AE_84_pretrain) encode + decode, 84 x 120 x 240 frames + 5 static channels, per precision mode (fp32 MFMA | bf16x3 split | bf16 single-term), random-init seed 1234"}
    for prec in ("fp32", "bf16x3", "bf16"):  # AutoencoderDC.set_gemm_precision: exact fp32 | split-bf16 | single-term bf16 convs / Linears
        g.set_gemm_precision(prec)
        for frames in (1, 4, 8, 32):
            x = torch.randn(frames, 84, 120, 240, device="cuda")
            st = torch.randn(1, 5, 120, 240, device="cuda")
            te, td = _time_pair(g, x, st, 3)
            res[f"gpu_{prec}_{frames}"] = _row(te, td, frames)
            if prec == "bf16x3" and frames <= g.GRAPH_MAX_FRAMES:  # the same through one hipGraph per direction (launch-bound sizes)
                g.enable_hip_graph(True)
                te, td = _time_pair(g, x, st, 5)
                g.enable_hip_graph(False)
                res[f"gpu_{prec}_graph_{frames}"] = _row(te, td, frames)
    if args.cpu_forwards > 0:
        from .cpu import dcae_baseline

        res["cpu_baseline"] = dcae_baseline(1)
    print(json.dumps(res))
