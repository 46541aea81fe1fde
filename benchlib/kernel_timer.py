"""HIP-event brackets around the C-ABI GEMM / attention calls of one instrumented step (the `roofline` block), and the live calibration
of what such a bracket adds per launch."""
import os
import statistics

import torch


class KernelTimer:
    """HIP-event bracket around every launch of selected C-ABI kernels on torch's current stream
    (the stream the kernels are launched on)."""

    WRAPPED = ("gemm", "attn_fwd", "gemm_grouped", "gemm_grouped_qkv", "gemm_grouped_qkv_f32", "attn_fwd_split")

    def __init__(self):
        self.records = {}  # name -> list of (start, end, work)

    def _bracket(self, name, work, fn, *a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **kw)
        e.record()
        self.records.setdefault(name, []).append((s, e, work))
        return out

    def install(self, hip):
        self._hip = hip
        self._orig = {k: getattr(hip, k) for k in self.WRAPPED}
        timer, orig = self, self._orig
        flops = lambda problems: sum(2.0 * p[0].d.M * p[0].d.N * p[0].d.K * p[0].d.batch for p in problems)  # noqa: E731

        def gemm_grouped(problems, split_bf16=False):
            # which kernel the C ABI dispatches to (gemm_streamk.hip, ldc_gemm_grouped_bf16x3): pre-split activations and K % 32 == 0 -> the
            # 16x16x32 ring kernel; exact fp32: the ring kernel's TERMS = 0 instance when K % 32 == 0, else the register-staged stream-K kernel
            if not split_bf16:
                ring = all(p[0].d.K % 32 == 0 and p[0].d.ldw == p[0].d.K for p in problems) and os.environ.get("LDC_F32_RING", "1") != "0"
                name = "gemm_bf16x3_v3_kernel<128, 0, false>" if ring else "gemm_streamk_kernel"
            elif all(p[0].d.K % 32 == 0 for p in problems):
                name = timer.v3_variant(problems) if all(p[0].d.flags & 1 for p in problems) else "gemm_streamk_bf16x3_kernel"
            else:
                name = "gemm_streamk_bf16x3_kernel"
            return timer._bracket(name, flops(problems), orig["gemm_grouped"], problems, split_bf16=split_bf16)

        def gemm(A, W, C, **kw):
            return timer._bracket("gemm_nt_f32_kernel", 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("batch", 1), orig["gemm"], A, W, C, **kw)

        def attn_fwd(Q, K, V, O, **kw):
            return timer._bracket("attn_fwd_f32_kernel", 4.0 * kw["B"] * kw["H"] * kw["S"] * kw["S"] * 128, orig["attn_fwd"], Q, K, V, O, **kw)

        def gemm_grouped_qkv(problems, epilogues, **kw):  # QKV projections with the attention-operand epilogue: the same kernel, same FLOPs
            return timer._bracket(timer.v3_variant(problems), flops(problems), orig["gemm_grouped_qkv"], problems, epilogues, **kw)

        def gemm_grouped_qkv_f32(problems, epilogues):  # the exact-fp32 QKV projection (RMSNorm + rotary epilogue): the TERMS = 0 ring instance
            return timer._bracket("gemm_bf16x3_v3_kernel<128, 0, false>", flops(problems), orig["gemm_grouped_qkv_f32"], problems, epilogues)

        def attn_fwd_split(Q, K, V, O, **kw):
            return timer._bracket("attn_fwd_split_kernel", 4.0 * kw["B"] * kw["H"] * kw["S"] * kw["S"] * 128, orig["attn_fwd_split"], Q, K, V, O, **kw)

        for k, fn in (("gemm", gemm), ("attn_fwd", attn_fwd), ("gemm_grouped", gemm_grouped), ("gemm_grouped_qkv", gemm_grouped_qkv),
                      ("gemm_grouped_qkv_f32", gemm_grouped_qkv_f32), ("attn_fwd_split", attn_fwd_split)):
            setattr(hip, k, fn)

    @staticmethod
    def v3_variant(problems):
        """the template instance gemm_v3_dispatch (csrc/gemm_bf16x3_v3.hip) launches for a grouped call - the kernel name rocprofv3 prints:
        128-row tiles while 256-row tiles would number fewer than 400, TERMS = 1 in the single-term bf16 mode"""
        t256 = sum(p[0].d.batch * -(-p[0].d.M // 256) * -(-p[0].d.N // 128) for p in problems)
        terms = 1 if (problems[0][0].d.flags & 4) else 3
        return f"gemm_bf16x3_v3_kernel<{128 if t256 < 400 else 256}, {terms}, false>"

    def uninstall(self):
        for k, fn in self._orig.items():
            setattr(self._hip, k, fn)

    def clear(self):
        self.records = {}

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            work = sum(w for _, _, w in recs)
            out[name] = dict(launches=len(recs), total_ms=ms, avg_us=1e3 * ms / len(recs), work_per_launch=work / len(recs),
                             tflops=work / (ms * 1e-3) / 1e12)
        return out


def bracket_overhead(hip, dev):
    """What a HIP-event pair adds around one launch on a busy stream, measured live: the bracket around a one-element kernel that is
    queued behind a long one (median of 20), minus what the same launch costs un-bracketed inside a back-to-back chain (200 launches
    replayed from one hipGraph between two events: kernel + the ~1.5 us dependent-launch boundary).  The instrumented step's averages
    contain this per launch: `roofline.achieved` is the MEASURED figure (with it), `achieved_net` the modelled one without it."""
    big, tiny, vals = torch.empty(64 << 20, device=dev), torch.empty(1, device=dev), []
    for _ in range(20):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.scale_f32(tiny, 1.0, tiny)
        e1.record()
        torch.cuda.synchronize()
        vals.append(1e3 * e0.elapsed_time(e1))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        hip.scale_f32(tiny, 1.0, tiny)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for _ in range(200):
            hip.scale_f32(tiny, 1.0, tiny)
    chain = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        chain.append(1e3 * e0.elapsed_time(e1) / 200)
    bracket, in_chain = statistics.median(vals), statistics.median(chain)
    return round(max(0.0, bracket - in_chain), 2), round(bracket, 2), round(in_chain, 2)


def rocprof_averages(root, pattern):
    """per kernel-template instance: (AverageNs in us, calls) from the newest committed `rocprofv3 --kernel-trace --stats` summary that
    matches `pattern` under profiles/; ({}, None) when there is none"""
    import csv
    import glob

    stats = sorted(glob.glob(os.path.join(root, "profiles", pattern)))
    prof = {}
    if not stats:
        return prof, None
    try:
        for row in csv.DictReader(open(stats[-1])):
            nm = row.get("Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
            prof[nm.split("(")[0].strip()] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    except Exception:
        return {}, None
    return prof, os.path.basename(stats[-1])


def roofline_rows(ks, dom, peak, bracket_us, prof_avg):
    """one row per template instance of the dominant kernel: launches, exact flops per launch, the MEASURED HIP-event average
    (`achieved` / `frac`), the same net of the calibrated bracket overhead (`achieved_net`; skipped when the launch is not well above the
    bracket) and - where the committed rocprofv3 summary holds that instance - its AverageNs there (`achieved_rocprof`)"""
    rows = []
    for vn in sorted(n for n in ks if n.startswith(dom.split("<")[0])):
        kv = ks[vn]
        row = dict(kernel=vn, launches=kv["launches"], avg_launch_us=round(kv["avg_us"], 2), flops_per_launch=kv["work_per_launch"],
                   achieved=round(kv["tflops"], 2), frac=round(kv["tflops"] / peak, 4))
        if bracket_us is not None and kv["avg_us"] > 4.0 * bracket_us:
            net = kv["avg_us"] - bracket_us
            row["avg_launch_us_net"] = round(net, 2)
            row["achieved_net"] = round(kv["work_per_launch"] / (net * 1e-6) / 1e12, 2)
        if vn in prof_avg:
            row["rocprof_avg_us"] = round(prof_avg[vn][0], 2)
            row["achieved_rocprof"] = round(kv["work_per_launch"] / (prof_avg[vn][0] * 1e-6) / 1e12, 2)
        rows.append(row)
    return rows
