#!/usr/bin/env python3
"""Headline benchmark: ensemble-member-steps/sec of the LaDCast AR rollout on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input on every rank: a full
autoregressive sampler chunk (BASELINE.json configs[1]: 375M AR transformer, 1 ensemble member per
GPU, 20 solver steps, 1 lead step) -- IC latent resident in HBM in, lead-step latents out.  With
the shipped return_seq_len = 4 the reference computes one 4-frame chunk for a 1-lead-step request and
keeps ``pred_selection = 1`` frame (pipelines/utils.py:535-536); the metric counts that 1 lead step.
Weak scaling (default): every rank owns ``--members-per-gpu`` members (global ids {k : k mod N == rank}, seeded by
member id as pipelines/utils.py:703-706), no data-path collective except one RCCL gather of the
result latents at the end of each step.  Strong scaling: ``--ensemble-size E`` fixes the ensemble at E members
whatever N is and deals them to the ranks (same rule; ranks may own different counts, or none) - BASELINE
configs[2] is ``--ensemble-size 16 --lead-steps 40``, the workload of north_star's ">= 6x at 8 GPUs vs 1";
``--ensemble-size 1`` at N = 1 is the default line.

Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIGS = {
    "375M": dict(
        in_channels=84, out_channels=84, num_attention_heads=12, attention_head_dim=128, num_layers=2, num_single_layers=4,
        num_refiner_layers=1, mlp_ratio=4, patch_size=1, patch_size_t=1, qk_norm="rms_norm", rope_theta=256.0, rope_axes_dim=(16, 56, 56),
        rope_spatial_grid_start_pos=(-499.5, 5.25), rope_spatial_grid_end_pos=(508.5, 353.25), spatial_deg2rad=True,
        conditioning_tensor_in_channels=84, conditioning_tensor_rope_axes_dim=(16, 56, 56), incl_time_elapsed=True,
    ),  # configs/ladcast_375M.yaml:1-30
}
CONFIGS["1.6B"] = dict(CONFIGS["375M"], num_attention_heads=16, num_layers=5, num_single_layers=10, num_refiner_layers=3)
CONFIG_DCAE_84 = dict(  # configs/DC_AE_84_pretrain.yaml:1-48
    in_channels=89, out_channels=89, latent_channels=84, attention_head_dim=32,
    encoder_block_types=("ResBlock", "ResBlock", "EfficientViTBlock", "EfficientViTBlock"),
    decoder_block_types=("ResBlock", "ResBlock", "EfficientViTBlock", "EfficientViTBlock"),
    encoder_block_out_channels=(252, 504, 504, 1008), decoder_block_out_channels=(252, 504, 504, 1008),
    encoder_layers_per_block=(4, 4, 4, 4), decoder_layers_per_block=(4, 4, 4, 4),
    encoder_qkv_multiscales=((), (), (5,), (5,)), decoder_qkv_multiscales=((), (), (5,), (5,)),
    upsample_block_type="pixel_shuffle", downsample_block_type="pixel_unshuffle", static_channels=5,
)


def dcae_workload(args):
    """`--workload dcae` (BASELINE configs[0] / the decode leg of configs[4]; not the headline): full-size DCAE encode +
    decode of 240x120x84 frames on the MI355X (HIP, NHWC) for 1 / 8 / 32 frames, and -- cpu_baseline leg -- one frame with
    the CPU oracle.  Prints one JSON object."""
    from ladcast_amd.models import AutoencoderDC

    torch.manual_seed(1234)
    g = AutoencoderDC.from_config(CONFIG_DCAE_84).cuda().eval()
    res = {"workload": "DCAE (DC_AE_84_pretrain) encode + decode, 84 x 120 x 240 frames + 5 static channels, per precision mode (fp32 MFMA | bf16x3 split | bf16 single-term), random-init seed 1234"}
    for prec in ("fp32", "bf16x3", "bf16"):  # AutoencoderDC.set_gemm_precision: exact fp32 | split-bf16 | single-term bf16 convs / Linears
        g.set_gemm_precision(prec)
        for frames in (1, 4, 8, 32):
            x = torch.randn(frames, 84, 120, 240, device="cuda"); st = torch.randn(1, 5, 120, 240, device="cuda")
            z = g.encode(x, static_conditioning_tensor=st).latent; g.decode(z); torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 3
            for _ in range(n): z = g.encode(x, static_conditioning_tensor=st).latent
            torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n): g.decode(z)
            torch.cuda.synchronize(); td = (time.perf_counter() - t0) / n
            res[f"gpu_{prec}_{frames}"] = dict(encode_ms=round(te * 1e3, 2), decode_ms=round(td * 1e3, 2), encode_tflops=round(0.695 * frames / te, 1),
                                               decode_tflops=round(0.7814 * frames / td, 1))
            if prec == "bf16x3" and frames <= g.GRAPH_MAX_FRAMES:  # the same through one hipGraph per direction (launch-bound sizes)
                g.enable_hip_graph(True)
                z = g.encode(x, static_conditioning_tensor=st).latent; g.decode(z); torch.cuda.synchronize()
                t0 = time.perf_counter(); n = 5
                for _ in range(n): z = g.encode(x, static_conditioning_tensor=st).latent
                torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
                t0 = time.perf_counter()
                for _ in range(n): g.decode(z)
                torch.cuda.synchronize(); td = (time.perf_counter() - t0) / n
                g.enable_hip_graph(False)
                res[f"gpu_{prec}_graph_{frames}"] = dict(encode_ms=round(te * 1e3, 2), decode_ms=round(td * 1e3, 2), encode_tflops=round(0.695 * frames / te, 1),
                                                         decode_tflops=round(0.7814 * frames / td, 1))
    if args.cpu_forwards > 0:
        from oracle.dcae import AutoencoderDC as OracleAE  # cpu_baseline leg only

        o = OracleAE.from_config(CONFIG_DCAE_84).eval()
        x = torch.randn(1, 84, 120, 240); st = torch.randn(1, 5, 120, 240)
        with torch.no_grad():
            o.encode(x, static_conditioning_tensor=st)
            t0 = time.perf_counter(); z = o.encode(x, static_conditioning_tensor=st).latent; te = time.perf_counter() - t0
            t0 = time.perf_counter(); o.decode(z); td = time.perf_counter() - t0
        res["cpu_baseline"] = dict(encode_ms=round(te * 1e3, 1), decode_ms=round(td * 1e3, 1), cores=torch.get_num_threads(), kind="port",
                                   sample="1 frame by the PyTorch CPU oracle")
    print(json.dumps(res))

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (the 2:1-sparsity figure is never used)


def model_flops_per_forward(cfg, R, T_in=1, hw=450, split=False):
    """Analytic 2*MAC count of one forward per member; returns (gemm_flops, attn_flops), or with `split` the pair
    ((gemm, attn) of the sample-dependent part, (gemm, attn) of the conditioning path: context embed + token refiner - the part a
    sampler chunk evaluates once per noise level instead of once per network evaluation, LaDCastTransformer3DModel.prepare_conditioning)."""
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    Nx, Nc = R * hw, T_in * hw
    S = Nx + Nc
    F = int(D * cfg["mlp_ratio"])
    lin = lambda m, n, k: 2.0 * m * n * k  # noqa: E731
    cg = lin(Nc, D, 84) + lin(Nc, D, D)  # context embed + refiner proj_in
    cg += cfg["num_refiner_layers"] * (lin(Nc, 3 * D, D) + lin(Nc, F, D) + lin(Nc, D, F))
    ca = cfg["num_refiner_layers"] * 4.0 * Nc * Nc * D
    g = lin(Nx, D, 84)  # sample embed
    g += cfg["num_layers"] * (lin(S, 3 * D, D) + lin(S, D, D) + lin(S, F, D) + lin(S, D, F))
    g += cfg["num_single_layers"] * (lin(S, 3 * D, D) + lin(S, F, D) + lin(S, D, D + F))
    g += lin(Nx, 84, D)
    a = (cfg["num_layers"] + cfg["num_single_layers"]) * 4.0 * S * S * D
    if split:
        return (g, a), (cg, ca)
    return g + cg, a + ca


def gpu_state(dev_index):
    """Clock / power / temperature of this rank's GPU from sysfs (hwmon of the device's PCI function; readable as an ordinary user):
    sampled right before and right after the timed region so that two runs whose `value` differs can be told apart by what the
    chip was doing (DVFS: MI355X_MICROARCH.md, 'DVFS give-back' - devices differ by up to 12 % on MFMA-dense loops).  The read
    happens outside the timed region; a missing file leaves its field out."""
    import glob

    out = {}
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        out["pci_bus_id"] = bdf
        out["name"] = pr.name
        out["gcn_arch"] = getattr(pr, "gcnArchName", None)
        out["uuid"] = str(getattr(pr, "uuid", ""))
        base = f"/sys/bus/pci/devices/{bdf}"

        def rd(path):
            try:
                with open(path) as f:
                    return f.read().strip()
            except OSError:
                return None

        for hw in glob.glob(base + "/hwmon/hwmon*"):
            for key, fn, scale in (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_input", 1e-6),
                                   ("power_w", "power1_average", 1e-6), ("power_cap_w", "power1_cap", 1e-6), ("temp_junction_c", "temp2_input", 1e-3),
                                   ("temp_edge_c", "temp1_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
                v = rd(f"{hw}/{fn}")
                if v is not None and key not in out:
                    try:
                        out[key] = round(float(v) * scale, 1)
                    except ValueError:
                        pass
        lv = rd(base + "/pp_dpm_sclk")
        if lv:
            cur = [ln for ln in lv.splitlines() if ln.strip().endswith("*")]
            out["pp_dpm_sclk"] = cur[0].strip() if cur else None
    except Exception as e:  # diagnostics only: never fail the bench over it
        out["error"] = repr(e)
    return out


class GpuStateSampler:
    """Background thread: `gpu_state` every `period` seconds while the timed region runs (a sysfs read on the host; the launching thread
    is asleep inside graph replays / the final synchronise, so the GPU never waits for it).  `summary()`: min / median / max of clock and
    power, max temperature, number of samples."""

    def __init__(self, dev_index, period=0.05):
        import threading

        self.dev_index, self.period, self.samples = dev_index, period, []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(gpu_state(self.dev_index))
            self._stop.wait(self.period)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=2.0)
        return self.summary()

    def summary(self):
        import statistics

        out = {"samples": len(self.samples)}
        for key in ("sclk_mhz", "power_w"):
            v = [x[key] for x in self.samples if key in x]
            if v:
                out[key] = dict(min=min(v), median=round(statistics.median(v), 1), max=max(v))
        for key in ("temp_junction_c", "temp_mem_c", "temp_edge_c"):
            v = [x[key] for x in self.samples if key in x]
            if v:
                out[key + "_max"] = max(v)
        return out


def host_description():
    """CPU model string, sockets x physical cores, threads (lscpu): printed with cpu_baseline."""
    import subprocess

    d = {"threads_online": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for ln in txt.splitlines():
            if ":" in ln:
                k, v = ln.split(":", 1)
                kv[k.strip()] = v.strip()
        d["model"] = kv.get("Model name")
        sockets, cps = int(kv.get("Socket(s)", "0") or 0), int(kv.get("Core(s) per socket", "0") or 0)
        d["sockets"], d["physical_cores"] = sockets, sockets * cps
        d["threads_per_core"] = int(kv.get("Thread(s) per core", "0") or 0)
    except Exception as e:
        d["error"] = repr(e)
    return d


class KernelTimer:
    """HIP-event bracket around every launch of selected C-ABI kernels on torch's current stream
    (the stream the kernels are launched on)."""

    def __init__(self):
        self.records = {}  # name -> list of (start, end, work)

    def install(self, hip):
        self._hip = hip
        self._orig = {"gemm": hip.gemm, "attn_fwd": hip.attn_fwd, "gemm_grouped": hip.gemm_grouped,
                      "gemm_grouped_qkv": hip.gemm_grouped_qkv, "attn_fwd_split": hip.attn_fwd_split}
        timer = self

        def gemm_grouped(problems, split_bf16=False):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig["gemm_grouped"](problems, split_bf16=split_bf16)
            e.record()
            work = sum(2.0 * p[0].d.M * p[0].d.N * p[0].d.K * p[0].d.batch for p in problems)
            # which kernel the C ABI dispatches to (gemm_streamk.hip, ldc_gemm_grouped_bf16x3): pre-split activations and
            # K % 32 == 0 -> the 16x16x32 kernel; fp32 activations and K % 32 == 0 -> the 32x32x16 LDS-DMA kernel
            if not split_bf16:  # exact fp32: the ring kernel (gemm_bf16x3_v3.hip, TERMS = 0) when K % 32 == 0, else the register-staged one
                ring = all(p[0].d.K % 32 == 0 and p[0].d.ldw == p[0].d.K for p in problems) and os.environ.get("LDC_F32_RING", "1") != "0"
                name = "gemm_bf16x3_v3_kernel<128, 0, false>" if ring else "gemm_streamk_kernel"
            elif all(p[0].d.K % 32 == 0 for p in problems):
                name = timer.v3_variant(problems) if all(p[0].d.flags & 1 for p in problems) else "gemm_bf16x3_dma_kernel"
            else:
                name = "gemm_streamk_bf16x3_kernel"
            timer.records.setdefault(name, []).append((s, e, work))

        def gemm(A, W, C, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig["gemm"](A, W, C, **kw)
            e.record()
            timer.records.setdefault("gemm_nt_f32_kernel", []).append((s, e, 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("batch", 1)))

        def attn_fwd(Q, K, V, O, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig["attn_fwd"](Q, K, V, O, **kw)
            e.record()
            timer.records.setdefault("attn_fwd_f32_kernel", []).append((s, e, 4.0 * kw["B"] * kw["H"] * kw["S"] * kw["S"] * 128))

        def gemm_grouped_qkv(problems, epilogues):  # QKV projections with the attention-operand epilogue: the same kernel, same FLOPs
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig["gemm_grouped_qkv"](problems, epilogues)
            e.record()
            work = sum(2.0 * p[0].d.M * p[0].d.N * p[0].d.K * p[0].d.batch for p in problems)
            timer.records.setdefault(timer.v3_variant(problems), []).append((s, e, work))

        def attn_fwd_split(Q, K, V, O, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig["attn_fwd_split"](Q, K, V, O, **kw)
            e.record()
            timer.records.setdefault("attn_fwd_split_kernel", []).append((s, e, 4.0 * kw["B"] * kw["H"] * kw["S"] * kw["S"] * 128))

        hip.gemm, hip.attn_fwd, hip.gemm_grouped = gemm, attn_fwd, gemm_grouped
        hip.gemm_grouped_qkv, hip.attn_fwd_split = gemm_grouped_qkv, attn_fwd_split

    @staticmethod
    def v3_variant(problems):
        """the template instance gemm_v3_dispatch (csrc/gemm_bf16x3_v3.hip) launches for a grouped call - the kernel name rocprofv3 prints:
        128-row tiles while 256-row tiles would number fewer than 400, TERMS = 1 in the single-term bf16 mode"""
        t256 = sum(p[0].d.batch * -(-p[0].d.M // 256) * -(-p[0].d.N // 128) for p in problems)
        terms = 1 if (problems[0][0].d.flags & 4) else 3
        return f"gemm_bf16x3_v3_kernel<{128 if t256 < 400 else 256}, {terms}, false>"

    def uninstall(self):
        self._hip.gemm, self._hip.attn_fwd, self._hip.gemm_grouped = self._orig["gemm"], self._orig["attn_fwd"], self._orig["gemm_grouped"]
        self._hip.gemm_grouped_qkv, self._hip.attn_fwd_split = self._orig["gemm_grouped_qkv"], self._orig["attn_fwd_split"]

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


def cpu_baseline(cfg_name, R, n_forwards):
    """Time the CPU oracle (kind "port": the reference's own path needs diffusers, absent here) on a
    bounded sample: `n_forwards` model forwards of the same workload, then scale to one sampler chunk."""
    from oracle.ar_model import LaDCastTransformer3DModel as OracleModel

    torch.manual_seed(1234)
    m = OracleModel.from_config(CONFIGS[cfg_name]).eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 84, R, 15, 30, generator=g)
    known = 0.5 * torch.randn(1, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))
    ts = torch.tensor([2018010100])
    times = []
    with torch.no_grad():
        m(x, torch.tensor([0.5]), known, time_elapsed=ts)  # warm-up (page-in, thread pool)
        for i in range(n_forwards):
            t0 = time.perf_counter()
            m(x, torch.tensor([0.5 - 0.05 * i]), known, time_elapsed=ts)
            times.append(time.perf_counter() - t0)
    return times


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process has touched no GPU API (importing torch
    does not), so it starts N fresh rank processes through `python -m torch.distributed.run` as a CHILD, waits, and exits
    with its code - never an exec from a process that has initialised the GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "8")  # torchrun's default of 1 would throttle nothing here but the noise draws
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        print(f"bench.py: the {n}-rank launch exited with code {rc}", file=sys.stderr)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="375M", choices=list(CONFIGS))
    ap.add_argument("--members-per-gpu", type=int, default=1)
    ap.add_argument("--ensemble-size", type=int, default=0, help="STRONG scaling: a fixed ensemble of E members dealt to the N ranks (member k on rank "
                    "k mod N; evaluate/pred_rollout.py:349-358,398-400 is the split / gather it replaces); 0 = weak scaling with --members-per-gpu")
    ap.add_argument("--return-seq-len", type=int, default=4)
    ap.add_argument("--lead-steps", type=int, default=1, help="lead steps requested per rollout call (1 chunk covers up to return_seq_len)")
    ap.add_argument("--solver-steps", type=int, default=20)
    ap.add_argument("--sampler", default="edm", choices=["edm", "pipeline", "ddim"], help="edm = Heun, 2N-1 forwards (reference default); pipeline = the "
                    "reference's scheduler loop with its EDM DPM-Solver++(2M) scheduler, N forwards; ddim = the same loop with ladcast_amd.schedulers.DDIMScheduler "
                    "(diffusers defaults, eta = 0), N forwards - BASELINE's literal '20-step DDIM'")
    ap.add_argument("--dump-output", default=None, help="rank 0 saves the (gathered) result tensor of the LAST timed step to this .pt file (tests)")
    ap.add_argument("--cpu-forwards", type=int, default=5, help="oracle forwards timed for cpu_baseline after one warm-up forward; the MEDIAN is reported (0 = skip)")
    ap.add_argument("--strong-cfg3", action="store_true", help="also run BASELINE configs[2] (a fixed ensemble of 16 members, 40 lead steps = 10 chained chunks per member) "
                    "as an extra leg and print it as `strong_cfg3` - always on with N > 1, where it is north_star's strong-scaling point; ~17 s per call at N = 1")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend; nccl (= RCCL over xGMI) is the real one, "
                    "gloo only lets the N > 1 code path be dry-run on a box with fewer GPUs than ranks")
    ap.add_argument("--no-strong-cfg3", action="store_true", help="skip the strong_cfg3 leg that N > 1 runs carry by default (tests of other fields)")
    ap.add_argument("--collective-timeout", type=float, default=300.0, help="seconds before a rendezvous / collective is declared dead (the rank then exits non-zero)")
    ap.add_argument("--share-gpus", action="store_true", help="dry-run aid: map ranks onto the available GPUs modulo their count")
    ap.add_argument("--host-outputs", action="store_true", help="A/B aid: return every step's latents on the host as the reference's roll_out_serial does (one device "
                    "-> host copy + synchronise per step) instead of leaving them in HBM")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying one captured hipGraph per model forward "
                    "(graph replay keeps the GPU fed regardless of host speed; kernels and numerics are identical)")
    ap.add_argument("--decode", action="store_true", help="secondary workload (BASELINE configs[4]): encode the IC field with the DCAE, roll out, DECODE "
                    "every lead step to 84 x 120 x 240 fields (DCAE in bf16x3 mode unless --precision fp32); not the headline")
    ap.add_argument("--workload", default="rollout", choices=["rollout", "dcae"], help="rollout = the headline (AR sampler chunk); dcae = DCAE "
                    "encode / decode timing (BASELINE configs[0], secondary)")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="token-stream GEMM / attention arithmetic: exact fp32 MFMA; split-bf16 (hi*hi+hi*lo+lo*hi, fp32 accumulate; the headline, "
                         "inside the 1e-4 budget); bf16 = ONE bf16 MFMA per product, the mixed-precision mode of BASELINE configs[4] (own tolerance)")
    ap.add_argument("--no-batched-conditioning", action="store_true", help="A/B aid: evaluate the sample-independent part of the network (context refiner, conditioning "
                    "embedding, AdaLN modulation vectors) inside every network evaluation as the reference does, instead of once per chunk as one batch over the "
                    "chunk's noise levels (LaDCastTransformer3DModel.prepare_conditioning)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="after the K timed steps keep stepping until this many seconds of "
                    "back-to-back chunks have run and report that window as `sustained` (clock under sustained load); 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    if args.workload == "dcae":
        return dcae_workload(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    dev_index = local_rank % torch.cuda.device_count() if args.share_gpus else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        from datetime import timedelta

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # never a hang: rendezvous and every collective carry a timeout (the RCCL watchdog aborts the rank when one expires), and a failure
        # to bring the communicator up ends the rank with the library's own error text and a non-zero exit code
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=args.collective_timeout))  # RCCL over xGMI
            else:
                dist.init_process_group("gloo", timeout=timedelta(seconds=args.collective_timeout))
            probe = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(probe)  # the first collective builds the communicator: fail here, with a message, rather than inside the timed region
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
        except Exception as e:  # noqa: BLE001
            print(f"bench.py rank {rank}: cannot bring up the {args.backend} process group over {world} ranks: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            sys.exit(3)

    import ladcast_amd.hip as hip
    from ladcast_amd.models import LaDCastTransformer3DModel
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from datetime import datetime

    cfg = CONFIGS[args.model]
    torch.manual_seed(1234)
    model = LaDCastTransformer3DModel.from_config(cfg).to(dev).eval().set_gemm_precision(args.precision)
    model.enable_hip_graph(not args.no_graph)
    model.batch_conditioning = not args.no_batched_conditioning
    from ladcast_amd.schedulers import DDIMScheduler

    pipes = {"edm": AutoRegressive2DPipeline(model, EDMDPMSolverMultistepScheduler()), "ddim": AutoRegressive2DPipeline(model, DDIMScheduler())}
    pipes["pipeline"] = pipes["edm"]
    pipe = pipes[args.sampler]
    sampler_type = "edm" if args.sampler == "edm" else "pipeline"  # roll_out_serial's switch: Heun sampler | the pipeline's scheduler loop
    from ladcast_amd.pipelines.distributed import shard_members
    from ladcast_amd.precision import tolerance

    strong = args.ensemble_size > 0
    total_members = args.ensemble_size if strong else args.members_per_gpu * world
    member_ids = shard_members(total_members, rank, world)  # rank r owns members {k : k mod world == r}
    m = len(member_ids)  # this rank's members: --members-per-gpu (weak) or its share of the fixed ensemble (strong; may be 0)
    R, lead = args.return_seq_len, args.lead_steps
    ic = (0.5 * torch.randn(84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).to(dev)  # IC latent, resident in HBM
    # results stay in HBM (with N > 1 the one collective gathers them there): no host copy and therefore no host stall per step - the host
    # prepares step k + 1 (noise draw, timestamps) while the GPU still runs step k; the timed region ends with a synchronise as the contract says
    out_dev = None if args.host_outputs else dev
    targs = {"mean": [0.0] * 84, "std": [1.0] * 84, "target_std": 0.5}
    from ladcast_amd.pipelines.distributed import gather_members

    dec_kw = {}
    if args.decode:  # end-to-end: IC field -> encode -> AR chunks -> decode (roll_out_serial's decoded-field mode)
        from ladcast_amd.models import AutoencoderDC

        ae = AutoencoderDC.from_config(CONFIG_DCAE_84).to(dev).eval().set_gemm_precision(args.precision).enable_hip_graph(not args.no_graph)
        g_ = torch.Generator().manual_seed(3)
        field = torch.randn(84, 1, 120, 240, generator=g_)
        static = torch.randn(5, 120, 240, generator=g_)
        dec_kw = dict(encdec_model=ae, encdec_model_type="ae", static_tensor4encdec=static,
                      normalization_param_dict={"mean": torch.zeros(84), "std": torch.ones(84)})

        def step_decode():
            return roll_out_serial(
                lambda t: field, [datetime(2018, 1, 1, 0)], pipe, ensemble_size=m, num_inference_steps=args.solver_steps, return_seq_len=R,
                latent_transform_args=targs, total_lead_time_hour=6 * lead, sampler_type=sampler_type, return_latent=False,
                member_ids=member_ids, output_device=out_dev, **dec_kw)

    def step_local(od=out_dev):
        return roll_out_serial(
            None, [datetime(2018, 1, 1, 0)], pipe, ensemble_size=m, num_inference_steps=args.solver_steps, return_seq_len=R,
            latent_transform_args=targs, total_lead_time_hour=6 * lead, sampler_type=sampler_type, return_latent=True,
            known_latents_override=ic, member_ids=member_ids, output_device=od,
        )

    phase = []  # per step of this rank: (event before the rollout, after it = before the gather, after the gather) + host times around the gather

    def step():
        if args.decode:
            return step_decode()
        if world > 1:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        out = step_local()
        if world > 1:  # the one collective of the path: gather the per-rank latents (evaluate/pred_rollout.py:398-400)
            e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e1.record()
            h0 = time.perf_counter()
            out = gather_members(out.to(dev) if args.backend == "nccl" else out.to("cpu"), total_members, member_dim=1)
            h1 = time.perf_counter()
            e2.record()
            phase.append((e0, e1, e2, h1 - h0))
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()  # set-up, not a warm-up step: packs the weights and captures the chunk graph (like model construction, it is never timed)
    for _ in range(args.warmup):
        step()
    timer = KernelTimer()
    state_before = gpu_state(dev_index)
    fence()
    sampler = GpuStateSampler(dev_index).start()
    t0 = time.perf_counter()
    marks, evs = [], [torch.cuda.Event(enable_timing=True)]
    evs[0].record()
    phase.clear()
    last_out = None
    for _ in range(args.steps):
        last_out = step()
        marks.append(time.perf_counter())  # --host-outputs: every step returns host tensors, i.e. is complete here
        evs.append(torch.cuda.Event(enable_timing=True))
        evs[-1].record()  # device outputs: the host runs ahead; the step boundaries are read from the stream afterwards
    fence()
    elapsed = time.perf_counter() - t0
    gpu_during = sampler.stop()
    state_after = gpu_state(dev_index)
    if out_dev is None:
        step_ms = [round(1e3 * (b - a), 2) for a, b in zip([t0] + marks[:-1], marks)]  # diagnostic only (rank 0's view)
    else:
        step_ms = [round(a.elapsed_time(b), 2) for a, b in zip(evs[:-1], evs[1:])]
    rank_stats = None
    if world > 1:
        # per-rank diagnostics (a SCALE run must be readable from the one line): this rank's own wall time, its rollout time per step
        # (events on the stream: start of the step -> result ready for the collective) and the gather's own time (stream events with
        # nccl - the collective is stream-ordered; host clock with gloo, whose all_gather blocks the host)
        roll_ms = sum(a.elapsed_time(b) for a, b, _, _ in phase) / max(len(phase), 1)
        gath_ms = (sum(b.elapsed_time(c) for _, b, c, _ in phase) if args.backend == "nccl" else 1e3 * sum(h for _, _, _, h in phase)) / max(len(phase), 1)
        mine = torch.tensor([1e3 * elapsed / args.steps, roll_ms, gath_ms, float(m)], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        rank_stats = dict(ms_per_step=[round(v, 3) for v in allr[:, 0].tolist()], ms_per_step_min=round(allr[:, 0].min().item(), 3),
                          ms_per_step_max=round(allr[:, 0].max().item(), 3), rollout_ms_per_step=[round(v, 3) for v in allr[:, 1].tolist()],
                          gather_ms_per_step=[round(v, 3) for v in allr[:, 2].tolist()], members=[int(v) for v in allr[:, 3].tolist()],
                          note="per rank: wall ms per step incl. barrier + synchronise; rollout = this rank's chunks up to the collective; gather = the one "
                               "all_gather of the result latents (a rank that finishes early waits here for the slowest)")
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if args.dump_output and rank == 0 and last_out is not None:
        torch.save(last_out.detach().to("cpu"), args.dump_output)
    # Sustained window (not `value`): keep running back-to-back chunks until >= --sustained-seconds have passed, so the number
    # reflects the clock the chip holds under sustained load; the step count is fixed up front so every rank runs the same.
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = max(1, int(-(-args.sustained_seconds // (elapsed / args.steps))))
        fence()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            step()
        fence()
        dt = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        sustained = dict(steps=n_sus, seconds=round(dt, 3), ms_per_step=round(1e3 * dt / n_sus, 3), value=round(total_members * lead * n_sus / dt, 4))
    # BASELINE configs[2] as an extra leg (not `value`): a FIXED ensemble of 16 members x 40 lead steps (10 chained chunks per member) dealt to
    # the ranks - north_star's strong-scaling workload (">= 6x at 8 GPUs vs 1").  Always with N > 1, so that the driver's SCALE record carries
    # it; at N = 1 behind --strong-cfg3 (17 s per call).  One call is timed, after a one-chunk call that captures this rank's chunk graph.
    strong_cfg3 = None
    if (world > 1 or args.strong_cfg3) and not args.decode and not args.no_strong_cfg3:
        E3, L3 = 16, 40
        ids3 = shard_members(E3, rank, world)

        def cfg3_call(lead_steps):
            return roll_out_serial(
                None, [datetime(2018, 1, 1, 0)], pipe, ensemble_size=len(ids3), num_inference_steps=args.solver_steps, return_seq_len=R,
                latent_transform_args=targs, total_lead_time_hour=6 * lead_steps, sampler_type=sampler_type, return_latent=True,
                known_latents_override=ic, member_ids=ids3, output_device=out_dev)

        cfg3_call(R)  # set-up: weight plan + chunk graph for this rank's batch size
        fence()
        e3a, e3b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t3 = time.perf_counter()
        e3a.record()
        out3 = cfg3_call(L3)
        e3b.record()
        if world > 1:
            out3 = gather_members(out3.to(dev) if args.backend == "nccl" else out3.to("cpu"), E3, member_dim=1)
        fence()
        dt3 = time.perf_counter() - t3
        roll3 = e3a.elapsed_time(e3b)
        per_rank3 = None
        if world > 1:
            mine3 = torch.tensor([dt3, roll3, float(len(ids3))], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            all3 = [torch.empty_like(mine3) for _ in range(world)]
            dist.all_gather(all3, mine3)
            all3 = torch.stack(all3).cpu()
            dt3 = all3[:, 0].max().item()
            per_rank3 = dict(seconds=[round(v, 3) for v in all3[:, 0].tolist()], rollout_ms=[round(v, 1) for v in all3[:, 1].tolist()],
                             members=[int(v) for v in all3[:, 2].tolist()])
        ref3 = None
        try:
            import glob as _glob

            cands = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg3_whole_job_16members_one_gpu.json")))
            if cands:
                d3 = json.loads(open(cands[-1]).read().strip().splitlines()[-1])
                v3 = d3["strong_cfg3"]["value"] if d3.get("strong_cfg3") else d3["value"]
                ref3 = dict(value=v3, source=os.path.relpath(cands[-1], ROOT) + " (committed; the same workload on ONE MI355X)")
        except Exception:
            ref3 = None
        v3now = E3 * L3 / dt3
        strong_cfg3 = dict(
            workload=f"BASELINE configs[2]: {args.model} AR, a FIXED ensemble of {E3} members over {world} GPU(s), {args.solver_steps} solver steps ({args.sampler}), {L3} lead steps = "
                     f"{-(-L3 // R)} chained chunks per member, one RCCL gather of the (16, 84, 41, 15, 30) latents at the end",
            value=round(v3now, 4), unit="member-steps/s", steps=1, ms_per_step=round(1e3 * dt3, 1), scaling="strong",
            members_on_rank=[len(shard_members(E3, r, world)) for r in range(world)], per_rank=per_rank3, n1_reference=ref3,
            # (only where the committed N = 1 figure is the same workload: 375M, edm, 20 solver steps, bf16x3)
            speedup_vs_n1=None if (ref3 is None or world == 1 or not (args.solver_steps == 20 and args.sampler == "edm" and args.model == "375M" and args.precision == "bf16x3"))
            else round(v3now / ref3["value"], 3),
            expected=("members are independent and a rank's members run as one batch: at N = 8 two members per rank (35.8 member-steps/s per GPU measured at that batch, "
                      "profiles/r04_x_bench_cfg3_share_2members_40leadsteps.json) -> ~286 against 37.5 at N = 1, i.e. ~7.6x; the gather moves 12.4 MB per rank"))
    other_sampler, ddim_sampler, host_outputs = None, None, None
    if not args.no_kernel_timers and rank == 0 and world == 1 and not args.decode:
        # Secondary numbers (not `value`): the same workload with the other samplers - BASELINE's metric says "20-step DDIM"; `edm` (the
        # reference's default, 39 forwards per chunk) is the headline, `pipeline` (the reference's scheduler loop with its EDM
        # DPM-Solver++(2M) scheduler, 20 forwards per chunk) and `ddim` (the same loop with DDIMScheduler, 20 forwards) are reported
        # beside it; and the headline workload with the reference's output placement (a host copy + synchronise per step).
        def timed(fn, n):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        def alt_block(alt):
            st = "edm" if alt == "edm" else "pipeline"

            def step_alt():
                return roll_out_serial(
                    None, [datetime(2018, 1, 1, 0)], pipes[alt], ensemble_size=m, num_inference_steps=args.solver_steps, return_seq_len=R,
                    latent_transform_args=targs, total_lead_time_hour=6 * lead, sampler_type=st, return_latent=True,
                    known_latents_override=ic, member_ids=member_ids,
                )

            dt_ = timed(step_alt, 2)
            return dict(sampler=alt, forwards_per_step=(-(-lead // R)) * (2 * args.solver_steps - 1 if alt == "edm" else args.solver_steps),
                        value=round(m * lead * 2 / dt_, 4), unit="member-steps/s", steps=2, ms_per_step=round(1e3 * dt_ / 2, 3))

        other_sampler = alt_block("pipeline" if args.sampler == "edm" else "edm")
        if args.sampler != "ddim":
            ddim_sampler = alt_block("ddim")
            ddim_sampler["scheduler"] = "ladcast_amd.schedulers.DDIMScheduler (diffusers defaults: linear betas, leading spacing, clip_sample, eta = 0), whole loop in one hipGraph"
        if out_dev is not None:
            dt_ = timed(lambda: step_local(None), 3)
            host_outputs = dict(value=round(m * lead * 3 / dt_, 4), unit="member-steps/s", steps=3, ms_per_step=round(1e3 * dt_ / 3, 3),
                                note="the headline workload with every step's result copied to the host + synchronised, as the reference's roll_out_serial returns it (--host-outputs)")

    def instrumented_step():
        """Kernel-level numbers for `roofline`: ONE more step of the same workload, right after the timed region, with a
        HIP-event pair around every GEMM / attention call on the stream they are launched on.  It runs eagerly (events
        cannot bracket nodes of a replayed hipGraph) and stays out of `value`, so the headline is not perturbed."""
        model.enable_hip_graph(False)
        timer.clear()
        timer.install(hip)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step_local()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t1)
        timer.uninstall()
        model.enable_hip_graph(not args.no_graph)
        return ms, timer.summary()

    def bracket_overhead():
        """What a HIP-event pair adds around one launch on a busy stream, measured live: the bracket around a one-element kernel that is
        queued behind a long one (median of 20), minus what the same launch costs un-bracketed inside a back-to-back chain (200 launches
        replayed from one hipGraph between two events: kernel + the ~1.5 us dependent-launch boundary).  The instrumented step's averages
        contain this per launch; `achieved` is quoted net of it, `achieved_raw` with it."""
        import statistics

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

    instrumented_ms, ks, bracket_us, bracket_raw_us, chain_us = None, {}, None, None, None
    if not args.no_kernel_timers and rank == 0:
        instrumented_ms, ks = instrumented_step()
        try:
            bracket_us, bracket_raw_us, chain_us = bracket_overhead()
        except Exception:
            bracket_us, bracket_raw_us, chain_us = None, None, None
    # The same workload in the exact-fp32 mode (fp32-input MFMA everywhere), beside the headline: value, ms per step and the
    # dominant GEMM kernel against the 157.3 TFLOP/s fp32 matrix peak.  Single-GPU runs only (no collective inside).
    fp32_mode = None
    if args.precision != "fp32" and not args.no_kernel_timers and world == 1 and not args.decode:
        model.set_gemm_precision("fp32")
        step_local()  # set-up: weight plan + graph capture of the fp32 chunk
        n32, w32 = max(10, args.steps), max(2, args.warmup)  # the headline's discipline: warm-up steps, then >= 10 timed steps with their own step_ms
        for _ in range(w32):
            step_local()
        torch.cuda.synchronize()
        ev32 = [torch.cuda.Event(enable_timing=True)]
        ev32[0].record()
        t1 = time.perf_counter()
        for _ in range(n32):
            step_local()
            ev32.append(torch.cuda.Event(enable_timing=True))
            ev32[-1].record()
        torch.cuda.synchronize()
        dt32 = time.perf_counter() - t1
        _, ks32 = instrumented_step()
        g32 = [n for n in ks32 if n.startswith("gemm_")]
        d32 = max(g32, key=lambda n: ks32[n]["total_ms"]) if g32 else None
        fp32_mode = dict(value=round(m * lead * n32 / dt32, 4), unit="member-steps/s", steps=n32, warmup=w32, ms_per_step=round(1e3 * dt32 / n32, 3), dtype="f32",
                         step_ms=[round(a_.elapsed_time(b_), 2) for a_, b_ in zip(ev32[:-1], ev32[1:])])
        if d32:
            k = ks32[d32]
            fp32_mode["roofline"] = dict(bound="mfma", kernel=d32, achieved=round(k["tflops"], 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                                         frac=round(k["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), launches=k["launches"], avg_launch_us=round(k["avg_us"], 2))
        if "attn_fwd_f32_kernel" in ks32:
            k = ks32["attn_fwd_f32_kernel"]
            fp32_mode["attention_kernel"] = dict(kernel="attn_fwd_f32_kernel", achieved=round(k["tflops"], 2), peak=PEAK_F32_MFMA_TFLOPS,
                                                 frac=round(k["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), avg_launch_us=round(k["avg_us"], 2))
        model.set_gemm_precision(args.precision)

    chunks = -(-lead // R)
    fwd_per_chunk = (2 * args.solver_steps - 1) if args.sampler == "edm" else args.solver_steps
    me = gpu_state(dev_index)
    rank_devices = [dict(rank=rank, device_index=dev_index, name=me.get("name"), pci_bus_id=me.get("pci_bus_id"), gcn_arch=me.get("gcn_arch"))]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered
    if rank == 0:
        gflops, aflops = model_flops_per_forward(cfg, R)
        (mg, ma), (cg, ca) = model_flops_per_forward(cfg, R, split=True)
        # flops that RAN per member and chunk: the sample-dependent part once per network evaluation, the conditioning path once per
        # noise level when it is batched per chunk (2N - 1 evaluations share N noise levels in the Heun sampler), else per evaluation
        cond_evals = args.solver_steps if model.batch_conditioning else fwd_per_chunk
        flops_per_chunk = fwd_per_chunk * (mg + ma) + cond_evals * (cg + ca)
        value = total_members * lead * args.steps / elapsed
        roof = None
        split = args.precision == "bf16x3"
        bf16_cores = args.precision in ("bf16x3", "bf16")
        gemm_names = [n for n in ks if n.startswith("gemm_")]
        dom = max(gemm_names, key=lambda n: ks[n]["total_ms"]) if gemm_names else None  # the GEMM kernel with the most time
        if dom in ks:
            k = ks[dom]
            kname = dom
            peak = PEAK_BF16_MFMA_TFLOPS if bf16_cores else PEAK_F32_MFMA_TFLOPS
            traffic, traffic_source, traffic_stale = None, None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(pmc):
                try:
                    from ladcast_amd.build_id import csrc_sha16

                    doc = json.load(open(pmc))
                    ent = doc.get(kname) or doc.get(kname.split("<")[0], {})
                    traffic = ent.get("hbm_bytes_per_launch")
                    if traffic is not None:
                        built = (doc.get("_build") or {}).get("csrc_sha16")
                        traffic_stale = built != csrc_sha16()  # the kernels changed since the counter passes (or the file predates build ids)
                        traffic_source = ("profiles/pmc_summary.json: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this command on kernel "
                                          f"sources {built}; this run's sources are {csrc_sha16()}; not re-measured in this run")
                except Exception:
                    traffic = None
            # every instance of the dominant kernel template on its own row: launches, HIP-event average, exact flops per launch, and - when the
            # committed rocprofv3 summary of the headline leg holds that instance - its AverageNs there and the rate that follows from it
            import csv
            import glob

            stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_bench_cfg2*.csv")))
            prof_avg = {}
            if stats:
                try:
                    for row in csv.DictReader(open(stats[-1])):
                        nm = row.get("Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
                        prof_avg[nm.split("(")[0].strip()] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
                except Exception:
                    prof_avg = {}
            variants = []
            for vn in sorted(n for n in ks if n.startswith(dom.split("<")[0])):
                kv = ks[vn]
                net_us = kv["avg_us"] - (bracket_us or 0.0)  # HIP-event time of the launch net of what the event pair itself adds
                rowv = dict(kernel=vn, launches=kv["launches"], avg_launch_us=round(net_us, 2), avg_launch_us_raw=round(kv["avg_us"], 2),
                            flops_per_launch=kv["work_per_launch"], achieved=round(kv["work_per_launch"] / (net_us * 1e-6) / 1e12, 2),
                            achieved_raw=round(kv["tflops"], 2), frac=round(kv["work_per_launch"] / (net_us * 1e-6) / 1e12 / peak, 4))
                if vn in prof_avg:
                    rowv["rocprof_avg_us"] = round(prof_avg[vn][0], 2)
                    rowv["achieved_rocprof"] = round(kv["work_per_launch"] / (prof_avg[vn][0] * 1e-6) / 1e12, 2)
                variants.append(rowv)
            dom_net_us = k["avg_us"] - (bracket_us or 0.0)
            dom_tf = k["work_per_launch"] / (dom_net_us * 1e-6) / 1e12
            roof = dict(bound="mfma", kernel=kname, achieved=round(dom_tf, 2), achieved_raw=round(k["tflops"], 2), peak=peak, unit="TFLOP/s", variants=variants,
                        event_bracket_overhead_us=bracket_us, event_bracket_us=bracket_raw_us, unbracketed_launch_us=chain_us,
                        kernel_stats_source=(os.path.relpath(stats[-1], ROOT) + ": rocprofv3 --kernel-trace --stats summary of the headline leg alone (bench.py "
                                             "--steps K --warmup W --no-kernel-timers --cpu-forwards 0 --sustained-seconds 0; committed).  Per template instance, "
                                             "`flops_per_launch` / its AverageNs there = `achieved_rocprof` in `variants`; `achieved` is the live HIP-event figure net of "
                                             "`event_bracket_overhead_us` (what an event pair adds per launch, calibrated in the same run), `achieved_raw` with it") if stats else None,
                        frac=round(dom_tf / peak, 4), traffic=traffic, traffic_source=traffic_source, traffic_stale=traffic_stale, launches=k["launches"],
                        avg_launch_us=round(dom_net_us, 2), avg_launch_us_raw=round(k["avg_us"], 2),
                        flops_per_launch=k["work_per_launch"],
                        note="one launch = one grouped stream-K GEMM call of the dominant template instance (`kernel`); achieved = ALGORITHMIC 2*M*N*K summed over the call's problems / "
                             "HIP-event time of the call, averaged over one instrumented step run right after the timed region (see instrumented_ms_per_step)."
                             + (" Split-bf16: the kernel issues 3 bf16 MFMA flops per algorithmic flop (hi*hi + hi*lo + lo*hi), so its "
                                "ceiling against this peak is 1/3; frac_of_attainable = achieved / (peak/3)." if split else ""))
            if split:
                roof["frac_of_attainable"] = round(3 * dom_tf / peak, 4)
                roof["traffic_note"] = ("fabric-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (profiles/pmc_summary.json, "
                                        "2*FETCH+WRITE KiB); served mostly by the 256 MiB Infinity Cache: panels are re-read per XCD")
        line = {
            "metric": "ensemble-member-steps/sec", "value": round(value, 4), "unit": "member-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "step_ms": step_ms, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3(split-fp32 operands, f32 accumulate; softmax/norms f32, sampler state f64)", "fp32": "f32",
                      "bf16": "bf16(single-term bf16 operands, f32 accumulate and outputs; temb/norms/softmax f32, sampler state f64; stated tolerance "
                              f"{tolerance('bf16', 'forward'):g} per forward, ladcast_amd/precision.py)"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": ("cfg5-style END-TO-END (DCAE encode -> AR -> DCAE decode of every lead step), " if args.decode else "") +
                            f"cfg2: {args.model} AR transformer, " + (f"a FIXED ensemble of {total_members} member(s) over {world} GPU(s)" if strong else f"{m} member/GPU") +
                            f", {args.solver_steps} solver steps ({args.sampler}: {fwd_per_chunk} forwards/chunk), "
                            f"{lead} lead step(s) = {chunks} chunk(s) of return_seq_len {R}, latent 84x15x30, fp32 weights random-init seed 1234, arithmetic {args.precision}",
                "sampler": args.sampler, "members_per_gpu": m if not strong else None, "ensemble_size": total_members,
                "members_on_rank": [len(shard_members(total_members, r, world)) for r in range(world)], "lead_steps": lead, "return_seq_len": R, "forwards_per_step": chunks * fwd_per_chunk,
                "tflop_per_forward_per_member": round((gflops + aflops) / 1e12, 4),
                "conditioning_path": ("per chunk: the sample-independent part of every forward (context refiner, conditioning embedding, AdaLN modulation GEMV: "
                                      "0.026 of the 0.979 TFLOP and 17 of the 52 launches of a 375M forward) runs once per chunk as ONE batch over the chunk's "
                                      f"{args.solver_steps} noise levels, inside the timed region, every chunk; --no-batched-conditioning runs it per evaluation")
                if model.batch_conditioning else "per network evaluation (as the reference)",
                "outputs": ("left in HBM (with N > 1: gathered there); no per-step host copy, so the host prepares step k + 1 while the GPU runs step k - the "
                            "timed region ends with barrier + synchronise (two alternating instances of the captured chunk); --host-outputs copies every step's result to the host as the reference's "
                            "roll_out_serial does") if out_dev is not None else "copied to the host after every step (as the reference)",
            },
            "gpu": {"device_index": dev_index, "before": state_before, "during_timed_region": gpu_during, "after": state_after,
                    "note": "rank 0's GPU from sysfs (hwmon of its PCI function): one reading before the timed region, one every 50 ms during it (background thread), one "
                            "after; two runs whose `value` differs on different boxes can be compared by the clock / power the chip held"},
            "strong_cfg3": strong_cfg3,
            "instrumented_ms_per_step": None if instrumented_ms is None else round(instrumented_ms, 3),
            "other_sampler": other_sampler,
            "ddim_sampler": ddim_sampler,
            "host_outputs": host_outputs,
            "sustained": sustained,
            "fp32_mode": fp32_mode,
            # the like-for-like-precision reader's roofline (exact fp32 everywhere), at the top level beside the split one
            "roofline_fp32": None if not fp32_mode else fp32_mode.get("roofline"),
            "ranks": {"world_size": world if dist is None else dist.get_world_size(), "backend": "none" if dist is None else args.backend, "devices": rank_devices,
                      "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if (dist is not None and args.backend == "nccl") else None,
                      "visible_gpus": torch.cuda.device_count(), "per_rank": rank_stats},
            "model_tflops": round(total_members * chunks * flops_per_chunk * args.steps / elapsed / 1e12, 2),
            "model_tflops_note": f"algorithmic flops that ran: {fwd_per_chunk} x {round((mg + ma) / 1e12, 4)} TFLOP (sample-dependent part) + {cond_evals} x "
                                 f"{round((cg + ca) / 1e12, 4)} TFLOP (conditioning path) per member and chunk",
            "roofline": roof,
        }
        for an, apeak in (("attn_fwd_f32_kernel", PEAK_F32_MFMA_TFLOPS), ("attn_fwd_split_kernel", PEAK_BF16_MFMA_TFLOPS)):
            if an in ks:
                k = ks[an]
                line["attention_kernel"] = dict(kernel=an, achieved=round(k["tflops"], 2), peak=apeak, unit="TFLOP/s", frac=round(k["tflops"] / apeak, 4),
                                                launches=k["launches"], avg_launch_us=round(k["avg_us"], 2))
                if an != "attn_fwd_f32_kernel" and split:
                    line["attention_kernel"]["frac_of_attainable"] = round(3 * k["tflops"] / apeak, 4)
        if args.cpu_forwards > 0 and world == 1:
            import statistics

            cores = torch.get_num_threads()
            times = cpu_baseline(args.model, R, args.cpu_forwards)
            dt = statistics.median(times)
            cpu_value = lead / (chunks * fwd_per_chunk * dt)
            line["cpu_baseline"] = dict(
                value=round(cpu_value, 5), unit="member-steps/s", cores=cores, kind="port", host=host_description(),
                forward_seconds=[round(t, 3) for t in times],
                sample=f"{args.cpu_forwards} forwards (after one warm-up forward) of the same {args.model} model (1 member, R={R}) by the PyTorch CPU oracle on {cores} torch "
                       f"threads, median {dt:.2f} s per forward, scaled to {chunks * fwd_per_chunk} forwards per step",
            )
            line["gpu_over_cpu"] = round(value / cpu_value, 1)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
