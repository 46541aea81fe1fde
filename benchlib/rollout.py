"""The rollout legs of the benchmark: the headline (BASELINE configs[1]: one sampler chunk per step), the sustained window, the strong
cfg 3 leg, the other samplers, the other arithmetic mode, cfg 5's chunk + decode, and the `roofline` rows of each mode's dominant kernel."""
import glob
import json
import os
import time
from datetime import datetime

import torch

from .configs import CONFIG_DCAE_84, CONFIGS, PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS
from .host import GpuStateSampler, gpu_state
from .kernel_timer import rocprof_averages, roofline_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T0 = [datetime(2018, 1, 1, 0)]
TARGS = {"mean": [0.0] * 84, "std": [1.0] * 84, "target_std": 0.5}

DTYPE_TEXT = {
    "fp32": "f32",
    "bf16x3": "bf16x3(split-fp32 operands hi*hi + hi*lo + lo*hi, f32 accumulate; softmax/norms f32, sampler state f64)",
    "bf16": "bf16(single-term bf16 operands, f32 accumulate and outputs; temb/norms/softmax f32, sampler state f64)",
}


def _peak(precision):
    return PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS


def _pmc_traffic(kname, precision):
    """fabric bytes per launch of `kname` from the committed counter passes of this precision's headline run (tools/collect_artifacts.sh:
    two separate rocprofv3 --pmc passes, FETCH_SIZE x 2 + WRITE_SIZE), stamped with the kernel-source hash they were taken on"""
    from ladcast_amd.build_id import csrc_sha16

    pmc = os.path.join(ROOT, "profiles", "pmc_summary_fp32.json" if precision == "fp32" else "pmc_summary.json")
    if not os.path.exists(pmc):
        return None, None, None
    try:
        doc = json.load(open(pmc))
        ent = doc.get(kname) or doc.get(kname.split("<")[0], {})
        traffic = ent.get("hbm_bytes_per_launch")
        if traffic is None:
            return None, None, None
        built = (doc.get("_build") or {}).get("csrc_sha16")
        src = (f"{os.path.relpath(pmc, ROOT)}: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; (2 * FETCH + WRITE) KiB) of `bench.py --precision {precision}` on "
               f"kernel sources {built}; this run's sources are {csrc_sha16()}; not re-measured in this run")
        return traffic, src, built != csrc_sha16()
    except Exception:
        return None, None, None


def _stats_pattern(precision):
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_kernel_stats_bench_cfg2_{precision}*.csv"))
                   + (glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_bench_cfg2_headline.csv")) if precision == "bf16x3" else []), key=os.path.basename)
    return os.path.basename(cands[-1]) if cands else "__none__"


def build_roofline(ks, precision, bracket, note_extra=""):
    """`roofline` of one arithmetic mode from the instrumented step's per-kernel summary `ks`: the GEMM template instance with the most time,
    quoted against the dense matrix peak of the mode's MFMA input type.  `achieved` / `frac` are the MEASURED HIP-event figures of this run;
    `achieved_net` (bracket overhead removed, a modelled figure) and `achieved_rocprof` (committed rocprofv3 AverageNs) sit beside them."""
    gemm_names = [n for n in ks if n.startswith("gemm_")]
    if not gemm_names:
        return None
    dom = max(gemm_names, key=lambda n: ks[n]["total_ms"])
    k, peak = ks[dom], _peak(precision)
    bracket_us, bracket_raw_us, chain_us = bracket
    prof_avg, stats_file = rocprof_averages(ROOT, _stats_pattern(precision))
    variants = roofline_rows(ks, dom, peak, bracket_us, prof_avg)
    traffic, traffic_source, traffic_stale = _pmc_traffic(dom, precision)
    split = precision == "bf16x3"
    roof = dict(bound="mfma", kernel=dom, achieved=round(k["tflops"], 2), peak=peak, unit="TFLOP/s", frac=round(k["tflops"] / peak, 4), traffic=traffic,
                launches=k["launches"], avg_launch_us=round(k["avg_us"], 2), flops_per_launch=k["work_per_launch"], variants=variants,
                event_bracket_overhead_us=bracket_us, event_bracket_us=bracket_raw_us, unbracketed_launch_us=chain_us,
                traffic_source=traffic_source, traffic_stale=traffic_stale,
                kernel_stats_source=(f"profiles/{stats_file}: rocprofv3 --kernel-trace --stats summary of this mode's headline leg alone (bench.py --precision {precision} --steps K "
                                     "--warmup W --no-kernel-timers --cpu-forwards 0 --sustained-seconds 0; committed): flops_per_launch / its AverageNs = achieved_rocprof") if stats_file else None,
                note="one launch = one grouped stream-K GEMM call of the dominant template instance (`kernel`); achieved = ALGORITHMIC 2*M*N*K summed over the call's problems / "
                     "HIP-event time of the call (on the launch stream), averaged over one instrumented eager step run right after the timed region; achieved_net removes "
                     "the calibrated per-launch cost of the event pair itself (a modelled figure, shown beside the measured one)" + note_extra)
    for v in variants:
        if v["kernel"] == dom and "achieved_net" in v:
            roof["achieved_net"], roof["avg_launch_us_net"] = v["achieved_net"], v["avg_launch_us_net"]
        if v["kernel"] == dom and "achieved_rocprof" in v:
            roof["achieved_rocprof"] = v["achieved_rocprof"]
    if split:
        roof["frac_of_attainable"] = round(3 * k["tflops"] / peak, 4)
        roof["note"] += (" Split-bf16: the kernel issues 3 bf16 MFMA flops per algorithmic flop (hi*hi + hi*lo + lo*hi), so its ceiling against this peak is 1/3; "
                         "frac_of_attainable = achieved / (peak / 3).")
        roof["traffic_note"] = ("fabric-side bytes per launch (FETCH_SIZE / WRITE_SIZE count at the fabric; served mostly by the 256 MiB Infinity Cache: panels are re-read per XCD)")
    return roof


def attention_row(ks, precision):
    for an, apeak in (("attn_fwd_f32_kernel", PEAK_F32_MFMA_TFLOPS), ("attn_fwd_split_kernel", PEAK_BF16_MFMA_TFLOPS)):
        if an in ks:
            k = ks[an]
            row = dict(kernel=an, achieved=round(k["tflops"], 2), peak=apeak, unit="TFLOP/s", frac=round(k["tflops"] / apeak, 4), launches=k["launches"],
                       avg_launch_us=round(k["avg_us"], 2))
            if an == "attn_fwd_split_kernel" and precision == "bf16x3":
                row["frac_of_attainable"] = round(3 * k["tflops"] / apeak, 4)
            return row
    return None


class Rollout:
    """model + pipelines + the step functions of one rank"""

    def __init__(self, args, dev, rank, world, dist):
        import ladcast_amd.hip as hip
        from ladcast_amd.models import LaDCastTransformer3DModel
        from ladcast_amd.pipelines import AutoRegressive2DPipeline
        from ladcast_amd.pipelines.distributed import shard_members
        from ladcast_amd.schedulers import DDIMScheduler, EDMDPMSolverMultistepScheduler

        self.args, self.dev, self.rank, self.world, self.dist, self.hip = args, dev, rank, world, dist, hip
        self.cfg = CONFIGS[args.model]
        torch.manual_seed(1234)
        self.model = LaDCastTransformer3DModel.from_config(self.cfg).to(dev).eval().set_gemm_precision(args.precision)
        self.model.enable_hip_graph(not args.no_graph)
        self.model.batch_conditioning = not args.no_batched_conditioning
        self.pipes = {"edm": AutoRegressive2DPipeline(self.model, EDMDPMSolverMultistepScheduler()), "ddim": AutoRegressive2DPipeline(self.model, DDIMScheduler())}
        self.pipes["pipeline"] = self.pipes["edm"]
        self.pipe = self.pipes[args.sampler]
        self.sampler_type = "edm" if args.sampler == "edm" else "pipeline"  # roll_out_serial's switch: Heun sampler | the pipeline's scheduler loop
        self.strong = args.ensemble_size > 0
        self.total_members = args.ensemble_size if self.strong else args.members_per_gpu * world
        self.member_ids = shard_members(self.total_members, rank, world)  # rank r owns members {k : k mod world == r}
        self.m = len(self.member_ids)  # this rank's members: --members-per-gpu (weak) or its share of the fixed ensemble (strong; may be 0)
        self.R, self.lead = args.return_seq_len, args.lead_steps
        self.ic = (0.5 * torch.randn(84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).to(dev)  # IC latent, resident in HBM
        # results stay in HBM (with N > 1 the one collective gathers them there): no host copy and therefore no host stall per step - the host
        # prepares step k + 1 (noise draw, timestamps) while the GPU still runs step k; the timed region ends with a synchronise as the contract says
        self.out_dev = None if args.host_outputs else dev
        self.phase = []  # per step of this rank: (event before the rollout, after it = before the gather, after the gather, host seconds of the gather)
        self.ae = None
        if args.decode:
            self.ae, self.field, self.static = self.make_ae(args.precision)

    def make_ae(self, precision):
        from ladcast_amd.models import AutoencoderDC

        ae = AutoencoderDC.from_config(CONFIG_DCAE_84).to(self.dev).eval().set_gemm_precision(precision).enable_hip_graph(not self.args.no_graph)
        g_ = torch.Generator().manual_seed(3)
        return ae, torch.randn(84, 1, 120, 240, generator=g_), torch.randn(5, 120, 240, generator=g_)

    def rollout(self, pipe=None, sampler_type=None, members=None, lead=None, od="default"):
        from ladcast_amd.pipelines import roll_out_serial

        ids = self.member_ids if members is None else members
        return roll_out_serial(
            None, T0, pipe or self.pipe, ensemble_size=len(ids), num_inference_steps=self.args.solver_steps, return_seq_len=self.R,
            latent_transform_args=TARGS, total_lead_time_hour=6 * (self.lead if lead is None else lead), sampler_type=sampler_type or self.sampler_type,
            return_latent=True, known_latents_override=self.ic, member_ids=ids, output_device=self.out_dev if od == "default" else od)

    def rollout_decoded(self, ae, field, static, lead=None):
        """end-to-end: IC field -> encode -> AR chunks -> decode (roll_out_serial's decoded-field mode)"""
        from ladcast_amd.pipelines import roll_out_serial

        return roll_out_serial(
            lambda t: field, T0, self.pipe, ensemble_size=self.m, num_inference_steps=self.args.solver_steps, return_seq_len=self.R,
            latent_transform_args=TARGS, total_lead_time_hour=6 * (self.lead if lead is None else lead), sampler_type=self.sampler_type, return_latent=False,
            member_ids=self.member_ids, output_device=self.out_dev, encdec_model=ae, encdec_model_type="ae", static_tensor4encdec=static,
            normalization_param_dict={"mean": torch.zeros(84), "std": torch.ones(84)}, decode_batch_frames=self.args.decode_batch_frames or None)

    def step(self):
        from ladcast_amd.pipelines.distributed import gather_members

        if self.args.decode:
            return self.rollout_decoded(self.ae, self.field, self.static)
        if self.world > 1:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        out = self.rollout()
        if self.world > 1:  # the one collective of the path: gather the per-rank latents (evaluate/pred_rollout.py:398-400)
            e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e1.record()
            h0 = time.perf_counter()
            out = gather_members(out.to(self.dev) if self.args.backend == "nccl" else out.to("cpu"), self.total_members, member_dim=1)
            h1 = time.perf_counter()
            e2.record()
            self.phase.append((e0, e1, e2, h1 - h0))
        return out

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()

    def coll_device(self):
        return self.dev if self.args.backend == "nccl" else "cpu"

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], device=self.coll_device(), dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    # ---- the legs ----------------------------------------------------------------------------------------------------------------
    def timed_region(self):
        """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronise on both sides; max over ranks"""
        a = self.args
        self.step()  # set-up, not a warm-up step: packs the weights and captures the chunk graph (like model construction, it is never timed)
        for _ in range(a.warmup):
            self.step()
        dev_index = self.dev.index
        state_before = gpu_state(dev_index)
        self.fence()
        sampler = GpuStateSampler(dev_index).start()
        t0 = time.perf_counter()
        marks, evs = [], [torch.cuda.Event(enable_timing=True)]
        evs[0].record()
        self.phase.clear()
        last_out = None
        for _ in range(a.steps):
            last_out = self.step()
            marks.append(time.perf_counter())  # --host-outputs: every step returns host tensors, i.e. is complete here
            evs.append(torch.cuda.Event(enable_timing=True))
            evs[-1].record()  # device outputs: the host runs ahead; the step boundaries are read from the stream afterwards
        self.fence()
        elapsed = time.perf_counter() - t0
        gpu_during = sampler.stop()
        state_after = gpu_state(dev_index)
        if self.out_dev is None:
            step_ms = [round(1e3 * (b - a_), 2) for a_, b in zip([t0] + marks[:-1], marks)]  # diagnostic only (rank 0's view)
        else:
            step_ms = [round(a_.elapsed_time(b), 2) for a_, b in zip(evs[:-1], evs[1:])]
        rank_stats = self.rank_stats(elapsed) if self.world > 1 else None
        elapsed = self.max_over_ranks(elapsed)
        if a.dump_output and self.rank == 0 and last_out is not None:
            torch.save(last_out.detach().to("cpu"), a.dump_output)
        return dict(elapsed=elapsed, step_ms=step_ms, rank_stats=rank_stats,
                    gpu={"device_index": dev_index, "before": state_before, "during_timed_region": gpu_during, "after": state_after,
                         "note": "rank 0's GPU from sysfs (hwmon of its PCI function): one reading before the timed region, one every 50 ms during it (background thread), one "
                                 "after; two runs whose `value` differs on different boxes can be compared by the clock / power the chip held"})

    def rank_stats(self, elapsed):
        """per-rank diagnostics (a SCALE run must be readable from the one line): this rank's own wall time, its rollout time per step (events on
        the stream: start of the step -> result ready for the collective) and the gather's own time (stream events with nccl - the collective is
        stream-ordered; host clock with gloo, whose all_gather blocks the host)"""
        a, ph = self.args, self.phase
        roll_ms = sum(x.elapsed_time(y) for x, y, _, _ in ph) / max(len(ph), 1)
        gath_ms = (sum(y.elapsed_time(z) for _, y, z, _ in ph) if a.backend == "nccl" else 1e3 * sum(h for _, _, _, h in ph)) / max(len(ph), 1)
        mine = torch.tensor([1e3 * elapsed / a.steps, roll_ms, gath_ms, float(self.m)], device=self.coll_device(), dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        return dict(ms_per_step=[round(v, 3) for v in allr[:, 0].tolist()], ms_per_step_min=round(allr[:, 0].min().item(), 3),
                    ms_per_step_max=round(allr[:, 0].max().item(), 3), rollout_ms_per_step=[round(v, 3) for v in allr[:, 1].tolist()],
                    gather_ms_per_step=[round(v, 3) for v in allr[:, 2].tolist()], members=[int(v) for v in allr[:, 3].tolist()],
                    note="per rank: wall ms per step incl. barrier + synchronise; rollout = this rank's chunks up to the collective; gather = the one "
                         "all_gather of the result latents (a rank that finishes early waits here for the slowest)")

    def sustained(self, per_step_s):
        """(not `value`) keep running back-to-back chunks until >= --sustained-seconds have passed, so the number reflects the clock the chip
        holds under sustained load; the step count is fixed up front so every rank runs the same"""
        a = self.args
        n_sus = max(1, int(-(-a.sustained_seconds // per_step_s)))
        self.fence()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            self.step()
        self.fence()
        dt = self.max_over_ranks(time.perf_counter() - t1)
        return dict(steps=n_sus, seconds=round(dt, 3), ms_per_step=round(1e3 * dt / n_sus, 3), value=round(self.total_members * self.lead * n_sus / dt, 4))

    def strong_cfg3(self):
        """BASELINE configs[2] as an extra leg (not `value`): a FIXED ensemble of 16 members x 40 lead steps (10 chained chunks per member) dealt to
        the ranks - north_star's strong-scaling workload (">= 6x at 8 GPUs vs 1").  Always with N > 1, so that the driver's SCALE record carries
        it; at N = 1 behind --strong-cfg3 (17 s per call).  One call is timed, after a one-chunk call that captures this rank's chunk graph."""
        from ladcast_amd.pipelines.distributed import gather_members, shard_members

        a, world = self.args, self.world
        E3, L3 = 16, 40
        ids3 = shard_members(E3, self.rank, world)
        self.rollout(members=ids3, lead=self.R)  # set-up: weight plan + chunk graph for this rank's batch size
        self.fence()
        e3a, e3b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t3 = time.perf_counter()
        e3a.record()
        out3 = self.rollout(members=ids3, lead=L3)
        e3b.record()
        if world > 1:
            out3 = gather_members(out3.to(self.dev) if a.backend == "nccl" else out3.to("cpu"), E3, member_dim=1)
        self.fence()
        dt3 = time.perf_counter() - t3
        roll3 = e3a.elapsed_time(e3b)
        per_rank3 = None
        if world > 1:
            mine3 = torch.tensor([dt3, roll3, float(len(ids3))], device=self.coll_device(), dtype=torch.float64)
            all3 = [torch.empty_like(mine3) for _ in range(world)]
            self.dist.all_gather(all3, mine3)
            all3 = torch.stack(all3).cpu()
            dt3 = all3[:, 0].max().item()
            per_rank3 = dict(seconds=[round(v, 3) for v in all3[:, 0].tolist()], rollout_ms=[round(v, 1) for v in all3[:, 1].tolist()],
                             members=[int(v) for v in all3[:, 2].tolist()])
        ref3 = None
        try:
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_cfg3_whole_job_16members_one_gpu_{a.precision}.json")), key=os.path.basename)
            if not cands and a.precision == "bf16x3":
                cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg3_whole_job_16members_one_gpu.json")), key=os.path.basename)
            if cands:
                d3 = json.loads(open(cands[-1]).read().strip().splitlines()[-1])
                v3 = d3["strong_cfg3"]["value"] if d3.get("strong_cfg3") else d3["value"]
                ref3 = dict(value=v3, source=os.path.relpath(cands[-1], ROOT) + " (committed; the same workload and arithmetic on ONE MI355X)")
        except Exception:
            ref3 = None
        v3now = E3 * L3 / dt3
        same = a.solver_steps == 20 and a.sampler == "edm" and a.model == "375M"
        return dict(
            workload=f"BASELINE configs[2]: {a.model} AR, a FIXED ensemble of {E3} members over {world} GPU(s), {a.solver_steps} solver steps ({a.sampler}), {L3} lead steps = "
                     f"{-(-L3 // self.R)} chained chunks per member, arithmetic {a.precision}, one RCCL gather of the (16, 84, 41, 15, 30) latents at the end",
            value=round(v3now, 4), unit="member-steps/s", steps=1, ms_per_step=round(1e3 * dt3, 1), scaling="strong", dtype=DTYPE_TEXT[a.precision].split("(")[0],
            members_on_rank=[len(shard_members(E3, r, world)) for r in range(world)], per_rank=per_rank3, n1_reference=ref3,
            # (only where the committed N = 1 figure is the same workload and arithmetic)
            speedup_vs_n1=None if (ref3 is None or world == 1 or not same) else round(v3now / ref3["value"], 3),
            expected=("members are independent and a rank's members run as one batch: at N = 8 two members per rank, the gather moves 12.4 MB per rank; with the per-GPU rate at "
                      "batch 2 measured on one GPU (profiles/r*_bench_cfg3_share_2members_40leadsteps*.json) the expected speed-up over N = 1 is ~7.6x (DESIGN.md section 6)"))

    def alt_samplers(self):
        """Secondary numbers (not `value`): the same workload with the other samplers - BASELINE's metric says "20-step DDIM"; `edm` (the
        reference's default, 39 forwards per chunk) is the headline, `pipeline` (the reference's scheduler loop with its EDM DPM-Solver++(2M)
        scheduler, 20 forwards per chunk) and `ddim` (the same loop with DDIMScheduler, 20 forwards) are reported beside it; and the headline
        workload with the reference's output placement (a host copy + synchronise per step)."""
        a, m, lead, R = self.args, self.m, self.lead, self.R

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
            dt_ = timed(lambda: self.rollout(pipe=self.pipes[alt], sampler_type=st, od=None), 2)
            return dict(sampler=alt, forwards_per_step=(-(-lead // R)) * (2 * a.solver_steps - 1 if alt == "edm" else a.solver_steps), dtype=DTYPE_TEXT[a.precision].split("(")[0],
                        value=round(m * lead * 2 / dt_, 4), unit="member-steps/s", steps=2, ms_per_step=round(1e3 * dt_ / 2, 3))

        other = alt_block("pipeline" if a.sampler == "edm" else "edm")
        ddim = None
        if a.sampler != "ddim":
            ddim = alt_block("ddim")
            ddim["scheduler"] = "ladcast_amd.schedulers.DDIMScheduler (diffusers defaults: linear betas, leading spacing, clip_sample, eta = 0), whole loop in one hipGraph"
        host = None
        if self.out_dev is not None:
            dt_ = timed(lambda: self.rollout(od=None), 3)
            host = dict(value=round(m * lead * 3 / dt_, 4), unit="member-steps/s", steps=3, ms_per_step=round(1e3 * dt_ / 3, 3),
                        note="the headline workload with every step's result copied to the host + synchronised, as the reference's roll_out_serial returns it (--host-outputs)")
        return other, ddim, host

    def instrumented_step(self, timer):
        """Kernel-level numbers for `roofline`: ONE more step of the same workload, right after the timed region, with a HIP-event pair around
        every GEMM / attention call on the stream they are launched on.  It runs eagerly (events cannot bracket nodes of a replayed hipGraph)
        and stays out of `value`, so the headline is not perturbed."""
        self.model.enable_hip_graph(False)
        timer.clear()
        timer.install(self.hip)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.rollout()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t1)
        timer.uninstall()
        self.model.enable_hip_graph(not self.args.no_graph)
        return ms, timer.summary()

    def other_mode(self, precision, timer, bracket):
        """The same workload in another arithmetic mode, beside the headline and timed with the headline's discipline (warm-up steps, then >= 10
        timed steps with their own step_ms, one synchronise on each side), with that mode's own roofline.  Single-GPU runs only."""
        a, m, lead = self.args, self.m, self.lead
        self.model.set_gemm_precision(precision)
        self.rollout()  # set-up: weight plan + graph capture of this mode's chunk
        n, w = max(10, a.steps), max(2, a.warmup)
        for _ in range(w):
            self.rollout()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True)]
        ev[0].record()
        t1 = time.perf_counter()
        for _ in range(n):
            self.rollout()
            ev.append(torch.cuda.Event(enable_timing=True))
            ev[-1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        _, ks = self.instrumented_step(timer)
        from ladcast_amd.precision import tolerance

        blk = dict(value=round(m * lead * n / dt, 4), unit="member-steps/s", steps=n, warmup=w, ms_per_step=round(1e3 * dt / n, 3), dtype=DTYPE_TEXT[precision],
                   step_ms=[round(x.elapsed_time(y), 2) for x, y in zip(ev[:-1], ev[1:])], roofline=build_roofline(ks, precision, bracket),
                   attention_kernel=attention_row(ks, precision),
                   tolerance=f"<= {tolerance(precision, 'chunk' if precision != 'bf16' else 'chunk_edm'):g} rel-L2 per sampler chunk against the fp32 CPU oracle (ladcast_amd/precision.py; tests/test_gpu_chain.py)")
        self.model.set_gemm_precision(a.precision)
        return blk

    def cfg5_block(self, precisions=("bf16x3", "bf16")):
        """BASELINE configs[4] in shape on one GPU (its share of one member): IC field -> DC-AE encode -> one 20-step AR chunk -> DC-AE decode of the
        chunk's 4 frames, AR model AND autoencoder in the given arithmetic; per call after a set-up call, 3 timed calls"""
        a = self.args
        out = {"workload": f"BASELINE configs[4] per member on one MI355X: encode the IC field (84 x 120 x 240 + 5 static), one {a.solver_steps}-step {a.sampler} "
                                         f"chunk of the {a.model} AR model (return_seq_len {self.R} = {self.R} lead steps), decode its {self.R} frames; {self.m} member(s)"}
        for prec in precisions:
            self.model.set_gemm_precision(prec)
            ae, field, static = self.make_ae(prec)
            self.rollout_decoded(ae, field, static, lead=self.R)
            self.rollout_decoded(ae, field, static, lead=self.R)
            torch.cuda.synchronize()
            n = 3
            t1 = time.perf_counter()
            for _ in range(n):
                self.rollout_decoded(ae, field, static, lead=self.R)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / n
            # the chunk alone, same arithmetic (latent mode), to show the autoencoder's share
            self.rollout(lead=self.R)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                self.rollout(lead=self.R)
            torch.cuda.synchronize()
            dt_lat = (time.perf_counter() - t1) / n
            out[prec] = dict(value=round(self.m * self.R / dt, 4), unit="member-steps/s", steps=n, ms_per_step=round(1e3 * dt, 2), dtype=DTYPE_TEXT[prec].split("(")[0],
                             chunk_only_ms=round(1e3 * dt_lat, 2), encode_plus_decode_ms=round(1e3 * (dt - dt_lat), 2))
            del ae
            torch.cuda.empty_cache()
        self.model.set_gemm_precision(a.precision)
        return out
