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

ARITHMETIC.  The top-level ``value`` is measured in the reference's own arithmetic: exact fp32 on the fp32-input matrix cores
(``--precision fp32``, the default since round 6; ``dtype: "f32"``).  The split-bf16 mode (``bf16x3``: hi*hi + hi*lo + lo*hi with
fp32 accumulation, ~16 mantissa bits per product, inside north_star's 1e-4 rel-L2 by 40x) is the FAST mode; its figure is the
``bf16x3_mode`` block of the same line, never ``value``.  ``like_for_like`` repeats which block is the fp32 one.

Prints ONE JSON line on rank 0.  This file parses the arguments and assembles that line; the legs live in ``benchlib/``:
``rollout.py`` (headline, sustained, strong cfg 3, samplers, other arithmetic mode, cfg 5, roofline rows), ``dcae.py`` (BASELINE
configs[0] + the conv roofline row), ``cpu.py`` (the oracle-timed cpu_baseline legs: cfg 1 and cfg 2), ``rccl_world1.py`` (RCCL executed
at world size 1), ``kernel_timer.py`` (HIP-event brackets), ``host.py`` (sysfs clock / power sampler), ``launch.py`` (self-launcher,
process-group bring-up).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

from benchlib.configs import CONFIG_DCAE_84, CONFIGS, PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, model_flops_per_forward  # noqa: E402,F401  (tools/ import these names from here)


def parse_args(argv=None):
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
    ap.add_argument("--cpu-forwards", type=int, default=5, help="oracle forwards timed for cfg 2's cpu_baseline after one warm-up forward; the MEDIAN is reported (0 = skip "
                    "every cpu_baseline leg, cfg 1's too)")
    ap.add_argument("--strong-cfg3", action="store_true", help="also run BASELINE configs[2] (a fixed ensemble of 16 members, 40 lead steps = 10 chained chunks per member) "
                    "as an extra leg and print it as `strong_cfg3` - always on with N > 1, where it is north_star's strong-scaling point")
    ap.add_argument("--no-kernel-timers", action="store_true", help="the timed region, sustained window and strong_cfg3 only: no instrumented step (roofline), no other "
                    "arithmetic mode, no dcae / cfg5 / rccl_world1 blocks")
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
                    "every lead step to 84 x 120 x 240 fields (AR model and DCAE in --precision); not the headline")
    ap.add_argument("--decode-batch-frames", type=int, default=32, help="decoded legs (--decode, the cfg5 block): decode the lead steps after the last chunk in batches of "
                    "up to this many frames (roll_out_serial's decode_batch_frames); 0 = decode every chunk right after it, as the reference")
    ap.add_argument("--workload", default="rollout", choices=["rollout", "dcae"], help="rollout = the headline (AR sampler chunk); dcae = the stand-alone DCAE "
                    "encode / decode sweep (the default line already carries a `dcae` block)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="arithmetic of `value`: fp32 = exact fp32 MFMA, the reference's own arithmetic (default); bf16x3 = split-bf16 (hi*hi+hi*lo+lo*hi, fp32 accumulate; the "
                         "fast mode, inside the 1e-4 budget); bf16 = ONE bf16 MFMA per product, the mixed-precision mode of BASELINE configs[4] (own tolerance)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the leg that runs the same workload in the other arithmetic mode (bf16x3_mode / fp32_mode)")
    ap.add_argument("--no-dcae-block", action="store_true", help="skip the `dcae` block (BASELINE configs[0] on the GPU + its cpu_baseline)")
    ap.add_argument("--no-cfg5-block", action="store_true", help="skip the `cfg5` block (encode -> chunk -> decode in bf16x3 and bf16)")
    ap.add_argument("--no-rccl-world1", action="store_true", help="skip the world-size-1 RCCL child process (N = 1 runs)")
    ap.add_argument("--no-batched-conditioning", action="store_true", help="A/B aid: evaluate the sample-independent part of the network (context refiner, conditioning "
                    "embedding, AdaLN modulation vectors) inside every network evaluation as the reference does, instead of once per chunk as one batch over the "
                    "chunk's noise levels (LaDCastTransformer3DModel.prepare_conditioning)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="after the K timed steps keep stepping until this many seconds of "
                    "back-to-back chunks have run and report that window as `sustained` (clock under sustained load); 0 = skip")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from benchlib.launch import self_launch

        return self_launch(args.gpus)
    if args.workload == "dcae":
        from benchlib.dcae import dcae_workload

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
        from benchlib.launch import init_process_group

        dist = init_process_group(args.backend, dev, world, rank, args.collective_timeout)

    from benchlib.host import gpu_state
    from benchlib.kernel_timer import KernelTimer, bracket_overhead
    from benchlib.rollout import DTYPE_TEXT, Rollout, attention_row, build_roofline
    from ladcast_amd.pipelines.distributed import shard_members
    from ladcast_amd.precision import tolerance

    ro = Rollout(args, dev, rank, world, dist)
    m, lead, R = ro.m, ro.lead, ro.R
    timed = ro.timed_region()
    elapsed = timed["elapsed"]
    sustained = ro.sustained(elapsed / args.steps) if args.sustained_seconds > 0 else None
    strong_cfg3 = ro.strong_cfg3() if ((world > 1 or args.strong_cfg3) and not args.decode and not args.no_strong_cfg3) else None
    extras = not args.no_kernel_timers and rank == 0 and world == 1 and not args.decode  # the secondary legs of the default N = 1 line
    other_sampler, ddim_sampler, host_outputs = ro.alt_samplers() if extras else (None, None, None)
    timer = KernelTimer()
    instrumented_ms, ks, bracket = None, {}, (None, None, None)
    if not args.no_kernel_timers and rank == 0:
        instrumented_ms, ks = ro.instrumented_step(timer)
        try:
            bracket = bracket_overhead(ro.hip, dev)
        except Exception:
            bracket = (None, None, None)
    other_name = {"fp32": "bf16x3", "bf16x3": "fp32", "bf16": "fp32"}[args.precision]
    other_mode = ro.other_mode(other_name, timer, bracket) if (extras and not args.no_other_mode) else None
    cfg5 = ro.cfg5_block() if (extras and not args.no_cfg5_block) else None
    dcae = None
    if extras and not args.no_dcae_block:
        from benchlib.dcae import dcae_block

        dcae = dcae_block(ro.hip, dev, cpu_passes=3 if args.cpu_forwards > 0 else 0)
    rccl = None
    if extras and not args.no_rccl_world1:
        from benchlib import rccl_world1

        rccl = rccl_world1.run()

    chunks = -(-lead // R)
    fwd_per_chunk = (2 * args.solver_steps - 1) if args.sampler == "edm" else args.solver_steps
    me = gpu_state(dev_index)
    rank_devices = [dict(rank=rank, device_index=dev_index, name=me.get("name"), pci_bus_id=me.get("pci_bus_id"), gcn_arch=me.get("gcn_arch"))]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered
    if rank == 0:
        cfg = ro.cfg
        gflops, aflops = model_flops_per_forward(cfg, R)
        (mg, ma), (cg, ca) = model_flops_per_forward(cfg, R, split=True)
        # flops that RAN per member and chunk: the sample-dependent part once per network evaluation, the conditioning path once per
        # noise level when it is batched per chunk (2N - 1 evaluations share N noise levels in the Heun sampler), else per evaluation
        cond_evals = args.solver_steps if ro.model.batch_conditioning else fwd_per_chunk
        flops_per_chunk = fwd_per_chunk * (mg + ma) + cond_evals * (cg + ca)
        value = ro.total_members * lead * args.steps / elapsed
        roof = build_roofline(ks, args.precision, bracket) if ks else None
        dtype = DTYPE_TEXT[args.precision] + (f" (stated tolerance {tolerance('bf16', 'forward'):g} per forward, ladcast_amd/precision.py)" if args.precision == "bf16" else "")
        # which block of this line is measured in the reference's arithmetic (exact fp32), and which is the fast mode
        fp32_src = "top level" if args.precision == "fp32" else ("fp32_mode" if other_mode else None)
        fp32_blk = (dict(value=round(value, 4), ms_per_step=round(1e3 * elapsed / args.steps, 3), steps=args.steps, warmup=args.warmup, roofline=roof)
                    if args.precision == "fp32" else ({k: other_mode[k] for k in ("value", "ms_per_step", "steps", "warmup", "roofline")} if other_mode else None))
        line = {
            "metric": "ensemble-member-steps/sec", "value": round(value, 4), "unit": "member-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "step_ms": timed["step_ms"], "higher_is_better": True,
            "scaling": "strong" if ro.strong else "weak", "vs_baseline": None, "dtype": dtype,
            "value_dtype_note": {
                "fp32": "`value` is measured in exact fp32 (fp32-input MFMA, fp32 accumulate): the reference's own arithmetic, like for like.  The split-bf16 fast mode (inside "
                        "north_star's 1e-4 rel-L2, ~16 mantissa bits per product) is the `bf16x3_mode` block - a different, narrower arithmetic, never `value`.",
                "bf16x3": "`value` is measured in the split-bf16 FAST mode (hi*hi + hi*lo + lo*hi, fp32 accumulate: ~16 mantissa bits per product, inside north_star's 1e-4 rel-L2) - "
                          "NARROWER than the reference's fp32; the like-for-like figure is `like_for_like` / `fp32_mode` (run without --precision for it as `value`).",
                "bf16": "`value` is measured in the single-term bf16 mode (BASELINE configs[4]'s mixed precision, own tolerance) - narrower than the reference's fp32; the "
                        "like-for-like figure is `like_for_like` / `fp32_mode`.",
            }[args.precision],
            "like_for_like": None if fp32_blk is None else dict(dtype="f32", source=fp32_src, **fp32_blk),
            "data": "synthetic",
            "config": {
                "workload": ("cfg5-style END-TO-END (DCAE encode -> AR -> DCAE decode of every lead step), " if args.decode else "") +
                            f"cfg2: {args.model} AR transformer, " + (f"a FIXED ensemble of {ro.total_members} member(s) over {world} GPU(s)" if ro.strong else f"{m} member/GPU") +
                            f", {args.solver_steps} solver steps ({args.sampler}: {fwd_per_chunk} forwards/chunk), "
                            f"{lead} lead step(s) = {chunks} chunk(s) of return_seq_len {R}, latent 84x15x30, fp32 weights random-init seed 1234, arithmetic {args.precision}",
                "sampler": args.sampler, "members_per_gpu": m if not ro.strong else None, "ensemble_size": ro.total_members,
                "members_on_rank": [len(shard_members(ro.total_members, r, world)) for r in range(world)], "lead_steps": lead, "return_seq_len": R, "forwards_per_step": chunks * fwd_per_chunk,
                "tflop_per_forward_per_member": round((gflops + aflops) / 1e12, 4),
                "conditioning_path": ("per chunk: the sample-independent part of every forward (context refiner, conditioning embedding, AdaLN modulation GEMV: "
                                      "0.026 of the 0.979 TFLOP and 17 of the 52 launches of a 375M forward) runs once per chunk as ONE batch over the chunk's "
                                      f"{args.solver_steps} noise levels, inside the timed region, every chunk; --no-batched-conditioning runs it per evaluation")
                if ro.model.batch_conditioning else "per network evaluation (as the reference)",
                "outputs": ("left in HBM (with N > 1: gathered there); no per-step host copy, so the host prepares step k + 1 while the GPU runs step k - the "
                            "timed region ends with barrier + synchronise (two alternating instances of the captured chunk); --host-outputs copies every step's result to the host as the reference's "
                            "roll_out_serial does") if ro.out_dev is not None else "copied to the host after every step (as the reference)",
            },
            "gpu": timed["gpu"],
            "strong_cfg3": strong_cfg3,
            "instrumented_ms_per_step": None if instrumented_ms is None else round(instrumented_ms, 3),
            "other_sampler": other_sampler,
            "ddim_sampler": ddim_sampler,
            "host_outputs": host_outputs,
            "sustained": sustained,
            f"{other_name}_mode": other_mode,
            "cfg5": cfg5,
            "dcae": dcae,
            "rccl_world1": rccl,
            "ranks": {"world_size": world if dist is None else dist.get_world_size(), "backend": "none" if dist is None else args.backend, "devices": rank_devices,
                      "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if (dist is not None and args.backend == "nccl")
                                       else (rccl or {}).get("rccl_version")),
                      "rccl_version_source": ("this run's process group" if (dist is not None and args.backend == "nccl")
                                              else ("the world-size-1 nccl child process (rccl_world1)" if (rccl or {}).get("rccl_version") else None)),
                      "visible_gpus": torch.cuda.device_count(), "per_rank": timed["rank_stats"]},
            "model_tflops": round(ro.total_members * chunks * flops_per_chunk * args.steps / elapsed / 1e12, 2),
            "model_tflops_note": f"algorithmic flops that ran: {fwd_per_chunk} x {round((mg + ma) / 1e12, 4)} TFLOP (sample-dependent part) + {cond_evals} x "
                                 f"{round((cg + ca) / 1e12, 4)} TFLOP (conditioning path) per member and chunk",
            "roofline": roof,
            "attention_kernel": attention_row(ks, args.precision) if ks else None,
        }
        if args.cpu_forwards > 0 and world == 1:
            from benchlib.cpu import ar_baseline

            line["cpu_baseline"] = ar_baseline(args.model, R, args.cpu_forwards, chunks * fwd_per_chunk, lead)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
            line["cpu_baseline_cfg1"] = None if not dcae else dcae.get("cpu_baseline")
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
