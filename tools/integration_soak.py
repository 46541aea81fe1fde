"""One full-scale integration run on ONE GPU at the literal scale of BASELINE configs[2] + configs[4], through the
driver-counterpart entry point `ladcast_amd.evaluate.pred_rollout.run_rollout`:

    synthetic 84 x 120 x 240 IC field (+ 5 static fields)
      -> DCAE encode (inside run_rollout)
      -> 16 members x 40 lead steps (10 chunks of 4; 375M AR model, 20-step Heun = 39 forwards per chunk)
      -> latent_YYYYMMDDHH.npy  (16, 84, 41, 15, 30), the reference's file layout (evaluate/pred_rollout.py:420-430)
      -> load_latent_npy -> DCAE decode of all 16 x 40 = 640 frames (decode_latent_ens, 40 frames per launch batch)
      -> ldc_ensemble_scores per lead time (ensemble-mean MSE / ACC / CRPS against a synthetic truth and climatology)

and records peak HBM, wall time per phase, all-finite checks and the file layout as one JSON object.
usage: python tools/integration_soak.py [--precision bf16x3|bf16|fp32] [--members 16] [--lead-steps 40] > profiles/r04_*_integration_cfg3_cfg5.json"""
import argparse, json, os, sys, tempfile, time
from datetime import datetime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ladcast_amd.evaluate import ensemble_scores, get_normalized_lat_weights_based_on_cos
from ladcast_amd.evaluate.pred_rollout import SST_CHANNEL, run_rollout
from ladcast_amd.models import AutoencoderDC, LaDCastTransformer3DModel
from ladcast_amd.pipelines import AutoRegressive2DPipeline, decode_latent_ens, list_latent_files, load_latent_npy
from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--members", type=int, default=16)
ap.add_argument("--lead-steps", type=int, default=40)
ap.add_argument("--sampler", default="edm")
args = ap.parse_args()
dev = torch.device("cuda", 0)
res = {"workload": f"run_rollout on one MI355X: DCAE encode -> {args.members} members x {args.lead_steps} lead steps (375M, 20 solver steps, {args.sampler}) -> "
                   f"latent .npy -> decode of all {args.members * args.lead_steps} frames -> ensemble scores per lead time; arithmetic {args.precision}; "
                   "synthetic field / static fields / truth, random-init weights (seed 1234)"}


def phase(name, t0):
    torch.cuda.synchronize()
    res.setdefault("seconds", {})[name] = round(time.perf_counter() - t0, 3)
    res.setdefault("peak_hbm_gib_after", {})[name] = round(torch.cuda.max_memory_allocated() / 2**30, 3)


t0 = time.perf_counter()
torch.manual_seed(1234)
ar = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to(dev).eval().set_gemm_precision(args.precision).enable_hip_graph(True)
ae = AutoencoderDC.from_config(bench.CONFIG_DCAE_84).to(dev).eval().set_gemm_precision(args.precision).enable_hip_graph(True)
pipe = AutoRegressive2DPipeline(ar, EDMDPMSolverMultistepScheduler())
g = torch.Generator().manual_seed(3)
field = torch.randn(84, 1, 120, 240, generator=g)
static = torch.randn(5, 120, 240, generator=g)
truth = torch.randn(84, args.lead_steps + 1, 120, 240, generator=g)
clim = 0.3 * torch.randn(84, 120, 240, generator=g)
truth[SST_CHANNEL, :, :30, :40] = float("nan")  # land points of the sea-surface-temperature channel: the nanmean rule of the scoring
# latent statistics of the kind the reference's JSON holds: here the IC latent's own per-channel mean / std
z = ae.encode(field.permute(1, 0, 2, 3).to(dev), static_conditioning_tensor=static.unsqueeze(0).to(dev)).latent[0]
targs = {"mean": z.mean(dim=(1, 2)).tolist(), "std": z.std(dim=(1, 2)).tolist(), "target_std": 0.5}
phase("build_models_and_inputs", t0)

out_dir = tempfile.mkdtemp(prefix="ladcast_soak_")
init = datetime(2018, 1, 1, 0)
t0 = time.perf_counter()
lat = run_rollout(lambda t: field, [init], pipe, ae, targs, normalization_param_dict={"mean": torch.zeros(84), "std": torch.ones(84)},
                  static_conditioning_tensor=static, output=out_dir, ensemble_size=args.members, num_inference_steps=20, return_seq_len=4,
                  total_lead_time_hour=6 * args.lead_steps, sampler_type=args.sampler, save_as_latent=True, device=dev)
phase("encode_rollout_save (run_rollout, includes graph capture)", t0)
t0 = time.perf_counter()
lat2 = run_rollout(lambda t: field, [init], pipe, ae, targs, normalization_param_dict={"mean": torch.zeros(84), "std": torch.ones(84)},
                   static_conditioning_tensor=static, output=out_dir, ensemble_size=args.members, num_inference_steps=20, return_seq_len=4,
                   total_lead_time_hour=6 * args.lead_steps, sampler_type=args.sampler, save_as_latent=True, device=dev)
phase("encode_rollout_save, second call (steady state)", t0)
steady = res["seconds"]["encode_rollout_save, second call (steady state)"]
res["rollout_member_steps_per_s"] = round(args.members * args.lead_steps / steady, 2)
res["second_call_bit_identical"] = bool(torch.equal(lat[0], lat2[0]))
files = list_latent_files(out_dir)
res["files"] = [os.path.basename(path) for _, path in files]
t0 = time.perf_counter()
arr, ens = load_latent_npy(os.path.join(out_dir, res["files"][0]), device=dev)
res["latent_file_shape"] = list(arr.shape)
res["latent_file_mib"] = round(os.path.getsize(os.path.join(out_dir, res["files"][0])) / 2**20, 1)
res["latents_all_finite"] = bool(torch.isfinite(arr).all())
res["latents_equal_returned_tensor"] = bool(torch.equal(arr.cpu().reshape(lat[0].shape), lat[0].cpu()))
phase("load_latent_npy", t0)

t0 = time.perf_counter()
x = arr.reshape(args.members, 84, args.lead_steps + 1, 15, 30)
dec = torch.empty(args.members, 84, args.lead_steps + 1, 120, 240, device=dev)
for k in range(args.members):  # one member's 41 frames per decode call (the reference decodes member by member too, evaluate_ens_gpu.py:300-330)
    dec[k : k + 1] = decode_latent_ens(ae, x[k : k + 1])
phase("decode_all_frames", t0)
res["decoded_frames"] = args.members * (args.lead_steps + 1)
res["decoded_shape"] = list(dec.shape)
res["decoded_all_finite"] = bool(torch.isfinite(dec).all())
res["decode_ms_per_frame"] = round(1e3 * res["seconds"]["decode_all_frames"] / res["decoded_frames"], 3)

t0 = time.perf_counter()
w = get_normalized_lat_weights_based_on_cos(torch.linspace(-88.5, 90.0, 120))
truth_d, clim_d = truth.to(dev), clim.to(dev)
scores = []
for t in range(1, args.lead_steps + 1):
    s = ensemble_scores(dec[:, :, t], truth_d[:, t], clim_d, w, SST_CHANNEL)
    scores.append({k: v for k, v in s.items()})
phase("ensemble_scores_all_lead_times", t0)
res["scores_all_finite"] = bool(all(torch.isfinite(v).all() for s in scores for v in s.values()))
res["scores_lead_step_1_and_last_channel_mean"] = {k: [round(scores[0][k].mean().item(), 4), round(scores[-1][k].mean().item(), 4)] for k in scores[0]}
res["score_ms_per_lead_time"] = round(1e3 * res["seconds"]["ensemble_scores_all_lead_times"] / args.lead_steps, 3)
res["peak_hbm_gib"] = round(torch.cuda.max_memory_allocated() / 2**30, 3)
res["hbm_reserved_gib"] = round(torch.cuda.memory_reserved() / 2**30, 3)
res["ok"] = bool(res["latents_all_finite"] and res["decoded_all_finite"] and res["scores_all_finite"] and res["latents_equal_returned_tensor"])
print(json.dumps(res, indent=1))
sys.exit(0 if res["ok"] else 1)
