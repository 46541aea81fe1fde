"""The PRODUCT's sampler path (HIP state-update kernels, scheduler, ensemble driver) against outputs of the reference's own
sampler code (tests/golden/sampler_ref.npz, see tests/test_oracle_reference_pins.py): the same elementwise toy network stands in
for the transformer on both sides, so what is compared is exactly pipelines/edm_sampler.py:60-113, pipelines/pipeline_AR.py:77-102
and pipelines/utils.py:682-741 as executed by the reference vs by ladcast_amd on the GPU.  Tolerance 1e-6 rel-L2 (the toy network's
torch ops run on the GPU here and on the CPU there; the fp64 / fp32 update kernels themselves are bit-exact, test_gpu_ops.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden.make_golden import ToyNet  # noqa: E402


def _gens(n):
    return [torch.Generator("cpu").manual_seed(k) for k in range(n)]


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.double()
    return ((a - b).norm() / b.norm()).item()


def test_product_samplers_reproduce_the_reference_code(golden_dir):
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, edm_AR_sampler, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    z = np.load(f"{golden_dir}/sampler_ref.npz")
    z = {k: torch.from_numpy(z[k]) for k in z.files}
    net = ToyNet(6, device="cuda")
    ts = torch.tensor([2018010100]).cuda()
    known1, known3 = z["known1"].cuda(), z["known3"].cuda()
    got = edm_AR_sampler(net, EDMDPMSolverMultistepScheduler(), batch_size=3, return_seq_len=2, num_inference_steps=5, known_latents=known3, timestamps=ts,
                         generator=_gens(3), device="cuda")
    assert _rel(got, z["edm_n5"]) < 1e-6
    got = edm_AR_sampler(net, EDMDPMSolverMultistepScheduler(), batch_size=1, return_seq_len=4, num_inference_steps=1, known_latents=known1, timestamps=None,
                         generator=_gens(1), device="cuda")
    assert _rel(got, z["edm_n1"]) < 1e-6
    pipe = AutoRegressive2DPipeline(net, EDMDPMSolverMultistepScheduler())
    got = pipe(batch_size=3, return_seq_len=2, known_latents=known3, timestamps=ts, generator=_gens(3), num_inference_steps=6, return_dict=False)[0]
    assert _rel(got, z["pipe_n6"]) < 1e-6
    got = pipe(batch_size=1, return_seq_len=1, known_latents=known1, timestamps=ts, generator=_gens(1), num_inference_steps=20).fields
    assert _rel(got, z["pipe_n20"]) < 1e-6
    got = ensemble_AR_sampler(pipe, 5, 3, 4, known_latents=known1, timestamps=ts, batch_size=2, sampler_type="edm", device="cuda")
    assert _rel(got, z["ens_edm"]) < 1e-6
    got = ensemble_AR_sampler(pipe, 4, 2, 4, known_latents=known1, timestamps=ts, batch_size=3, sampler_type="pipeline", device="cuda")
    assert _rel(got, z["ens_pipe"]) < 1e-6
