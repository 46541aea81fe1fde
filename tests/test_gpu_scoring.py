"""HIP ensemble scoring (ladcast_amd.evaluate, C ABI ldc_ensemble_scores) against the pinned oracle (oracle/scoring.py)
and the reference outputs in tests/golden/scoring_ref.npz.  Tolerance: 1e-5 relative per metric (fp32 sums in a
different order than torch's reductions); point maps 2e-6."""
import numpy as np
import pytest
import torch

from oracle import scoring as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import ladcast_amd.evaluate as ev
    return ev


def _close(a, b, tol):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape
    nan_a, nan_b = torch.isnan(a), torch.isnan(b)
    assert bool((nan_a == nan_b).all()), "NaN pattern differs"
    a, b = a[~nan_a], b[~nan_b]
    if a.numel() == 0:
        return
    assert ((a - b).abs() <= tol * (b.abs() + b.abs().mean())).all(), float(((a - b).abs() / (b.abs() + b.abs().mean())).max())


def test_scoring_against_reference_outputs(E, golden_dir):
    z = np.load(f"{golden_dir}/scoring_ref.npz")
    for i in range(4):
        fc, tr, cl, lat = (torch.from_numpy(z[f"s{i}_{k}"]) for k in ("fc", "tr", "cl", "lat"))
        d = lambda t: t.cuda()
        _close(E.pointwise_crps_skill(d(fc), d(tr).unsqueeze(0), 0), z[f"s{i}_skill"], 2e-6)
        _close(E.pointwise_crps_spread(d(fc), 0), z[f"s{i}_spread"], 2e-5)  # sum of +- terms: absolute scale of the members
        _close(E.get_crps(d(fc), d(tr).unsqueeze(0), 0), z[f"s{i}_crps"], 2e-5)
        w = E.get_normalized_lat_weights_based_on_cos(lat)
        _close(w, z[f"s{i}_w_cos"], 0)
        _close(E.get_lat_weights_from_lat_tensor(lat[None])[0], z[f"s{i}_w_area"], 0)
        _close(E.get_acc(d(fc.mean(dim=0)), d(tr), d(cl), d(w).view(1, -1, 1)), z[f"s{i}_acc_w"], 1e-5)
        _close(E.get_acc(d(fc.mean(dim=0)), d(tr), d(cl), None), z[f"s{i}_acc"], 1e-5)


@pytest.mark.parametrize("M,C,H,W,sst", [(5, 3, 6, 8, 1), (16, 7, 33, 17, 0), (50, 84, 120, 240, 68), (64, 2, 30, 60, -1), (1, 2, 5, 4, 1), (2, 1, 3, 3, 0)])
def test_ensemble_scores_block(E, M, C, H, W, sst):
    """one lead time of evaluate_ens_gpu.py:339-425, on a [:, :, t] view of the (ens, C, T, H, W) array (no copy), with
    land NaNs in the SST channel; full-size case = BASELINE field 84 x 120 x 240 with 50 members"""
    g = torch.Generator().manual_seed(3)
    T = 3
    dec = torch.randn(M, C, T, H, W, generator=g) * 2 + 0.5
    ref = torch.randn(C, T, H, W, generator=g)
    clim = torch.randn(C, T, H, W, generator=g) * 0.3
    if sst >= 0:
        land = torch.rand(H, W, generator=g) < 0.3
        dec[:, sst][:, :, land] = float("nan")
        ref[sst][:, land] = float("nan")
    lat = torch.linspace(-89.0, 89.0, H)
    w = S.get_normalized_lat_weights_based_on_cos(lat)
    t = 1
    want = S.ensemble_scores(dec[:, :, t], ref[:, t], clim[:, t], w, sst_channel=max(sst, 0)) if sst >= 0 else None
    if want is None:  # no NaN channel: plain means everywhere
        want = S.ensemble_scores(dec[:, :, t], ref[:, t], clim[:, t], w, sst_channel=0)
    dd, dr, dc = dec.cuda(), ref.cuda(), clim.cuda()
    got = E.ensemble_scores(dd[:, :, t], dr[:, t], dc[:, t], w.cuda(), sst_channel=sst)
    for k in ("ens_acc", "ens_mse", "crps_spread", "crps_skill", "crps"):
        _close(got[k], want[k], 1e-5)
    got2 = E.ensemble_scores(dd[:, :, t], dr[:, t], dc[:, t], w.cuda(), sst_channel=sst)
    for k in got:
        assert torch.equal(torch.nan_to_num(got[k]), torch.nan_to_num(got2[k]))  # fixed reduction order


def test_nan_outside_the_nan_channel_propagates(E):
    g = torch.Generator().manual_seed(5)
    dec = torch.randn(4, 2, 6, 8, generator=g)
    ref, clim = torch.randn(2, 6, 8, generator=g), torch.randn(2, 6, 8, generator=g)
    dec[2, 1, 3, 3] = float("nan")
    w = torch.ones(6)
    want = S.ensemble_scores(dec, ref, clim, w, sst_channel=0)
    got = E.ensemble_scores(dec.cuda(), ref.cuda(), clim.cuda(), w.cuda(), sst_channel=0)
    assert torch.isnan(want["crps"][1]) and torch.isnan(got["crps"][1].cpu()) and torch.isnan(got["ens_mse"][1].cpu())
    for k in got:
        _close(got[k], want[k], 1e-5)


def test_scoring_rejects_host_tensors_and_large_ensembles(E):
    with pytest.raises(RuntimeError):
        E.ensemble_scores(torch.zeros(2, 1, 4, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4, 4), torch.ones(4), 0)
    with pytest.raises(RuntimeError):
        E.ensemble_scores(torch.zeros(65, 1, 4, 4).cuda(), torch.zeros(1, 4, 4).cuda(), torch.zeros(1, 4, 4).cuda(), torch.ones(4).cuda(), 0)
