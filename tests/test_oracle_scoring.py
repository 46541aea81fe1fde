"""The scoring oracle (oracle/scoring.py) against outputs of the reference's own functions
(tests/golden/scoring_ref.npz, made by tests/golden/make_golden.py from ladcast/evaluate/utils.py:9-149): bit for bit."""
import numpy as np
import torch

from oracle import scoring as S


def _eq(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return a.shape == b.shape and bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


def test_scoring_oracle_matches_reference_outputs(golden_dir):
    z = np.load(f"{golden_dir}/scoring_ref.npz")
    n = len([k for k in z.files if k.endswith("_fc")])
    assert n == 4
    for i in range(n):
        fc, tr, cl, lat = (torch.from_numpy(z[f"s{i}_{k}"]) for k in ("fc", "tr", "cl", "lat"))
        w_cos = S.get_normalized_lat_weights_based_on_cos(lat)
        assert _eq(w_cos, z[f"s{i}_w_cos"])
        assert _eq(S.get_lat_weights_from_lat_tensor(lat[None])[0], z[f"s{i}_w_area"])
        assert _eq(S.pointwise_crps_skill(fc, tr.unsqueeze(0), 0), z[f"s{i}_skill"])
        assert _eq(S.pointwise_crps_spread(fc, 0), z[f"s{i}_spread"])
        assert _eq(S.get_crps(fc, tr.unsqueeze(0), 0), z[f"s{i}_crps"])
        assert _eq(S.get_acc(fc.mean(dim=0), tr, cl, w_cos.view(1, -1, 1)), z[f"s{i}_acc_w"])
        assert _eq(S.get_acc(fc.mean(dim=0), tr, cl, None), z[f"s{i}_acc"])


def test_ensemble_scores_block_is_the_composition(golden_dir):
    """the fused per-lead-time block (evaluate_ens_gpu.py:339-425) restated from the pinned pieces"""
    z = np.load(f"{golden_dir}/scoring_ref.npz")
    fc, tr, cl, lat = (torch.from_numpy(z[f"s0_{k}"]) for k in ("fc", "tr", "cl", "lat"))
    w = S.get_normalized_lat_weights_based_on_cos(lat)
    r = S.ensemble_scores(fc, tr, cl, w, sst_channel=1)
    wv = w.view(1, -1, 1)
    assert _eq(r["ens_acc"], torch.from_numpy(z["s0_acc_w"]))
    skill = torch.from_numpy(z["s0_skill"]) * wv
    assert _eq(r["crps_skill"][0], skill[0].mean()) and _eq(r["crps_skill"][1], torch.nanmean(skill[1])) and _eq(r["crps_skill"][2], skill[2].mean())
    assert torch.isfinite(r["crps"][1]) and torch.isfinite(r["ens_mse"][1])  # nanmean skips the land points
