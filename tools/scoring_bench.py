"""ensemble scoring of one lead time at the BASELINE field size: fused HIP kernel vs the torch op sequence of the
reference (evaluate_ens_gpu.py:339-425) run on the same GPU, and the kernel's HBM rate.  usage: python tools/scoring_bench.py [ens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.evaluate as E

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C, H, W, sst = 84, 120, 240, 68
g = torch.Generator().manual_seed(0)
dec = (torch.randn(M, C, H, W, generator=g) * 2).cuda(); ref = torch.randn(C, H, W, generator=g).cuda(); clim = torch.randn(C, H, W, generator=g).cuda()
w = E.get_normalized_lat_weights_based_on_cos(torch.linspace(-89, 89, H)).cuda()

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

def torch_ops():  # the same quantities as a sequence of torch ops on the device (what the reference driver issues)
    wv = w.view(1, -1, 1)
    mean_t = dec.mean(dim=0)
    fa, ta = mean_t - clim, ref - clim
    (fa * ta * wv).nanmean(dim=(-2, -1)) / torch.sqrt((fa**2 * wv).nanmean(dim=(-2, -1)) * (ta**2 * wv).nanmean(dim=(-2, -1)))
    ((mean_t - ref) ** 2 * wv).mean(dim=(1, 2))
    srt, _ = torch.sort(dec, dim=0)
    wts = (2 * torch.arange(1, M + 1, device=dec.device, dtype=dec.dtype) - M - 1).view(-1, 1, 1, 1)
    sp = 2 * (srt * wts).sum(dim=0) / (M * (M - 1)) * wv
    sk = torch.abs(ref.unsqueeze(0) - dec).mean(dim=0) * wv
    (sk - 0.5 * sp).mean(dim=(1, 2)); sp.mean(dim=(1, 2)); sk.mean(dim=(1, 2))

t_hip = timed(lambda: E.ensemble_scores(dec, ref, clim, w, sst))
t_torch = timed(torch_ops, 5)
nbytes = dec.numel() * 4 + 2 * ref.numel() * 4
print(f"ens={M}: ldc_ensemble_scores {t_hip * 1e6:.1f} us = {nbytes / t_hip / 1e12:.2f} TB/s of algorithmic bytes ({nbytes / 1e6:.0f} MB read once); "
      f"torch op sequence on the same GPU {t_torch * 1e6:.1f} us ({t_torch / t_hip:.1f}x)")
