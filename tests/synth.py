"""Seeded synthetic configs / weights / inputs shared by tests, smoke and bench
(weights = default torch init under manual_seed(1234), IC latent
0.5*N(0,1) seed 2, fields N(0,1) seed 0/1, timestamp 2018010100)."""
import torch

from oracle.ar_model import CONFIG_375M, LaDCastTransformer3DModel
from oracle.dcae import CONFIG_DCAE_84, AutoencoderDC


def tiny_ar_config(heads=2, layers=1, single=1, refiner=1):
    return dict(CONFIG_375M, num_attention_heads=heads, num_layers=layers, num_single_layers=single, num_refiner_layers=refiner)


def tiny_dcae_config():
    return dict(
        CONFIG_DCAE_84,
        in_channels=13,
        out_channels=13,
        latent_channels=8,
        encoder_block_out_channels=(32, 64, 64, 128),
        decoder_block_out_channels=(32, 64, 64, 128),
        encoder_layers_per_block=(1, 1, 1, 1),
        decoder_layers_per_block=(1, 1, 1, 1),
    )


def randomize_norms(model, seed=4321):
    """Default init leaves every norm weight at 1 and bias at 0, which would hide a
    dropped affine term; perturb all 1-D parameters deterministically."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return model


def make_ar(cfg, seed=1234):
    torch.manual_seed(seed)
    return randomize_norms(LaDCastTransformer3DModel.from_config(cfg)).eval()


def make_dcae(cfg, seed=1234):
    torch.manual_seed(seed)
    return randomize_norms(AutoencoderDC.from_config(cfg)).eval()


def synth_known(batch=1, t_in=1, seed=2):
    return 0.5 * torch.randn(batch, 84, t_in, 15, 30, generator=torch.Generator().manual_seed(seed))


def synth_field(batch, channels, h, w, seed=0):
    return torch.randn(batch, channels, h, w, generator=torch.Generator().manual_seed(seed))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()
