from .edm_sampler import edm_AR_sampler  # noqa: F401
from .pipeline_AR import AutoRegressive2DPipeline  # noqa: F401
from .utils import (  # noqa: F401
    Fields2DPipelineOutput,
    decode_latent_ens,
    ensemble_AR_sampler,
    roll_out_serial,
)
from .io import latent_file_name, list_latent_files, load_latent_npy, save_latent_npy  # noqa: F401
from .encode import encode_latents  # noqa: F401
