from .scheduling_edm_dpmsolver_multistep import EDMDPMSolverMultistepScheduler  # noqa: F401
from .scheduling_ddim_ddpm import DDIMScheduler, DDPMScheduler  # noqa: F401
