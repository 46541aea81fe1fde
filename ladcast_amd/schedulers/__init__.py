from .scheduling_edm_dpmsolver_multistep import EDMDPMSolverMultistepScheduler  # noqa: F401
