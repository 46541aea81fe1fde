from .LaDCast_3D_model import LaDCastTransformer3DModel  # noqa: F401
