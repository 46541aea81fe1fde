from .DCAE import AutoencoderDC, SanaMultiscaleAttnProcessor2_0  # noqa: F401
from .LaDCast_3D_model import LaDCastAttnProcessor2_0, LaDCastTransformer3DModel  # noqa: F401
from .sphere_conv import SphereConv2d  # noqa: F401
