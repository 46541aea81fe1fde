from .DCAE import AutoencoderDC  # noqa: F401
from .LaDCast_3D_model import LaDCastTransformer3DModel  # noqa: F401
from .sphere_conv import SphereConv2d  # noqa: F401
