"""Drop-in for the reference's ``src/networks`` package (same module / class names)."""
from .GAN import (BoundaryDiscriminator, BoundaryEntDiscriminator, Discriminator,  # noqa: F401
                  OutputDiscriminator, UncertaintyDiscriminator)
from .PointNetCls import (PointNetCls, PointNetfeat, STN3d, STNkd,  # noqa: F401
                          feature_transform_regularizer)
from .unet import (Bottleneck, Decoder, Encoder, PointNet, Segmentation_model,  # noqa: F401
                   Segmentation_model_Point)
