"""GMLight's geometric mover's loss (reference ``RegressionNetwork/gmloss``): the Sinkhorn divergence of ``geomloss``
over DEPTH-SCALED anchors, whose chord matrix changes with every scene."""
from .samples_loss import SamplesLoss, geometric_points

__all__ = ["SamplesLoss", "geometric_points"]
