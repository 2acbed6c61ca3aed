"""Mirror of the reference's ``RegressionNetwork/geomloss`` package (``from geomloss import SamplesLoss``)."""
from .samples_loss import SamplesLoss

__all__ = ["SamplesLoss"]
