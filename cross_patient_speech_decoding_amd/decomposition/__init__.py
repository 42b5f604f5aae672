from .DimRedReshape import DimRedReshape  # noqa: F401
from .NoCenterPCA import NoCenterPCA  # noqa: F401
