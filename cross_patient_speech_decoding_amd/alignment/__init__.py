"""Latent alignment (CCA / MCCA / joint PCA / PCA) on the MI355X behind the reference's
sklearn-style surfaces.  Importing needs no GPU; fitting does."""
from .AlignCCA import AlignCCA, CCA_align  # noqa: F401
from .AlignMCCA import AlignMCCA  # noqa: F401
from .JointPCA import JointPCA  # noqa: F401
from .alignment_utils import cnd_avg, extract_group_conditions, label2str  # noqa: F401
from .pca import PCA  # noqa: F401
