from .cross_pt_decoders import (crossPtDecoder, crossPtDecoder_jointDimRed, crossPtDecoder_mcca,  # noqa: F401
                                crossPtDecoder_sepAlign, crossPtDecoder_sepDimRed)
from .svm import SVC  # noqa: F401,E402
