from .cross_pt_decoders import (crossPtDecoder, crossPtDecoder_jointDimRed, crossPtDecoder_mcca,  # noqa: F401
                                crossPtDecoder_sepAlign, crossPtDecoder_sepDimRed)
