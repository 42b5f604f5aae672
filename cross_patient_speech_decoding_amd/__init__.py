"""MI355X-native hot path of coganlab/cross_patient_speech_decoding: latent alignment
(CCA / MCCA / joint PCA) and the seq2seq GRU trainer, behind the reference's own Python
surfaces.  All arithmetic on the path runs in libxps.so (hand-written HIP for gfx950,
C ABI in include/xps.h); importing the package does not need a GPU, calling it does."""
__version__ = '0.1.0'
