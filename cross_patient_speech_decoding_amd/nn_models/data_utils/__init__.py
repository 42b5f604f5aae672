"""Data modules and batch augmentations of the seq2seq training path (host-side counterparts of nn_models/data_utils)."""
