from .models import (BaseLightningModel, DecoderRNN, EncoderRNN, Seq2SeqRNN,  # noqa: F401
                     TemporalConv, cmat_acc)
