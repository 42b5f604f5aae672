from .ctc_decoder import greedy_decode_batch  # noqa: F401
from .realtime_nn_model import DenseClassifier, RealtimeRNNModel, StackedRNN, StreamingDecoder  # noqa: F401
