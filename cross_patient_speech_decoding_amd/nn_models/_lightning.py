"""LightningModule base: the real ``lightning`` class when it is installed (the reference's
environment), otherwise a minimal stand-in with the hooks the reference's modules use
(log / log_dict / save_hyperparameters), driven by ``trainer.Trainer``."""
import torch.nn as nn

try:                                        # pragma: no cover - not installed in the build image
    import lightning as _L
    LightningModule = _L.LightningModule
    HAVE_LIGHTNING = True
except ImportError:
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._xps_logged = {}
            self.trainer = None

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                import torch
                return torch.device('cpu')

        def log(self, name, value, *args, **kwargs):
            self._xps_logged[name] = value.detach() if hasattr(value, 'detach') else value

        def log_dict(self, d, *args, **kwargs):
            for k, v in d.items():
                self.log(k, v)

        def save_hyperparameters(self, *args, **kwargs):
            pass
