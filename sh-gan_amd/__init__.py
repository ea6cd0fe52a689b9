"""MI355X-native generator forward path of SH-GAN (package directory ``sh-gan_amd``; import it as
``shgan_amd`` through the alias module at the repository root)."""
from . import _lib  # noqa: F401

__all__ = ['_lib', 'kernels', 'model_zoo', 'eval_harness', 'configs', 'masks', 'fid_stats', 'grad_sync', 'losses', 'train_stage', 'datasets']
