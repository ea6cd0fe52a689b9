"""Shape guards used by the ops (interface of lib/model_zoo/stylegan_utils/misc.py:19-33)."""
import contextlib


@contextlib.contextmanager
def suppress_tracer_warnings():
    """Kept for call-site compatibility; nothing here is ever traced."""
    yield


def assert_shape(tensor, ref_shape):
    """Raise AssertionError unless ``tensor.shape`` matches ``ref_shape`` (None = wildcard)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for d, (have, want) in enumerate(zip(tensor.shape, ref_shape)):
        if want is not None and int(have) != int(want):
            raise AssertionError(f'Wrong size for dimension {d}: got {have}, expected {want}')
