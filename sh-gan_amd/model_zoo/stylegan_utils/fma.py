"""a*b+c (interface of lib/model_zoo/stylegan_utils/fma.py:15); forward only, on the HIP kernel."""
from ... import kernels


def fma(a, b, c):  # => a * b + c with NumPy-style broadcasting
    return kernels.fma(a, b, c)
