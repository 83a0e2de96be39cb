"""Padding arithmetic (host-side shape logic) with the reference's names (blvm/utils/padding.py:72-117)."""


def get_modulo_padding(length: int, stride: int, kernel_size: int = 0, dilation: int = 1, pad_to_kernel_size: bool = False) -> int:
    """`p` such that `(length + p - kernel_size) % stride == 0` (padding.py:72-91)."""
    if dilation > 1:
        raise NotImplementedError(f"Dilation greater than 1 not yet supported but got {dilation=}.")
    if length < kernel_size:
        if pad_to_kernel_size:
            return kernel_size - length
        raise ValueError(f"Input {length=} was shorter than {kernel_size=} and {pad_to_kernel_size=}.")
    missing = (length - kernel_size) % stride
    return stride - missing if missing else 0


def get_modulo_length(length: int, stride: int, kernel_size: int = 0) -> int:
    """Smallest number >= `length` that a (kernel_size, stride) window tiles exactly (padding.py:94-97)."""
    return length + get_modulo_padding(length, stride, kernel_size)


def get_same_padding(length: int, stride: int, kernel_size: int, dilation: int = 1) -> int:
    """Padding that makes the convolved length `ceil(length / stride)` (padding.py:100-117)."""
    return max(0, dilation * (kernel_size - 1) - (length - 1) % stride)
