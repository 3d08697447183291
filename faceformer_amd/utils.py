"""dtype helpers of the decode path (mirror of reference `faceformer/utils.py`).

`min_value_of_dtype` is the pointer head's mask-fill value (reference `utils.py:16-20`, used at
`models/model.py:165` and `models/model_para.py:177`): `finfo(dtype).min`, NOT -inf.  The HIP pointer
kernel hard-codes the fp32 instance of it (-FLT_MAX).
"""
import torch

__all__ = ["info_value_of_dtype", "min_value_of_dtype", "max_value_of_dtype",
           "tiny_value_of_dtype", "flatten_list"]


def info_value_of_dtype(dtype):
    """finfo / iinfo of a torch dtype; bool is rejected like in the reference."""
    if dtype == torch.bool:
        raise TypeError("Does not support torch.bool")
    return torch.finfo(dtype) if dtype.is_floating_point else torch.iinfo(dtype)


def min_value_of_dtype(dtype):
    return info_value_of_dtype(dtype).min


def max_value_of_dtype(dtype):
    return info_value_of_dtype(dtype).max


_TINY = {torch.float: 1e-13, torch.double: 1e-13, torch.half: 1e-4}


def tiny_value_of_dtype(dtype):
    """Small positive constant used against division by zero (fp16: 1e-4, fp32/fp64: 1e-13)."""
    if not dtype.is_floating_point:
        raise TypeError("Only supports floating point dtypes.")
    if dtype not in _TINY:
        raise TypeError("Does not support dtype " + str(dtype))
    return _TINY[dtype]


def flatten_list(nested):
    """[[a, b], [c]] -> [a, b, c]"""
    out = []
    for sub in nested:
        out.extend(sub)
    return out
