from .model import SurfaceFormer
from .model_para import SurfaceFormer_Parallel

__all__ = ["SurfaceFormer", "SurfaceFormer_Parallel"]
