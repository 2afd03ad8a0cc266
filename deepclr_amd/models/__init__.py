"""Mirror of the reference's ``deepclr.models`` public interface
(/root/reference/deepclr/models/__init__.py:1-5) on top of the HIP kernels."""
from .base import BaseModel, ModelInferenceHelper
from .build import build_model, load_trained_model, ModelType, store_models_code

__all__ = ['BaseModel', 'ModelInferenceHelper',
           'build_model', 'load_trained_model', 'ModelType', 'store_models_code']
