from deepclr_amd.models import BaseModel, ModelInferenceHelper, build_model, load_trained_model, ModelType, \
    store_models_code

__all__ = ['BaseModel', 'ModelInferenceHelper', 'build_model', 'load_trained_model', 'ModelType', 'store_models_code']
