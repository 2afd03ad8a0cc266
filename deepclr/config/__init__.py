from deepclr_amd.config import Config, ConfigEnum, Mode, load_config, load_model_config

__all__ = ['Config', 'ConfigEnum', 'Mode', 'load_config', 'load_model_config']
