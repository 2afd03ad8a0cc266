"""deepclr_amd -- MI355X-native forward hot path of DeepCLR.

Importing the package is cheap and GPU-free; the HIP library
(``deepclr_amd/csrc/libdeepclr_amd.so``) is loaded on first use by
``deepclr_amd.lib`` and a missing library is a hard error, never a fallback.
"""
__version__ = '0.1.0'
