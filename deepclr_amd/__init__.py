"""deepclr_amd -- MI355X-native forward hot path of DeepCLR.

Importing the package is cheap and GPU-free; the HIP library
(``deepclr_amd/csrc/libdeepclr_amd.so``) is loaded on first use by
``deepclr_amd.lib`` and a missing library is a hard error, never a fallback.
"""
import os as _os

# The HIP runtime multiplexes the streams of a process onto 4 hardware queues by default. The pipelined
# runner keeps the caller's stream + 3 side streams busy; one more stream (RCCL's, a copy stream) then shares a
# queue with a ~1 ms sampling launch and waits behind it (measured: -35 % throughput). Eight queues avoid that.
# Only effective before the runtime initialises, i.e. when this package is imported before the first GPU call;
# an explicit setting in the environment wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
