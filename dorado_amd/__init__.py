"""dorado_amd — MI355X-native simplex basecalling hot path (conv -> LSTM stack -> CRF head ->
CRF beam-search decode) behind the reference's ModelRunnerBase boundary.

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C-ABI shared library (include/mibc.h)
  config.py  host mirror of dorado/config (BasecallModelConfig, BatchParams)
  capi.py    ctypes binding of the C-ABI (the only way Python reaches the kernels)
  host/      C++ mirror of basecall::ModelRunnerBase / CudaCaller / BasecallerNode (libmibc_host.so)
  hostapi.py ctypes binding of the host layer
  dist.py    one-process-per-GPU plumbing (replicas only, no data-path collective)
  synth.py   seeded synthetic weights and 5 kHz signal chunks (no network, no real models)
"""
__version__ = "0.1.0"
