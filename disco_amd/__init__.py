"""disco_amd — MI355X-native BuildGraph (overlap-graph construction) stage for the DISCO assembler.

Only what the hot path needs lives here: HIP kernels + C-ABI (csrc/), the C++ `buildG` host (host/),
a ctypes mirror of the C-ABI (buildgraph.py) and the synthetic read generator (readgen.py).
The CPU oracle is test infrastructure and lives in /oracle; nothing in this package imports it.
"""
__version__ = "0.1.0"
