"""diamond_amd -- MI355X-native seed-and-extend hot path behind the DIAMOND operator interface.

The product is the C-ABI shared library ``libdiamond_hip.so`` (include/diamond_hip.h) built from
``diamond_amd/csrc``; this package is the thin Python host mirror used by tests and bench.py.
"""
