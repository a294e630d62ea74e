"""CPU oracle for the Silero-VAD hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .oracle import Oracle, load_weights_blob, build_oracle  # noqa: F401
