"""MI355X-native hot path of PCSeqLearning (see README.md / DESIGN.md)."""
import os

# The sequence tracker drives a few lock-step groups of tracking units, one HIP stream each. ROCm
# maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); measured on MI355X
# (tools/stage_times.py) four groups on 8 queues track a sequence in 232 ms where 4 queues need
# 316 ms. The variable is read when the HIP runtime initialises, i.e. at the first device call after
# this import; an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
