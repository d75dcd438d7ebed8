"""MI355X-native hot path of PCSeqLearning (see README.md / DESIGN.md)."""
import os

# The tracking units run on one HIP stream each. ROCm maps streams onto at most GPU_MAX_HW_QUEUES
# hardware queues (default 4); measured on MI355X (tools/stage_times.py) the tracking stage is
# fastest with 4-5 streams on 6+ hardware queues (190 ms against 207 ms for 9 streams on 4 queues,
# 275 ms for 9 streams on 8). The variable is read when the HIP runtime initialises, i.e. at the
# first device call after this import; an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
