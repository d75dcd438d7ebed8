"""MI355X-native hot path of PCSeqLearning (see README.md / DESIGN.md)."""
# NB: this package does NOT touch GPU_MAX_HW_QUEUES. Round 2 exported 8 at import (its four lock-step groups
# per sequence overlapped better on 8 hardware queues than on HIP's default 4). With 8, batches of sequences
# in flight froze for 3-110 s once in ~10 steps -- every queue of the process standing still at once, no
# error -- while 30-step soaks with 4 (the default) and with 24 showed none (gpurun_out/r3f, DESIGN.md
# section 4 "The freeze"); throughput is the same. The default is what the vendor tests.
