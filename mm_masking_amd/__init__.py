"""mm_masking_amd — MI355X-native implementation of the mm_masking hot path
(learned radar mask -> differentiable ICP, forward + backward).

Host-side mirrors of the reference's Python interface:
  mm_masking_amd.radar_utils         <- mm_masking/radar_utils.py
  mm_masking_amd.dICP.ICP            <- external/dICP (dICP.ICP.ICP)
  mm_masking_amd.icp_weight_policy   <- mm_masking/icp_weight_policy.py
  mm_masking_amd.train_icp_weights   <- mm_masking/train_icp_weights.py (step, losses)
Device code: csrc/*.hip behind the C ABI of include/mmk.h (libmmk_hip.so).
"""
__version__ = "0.1.0"

import os as _os

# Hardware queues.  A training step uses the caller's stream, the weight-gradient side stream of the U-Net backward
# (csrc/mmk_unet_driver.hip) and, data-parallel, a communication stream plus RCCL's own.  ROCm maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; with a process group initialised first the side stream
# ends up SHARING a queue with the caller's stream and its overlap is gone: U-Net backward 4.66 -> 5.17 ms at B = 32
# (bench.py --gpus 1 --force-dist, round 5; 4.70 with 8 queues).  The runtime reads the variable when it initialises, i.e.
# at the first HIP call: importing this package before anything touches the GPU is enough.  An explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
