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

