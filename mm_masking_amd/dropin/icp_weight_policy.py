"""Drop-in for the reference's flat module name (`from icp_weight_policy import LearnICPWeightPolicy`)."""
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy, weights_init  # noqa: F401
