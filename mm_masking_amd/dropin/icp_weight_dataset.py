"""Drop-in for the reference's flat module name (`from icp_weight_dataset import ICPWeightDataset`)."""
from mm_masking_amd.icp_weight_dataset import ICPWeightDataset  # noqa: F401
