from mm_masking_amd.dICP.ICP import ICP  # noqa: F401
