"""The drop-in import shims (mm_masking_amd/dropin): the reference's own import statements resolve against them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dropin_modules_accept_the_reference_import_lines():
    """mm_masking_amd/dropin on sys.path: the reference's own import statements
    (icp_weight_policy.py:6-7, icp_weight_dataset.py:11-12, train_icp_weights.py:3,5,17) resolve."""
    code = "\n".join([
        "import sys",
        "sys.path.insert(0, %r)" % os.path.join(ROOT, "mm_masking_amd", "dropin"),
        "sys.path.insert(1, %r)" % ROOT,
        "from dICP.ICP import ICP",
        "from radar_utils import load_pc_from_file, cfar_mask, extract_pc, radar_polar_to_cartesian_diff, "
        "radar_cartesian_to_polar, radar_polar_to_cartesian, extract_weights, point_to_cart_idx, "
        "form_cart_range_angle_grid, form_polar_range_grid",
        "from radar_utils import load_radar, cfar_mask, extract_pc, load_pc_from_file, radar_cartesian_to_polar, "
        "radar_polar_to_cartesian_diff, extract_bev_from_pts, point_to_cart_idx",
        "from icp_weight_dataset import ICPWeightDataset",
        "from icp_weight_policy import LearnICPWeightPolicy",
        "from radar_utils import extract_bev_from_pts",
        "import mm_masking_amd.icp_weight_policy as p, mm_masking_amd.radar_utils as r",
        "assert LearnICPWeightPolicy is p.LearnICPWeightPolicy and cfar_mask is r.cfar_mask",
        "assert ICP.__module__ == 'mm_masking_amd.dICP.ICP' and ICPWeightDataset.__module__ == 'mm_masking_amd.icp_weight_dataset'",
        "icp = ICP(icp_type='pt2pt', config_path='../external/dICP/config/dICP_config.yaml')",
        "assert icp.target_pad_val == 1000.0",
    ])
    subprocess.check_call([sys.executable, "-c", code], cwd="/")
