"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol
that include/mmk.h declares (no compute calls without a GPU); host-side argument
checking; the product fails loudly instead of falling back when no GPU exists."""
import ctypes
import os
import re

import pytest
import torch

from mm_masking_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _lib.build()
    return _lib.lib()


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "mmk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(mmk_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 15
    for n in sorted(names):
        assert hasattr(L, n), "symbol %s declared in mmk.h is not exported" % n
    assert names == set(_lib.EXPORTED.keys()), names ^ set(_lib.EXPORTED.keys())
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mmk.h")).read()
    assert L.mmk_version() == int(re.search(r"#define\s+MMK_VERSION\s+(\d+)", hdr).group(1))


def test_host_side_argument_checks(L):
    p = _lib.IcpParams(B=2, N=100, M=300, tgt_cols=6, dim=2, icp_type=1, loss=2, loss_k=1.0, trim_dist=5.0,
                       tolerance=1e-5, max_iter=10, save_state=1, check_every=0, nn_method=0)
    need = L.mmk_icp_workspace_bytes(ctypes.byref(p))
    assert need > 2 * 2 * 1024 * 4
    assert L.mmk_nn_padded_m(20000) == 20480 and L.mmk_nn_padded_m(1) == 1024
    assert L.mmk_nn_workspace_bytes(32, 5120, 20000, 2) >= 2 * 32 * 5120 * 4
    bad = _lib.IcpParams(B=2, N=100, M=300, tgt_cols=3, dim=2, icp_type=1, loss=2, loss_k=1.0, trim_dist=5.0,
                         tolerance=1e-5, max_iter=10, save_state=1, check_every=0, nn_method=0)
    assert L.mmk_icp_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"normals" in L.mmk_last_error()
    bad.tgt_cols, bad.dim = 6, 4
    assert L.mmk_icp_workspace_bytes(ctypes.byref(bad)) == 0 and b"dim" in L.mmk_last_error()
    bad.dim, bad.nn_method = 2, 7
    assert L.mmk_icp_workspace_bytes(ctypes.byref(bad)) == 0 and b"nn_method" in L.mmk_last_error()
    p.nn_method = 1
    assert L.mmk_icp_workspace_bytes(ctypes.byref(p)) > need      # the grid engine carves its cell tables
    null = ctypes.c_void_p(0)
    assert L.mmk_cfar_mask(null, 1, 1, 1, 50, 5, 56, 1, 1.0, 0.09, 0, 10.0, null, null) == -1
    assert L.mmk_nn_search(null, null, null, 1, 1, 1, 2, null, null, null, 0, null) == -1
    assert L.mmk_bev_raster(null, 1, 1, 6, 640, 0.2384, null, null) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU error path")
def test_no_cpu_fallback():
    from mm_masking_amd import radar_utils as ru
    from mm_masking_amd.dICP.ICP import ICP
    with pytest.raises(_lib.MmkError):
        ru.cfar_mask(torch.zeros(1, 4, 400), 0.0596)
    with pytest.raises(_lib.MmkError):
        ru.extract_weights(torch.zeros(1, 640, 640), torch.zeros(1, 4, 3))
    with pytest.raises(_lib.MmkError):
        ICP("pt2pt").icp(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), dim=2)
    icp = ICP(icp_type="pt2pl", config_path="../external/dICP/config/dICP_config.yaml")
    assert icp.target_pad_val == 1000.0


def test_product_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; import mm_masking_amd.train_icp_weights, mm_masking_amd.radar_utils, "
            "mm_masking_amd.icp_weight_policy, mm_masking_amd.dICP.ICP; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for root, _, files in os.walk(os.path.join(ROOT, "mm_masking_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt, f

