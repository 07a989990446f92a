"""CPU-only checks of the real-data loader: ``ICPWeightDataset`` over a plain-file export against the
dictionaries the reference's own ``__getitem__`` / ``load_graph_data`` returned (tests/golden/dataset_item.npz), the
worker-safe item mode (``params["batched_prepare"]``: CPU-only items that a ``DataLoader(num_workers=4)`` produces in parallel as
/root/reference/mm_masking/train_icp_weights.py:454-455 does, finished per batch by ``finish_batch``), the native batch fill
(``mmk_host_read_rows{,_batch}``) and ``DeviceLoader`` on a CPU device."""
import os

import numpy as np
import torch

from mm_masking_amd import icp_weight_dataset as ds

from export_util import assert_same, dataset_params, write_fixture_export


def test_dataset_item_matches_reference_polar(golden_dir, tmp_path):
    """ICPWeightDataset.__getitem__ (polar network input, no augmentation: no HIP call) on the export written from
    the fixture, against the dictionaries the reference's __getitem__ / load_graph_data returned."""
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = write_fixture_export(str(tmp_path), g)
    d = ds.ICPWeightDataset(pairs, dataset_params(), dataset_type="train", data_dir=str(tmp_path))
    assert len(d) == 2 and d.target_pad_val == 1000.0 and d.augment is False
    # T_init is exp of a seeded uniform draw (icp_weight_dataset.py:261-277): planar, inside the envelope
    for T in d.T_loc_init:
        assert abs(float(T[0, 3])) <= 2.0 and abs(float(T[1, 3])) <= 2.0 and float(T[2, 3]) == 0.0
        assert abs(np.arctan2(float(T[1, 0]), float(T[0, 0]))) <= 0.6 + 1e-6
    d.T_loc_init = torch.from_numpy(g["T_init"])             # the fixture's initial guesses
    for i in range(2):
        it = d[i]
        pre = "p%d_" % i
        for key, val in (("raw_pc", it["loc_data"]["raw_pc"]), ("filtered_pc", it["loc_data"]["filtered_pc"]),
                         ("fft_sub", it["loc_data"]["fft_data"]), ("cfar_sub", it["loc_data"]["fft_cfar"]),
                         ("map_pc", it["map_data"]["pc"]), ("T_init", it["transforms"]["T_ml_init"]),
                         ("T_gt", it["transforms"]["T_ml_gt"])):
            assert val.dtype == torch.float32
            assert np.array_equal(val.numpy(), g[pre + key]), key
        assert [it["loc_data"]["timestamp"], it["map_data"]["timestamp"]] == g[pre + "stamps"].tolist()
        assert it["map_data"]["pc"].shape == (80, 6) and it["loc_data"]["raw_pc"].shape == (40, 3)
    # a DataLoader batches the items into the Row-D dictionary of SURVEY.md §8a
    batch = next(iter(torch.utils.data.DataLoader(d, batch_size=2, shuffle=False, num_workers=0)))
    assert batch["loc_data"]["fft_data"].shape == (2, 40, 336) and batch["map_data"]["pc"].shape == (2, 80, 6)
    assert batch["transforms"]["T_ml_init"].shape == (2, 4, 4)
    # padding sizes derived from the data when params does not fix them
    d2 = ds.ICPWeightDataset(pairs, dataset_params(max_loc_pts=0, max_map_pts=0, num_val=1), dataset_type="test",
                             data_dir=str(tmp_path))
    assert len(d2) == 1 and d2.max_loc_pts == 30 and 0 < d2.max_map_pts <= 90
    assert d.get_item_from_loc_timestamp(int(g["loc_stamp"][1]))["loc_data"]["timestamp"] == int(g["loc_stamp"][1])


def test_png_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (13, 29), dtype=np.uint8)
    ds.write_png_gray(str(tmp_path / "a.png"), img)
    assert np.array_equal(ds.read_png_gray(str(tmp_path / "a.png")), img)


def test_worker_items_finish_to_the_reference_batch(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = write_fixture_export(str(tmp_path), g)
    ref = ds.ICPWeightDataset(pairs, dataset_params(), dataset_type="train", data_dir=str(tmp_path))
    wrk = ds.ICPWeightDataset(pairs, dataset_params(batched_prepare=True), dataset_type="train", data_dir=str(tmp_path))
    wrk.T_loc_init = ref.T_loc_init.clone()
    want = next(iter(torch.utils.data.DataLoader(ref, batch_size=2, shuffle=False, num_workers=0)))
    it = wrk[0]
    assert it["loc_data"]["fft_u8"].dtype == torch.uint8 and it["loc_data"]["azimuths"].dtype == torch.float32
    assert "fft_data" not in it["loc_data"]                                   # nothing float, nothing on a device
    # four worker processes, as upstream; twice: the second pass reads the decoded-byte cache the first one wrote
    for rep in range(2):
        got = next(iter(torch.utils.data.DataLoader(wrk, batch_size=2, shuffle=False, num_workers=4)))
        fin = ds.finish_batch(got, "cpu", network_input_type="polar")
        assert_same(fin, want)
        assert os.path.exists(wrk.loc_radar_path_list[0] + ".u8") and os.path.exists(wrk.loc_cfar_path_list[0] + ".u8")
    # the items of the default mode equal the reference's own (test_dataset_item_matches_reference_polar); spot-check the chain here too
    assert np.array_equal(fin["loc_data"]["fft_data"][0].numpy(), g["p0_fft_sub"])
    # a stale cache (older than its PNG) is ignored and rewritten
    cpath = wrk.loc_radar_path_list[1] + ".u8"
    with open(cpath, "wb") as f:
        f.write(b"\x00" * 16)
    os.utime(cpath, (1, 1))
    assert_same(ds.finish_batch(torch.utils.data.default_collate([wrk[0], wrk[1]]), "cpu", network_input_type="polar"), want)
    assert os.path.getsize(cpath) > 16
    # DeviceLoader on a CPU device: same batches, in order
    for mode in ("threads", "processes"):
        dl = ds.DeviceLoader(wrk, batch_size=1, device="cpu", num_workers=2, mode=mode)
        got = list(dl)
        assert len(got) == len(dl) == 2
        assert_same(torch.utils.data.default_collate([ref[0]]), got[0])
        assert_same(torch.utils.data.default_collate([ref[1]]), got[1])
    assert_same(want, next(iter(ds.DeviceLoader(wrk, batch_size=2, device="cpu", num_workers=4))))


def test_worker_items_augmentation_matches_default_mode(golden_dir, tmp_path):
    """Same yaw draw -> same rolled rows, azimuths and rotated clouds in both item modes (icp_weight_dataset.py:425-452)."""
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = write_fixture_export(str(tmp_path), g)
    ref = ds.ICPWeightDataset(pairs, dataset_params(augment=True), dataset_type="train", data_dir=str(tmp_path))
    wrk = ds.ICPWeightDataset(pairs, dataset_params(augment=True, batched_prepare=True), dataset_type="train",
                              data_dir=str(tmp_path))
    wrk.T_loc_init = ref.T_loc_init.clone()
    for i in range(2):
        torch.manual_seed(77 + i)
        a = ref[i]
        torch.manual_seed(77 + i)
        b = ds.finish_batch(torch.utils.data.default_collate([wrk[i]]), "cpu", network_input_type="polar")
        assert_same(torch.utils.data.default_collate([a]), b)


def test_native_fill_matches_default_mode_with_augmentation(golden_dir, tmp_path):
    """DeviceLoader's thread mode: ``fill_batch`` (every tensor moved by mmk_host_read_rows_batch: column cut + azimuth roll in C,
    clouds from the prepared-cloud cache) + ``finish_batch`` (rotation of the clouds on the device) against the default
    item mode under the same yaw draw: images, azimuths, poses, stamps bit-equal; rotated clouds to fp32 rounding."""
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = write_fixture_export(str(tmp_path), g)
    ref = ds.ICPWeightDataset(pairs, dataset_params(augment=True), dataset_type="train", data_dir=str(tmp_path))
    wrk = ds.ICPWeightDataset(pairs, dataset_params(augment=True, batched_prepare=True), dataset_type="train",
                              data_dir=str(tmp_path))
    wrk.T_loc_init = ref.T_loc_init.clone()
    spec = wrk.native_item_spec()
    for rep in range(2):                      # second round: every cache exists
        for i in range(2):
            bufs = {grp: {k: torch.empty((1,) + tuple(shape), dtype=dt) for k, (shape, dt) in d.items()} for grp, d in spec.items()}
            torch.manual_seed(91 + i)
            a = torch.utils.data.default_collate([ref[i]])
            torch.manual_seed(91 + i)
            wrk.fill_batch([i], bufs, threads=2)
            b = ds.finish_batch(bufs, "cpu", network_input_type="polar")
            for key in ("fft_data", "fft_cfar", "timestamp"):
                assert_same(a["loc_data"][key], b["loc_data"][key], key)
            assert_same(a["transforms"], b["transforms"])
            assert_same(a["map_data"]["timestamp"], b["map_data"]["timestamp"])
            for x, y in ((a["loc_data"]["raw_pc"], b["loc_data"]["raw_pc"]), (a["loc_data"]["filtered_pc"], b["loc_data"]["filtered_pc"]),
                         (a["map_data"]["pc"], b["map_data"]["pc"])):
                assert x.shape == y.shape
                np.testing.assert_allclose(y.numpy(), x.numpy(), rtol=2e-6, atol=2e-4)       # |pad value| = 1000: one fp32 ulp = 6e-5
    assert os.path.isdir(os.path.join(wrk.pair_dirs[0], "prepared"))
    # without augmentation the native path is bit-equal throughout
    ref0 = ds.ICPWeightDataset(pairs, dataset_params(), dataset_type="train", data_dir=str(tmp_path))
    wrk0 = ds.ICPWeightDataset(pairs, dataset_params(batched_prepare=True), dataset_type="train", data_dir=str(tmp_path))
    wrk0.T_loc_init = ref0.T_loc_init.clone()
    got = next(iter(ds.DeviceLoader(wrk0, batch_size=2, device="cpu", num_workers=3)))
    assert_same(torch.utils.data.default_collate([ref0[0], ref0[1]]), got)
    # the prepared-cloud cache is keyed on everything the clouds depend on: another padding value (or re-exported cloud files)
    # gets clouds of its own instead of the stale file
    n_before = len(os.listdir(os.path.join(wrk0.pair_dirs[0], "prepared")))
    wrk1 = ds.ICPWeightDataset(pairs, dataset_params(batched_prepare=True), dataset_type="train", data_dir=str(tmp_path))
    wrk1.T_loc_init = ref0.T_loc_init.clone()
    wrk1.target_pad_val = 500.0
    got1 = next(iter(ds.DeviceLoader(wrk1, batch_size=2, device="cpu", num_workers=2)))
    pad_rows = got["map_data"]["pc"][..., 0] == wrk0.target_pad_val
    assert pad_rows.any() and bool((got1["map_data"]["pc"][..., 0][pad_rows] == 500.0).all())
    assert len(os.listdir(os.path.join(wrk1.pair_dirs[0], "prepared"))) > n_before
    src = wrk0._cloud_sources(0)[2]
    before = wrk0._prepared_clouds(0)
    os.utime(src, ns=(os.stat(src).st_atime_ns, os.stat(src).st_mtime_ns + 10 ** 9))
    wrk2 = ds.ICPWeightDataset(pairs, dataset_params(batched_prepare=True), dataset_type="train", data_dir=str(tmp_path))
    assert wrk2._prepared_clouds(0) != before


def test_device_loader_draws_from_its_own_generator(golden_dir, tmp_path):
    """Shuffle order and augmentation yaws come from the loader's private generator: the same seed gives the same batches
    whatever is drawn from the global generator in between (the training thread draws from it concurrently)."""
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = write_fixture_export(str(tmp_path), g)
    dset = ds.ICPWeightDataset(pairs, dataset_params(augment=True, batched_prepare=True), dataset_type="train", data_dir=str(tmp_path))

    def run(noise):
        out = []
        for b in ds.DeviceLoader(dset, batch_size=1, device="cpu", num_workers=2, shuffle=True, seed=77):
            out.append(b)
            if noise:
                torch.rand(5)
        return out
    torch.manual_seed(1)
    a = run(False)
    torch.manual_seed(2)
    b = run(True)
    for x, y in zip(a, b):
        assert_same(x, y)


def test_host_read_rows_roll_and_columns(tmp_path):
    from mm_masking_amd import _lib
    import host_read_checks
    host_read_checks.run(_lib.lib(), str(tmp_path), ReadJob=_lib.ReadJob)


def test_host_read_rows_under_address_and_ub_sanitizers(tmp_path):
    """The host half of the loader (csrc/mmk_loader_host.inc: pread loops, a stack block, C threads drawing jobs from a shared
    counter) compiled for the CPU with -fsanitize=address,undefined and driven through the same checks as the product library
    (tests/host_read_checks.py), in a child interpreter with the sanitizer runtime pre-loaded.  GPU sanitizer runs are not
    available on the pool; this half needs no GPU."""
    import shutil
    import subprocess
    import sys
    import pytest
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libmmk_loader_san.so")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-shared", "-fPIC", "-I", os.path.join(root, "include"), "-o", so,
                           os.path.join(root, "mm_masking_amd", "csrc", "mmk_loader_host_san.cpp"), "-lpthread"])
    asan = subprocess.check_output([gxx, "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "host_read_checks.py"), so, str(work)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "host_read_checks ok" in r.stdout, r.stdout[-3000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-3000:]
