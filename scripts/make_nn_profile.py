"""profiles/<tag>_nn_pmc_counters.json + <tag>_nn_traffic.json out of the three rocprofv3 --pmc passes of scripts/pmc_nn.sh
(gpurun_out/pmc_nn_<tag>_{1,2,3}.json: per-kernel means over the launches of scripts/prof_nn.py = the dICP alone, B=32, 10
iterations x 2 passes).   python scripts/make_nn_profile.py <pmc tag> <profile tag> [valid scan points per pair]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
valid = float(sys.argv[3]) if len(sys.argv) > 3 else 4300.0
c, coarse = {}, {}
dur, cdur = [], []
for i in (1, 2, 3):
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_nn_%s_%d.json" % (tag, i))))
    k = [n for n in d if n.startswith("nn_") and "coarse" not in n][0]          # the scan itself (nn_mfma_kernel / nn_prefilter_kernel)
    for name, v in d[k].items():
        if name == "duration_ns":
            dur.append(v["mean"])
        else:
            c[name] = v["mean"]
    for kc in [n for n in d if "coarse" in n]:                                # the first iteration's seed pass (1 launch in 10)
        for name, v in d[kc].items():
            if name == "duration_ns":
                cdur.append(v["mean"])
            else:
                coarse[name] = v["mean"]
kern = k
dur_ns = sum(dur) / len(dur)
B, N, Mpad, dim = 32, 5120, 20480, 2
alg = B * (4 * dim * N + 4 * dim * Mpad + 8 * N)
clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (dur_ns * 1e-9)
mfma = c.get("SQ_INSTS_MFMA", 0.0)
pmc = {"kernel": kern, "shape": "B=32, N=5120 padded (~%d valid rows per pair, zero rows scanned once), Mpad=20480, dim 2; scripts/prof_nn.py: the dICP alone, "
       "10 iterations x 2 passes (2 unseeded launches of 20)" % valid, "avg_launch_us": dur_ns / 1e3, "counters_per_launch": c,
       "derived": {"effective_clock_GHz_from_GRBM_GUI_ACTIVE": clock / 1e9,
                   "mfma_instructions": mfma, "pairs_priced": mfma * 1024,
                   "matrix_pipe_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0),
                   "valu_wave_instructions_per_mfma": (c["SQ_INSTS_VALU"] - mfma) / mfma if mfma else None,
                   "valu_wave_instructions_per_64_pairs": (c["SQ_INSTS_VALU"] - mfma) / (mfma * 16) if mfma else None,
                   "wave_cycles_share": {"active": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stall": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                         "waitcnt_or_barrier": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]},
                   "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]},
       "note": "separate rocprofv3 --pmc passes (scripts/pmc_nn.sh; FETCH_SIZE and WRITE_SIZE in passes of their own). SQ_* cycle counters are "
               "quad-cycles summed over the waves; SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16. The LDS bank conflicts are "
               "the exact re-scan's per-lane reads of the fp32 planes (each lane its own chunk)."}
if coarse:
    pmc["coarse_seed_pass"] = {"kernel": "nn_coarse_seed_kernel<2>", "launches": "one per icp() call, in front of the first (unseeded) scan",
                               "avg_launch_us": sum(cdur) / len(cdur) / 1e3, "counters_per_launch": coarse,
                               "note": "4-byte gathers of every 64th target (FETCH_SIZE not doubled: narrow reads)"}
json.dump(pmc, open(os.path.join(ROOT, "profiles", out + "_nn_pmc_counters.json"), "w"), indent=1)
fetch, write = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
tr = {"kernel": kern, "shape": pmc["shape"], "density": "survey (%d valid scan points per pair)" % valid, "avg_launch_us": round(dur_ns / 1e3, 1),
      "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
      "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg,
      "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads on gfx950 -> doubled; WRITE_SIZE exact (the 64-bit "
              "atomic mins are counted on the write side, as the 64-byte requests they leave L2 in). Fetch = the target planes once per source block "
              "class that misses L2 + the scanned source rows + the previous iteration's indices and their targets (seed); writes = 8-byte atomic mins, one "
              "per (scanned point, unit of 4 tiles that found a candidate below the running bound)."}
json.dump(tr, open(os.path.join(ROOT, "profiles", out + "_nn_traffic.json"), "w"), indent=1)
print(json.dumps(pmc["derived"], indent=1))
print("traffic %.2f MB per launch (fetch x2 %.2f + write %.2f) vs algorithmic %.2f MB" % ((fetch + write) / 1e6, fetch / 1e6, write / 1e6, alg / 1e6))
