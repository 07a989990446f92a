"""Per-launch report of one U-Net forward + backward pass: zips the program-order schedule (scripts/unet_schedule.py) with the
dispatches of rocprofv3 runs of scripts/prof_unet_pass.py (rocpd SQLite files):
   python scripts/unet_layers_report.py out_prefix trace.db [pmc.db ...]
-> <out_prefix>_unet_layers.json   per launch {layer, role, kernel, bytes, FLOP, us, TB/s, TFLOP/s, bound, + counters}
   <out_prefix>_conv_mfma_busy.json / _conv_traffic.json   per kernel class
FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM);
WRITE_SIZE as reported; both arrive in KB.  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x duration x clock),
clock from GRBM_GUI_ACTIVE / 8 / duration of the same dispatch."""
import collections, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import unet_schedule

OURS = ("(anonymous namespace)", "_GLOBAL__N_1")
SKIP = ("channel_minmax",)


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


def demangled(name):
    # the few kernels rocprofv3 leaves mangled
    for k in ("final_bwd_kernel", "final_fwd_kernel", "final_bwd_reduce_kernel"):
        if k in name:
            return k
    return short(name)


def last_pass(rows, n):
    """rows: (name, payload) in dispatch order -> the last n of OUR kernels."""
    mine = [(demangled(nm), pl) for nm, pl in rows if any(t in nm for t in OURS) and not any(s in nm for s in SKIP)]
    return mine[-n:]


def main():
    prefix, trace = sys.argv[1], sys.argv[2]
    sch, nf = unet_schedule.schedule()
    db = sqlite3.connect(trace)
    rows = [(r[0], r[2] - r[1]) for r in db.execute("select name, start, end from kernels order by start").fetchall()]
    got = last_pass(rows, len(sch))
    assert len(got) == len(sch), (len(got), len(sch))
    for e, (nm, dur) in zip(sch, got):
        want = e["kernel"].split("<")[0]
        assert nm.startswith(want), ("schedule / trace mismatch", e, nm)
        e["kernel"] = nm if "<" in nm else e["kernel"]
        e["us"] = dur / 1e3
    for pmc in sys.argv[3:]:
        d = sqlite3.connect(pmc)
        per = collections.OrderedDict()
        for name, disp, cnt, val, dur in d.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection order by dispatch_id"):
            ent = per.setdefault(disp, [name, {}, dur])
            ent[1][cnt] = ent[1].get(cnt, 0.0) + val
        prow = [(v[0], (v[1], v[2])) for v in per.values()]
        gotp = last_pass(prow, len(sch))
        assert len(gotp) == len(sch), (pmc, len(gotp))
        for e, (nm, (cs, dur)) in zip(sch, gotp):
            assert nm.startswith(e["kernel"].split("<")[0]), ("schedule / pmc mismatch", e, nm)
            e.setdefault("counters", {}).update(cs)
            e.setdefault("us_under_pmc", {})[os.path.basename(pmc)] = dur / 1e3
    for e in sch:
        by = e["read_bytes"] + e["write_bytes"]
        e["TBps"] = by / e["us"] * 1e-6 if e["us"] > 0 else 0.0
        e["TFLOPs"] = e["flop"] / e["us"] * 1e-6 if e["us"] > 0 else 0.0
        t_hbm, t_mfma = by / 6.3e12 * 1e6, e["flop"] / 2.5e15 * 1e6       # us at the achievable HBM rate / dense bf16 peak
        e["bound"] = "hbm" if t_hbm >= t_mfma else "mfma"
        e["floor_us"] = max(t_hbm, t_mfma)
        c = e.get("counters", {})
        if "FETCH_SIZE" in c:
            e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    json.dump({"shape": "B=32, 640x640, one forward + backward pass on one stream (scripts/prof_unet_pass.py)", "forward_launches": nf,
               "sum_us": {"forward": sum(e["us"] for e in sch[:nf]), "backward": sum(e["us"] for e in sch[nf:])},
               "launches": sch}, open(prefix + "_unet_layers.json", "w"), indent=1)
    # per kernel class
    cls = collections.OrderedDict()
    for e in sch:
        k = e["kernel"]
        a = cls.setdefault(k, {"launches": 0, "us": 0.0, "flop": 0.0, "alg_read": 0, "alg_write": 0, "hbm_read": 0.0, "hbm_write": 0.0,
                               "mfma_busy_cycles": 0.0, "gui_active": 0.0})
        a["launches"] += 1; a["us"] += e["us"]; a["flop"] += e["flop"]; a["alg_read"] += e["read_bytes"]; a["alg_write"] += e["write_bytes"]
        a["hbm_read"] += e.get("hbm_read_bytes", 0.0); a["hbm_write"] += e.get("hbm_write_bytes", 0.0)
        c = e.get("counters", {})
        a["mfma_busy_cycles"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a["gui_active"] += c.get("GRBM_GUI_ACTIVE", 0.0)
        for nm in ("SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE",
                   "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"):
            a[nm] = a.get(nm, 0.0) + c.get(nm, 0.0)
    busy, traf = [], []
    for k, a in cls.items():
        if a["flop"] > 0 and a["gui_active"] > 0:
            ent = {"kernel": k, "launches_per_pass": a["launches"], "us_per_pass": a["us"], "TFLOPs": a["flop"] / a["us"] * 1e-6,
                   "mfma_busy_frac": a["mfma_busy_cycles"] / (1024.0 * a["gui_active"] / 8.0)}
            # round 5: overlap and LDS counters of the second SQ pass (absent when that pass did not run)
            if a.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) > 0 and a["mfma_busy_cycles"] > 0:
                ent["valu_coexec_frac_of_mfma_busy"] = a["SQ_VALU_MFMA_COEXEC_CYCLES"] / a["mfma_busy_cycles"]
            if a.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
                ent["lds_bank_conflict_frac"] = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"]
                ent["lds_array_active_frac"] = a["SQ_LDS_IDX_ACTIVE"] / (256.0 * a["gui_active"] / 8.0)
            if a.get("SQ_WAVE_CYCLES", 0.0) > 0:
                ent["wave_cycles_waiting_frac"] = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
                ent["wave_cycles_issue_stalled_frac"] = a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
                if a.get("SQ_WAIT_INST_LDS", 0.0) > 0:
                    ent["wave_cycles_lds_issue_stall_frac"] = a["SQ_WAIT_INST_LDS"] / a["SQ_WAVE_CYCLES"]
            if a.get("SQ_INSTS_MFMA", 0.0) > 0:
                ent["valu_insts_per_mfma"] = a.get("SQ_INSTS_VALU", 0.0) / a["SQ_INSTS_MFMA"]
                if a.get("SQ_INSTS_LDS", 0.0) > 0:
                    ent["lds_insts_per_mfma"] = a["SQ_INSTS_LDS"] / a["SQ_INSTS_MFMA"]
            busy.append(ent)
        if a["alg_read"] + a["alg_write"] > 0:
            traf.append({"kernel": k, "launches_per_pass": a["launches"], "us_per_pass": a["us"],
                         "algorithmic_read_MB": a["alg_read"] / 1e6, "algorithmic_write_MB": a["alg_write"] / 1e6,
                         "hbm_read_MB_fetch_x2": a["hbm_read"] / 1e6, "hbm_write_MB": a["hbm_write"] / 1e6,
                         "read_ratio": a["hbm_read"] / a["alg_read"] if a["alg_read"] else None,
                         "write_ratio": a["hbm_write"] / a["alg_write"] if a["alg_write"] else None,
                         "algorithmic_TBps": (a["alg_read"] + a["alg_write"]) / a["us"] * 1e-6})
    note = "one U-Net forward + backward pass, B=32, 640x640; counters from separate rocprofv3 --pmc passes (scripts/pmc_unet.sh)"
    json.dump({"note": note, "kernels": busy}, open(prefix + "_conv_mfma_busy.json", "w"), indent=1)
    json.dump({"note": note + "; FETCH_SIZE x 2 per the gfx950 correction, WRITE_SIZE as reported", "kernels": traf},
              open(prefix + "_conv_traffic.json", "w"), indent=1)
    print("forward %.0f us, backward %.0f us over %d launches" % (sum(e["us"] for e in sch[:nf]), sum(e["us"] for e in sch[nf:]), len(sch)))
    for t in traf[:40]:
        print("%-52s x%2d %8.0f us  alg %7.0f MB  read x%s write x%s" % (t["kernel"][:52], t["launches_per_pass"], t["us_per_pass"],
              t["algorithmic_read_MB"] + t["algorithmic_write_MB"], "%.2f" % t["read_ratio"] if t["read_ratio"] else "-",
              "%.2f" % t["write_ratio"] if t["write_ratio"] else "-"))


if __name__ == "__main__":
    main()
