"""Instructions per tile in the main loop of the thin-layer kernels, from the gfx950 assembly (no GPU needed):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I include --cuda-device-only -S -o /tmp/u.s mm_masking_amd/csrc/mmk_unet.hip
  python scripts/loop_isa.py /tmp/u.s
The main loop = the innermost loop with the most MFMAs; a ring kernel's loop body holds RD tiles (its third template argument)."""
import collections
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
print("%-64s %5s %5s %5s %5s %5s %5s  per tile" % ("kernel", "VALU", "SALU", "MFMA", "LDS", "VMEM", "wait"))
for st in starts:
    name = lines[st].split(":")[0]
    if not re.search(r"conv3x3_ring_kernel|conv_bwd_fused_kernel|conv3x3_wgrad_kernel", name):
        continue
    en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    # loop bodies: every basic block is introduced by its label line, which says which loop it belongs to
    blocks, cur = collections.OrderedDict(), None          # label -> (loop header it belongs to or None, [lines])
    for i in range(st + 1, en):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", lines[i])
        if m:
            cur = m.group(1)
            hdr = None
            if "Loop Header" in m.group(2):
                hdr = cur
            mm = re.search(r"Header=(BB\d+_\d+)", m.group(2))
            if mm:
                hdr = ".L" + mm.group(1)
            blocks[cur] = (hdr, [])
        elif cur is not None:
            blocks[cur][1].append(lines[i])
    loops = collections.defaultdict(list)
    for lab, (hdr, body) in blocks.items():
        if hdr is not None:
            loops[hdr] += body
    best = None
    for hdr, body in loops.items():
        c = collections.Counter()
        for l in body:
            t = l.strip()
            if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
                continue
            op = t.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("s_waitcnt"): c["wait"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("global_", "buffer_")): c["vmem"] += 1
        if best is None or c["mfma"] > best["mfma"]:
            best = c
    if best is None:
        continue
    m = re.search(r"ring_kernel<\d+, \d+, (\d+)", dem)
    tiles = int(m.group(1)) if m else 1
    print("%-64s %5.0f %5.0f %5.0f %5.0f %5.0f %5.0f" % (dem[:64], best["valu"] / tiles, best["salu"] / tiles, best["mfma"] / tiles,
                                                       best["lds"] / tiles, best["vmem"] / tiles, best["wait"] / tiles))
