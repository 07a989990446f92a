"""Development tool: per-kernel instruction statistics of a HIP source compiled for gfx950
(hipcc --cuda-device-only -S), to compare two versions of a kernel without a GPU.
  python scripts/isa_count.py mm_masking_amd/csrc/mmk_unet.hip [filter]"""
import collections, os, re, subprocess, sys

def asm(src, out):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-I", os.path.join(root, "include"), "--cuda-device-only", "-S", "-o", out, src]
    subprocess.check_call(cmd)

def stats(path, flt=None):
    cur, res = None, collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); res[cur] = collections.Counter(); continue
        if cur is None: continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None if line.startswith(".Lfunc_end") else cur
            continue
        t = line.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"): continue
        op = t.split()[0]
        c = res[cur]
        c["total"] += 1
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): c["branch"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1; c["scratch"] += op.startswith("scratch_")
    # register use from the metadata
    txt = open(path).read()
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt):
        pass
    return res

if __name__ == "__main__":
    src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else None
    out = "/tmp/isa/%s.s" % os.path.basename(src)
    asm(src, out)
    dem = subprocess.run(["/usr/bin/c++filt"], input="\n".join(stats(out).keys()), capture_output=True, text=True).stdout.split("\n")
    for (k, c), d in zip(stats(out).items(), dem):
        if flt and flt not in d: continue
        print("%-110s tot %5d valu %5d mfma %3d salu %4d lds %3d vmem %3d wait %3d br %3d scr %d" % (
            d[:110], c["total"], c["valu"], c["mfma"], c["salu"], c["lds"], c["vmem"], c["waitcnt"], c["branch"], c["scratch"]))
