"""Current kernel inventory of liblmnet_hip.so: resource usage of every instantiated kernel (hipcc -Rpass-analysis=kernel-resource-usage
over the product sources, no GPU needed) joined with the serial (alone) and in-step durations of the committed profiles.
    python tools/kernel_inventory.py [serial_table.txt [timeline.txt]] > profiles/rNN_kernel_inventory.md"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "lm_net_amd", "csrc")
SRCS = ["conv_tile_1x1.hip", "conv_tile_3x3.hip", "conv_tileM.hip", "conv_dma1.hip", "conv_dma3.hip", "conv_dmaM.hip", "conv_wgrad.hip", "conv_fwd.hip", "dwconv.hip", "na.hip",
        "gattn.hip", "rows.hip", "runtime.hip"]
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form=1 --cuda-device-only -c -o /dev/null -Rpass-analysis=kernel-resource-usage".split()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0].replace("void ", "") for o in out]


def resources():
    res = {}
    for src in SRCS:
        extra = ["-fno-slp-vectorize"] if src in ("dwconv.hip", "na.hip") else []
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [os.path.join(CS, src)], capture_output=True, text=True, cwd=CS)
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "Function Name":
                cur = res.setdefault(v, {"file": src})
            elif cur is not None:
                cur[k] = v
    names = list(res)
    return {d: res[n] for n, d in zip(names, demangle(names))}


def table(path, col_n, col_avg):
    t = {}
    if not path or not os.path.isfile(path):
        return t
    for line in open(path):
        m = re.match(r"\s*(\S.*?>?)\s+(\d+)\s+x?\s*([\d.]+)", line)
        m2 = re.match(r"^(\S.*?)\s{2,}(\d+)\s+([\d.]+)\s+([\d.]+)", line)
        if m2:
            t[m2.group(1).strip()] = (int(m2.group(2)), float(m2.group(4)))
    return t


def main():
    res = resources()
    serial = table(sys.argv[1] if len(sys.argv) > 1 else None, 1, 3)
    print("| kernel instance | file | VGPRs | SGPR spills | scratch B/lane | static LDS B | waves/SIMD (registers) | launches/step | alone us (avg) |")
    print("|---|---|---|---|---|---|---|---|---|")
    def key(k):
        kk = k.replace(" ", "")
        for s, v in serial.items():
            ss = s.replace(" ", "")
            if kk == ss or (len(ss) >= 70 and kk.startswith(ss)):
                return v
        return None
    for k in sorted(res, key=lambda k: (res[k]["file"], k)):
        r = res[k]
        sv = key(k)
        if serial and not sv:
            continue              # (with a serial table: only the instances the profiled step launches)
        print("| `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (k, r["file"], r.get("VGPRs", "?"), r.get("SGPRs Spill", "0"), r.get("ScratchSize [bytes/lane]", "0"),
              r.get("LDS Size [bytes/block]", "0"), r.get("Occupancy [waves/SIMD]", "?"), sv[0] if sv else "", ("%.1f" % sv[1]) if sv else ""))
    print("\n%d kernel instances." % len(res))


if __name__ == "__main__":
    main()
