"""Register / scratch usage of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py local-hyperdb_amd/csrc/hdb_mfma.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_ru.o"] + sys.argv[3:], capture_output=True, text=True)
txt = r.stderr
if r.returncode != 0:
    print(txt[-3000:]); sys.exit(1)
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    if flt and flt not in d:
        continue
    m = re.search(r"<(.*)>", d)
    print((m.group(0) if m else d)[:90].ljust(90), "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "SGPR", g("SGPRs"), "LDS", g(r"LDS Size \[bytes/block\]"))
