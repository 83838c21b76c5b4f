"""Audit of the MFMA loops of one kernel in the device assembly: per basic block the MFMA / scratch / barrier / global-load counts and
the s_waitcnt vmcnt(0) that sit IN FRONT of the block's first MFMA (an exposed memory round trip per k-step).
usage: python tools/asm_hotloop.py <file.hip> <mangled-name-substring> [--dump LABEL]"""
import re, subprocess, sys, os
src, name = sys.argv[1], sys.argv[2]
out = "/tmp/asm_hotloop.s"
flags = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if os.path.basename(src) in ("kernels_panel.hip", "kernels_mchol.hip") else []
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, os.path.abspath(src)] + flags,
                      cwd=os.path.dirname(os.path.abspath(src)) or ".", stderr=subprocess.DEVNULL)
s = open(out).read()
starts = [m.start() for m in re.finditer(r"^(_Z\S*%s\S*):" % re.escape(name), s, re.M)]
for i in starts:
    j = s.index("s_endpgm", i)
    body = s[i:j].split("\n")
    print(body[0])
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = {"label": m.group(1), "loop": "Depth=" in m.group(2) or "Loop" in m.group(2), "lines": []}
            blocks.append(cur)
            continue
        if cur is None:
            cur = {"label": "entry", "loop": False, "lines": []}
            blocks.append(cur)
        if "in Loop" in l or "Loop Header" in l: cur["loop"] = True
        cur["lines"].append(l)
    for b in blocks:
        L = b["lines"]
        mf = sum("v_mfma" in l for l in L); sc = sum("scratch_" in l for l in L)
        first = next((n for n, l in enumerate(L) if "v_mfma" in l), len(L))
        early0 = sum(("vmcnt(0)" in l) for l in L[:first]) if mf else 0
        if (mf or sc) and b["loop"]:
            print("  %-12s mfma %3d  scratch %2d  barrier %d  global_load %d  vmcnt(0)-before-first-mfma %d" % (
                b["label"], mf, sc, sum("s_barrier" in l for l in L), sum("global_load" in l for l in L), early0))
    if "--dump" in sys.argv:
        lab = sys.argv[sys.argv.index("--dump") + 1]
        for b in blocks:
            if b["label"] == lab: print("\n".join(b["lines"]))
