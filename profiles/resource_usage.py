"""Register / spill / scratch figures of the render kernels, from the compiler's own remarks:
    python3 profiles/resource_usage.py [group ...]          (default: every group of the strict flavour)
One line per instantiation: <STACK, POW, STAGED, BVH, CULL, EDGES, ORDER, FEEDBACK>, VGPRs, SGPR spills, VGPR spills,
scratch bytes per lane, waves per SIMD.  (make -C rusty-marcher_amd/csrc resource-usage prints the raw remarks.)"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rusty-marcher_amd", "csrc")
groups = sys.argv[1:] or ["0", "1", "2", "3", "4"]
flav = os.environ.get("FLAVOUR", "0")
for g in groups:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-DRM_KERNEL_FAST=" + flav, "-DRM_KERNEL_GROUP=" + g,
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, "rm_kernels.hip"), "-o", "/dev/null"] + os.environ.get("DEFS", "").split()
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rec = {}
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1); rec[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur:
            rec[cur][m.group(1).strip()] = int(m.group(2))
    for name, r in rec.items():
        m = re.search(r"rm_render_staticILi(\d+)ELi(\d)ELi1ELi1E((?:Lb[01]E)+)", name)
        if not m:
            continue
        flags = re.findall(r"Lb([01])E", m.group(3))
        tags = ["STAGED", "BVH", "CULL", "EDGES", "ORDER", "FEEDBACK"]
        on = "+".join(t for t, f in zip(tags, flags) if f == "1") or "-"
        print("g%s stack %2s pow %s %-32s VGPRs %3d  SGPR spills %3d  VGPR spills %3d  scratch %4d B/lane  %d waves/SIMD" % (
            g, m.group(1), "int" if m.group(2) == "1" else "gen", on, r.get("VGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1),
            r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
