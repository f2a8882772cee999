#!/usr/bin/env python3
"""Register / scratch / occupancy figures of the kernels of one problem's callback module
(hipcc -Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.py [problem ...]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengoddard_amd import build, codegen, problems     # noqa: E402

for name in sys.argv[1:] or ["polar_tsto"]:
    prob, obj = problems.build(name)
    src = codegen.emit_header(codegen.trace_problem(prob, obj))
    hdr = os.path.join(build.JITDIR, "og_gen_%s.h" % build.module_digest(src))
    os.makedirs(build.JITDIR, exist_ok=True)
    with open(hdr, "w") as fh:
        fh.write(src)
    cmd = [build.hipcc()] + build.HIP_FLAGS + ["-I" + build.CSRC, '-DOG_GEN_HEADER="%s"' % hdr,
                                               os.path.join(build.CSRC, "ogk_kernels.hip"), "-o", "/dev/null",
                                               "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    print(name)
    for blk in out.split("Function Name: ")[1:]:
        lines = blk.splitlines()
        keep = [l.split("remark:")[-1].split("[-R")[0].strip() for l in lines
                if any(k in l for k in ("VGPRs:", "VGPRs Spill", "Occupancy", "ScratchSize", "LDS Size"))]
        print("  %-28s %s" % (lines[0].split("[")[0].strip()[-28:], " | ".join(keep)))
