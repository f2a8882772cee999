"""Compile the HIP sources of this package for gfx950 (in-tree, no JIT cache elsewhere).

Two artefacts:

* ``lib/libogpsx.so`` - the generic runtime and public C ABI (``include/ogpsx.h``), from
  ``csrc/ogpsx_core.hip``.
* ``lib/jit/libogk_<hash>.so`` - one callback module per traced problem: the hand-written
  sweep kernels ``csrc/ogk_kernels.hip`` instantiated with the generated device functions of
  that problem (``codegen.emit_header``).  Keyed by a hash of the generated source, so a
  problem is compiled once and the shared object travels with the tree.

``-ffp-contract=off`` is part of the numerical contract (bit parity with the CPU oracle,
SURVEY.md section 7.4 item 2); fused multiply-adds only happen where the code asks for them.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
JITDIR = os.path.join(LIBDIR, "jit")
CORE_LIB = os.path.join(LIBDIR, "libogpsx.so")
ARCH = "gfx950"
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wno-unused-value"] + os.environ.get("OG_EXTRA_HIPFLAGS", "").split()


class BuildError(RuntimeError):
    pass


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise BuildError("hipcc not found: the HIP extension cannot be built on this machine")
    return exe


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise BuildError("command failed: %s\n%s" % (" ".join(cmd), proc.stdout[-4000:]))
    return proc.stdout


def _digest_files(paths, extra=""):
    """Content hash of the sources an artefact is built from (mtimes do not survive the
    snapshot copy to the GPU box, content does)."""
    hsh = hashlib.sha256(extra.encode())
    for path in paths:
        with open(path, "rb") as fh:
            hsh.update(fh.read())
    hsh.update(" ".join(HIP_FLAGS).encode())
    return hsh.hexdigest()[:16]


def _core_sources():
    return [os.path.join(CSRC, f) for f in ("ogpsx_core.hip", "og_lgl.h", "og_math.h", "ogk.h")] + \
        [os.path.join(HERE, "..", "include", "ogpsx.h")]


def _kernel_sources():
    return [os.path.join(CSRC, f) for f in ("ogk_kernels.hip", "ogk.h", "og_math.h", "og_dual.h")]


def build_core(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_path = CORE_LIB + ".stamp"
    want = _digest_files(_core_sources())
    if not force and os.path.exists(CORE_LIB) and os.path.exists(stamp_path):
        with open(stamp_path) as fh:
            if fh.read().strip() == want:
                return CORE_LIB
    tmp = CORE_LIB + ".tmp%d" % os.getpid()
    _run([hipcc()] + HIP_FLAGS + [os.path.join(CSRC, "ogpsx_core.hip"), "-o", tmp, "-ldl"])
    os.replace(tmp, CORE_LIB)
    with open(stamp_path, "w") as fh:
        fh.write(want)
    return CORE_LIB


SQP_LIB = os.path.join(LIBDIR, "libogsqp.so")


def build_sqp(force=False):
    """``lib/libogsqp.so``: the QP subproblem / BFGS kernels of the SQP driver (``include/ogsqp.h``)."""
    os.makedirs(LIBDIR, exist_ok=True)
    sources = [os.path.join(CSRC, "ogsqp.hip"), os.path.join(CSRC, "ogsqp_rows.h"), os.path.join(CSRC, "ogsqp_lq16.h"),
               os.path.join(HERE, "..", "include", "ogsqp.h")]
    stamp_path = SQP_LIB + ".stamp"
    want = _digest_files(sources)
    if not force and os.path.exists(SQP_LIB) and os.path.exists(stamp_path):
        with open(stamp_path) as fh:
            if fh.read().strip() == want:
                return SQP_LIB
    tmp = SQP_LIB + ".tmp%d" % os.getpid()
    _run([hipcc()] + HIP_FLAGS + [sources[0], "-o", tmp])
    os.replace(tmp, SQP_LIB)
    with open(stamp_path, "w") as fh:
        fh.write(want)
    return SQP_LIB


def module_digest(header_source):
    """Key of a callback module: generated header + kernel sources + flags."""
    return _digest_files(_kernel_sources(), extra=header_source)


def module_path(digest):
    return os.path.join(JITDIR, "libogk_%s.so" % digest)


def build_module(header_source, digest=None, force=False):
    """Compile the sweep kernels against one generated header -> shared object path."""
    os.makedirs(JITDIR, exist_ok=True)
    digest = module_digest(header_source)
    out = module_path(digest)
    kernels = os.path.join(CSRC, "ogk_kernels.hip")
    if not force and os.path.exists(out):
        return out
    header = os.path.join(JITDIR, "og_gen_%s.h" % digest)
    with tempfile.NamedTemporaryFile("w", dir=JITDIR, suffix=".h", delete=False) as fh:
        fh.write(header_source)
        tmp_header = fh.name
    os.replace(tmp_header, header)
    tmp = out + ".tmp%d" % os.getpid()
    _run([hipcc()] + HIP_FLAGS + ["-I" + CSRC, "-DOG_GEN_HEADER=\"%s\"" % header, kernels,
                                  "-o", tmp])
    os.replace(tmp, out)
    return out
