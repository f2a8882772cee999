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


# flags for the callback modules only (a stamped build of one workload's kernels - -DOGK_TRACE=1 - must not change the
# digest of libogpsx.so / libogsqp.so and make every later process rebuild them)
MODULE_FLAGS = os.environ.get("OG_MODULE_HIPFLAGS", "").split()


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
    if os.environ.get("OG_CORE_LIB"):                   # diagnostics: a build with other flags (tools/sanitize.sh)
        return os.environ["OG_CORE_LIB"]
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
    if os.environ.get("OG_SQP_LIB"):                    # diagnostics: a build with other flags (-DOGSQP_TRACE)
        return os.environ["OG_SQP_LIB"]
    os.makedirs(LIBDIR, exist_ok=True)
    sources = [os.path.join(CSRC, "ogsqp.hip"), os.path.join(CSRC, "ogsqp_rows.h"), os.path.join(CSRC, "ogsqp_lq16.h"),
               os.path.join(CSRC, "ogsqp_lqwide.h"), os.path.join(CSRC, "ogsqp_resident.h"),
               os.path.join(HERE, "..", "include", "ogsqp.h")]
    stamp_path = SQP_LIB + ".stamp"
    want = _digest_files(sources)
    if not force and os.path.exists(SQP_LIB) and os.path.exists(stamp_path):
        with open(stamp_path) as fh:
            if fh.read().strip() == want:
                return SQP_LIB
    tmp = SQP_LIB + ".tmp%d" % os.getpid()
    _run([hipcc()] + HIP_FLAGS + [sources[0], "-o", tmp, "-ldl"])
    os.replace(tmp, SQP_LIB)
    with open(stamp_path, "w") as fh:
        fh.write(want)
    return SQP_LIB


def module_digest(header_source):
    """Key of a callback module: generated header + kernel sources + flags."""
    return _digest_files(_kernel_sources(), extra=header_source + " ".join(MODULE_FLAGS))


def module_path(digest):
    return os.path.join(JITDIR, "libogk_%s.so" % digest)


# The kernel source instantiates the generated callbacks once per kernel, and code generation for them is what a
# module's build time consists of (10-15 s at C3, 40 s at C5 as one translation unit).  The kernels are therefore
# compiled as PARTS side by side (-DOGK_PART=k, csrc/ogk_kernels.hip): part 0 (evaluation, pattern, pack, unpack)
# is <module>.so, the others <module>.p<k>.so next to it; the runtime loads what is there (ogpsx_core.hip:
# ogk_module).  Wall-clock of a cold start = the slowest part.
MODULE_PARTS = (0, 2, 3, 1)     # OGK_PART of <module>.so, .p1.so, .p2.so, .p3.so: main, one-launch sweep, sweep, aux


def part_path(out, index):
    return out if index == 0 else out[:-3] + ".p%d.so" % index


def build_module(header_source, digest=None, force=False, out_suffix="", parts=None):
    """Compile the sweep kernels against one generated header -> path of the module (its part 0; the other
    parts are built next to it, in parallel).  ``out_suffix``: write the result next to the cached module instead
    of over it (cold-start measurements: ``force=True``).  ``OG_MODULE_PARTS=1`` builds one translation unit."""
    os.makedirs(JITDIR, exist_ok=True)
    digest = module_digest(header_source)
    out = module_path(digest)
    if out_suffix:
        out = out[:-3] + out_suffix + ".so"
    kernels = os.path.join(CSRC, "ogk_kernels.hip")
    split = os.environ.get("OG_MODULE_PARTS", "") != "1" if parts is None else bool(parts)
    wanted = [part_path(out, i) for i in range(len(MODULE_PARTS))] if split else [out]
    # <module>.so is part 0 of a split build or a whole module, and only a stamp file tells which: a one-piece request
    # must not take a part 0 left by a split (or interrupted) build for the whole thing
    whole_stamp = out + ".whole"
    cached = all(os.path.exists(path) for path in wanted) and (split or os.path.exists(whole_stamp))
    if split and os.path.exists(whole_stamp) and os.path.exists(out):
        cached = True                   # a whole module serves a split request too: it holds every kernel
    if not force and cached:
        return out
    header = os.path.join(JITDIR, "og_gen_%s.h" % digest)
    with tempfile.NamedTemporaryFile("w", dir=JITDIR, suffix=".h", delete=False) as fh:
        fh.write(header_source)
        tmp_header = fh.name
    os.replace(tmp_header, header)

    def compile_part(index):
        """-> (temporary file, target); the caller renames, in an order a concurrent reader can rely on"""
        target = wanted[index]
        tmp = target + ".tmp%d" % os.getpid()
        define = ["-DOGK_PART=%d" % MODULE_PARTS[index]] if split else []
        _run([hipcc()] + HIP_FLAGS + MODULE_FLAGS + define + ["-I" + CSRC, "-DOG_GEN_HEADER=\"%s\"" % header, kernels, "-o", tmp])
        return tmp, target

    if not split:
        for stale in (part_path(out, i) for i in range(1, len(MODULE_PARTS))):      # a one-piece module has no parts
            if os.path.exists(stale):
                os.remove(stale)
        if os.path.exists(whole_stamp):
            os.remove(whole_stamp)
        os.replace(*compile_part(0))
        with open(whole_stamp, "w") as fh:
            fh.write(digest + "\n")
        return out
    if os.path.exists(whole_stamp):
        os.remove(whole_stamp)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(wanted)) as pool:
        built = list(pool.map(compile_part, range(len(wanted))))
    # every part is compiled before any lands; parts 1.. first and part 0 last: a reader that finds <module>.so finds
    # its parts (the loader opens <module>.so and then looks for the parts next to it)
    for tmp, target in built[1:] + built[:1]:
        os.replace(tmp, target)
    return out
