#!/usr/bin/env python3
"""Golden CSV files from the *reference implementation's* ``Problem.to_csv`` (``OpenGoddard/optimize.py:844-863``).

Build container only (the reference at /root/reference never travels):

    python tools/make_golden_csv.py

For each listed configuration this repo's problem definition is built by the reference engine, given a
deterministic decision vector (seeded uniform numbers, increasing final times) and written with the reference's
``to_csv`` - once with the default delimiter, once with ``;``.  The files under ``tests/golden/`` are the reference's
output bytes: header text, delimiter, ``%.18e`` number format, column order (time, states of phase 0's count, controls).
``tests/test_cabi_and_solve.py::test_to_csv_writes_the_reference_bytes`` compares this package's ``to_csv`` with them.
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REF)
sys.path.append(REPO)

import numpy as np                                   # noqa: E402
import OpenGoddard.optimize as ref                   # noqa: E402
from opengoddard_amd import problems                 # noqa: E402

assert ref.__file__.startswith(REF), ref.__file__

CASES = [("brachistochrone", ","), ("polar_tsto_shipped", ","), ("polar_tsto_shipped", ";")]


def move(prob):
    """The point the CSV is written at (the test does the same): seeded, reproducible, every column distinct."""
    rng = np.random.default_rng(20260928)
    prob.p = rng.uniform(0.1, 1.0, prob.p.size)            # (not built on the guess: that already went through tau)
    prob.p[-prob.number_of_section:] = np.cumsum(rng.uniform(0.2, 0.7, prob.number_of_section))

def main():
    for name, delimiter in CASES:
        prob, obj = problems.build(name, api=ref)
        move(prob)
        tag = "" if delimiter == "," else "_semicolon"
        path = os.path.join(OUT, "to_csv_%s%s.csv" % (name, tag))
        with contextlib.redirect_stdout(io.StringIO()):
            prob.to_csv(path, delimiter=delimiter)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
